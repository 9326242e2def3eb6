"""Timing experiment: the PatchGAN's first layer (1 -> 64, 4 x 4, s 2) and head (512 -> 1, 4 x 4, s 1) at Athena's batches."""
import sys
import torch
sys.path.insert(0, '.')
from neuroclear_amd import ops


def timeit(f, n=10):
    f(); f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for B in (108, 216):
    for name, C, K, H, s in (('first', 1, 64, 108, 2), ('head', 512, 1, 12, 1)):
        x = torch.randn(B, C, H, H, device='cuda')
        w = torch.randn(K, C, 4, 4, device='cuda') * 0.1
        b = torch.randn(K, device='cuda')
        y = ops.conv_fwd_raw(x, w, b, s, 1)
        dy = torch.randn_like(y)
        mb = max(x.numel(), y.numel()) * 4 / 1e6
        tf = timeit(lambda: ops.conv_fwd_raw(x, w, b, s, 1))
        td = timeit(lambda: ops.conv_dgrad_raw(dy, w, x.shape, s, 1))
        tw = timeit(lambda: ops.conv_wgrad_raw(x, dy, w.shape, s, 1, True))
        print('B %3d %-5s big tensor %6.1f MB | fwd %6.1f us (%5.0f GB/s)  dgrad %6.1f us (%5.0f)  wgrad+db %6.1f us (%5.0f)' % (
            B, name, mb, tf, mb / tf * 1e3, td, mb / td * 1e3, tw, mb / tw * 1e3))
    a = torch.randn(B, 64, 54, 54, device='cuda')
    t1 = timeit(lambda: ops.leaky_relu(a, 0.2))
    print('B %3d lrelu fwd on the first activation: %6.1f us' % (B, t1))
