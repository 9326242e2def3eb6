"""Timing experiment: the PatchGAN layers at configs[3]'s discriminator batches (8-16 planes of 148^2)."""
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


for N in (8, 16):
    for C, K, H, s in [(64, 128, 74, 2), (128, 256, 37, 2), (256, 512, 18, 1)]:
        x = torch.randn(N, C, H, H, device='cuda')
        w = torch.randn(K, C, 4, 4, device='cuda') * 0.02
        y = ops.conv_fwd_raw(x, w, None, s, 1)
        dy = torch.randn_like(y)
        fl = 2.0 * C * K * 16 * y.numel() / K
        tf = timeit(lambda: ops.conv_fwd_raw(x, w, None, s, 1))
        td = timeit(lambda: ops.conv_dgrad_raw(dy, w, x.shape, s, 1))
        tw = timeit(lambda: ops.conv_wgrad_raw(x, dy, w.shape, s, 1, False))
        print('B %3d %3d->%3d %3d^2 s%d %5.1f GF | fwd %6.1f us %5.1f TF  dgrad %6.1f %5.1f  wgrad %6.1f %5.1f' % (
            N, C, K, H, s, fl / 1e9, tf, fl / tf / 1e6, td, fl / td / 1e6, tw, fl / tw / 1e6), flush=True)
