"""Timing experiment: the PatchGAN layers at Athena's batches (108 and 216 slices of 108^2), fwd / dgrad / wgrad on the
image-staged kernels (conv2d_img.hip).  Run twice with NC_SCONV=0 / NC_SCONV_WGRAD=0 for the gather GEMM."""
import sys
import torch
sys.path.insert(0, '.')
from neuroclear_amd import ops

LAYERS = [(64, 128, 54, 2), (128, 256, 27, 2), (256, 512, 13, 1)]


def timeit(f, n=10):
    f(); f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for N in (108, 216):
    for C, K, H, s in LAYERS:
        x = torch.randn(N, C, H, H, device='cuda')
        w = torch.randn(K, C, 4, 4, device='cuda') * 0.02
        y = ops.conv_fwd_raw(x, w, None, s, 1)
        dy = torch.randn_like(y)
        fl = 2.0 * C * K * 16 * y.numel() / K
        tf = timeit(lambda: ops.conv_fwd_raw(x, w, None, s, 1))
        td = timeit(lambda: ops.conv_dgrad_raw(dy, w, x.shape, s, 1))
        tw = timeit(lambda: ops.conv_wgrad_raw(x, dy, w.shape, s, 1, False))
        print('B %3d %3d->%3d %3d^2 s%d  %6.1f GF | fwd %.3f ms %5.1f TF  dgrad %.3f %5.1f  wgrad %.3f %5.1f' % (
            N, C, K, H, s, fl / 1e9, tf, fl / tf / 1e9, td, fl / td / 1e9, tw, fl / tw / 1e9), flush=True)
