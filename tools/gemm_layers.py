"""Timing experiment: the PatchGAN layers at Athena's batch (108 slices of 108^2) on the gather GEMM, fwd / dgrad / wgrad,
with the output-tile rows capped at 256 (default), 128 and 64 (NC_GEMM_TM)."""
import os
import sys
import torch
sys.path.insert(0, '.')
from neuroclear_amd import ops

LAYERS = [(1, 64, 108, 2), (64, 128, 54, 2), (128, 256, 27, 2), (256, 512, 13, 1), (512, 1, 12, 1)]
N = 108


def timeit(f, n=5):
    f(); f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for C, K, H, s in LAYERS:
    x = torch.randn(N, C, H, H, device='cuda')
    w = torch.randn(K, C, 4, 4, device='cuda') * 0.02
    y = ops.conv_fwd_raw(x, w, None, s, 1)
    dy = torch.randn_like(y)
    fl = 2.0 * C * K * 16 * y.numel() / K
    row = '%3d->%3d %3d^2 s%d  %6.1f GF ' % (C, K, H, s, fl / 1e9)
    for tm in ('256', '128'):
        os.environ['NC_GEMM_TM'] = tm
        tf = timeit(lambda: ops.conv_fwd_raw(x, w, None, s, 1))
        td = timeit(lambda: ops.conv_dgrad_raw(dy, w, x.shape, s, 1))
        tw = timeit(lambda: ops.conv_wgrad_raw(x, dy, w.shape, s, 1, False))
        row += ' | %s fwd %.3f ms %5.1f TF  dgrad %.3f %5.1f  wgrad %.3f %5.1f' % (
            'TM<=' + tm, tf, fl / tf / 1e9, td, fl / td / 1e9, tw, fl / tw / 1e9)
    print(row, flush=True)
