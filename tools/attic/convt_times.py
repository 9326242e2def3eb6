"""Timing experiment: ConvTranspose3d(k 2, s 2) forward (fp32) at the U-Net's shapes (training 108^3, inference 140^3)."""
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
    return e0.elapsed_time(e1) / n


for C, K, E in ((256, 128, 27), (128, 64, 54), (256, 128, 35), (128, 64, 70)):
    x = torch.randn(1, C, E, E, E, device='cuda')
    w = torch.randn(C, K, 2, 2, 2, device='cuda') * 0.05
    b = torch.randn(K, device='cuda')
    t = timeit(lambda: ops.conv_transpose_k2s2(x, w, b))
    fl = 2.0 * C * K * 8 * E ** 3
    by = (C * E ** 3 + K * 8 * E ** 3) * 4
    print('convT %3d->%3d %3d^3: %.3f ms  %.1f TFLOP/s  %.0f GB/s (algorithmic bytes)' % (C, K, E, t, fl / t / 1e9, by / t / 1e6))
