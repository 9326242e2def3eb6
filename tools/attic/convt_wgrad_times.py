"""Timing experiment: ConvTranspose3d(k 2, s 2) weight gradient on C8 (bf16) at configs[3]'s two layers (4 x 148^3 output)."""
import ctypes
import sys
import torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from neuroclear_amd import ops
from neuroclear_amd._lib import lib
import test_gpu_c8 as T

L = lib()
P = ops._ptr


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


N = 4
for C, K, E in ((256, 128, 37), (128, 64, 74)):
    x = torch.randn(N, C, E, E, E, device='cuda')
    dy = torch.randn(N, K, 2 * E, 2 * E, 2 * E, device='cuda')
    xh, dyh = T.to_c8(x, T.BF), T.to_c8(dy, T.BF)
    dw = torch.empty(C, K, 2, 2, 2, device='cuda')
    db = torch.empty(K, device='cuda')
    nb = L.nc_convT_c8_ws_bytes(N, C, E, E, E, K)
    wsb = torch.empty(nb, dtype=torch.uint8, device='cuda')
    t = timeit(lambda: L.nc_convT_k2s2_wgrad_c8(P(xh), P(dyh), K, 0, P(dw), P(db), N, C, E, E, E, K, P(wsb), ctypes.c_size_t(nb), None))
    w = torch.randn(C, K, 2, 2, 2, device='cuda') * 0.05
    b = torch.randn(K, device='cuda')
    out = torch.empty(N, K // 8, 8 * E ** 3, 8, dtype=torch.bfloat16, device='cuda')
    dxh = torch.empty(N, C // 8, E ** 3, 8, dtype=torch.bfloat16, device='cuda')
    tf = timeit(lambda: L.nc_convT_k2s2_fwd_c8(P(xh), P(w), P(b), P(out), K, 0, N, C, E, E, E, K, T.BF, P(wsb), ctypes.c_size_t(nb), None))
    td = timeit(lambda: L.nc_convT_k2s2_dgrad_c8(P(dyh), K, 0, P(w), P(dxh), N, C, E, E, E, K, P(wsb), ctypes.c_size_t(nb), None))
    by = (x.numel() + dy.numel()) * 2
    print('convT fwd %.3f ms (%.0f GB/s)  dgrad %.3f ms (%.0f GB/s)' % (tf, by / tf / 1e6, td, by / td / 1e6))
    print('convT wgrad %3d->%3d %3d^3 x %d: %.3f ms  (%.2f GB of operands: %.0f GB/s)' % (C, K, E, N, t, by / 1e9, by / t / 1e6))
