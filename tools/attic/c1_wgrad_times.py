"""Timing experiment: weight gradient of the one-channel layers at configs[3]'s size (4 x 148^3): the 16-bit kernel
(nc_conv_c1_wgrad_c8, planar copies included) against the fp32 tap-axis kernel (nc_conv_wgrad) + its C8 -> fp32 conversion."""
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


N, E = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4, 148)
x = torch.randn(N, 1, E, E, E, device='cuda')
dy = torch.randn(N, 64, E, E, E, device='cuda')
dyh = T.to_c8(dy, T.BF)
for ks in (7, 3):
    nb = L.nc_conv_c1_wgrad_c8_ws_bytes(N, E, E, E, ks)
    wsw = torch.empty(nb, dtype=torch.uint8, device='cuda')
    dw = torch.empty(64, 1, ks, ks, ks, device='cuda')
    t16 = timeit(lambda: L.nc_conv_c1_wgrad_c8(P(x), P(dyh), P(dw), N, E, E, E, ks, P(wsw), ctypes.c_size_t(nb), None))
    w = torch.empty(64, 1, ks, ks, ks, device='cuda')
    t32 = timeit(lambda: ops.conv_wgrad_raw(x, dy, w.shape, 1, ks // 2, False))
    print('ks %d: 16-bit %.3f ms (ws %.2f GB)   fp32 kernel %.3f ms (+ C8 -> fp32 conversion)' % (ks, t16, nb / 1e9, t32))
