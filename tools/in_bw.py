"""Timing experiment: the InstanceNorm passes of the U-Net at their sizes (N x C x S), achieved HBM GB/s per kernel."""
import sys
import torch
sys.path.insert(0, '.')
from neuroclear_amd import ops
from neuroclear_amd._lib import lib, I, L_, F, Z

L = lib()
P = ops._ptr


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


for C, side in [(64, 108), (128, 54), (256, 27), (512, 13), (64, 148)]:
    S = side ** 3
    x = torch.randn(1, C, S, device='cuda')
    dy = torch.randn(1, C, S, device='cuda')
    y = torch.empty_like(x)
    dx = torch.empty_like(x)
    m = torch.empty(C, device='cuda')
    r = torch.empty(C, device='cuda')
    db = torch.empty(C, device='cuda')
    nb = L.nc_instnorm_bwd_dbias_ws_bytes(I(C), L_(S))
    ws = torch.empty(nb, dtype=torch.uint8, device='cuda')
    B = x.numel() * 4 / 1e9
    t0 = timeit(lambda: L.nc_instnorm_stats(P(x), I(C), L_(S), F(1e-5), P(m), P(r), P(ws), Z(nb), None))
    t1 = timeit(lambda: L.nc_instnorm_act_fwd(P(x), P(m), P(r), F(0.0), P(y), I(C), L_(S), None))
    t2 = timeit(lambda: L.nc_instnorm_act_bwd_dbias(P(dy), P(x), P(m), P(r), F(0.0), P(dx), P(db), I(1), I(C), L_(S), P(ws), Z(nb), None))
    print('C %3d %3d^3 (%.2f GB): stats %.3f ms %5.0f GB/s | act_fwd %.3f ms %5.0f GB/s | bwd(+dbias) %.3f ms %5.0f GB/s (5 passes)' % (
        C, side, B, t0, B / t0 * 1e3, t1, 2 * B / t1 * 1e3, t2, 5 * B / t2 * 1e3))
