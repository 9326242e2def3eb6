"""Does batching cubes help the inference convolutions?  Time the two-term forward of the 140^3 cube's 3^3 layer shapes through nc_conv_fwd
(conversion + pack included) at N = 1 (x 3 calls) and at N = 3 (one call): tile quantisation of the persistent 256-workgroup launches
(a 35^3 layer is 1.64 rounds at N = 1, 4.92 at N = 3).  usage: python tools/h2_batch_time.py"""
import sys
import torch
sys.path.insert(0, '.')
from neuroclear_amd import ops
from neuroclear_amd._lib import lib, I

L = lib()
L.nc_set_split_terms(I(2))


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


g = torch.Generator(device='cuda').manual_seed(3)
cases = [('64->64 140', 64, 64, 140), ('128->64 140', 128, 64, 140), ('64->128 70', 64, 128, 70), ('128->128 70', 128, 128, 70),
         ('256->128 70', 256, 128, 70), ('128->256 35', 128, 256, 35), ('256->256 35', 256, 256, 35)]
mult = {'64->64 140': 1, '128->64 140': 1, '64->128 70': 1, '128->128 70': 2, '256->128 70': 1, '128->256 35': 1, '256->256 35': 2}
for rep in range(2):
    t1s = t3s = 0.0
    for name, C, K, E in cases:
        x = torch.randn(3, C, E, E, E, device='cuda', generator=g).clamp_min(0)
        w = torch.randn(K, C, 3, 3, 3, device='cuda', generator=g) * 0.02
        x1 = x[:1].contiguous()
        t1 = timeit(lambda: ops.conv_fwd_raw(x1, w, None, 1, 1)) * 3
        t3 = timeit(lambda: ops.conv_fwd_raw(x, w, None, 1, 1))
        t1s += t1 * mult[name]; t3s += t3 * mult[name]
        print('%-14s 3 x N=1 %.3f ms   N=3 %.3f ms   ratio %.3f' % (name, t1, t3, t3 / t1), flush=True)
        del x, w, x1
    print('rep %d: cube mix (3 cubes) N=1 %.2f ms  N=3 %.2f ms  ratio %.3f' % (rep, t1s, t3s, t3s / t1s), flush=True)
