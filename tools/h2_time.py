"""Time the default (two-term) forward of the step's / the inference cube's 3^3 / 5^3 layer shapes through nc_conv_fwd (conversion + weight pack
included: constant across kernel variants) and check one slab against fp64.  usage: [NC_HIP_LIB=variant.so] python tools/h2_time.py [reps]"""
import os
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, '.')
from neuroclear_amd import ops
from neuroclear_amd._lib import lib, I

L = lib()
L.nc_set_split_terms(I(2))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
tag = os.path.basename(os.environ.get('NC_HIP_LIB', 'default'))


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
for ks in (3, 5):
    x = torch.randn(1, 64, 12, 30, 108, device='cuda', generator=g)
    w = torch.randn(64, 64, ks, ks, ks, device='cuda', generator=g) * 0.03
    ref = F.conv3d(x.double(), w.double(), padding=ks // 2)
    y = ops.conv_fwd_raw(x, w, None, 1, ks // 2)
    e = y.double() - ref
    sc = ref.pow(2).mean().sqrt()
    print('[%s] ks %d err vs fp64: max %.2e rms %.2e' % (tag, ks, float(e.abs().max() / sc), float(e.pow(2).mean().sqrt() / sc)), flush=True)
cases = [('64->64 108', 64, 64, 108, 3), ('128->64 108', 128, 64, 108, 3), ('128->128 54', 128, 128, 54, 3), ('256->128 54', 256, 128, 54, 3),
         ('256->256 27', 256, 256, 27, 3), ('5^3 64->64 108', 64, 64, 108, 5), ('64->64 140', 64, 64, 140, 3), ('128->64 140', 128, 64, 140, 3),
         ('128->128 70', 128, 128, 70, 3), ('256->256 35', 256, 256, 35, 3)]
for rep in range(reps):
    tot = 0.0
    out = []
    for name, C, K, E, ks in cases:
        x = torch.randn(1, C, E, E, E, device='cuda', generator=g).clamp_min(0)
        w = torch.randn(K, C, ks, ks, ks, device='cuda', generator=g) * 0.02
        t = timeit(lambda: ops.conv_fwd_raw(x, w, None, 1, ks // 2))
        tot += t
        out.append('%s %.3f' % (name, t))
        del x, w
    print('[%s] rep %d total %.3f ms | ' % (tag, rep, tot) + ' | '.join(out), flush=True)
