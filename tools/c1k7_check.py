"""Conv3d(1, 64, 7, padding 3) forward: the two-term pseudo-channel kernel (k_conv_s3x PC, round 6) against fp64 and against the fp32 matrix kernel
(nc_set_split_terms(3) sends the layer back to it), with timings.  usage: python tools/c1k7_check.py"""
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, '.')
from neuroclear_amd import ops
from neuroclear_amd._lib import lib, I

L = lib()


def timeit(f, n=20):
    f(); f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


g = torch.Generator(device='cuda').manual_seed(5)
for shape in ((1, 40, 40, 40), (2, 30, 44, 52), (1, 36, 36, 36), (1, 108, 108, 108), (1, 24, 148, 148)):
    N, D, H, W = shape
    x = torch.rand(N, 1, D, H, W, device='cuda', generator=g)
    w = torch.randn(64, 1, 7, 7, 7, device='cuda', generator=g) * 0.05
    b = torch.randn(64, device='cuda', generator=g) * 0.1
    ref = F.conv3d(x.double(), w.double(), b.double(), padding=3)
    sc = ref.pow(2).mean().sqrt()
    out = {}
    for terms in (2, 3):
        L.nc_set_split_terms(I(terms))
        y = ops.conv_fwd_raw(x, w, b, 1, 3)
        y2 = ops.conv_fwd_raw(x, w, b, 1, 3)
        e = y.double() - ref
        t = timeit(lambda: ops.conv_fwd_raw(x, w, b, 1, 3))
        out[terms] = (float(e.abs().max() / sc), float(e.pow(2).mean().sqrt() / sc), bool(torch.equal(y, y2)), t)
    L.nc_set_split_terms(I(2))
    print(shape, 'two-term: max %.2e rms %.2e det %s %.3f ms | fp32 MFMA: max %.2e rms %.2e det %s %.3f ms' % (out[2] + out[3]), flush=True)

print('--- data gradient (two-term: k_conv_s3x PC = 2 + k_fold_c1k7) against fp64 autograd', flush=True)
for shape in ((1, 40, 40, 40), (2, 30, 44, 52), (1, 16, 16, 16), (1, 108, 108, 108), (1, 24, 148, 148)):
    N, D, H, W = shape
    dy = torch.randn(N, 64, D, H, W, device='cuda', generator=g)
    w = torch.randn(64, 1, 7, 7, 7, device='cuda', generator=g) * 0.05
    ref = F.conv_transpose3d(dy.double(), w.double(), padding=3)
    sc = ref.pow(2).mean().sqrt()
    out = {}
    for terms in (2, 3):
        L.nc_set_split_terms(I(terms))
        dx = ops.conv_dgrad_raw(dy, w, (N, 1, D, H, W), 1, 3)
        dx2 = ops.conv_dgrad_raw(dy, w, (N, 1, D, H, W), 1, 3)
        e = dx.double() - ref
        t = timeit(lambda: ops.conv_dgrad_raw(dy, w, (N, 1, D, H, W), 1, 3))
        out[terms] = (float(e.abs().max() / sc), float(e.pow(2).mean().sqrt() / sc), bool(torch.equal(dx, dx2)), t)
    L.nc_set_split_terms(I(2))
    print(shape, 'two-term: max %.2e rms %.2e det %s %.3f ms | fp32 MFMA: max %.2e rms %.2e det %s %.3f ms' % (out[2] + out[3]), flush=True)
