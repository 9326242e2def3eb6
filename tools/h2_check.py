"""Two-term fp16 split (nc_set_split_terms(2), csrc/conv_s3x.hip NT = 2) against the three-term bf16 split and the fp32 MFMA kernels:
error against an fp64 convolution and time per layer.  usage: python tools/h2_check.py [quick]"""
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, '.')
from neuroclear_amd import ops
from neuroclear_amd._lib import lib, I

L = lib()


def timeit(f, n=8):
    f(); f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def err(a, r):
    s = r.pow(2).mean().sqrt()
    e = a.double() - r
    return float(e.abs().max() / s), float(e.pow(2).mean().sqrt() / s)


def data(kind, shape, g):
    x = torch.randn(shape, device='cuda', generator=g)
    if kind == 'relu':
        return x.clamp_min(0)
    if kind == 'grad':   # gradient-like: tiny, log-normal magnitudes
        return x * 1e-5 * torch.exp(2 * torch.randn(shape, device='cuda', generator=g))
    if kind == 'outlier':
        x.view(-1)[12345] = 900.0
        return x
    return x


quick = len(sys.argv) > 1
acc_cases = [(1, 64, 64, 40, 3, 'relu'), (1, 64, 64, 40, 3, 'grad'), (1, 64, 64, 40, 3, 'outlier'), (1, 128, 128, 27, 3, 'relu'), (1, 64, 64, 32, 5, 'randn')]
for N, C, K, E, ks, kind in acc_cases:
    g = torch.Generator(device='cuda').manual_seed(3)
    x = data(kind, (N, C, E, E, E), g)
    w = torch.randn(K, C, ks, ks, ks, device='cuda', generator=g) * (2.0 / (C * ks ** 3)) ** 0.5
    b = torch.randn(K, device='cuda', generator=g) * 0.1
    ref = F.conv3d(x.double(), w.double(), b.double(), padding=ks // 2)
    dy = data('grad' if kind == 'grad' else 'randn', ref.shape, g)
    refd = torch.nn.grad.conv3d_input(x.shape, w.double(), dy.double(), padding=ks // 2)
    out = {}
    for name, split, terms in (('fp32mfma', False, 3), ('bf16x3', True, 3), ('fp16x2', True, 2)):
        ops.set_conv_split(split); L.nc_set_split_terms(I(terms))
        y = ops.conv_fwd_raw(x, w, b, 1, ks // 2)
        dx = ops.conv_dgrad_raw(dy, w, x.shape, 1, ks // 2)
        y2 = ops.conv_fwd_raw(x, w, b, 1, ks // 2)
        out[name] = err(y, ref) + err(dx, refd) + (torch.equal(y, y2),)
    ops.set_conv_split(True); L.nc_set_split_terms(I(3))
    print('%d x %d->%d %d^3 k%d %-8s' % (N, C, K, E, ks, kind) + ' | '.join('%s fwd %.2e/%.2e dgrad %.2e/%.2e det %s' % ((k,) + v) for k, v in out.items()), flush=True)

time_cases = [(1, 64, 64, 108, 3), (1, 128, 128, 54, 3), (1, 256, 256, 27, 3), (1, 64, 64, 108, 5)] if not quick else [(1, 64, 64, 108, 3)]
for N, C, K, E, ks in time_cases:
    g = torch.Generator(device='cuda').manual_seed(3)
    x = torch.randn(N, C, E, E, E, device='cuda', generator=g).clamp_min(0)
    w = torch.randn(K, C, ks, ks, ks, device='cuda', generator=g) * 0.02
    fl = 2.0 * C * K * ks ** 3 * N * E ** 3
    res = []
    for rep in range(2):
        for terms in (3, 2):
            L.nc_set_split_terms(I(terms))
            t = timeit(lambda: ops.conv_fwd_raw(x, w, None, 1, ks // 2))
            res.append('terms %d: %.3f ms %.0f TF' % (terms, t, fl / t / 1e9))
    L.nc_set_split_terms(I(3))
    print('%d x %d->%d %d^3 k%d (fwd incl. conversion + pack) ' % (N, C, K, E, ks) + ' | '.join(res), flush=True)
