import sys, torch
sys.path.insert(0, '.')
from neuroclear_amd import ops
from neuroclear_amd._lib import lib
L = lib()
ops.set_conv_precision('bf16')
def run(shape, K, ks):
    g = torch.Generator(device='cuda').manual_seed(3)
    x = torch.randn(*shape, device='cuda', generator=g)
    w = torch.randn(K, shape[1], ks, ks, ks, device='cuda', generator=g) / (shape[1] * ks ** 3) ** 0.5
    out = {}
    for mode in (0, 2):
        L.nc_set_c8x_mode(mode)
        out[mode] = ops.conv_fwd_raw(x, w, None, 1, ks // 2)
    d = (out[0] - out[2]).abs()
    bad = d > 1e-3 * out[0].abs().max()
    print(shape, K, ks, 'maxdiff %.3g' % d.max().item(), 'nbad', int(bad.sum()), 'of', d.numel())
    if bad.any():
        idx = bad.nonzero()
        print('  first bad', idx[0].tolist(), 'last bad', idx[-1].tolist())
        for dim, nm in enumerate('nkzyx'):
            u = idx[:, dim].unique()
            print('  ', nm, 'count', len(u), 'min', int(u.min()), 'max', int(u.max()), u[:12].tolist())
run((8, 64, 40, 40, 52), 64, 5)
run((8, 192, 20, 40, 52), 64, 5)
run((8, 64, 40, 40, 52), 64, 3)
run((2, 64, 148, 148, 148), 64, 5)
