import sys, torch
sys.path.insert(0, '.')
from neuroclear_amd import ops
from neuroclear_amd._lib import lib
L = lib()
for ks, shp in ((3, (1, 64, 20, 20, 20)), (5, (1, 64, 20, 20, 20)), (5, (1, 64, 4, 40, 52))):
    g = torch.Generator(device='cuda').manual_seed(3)
    x = torch.randn(*shp, device='cuda', generator=g)
    w = torch.randn(64, 64, ks, ks, ks, device='cuda', generator=g) / (64 * ks ** 3) ** 0.5
    ops.set_conv_precision('bf16')
    out = {}
    for mode in (0, 2):
        L.nc_set_c8x_mode(mode)
        print(ks, shp, 'mode', mode, 'uses', L.nc_conv_lp_uses_c8x(0, 1, *shp[:1], shp[1], *shp[2:], 64, ks))
        out[mode] = (ops.conv_fwd_raw(x, w, None, 1, ks // 2), ops.conv_dgrad_raw(x, w, x.shape, 1, ks // 2))
    for i, nm in enumerate(('fwd', 'dgrad')):
        d = (out[0][i] - out[2][i]).abs()
        print('  ', nm, 'equal', torch.equal(out[0][i], out[2][i]), 'max diff', d.max().item(), 'ndiff', int((d > 0).sum()), 'of', d.numel())
