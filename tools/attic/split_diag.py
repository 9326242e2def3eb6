"""Where do the split kernel and the fp32 MFMA kernel differ at 108^3, and which one is off (fp64 reference on a slab)?"""
import sys
import numpy as np
import torch
import torch.nn.functional as F
sys.path.insert(0, '.')
from neuroclear_amd import ops

torch.manual_seed(2)
E, C, K, ks = 108, 64, 64, 3
x = torch.randn(1, C, E, E, E, device='cuda')
w = torch.randn(K, C, ks, ks, ks, device='cuda') * (1.0 / np.sqrt(C * ks ** 3))
ops.set_conv_split(True)
y = ops.conv_fwd_raw(x, w, None, 1, 1)
ops.set_conv_split(False)
y32 = ops.conv_fwd_raw(x, w, None, 1, 1)
d = (y - y32).abs()
print('max diff', d.max().item(), 'rms diff', d.pow(2).mean().sqrt().item())
idx = torch.nonzero(d > 3e-6)
print('positions with diff > 3e-6:', idx.shape[0])
print(idx[:20].tolist())
if idx.shape[0]:
    zs = idx[:, 2].unique().tolist()
    print('z planes', zs[:30], 'channels', idx[:, 1].unique().tolist()[:40])
    print('y', idx[:, 3].unique().tolist()[:40], 'x', idx[:, 4].unique().tolist()[:40])
    z0 = max(int(idx[0, 2]) - 2, 0)
    z1 = min(z0 + 6, E)
    lo, hi = max(z0 - 1, 0), min(z1 + 1, E)
    ref = F.conv3d(x[:, :, lo:hi].double().cpu(), w.double().cpu(), padding=1)[:, :, z0 - lo:z0 - lo + (z1 - z0)]
    for name, t in (('split', y), ('fp32', y32)):
        e = (t[:, :, z0:z1].double().cpu() - ref).abs()
        print(name, 'slab z %d..%d: max err %.3e rms %.3e' % (z0, z1, e.max().item(), e.pow(2).mean().sqrt().item()))
