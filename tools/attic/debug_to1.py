"""Debug helper: 64->1 7^3 data gradient (MFMA kernel) against torch, error by output coordinate."""
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, '.')
from neuroclear_amd import ops
shape = tuple(int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (10, 9, 20)
torch.manual_seed(0)
dy = torch.randn(1, 64, *shape, device='cuda')
w = torch.randn(64, 1, 7, 7, 7, device='cuda') * 0.05
ref = torch.nn.grad.conv3d_input((1, 1) + shape, w, dy, padding=3)
got = ops.conv_dgrad_raw(dy, w, (1, 1) + shape, 1, 3)
err = (got - ref).abs()[0, 0]
print('max err', float(err.max()), 'ref max', float(ref.abs().max()))
bad = (err > 1e-3 * ref.abs().max()).nonzero()
print('bad count', len(bad), 'of', err.numel())
print('bad z', sorted(set(bad[:, 0].tolist())))
print('bad y', sorted(set(bad[:, 1].tolist())))
print('bad x', sorted(set(bad[:, 2].tolist())))
if len(bad):
    z, y, x = bad[0].tolist()
    print('first', (z, y, x), float(got[0, 0, z, y, x]), float(ref[0, 0, z, y, x]))
