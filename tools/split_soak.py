"""Development check: 40 Apollo steps at 64^3 from the same seeds with the fp32 3^3 / 5^3 layers on the split-operand kernels
(default) and on the fp32 MFMA kernels (nc_set_conv_split(0)): two fp32 computations of the same step -- the loss trajectories
separate only as fast as a GAN step amplifies last-bit differences (a second run of the SAME kernels with another summation
order would do the same)."""
import contextlib
import io
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
import bench
from neuroclear_amd import ops
from neuroclear_amd.models import create_model
from neuroclear_amd.util import seed as S

res = {}
for name, on in (('split', True), ('fp32 mfma', False)):
    ops.set_conv_split(on)
    torch.manual_seed(3)
    np.random.seed(3)
    o = bench.apollo_opt(0)
    with contextlib.redirect_stdout(io.StringIO()):
        m = create_model(o)
    traj = []
    for it in range(40):
        v = S.random_volume(200 + it % 4, 64)
        real = torch.from_numpy((v.astype(np.float64) / 65535.0).astype(np.float32))[None, None].cuda()
        m.set_input({'A': real, 'A_paths': 'x'})
        m.optimize_parameters()
        L = m.get_current_losses()
        traj.append((L['cycle'], L['G_A'], L['G_B'], L['D_A_lateral'], L['D_B_lateral']))
    res[name] = np.array(traj)
ops.set_conv_split(True)
for it in (0, 1, 2, 5, 10, 20, 39):
    print('step %2d ' % it + ' | '.join('%s cycle %.5f G_A %.5f G_B %.5f D_A %.5f D_B %.5f' % ((p,) + tuple(res[p][it])) for p in res))
d = np.abs(res['split'] - res['fp32 mfma']) / (np.abs(res['fp32 mfma']) + 1e-3)
for it in (0, 1, 2, 5, 10, 20, 39):
    print('step %2d max relative difference of the 5 tracked losses %.2e' % (it, d[it].max()))
