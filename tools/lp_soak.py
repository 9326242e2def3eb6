"""Development check: 40 Apollo steps at 64^3 from the same seeds in fp32, bf16 and fp16 (= fp16 forward / bf16 backward
operands): the loss trajectories should stay together (the 16-bit path must not drift or blow up)."""
import contextlib
import io
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
import bench
from neuroclear_amd.models import create_model
from neuroclear_amd.util import seed as S

res = {}
for prec in ('fp32', 'bf16', 'fp16'):
    torch.manual_seed(3)
    np.random.seed(3)
    o = bench.apollo_opt(0)
    o.precision = prec
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
    res[prec] = np.array(traj)
for it in (0, 1, 5, 10, 20, 39):
    print('step %2d ' % it + ' | '.join('%s cycle %.4f G_A %.4f G_B %.4f D_A %.4f D_B %.4f' % ((p,) + tuple(res[p][it])) for p in res))
for p in ('bf16', 'fp16'):
    d = np.abs(res[p] - res['fp32']) / (np.abs(res['fp32']) + 1e-3)
    print(p, 'max relative deviation of the 5 tracked losses over 40 steps: %.3f' % d.max(), 'finite:', np.isfinite(res[p]).all())
