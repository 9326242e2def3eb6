"""How far ahead of the GPU is the host?  Enqueue time of K Apollo steps (no sync) vs their completion time."""
import sys
import time
import contextlib
import io

import numpy as np
import torch

sys.path.insert(0, '.')
import bench  # noqa: E402
from neuroclear_amd.models import create_model  # noqa: E402
from neuroclear_amd.util import seed as S  # noqa: E402

torch.manual_seed(0)
np.random.seed(0)
with contextlib.redirect_stdout(io.StringIO()):
    o = bench.apollo_opt(0)
    o.precision = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
    model = create_model(o)
vol = S.random_volume(100, 108)
real = torch.from_numpy((vol.astype(np.float64) / 65535.0).astype(np.float32))[None, None].cuda()
data = {'A': real, 'A_paths': 'x'}
for _ in range(3):
    model.set_input(data)
    model.optimize_parameters()
torch.cuda.synchronize()
K = 10
t0 = time.perf_counter()
for _ in range(K):
    model.set_input(data)
    model.optimize_parameters()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('enqueue %.1f ms/step, complete %.1f ms/step' % ((t1 - t0) / K * 1e3, (t2 - t0) / K * 1e3))
