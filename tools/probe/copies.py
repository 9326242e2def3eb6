"""Who issues the device-to-device copies of an Apollo step?  torch.profiler with stacks over one step; prints the Python call sites of
aten::copy_ / aten::clone with their CUDA time.  usage: python tools/probe/copies.py"""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
import bench
from neuroclear_amd.models import create_model

opt = bench.apollo_opt(0)
torch.manual_seed(1); np.random.seed(1)
model = create_model(opt)
real = torch.rand(1, 1, 108, 108, 108, device='cuda')
for _ in range(3):
    model.set_input({'A': real, 'A_paths': 'x'})
    model.optimize_parameters()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    model.set_input({'A': real, 'A_paths': 'x'})
    model.optimize_parameters()
    torch.cuda.synchronize()
rows = []
for e in prof.events():
    if e.name in ('aten::copy_', 'aten::clone', 'aten::contiguous', 'aten::cat', 'aten::zero_', 'aten::fill_', 'aten::mul', 'aten::add', 'aten::add_', 'aten::mul_'):
        dt = getattr(e, 'device_time_total', None) or getattr(e, 'cuda_time_total', 0)
        if dt > 5:
            st = [s for s in e.stack if 'neuroclear_amd' in s or 'bench.py' in s][:2]
            rows.append((dt, e.name, str(e.input_shapes)[:60], ' <- '.join(s.split('/')[-1] for s in st)))
rows.sort(reverse=True)
for r in rows[:40]:
    print('%8.1f us  %-16s %-60s %s' % r)
