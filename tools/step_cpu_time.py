"""How long does the HOST take to enqueue one Apollo step (no synchronisation inside), against the step's GPU time?  If the enqueue time of a stretch
of the step exceeds the GPU time of the kernels in front of it, the main stream starves there.  usage: python tools/step_cpu_time.py [apollo|athena]"""
import contextlib, io, sys, time
import numpy as np
import torch
sys.path.insert(0, '.')
import bench
from neuroclear_amd.models import create_model
from neuroclear_amd.util import seed as S

dev = torch.device('cuda', 0)
torch.manual_seed(1234); np.random.seed(1234)
with contextlib.redirect_stdout(io.StringIO()):
    model = create_model(bench.apollo_opt(0, sys.argv[1] if len(sys.argv) > 1 else 'apollo'))
real = torch.from_numpy((S.random_volume(101, 108).astype(np.float64) / 65535.0).astype(np.float32))[None, None].to(dev)
data = {'A': real, 'A_paths': 'x'}
for _ in range(4):
    model.set_input(data); model.optimize_parameters()
torch.cuda.synchronize()
marks = {}
orig = {}
def wrap(name):
    f = getattr(model, name)
    orig[name] = f
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); marks.setdefault(name, []).append(time.perf_counter() - t); return r
    setattr(model, name, g)
for nm in ('forward', 'backward_G', 'backward_D_all', 'backward_D_basic'):
    if hasattr(model, nm): wrap(nm)
cpu, tot = [], []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.set_input(data); model.optimize_parameters()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    cpu.append(t1 - t0); tot.append(t2 - t0)
print('host enqueue time per step: median %.2f ms (min %.2f); step with sync: median %.2f ms' % (np.median(cpu) * 1e3, min(cpu) * 1e3, np.median(tot) * 1e3))
for k, v in marks.items():
    print('  %-16s host %.2f ms' % (k, np.median(v) * 1e3))
