"""GPU time of the phases of one Athena (or Apollo) step, by events on the calling stream: forward | discriminator passes of the generator loss |
generators' backward | optimizer_G | discriminators' update | optimizer_D.  usage: python tools/athena_phases.py [athena|apollo] [crop batch precision]"""
import contextlib, io, sys
import numpy as np
import torch
sys.path.insert(0, '.')
import bench
from neuroclear_amd.models import create_model
from neuroclear_amd.util import seed as S

which = sys.argv[1] if len(sys.argv) > 1 else 'athena'
crop, batch = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (108, 1)
precision = sys.argv[4] if len(sys.argv) > 4 else 'fp32'
dev = torch.device('cuda', 0)
torch.manual_seed(1234); np.random.seed(1234)
with contextlib.redirect_stdout(io.StringIO()):
    o = bench.apollo_opt(0, which)
    if precision != 'fp32':
        o.precision = precision
    model = create_model(o)
real = torch.from_numpy((S.random_volume(101, crop).astype(np.float64) / 65535.0).astype(np.float32))[None, None].to(dev).repeat(batch, 1, 1, 1, 1)
data = {'A': real, 'A_paths': 'x'}
for _ in range(4):
    model.set_input(data); model.optimize_parameters()
torch.cuda.synchronize()
marks = []
def mark(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((name, e))
def wrap(obj, name, before=None, after=None):
    f = getattr(obj, name)
    def g(*a, **k):
        if before: mark(before)
        r = f(*a, **k)
        if after: mark(after)
        return r
    setattr(obj, name, g)
wrap(model, 'forward', 'step start', 'forward done')
many = '_on_streams' if hasattr(model, '_on_streams') else '_D_many'
wrap(model, many, None, 'discriminator jobs joined')
wrap(model.optimizer_G, 'step', 'generators backward done', 'optimizer_G done')
wrap(model.optimizer_D, 'step', None, 'optimizer_D done')
acc = {}
for it in range(8):
    marks.clear()
    model.set_input(data); model.optimize_parameters()
    torch.cuda.synchronize()
    for (n0, e0), (n1, e1) in zip(marks, marks[1:]):
        acc.setdefault('%s -> %s' % (n0, n1), []).append(e0.elapsed_time(e1))
tot = 0.0
for k, v in acc.items():
    print('%-64s %7.2f ms' % (k, np.median(v))); tot += np.median(v)
print('sum %.2f ms' % tot)
