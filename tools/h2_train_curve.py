"""Apollo training on the two-term and on the three-term form of the split-operand convolutions with the same seeds and data: the losses of both
runs at steps 1, 2, 5, 10, 20, 30 and their largest relative difference per step (an fp32-MFMA run beside them as the yardstick: how far do two
VALID fp32 evaluations of the same training drift apart).  usage: python tools/h2_train_curve.py [steps] [crop]"""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
import bench
from neuroclear_amd import ops
from neuroclear_amd._lib import lib
from neuroclear_amd.models import create_model

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
crop = int(sys.argv[2]) if len(sys.argv) > 2 else 72


def run(terms, split=True):
    ops.set_conv_split(split)
    lib().nc_set_split_terms(terms)
    torch.manual_seed(3); np.random.seed(3)
    model = create_model(bench.apollo_opt(0))
    g = torch.Generator(device='cuda').manual_seed(11)
    out = []
    for it in range(steps):
        real = torch.rand(1, 1, crop, crop, crop, device='cuda', generator=g)
        model.set_input({'A': real, 'A_paths': 'x'})
        model.optimize_parameters()
        out.append(dict(model.get_current_losses()))
    ops.set_conv_split(True)
    lib().nc_set_split_terms(2)
    return out


a, b, c = run(2), run(3), run(3, split=False)
keys = list(a[0].keys())
print('crop %d^3, %d steps; max over the %d losses of |x - fp32mfma| / |fp32mfma|' % (crop, steps, len(keys)))
for it in (0, 1, 4, 9, 19, 29):
    if it >= steps:
        break
    d2 = max(abs(a[it][k] - c[it][k]) / max(abs(c[it][k]), 1e-12) for k in keys)
    d3 = max(abs(b[it][k] - c[it][k]) / max(abs(c[it][k]), 1e-12) for k in keys)
    print('step %2d  two-term %.2e  three-term %.2e   G_A %.5f / %.5f / %.5f  cycle %.5f / %.5f / %.5f' % (
        it + 1, d2, d3, a[it]['G_A'], b[it]['G_A'], c[it]['G_A'], a[it]['cycle'], b[it]['cycle'], c[it]['cycle']))
