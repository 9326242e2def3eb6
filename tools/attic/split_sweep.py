"""Randomised shape sweep of the split-operand kernels against the fp32 kernels (NC_CONV_SPLIT off) -- odd sizes, tiny volumes,
batches; prints the worst relative difference per direction."""
import random
import sys

import torch

sys.path.insert(0, '.')
from neuroclear_amd import ops  # noqa: E402
from neuroclear_amd._lib import I, lib  # noqa: E402

random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
torch.manual_seed(0)
worst = {'fwd': 0.0, 'dgrad': 0.0, 'wgrad': 0.0}
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
for it in range(n_cases):
    ks = random.choice([3, 3, 5])
    N = random.choice([1, 1, 2, 3])
    C = random.choice([8, 16, 32, 64, 96, 128])
    K = random.choice([64, 128])
    D, H, W = random.randint(1, 12), random.randint(1, 40), random.randint(1, 60)
    if random.random() < 0.15:
        W = random.randint(100, 200)
        H = random.randint(1, 6)
    x = torch.randn(N, C, D, H, W, device='cuda')
    w = torch.randn(K, C, ks, ks, ks, device='cuda') * 0.05
    b = torch.randn(K, device='cuda')
    dy = torch.randn(N, K, D, H, W, device='cuda')
    sup = [lib().nc_conv_split_supported(I(wh), I(N), I(C), I(D), I(H), I(W), I(K), I(ks), I(ks), I(ks), I(1), I(ks // 2)) for wh in range(3)]
    res = {}
    for on in (True, False):
        ops.set_conv_split(on)
        y = ops.conv_fwd_raw(x, w, b, 1, ks // 2)
        dx = ops.conv_dgrad_raw(dy, w, x.shape, 1, ks // 2)
        dw, db = ops.conv_wgrad_raw(x, dy, w.shape, 1, ks // 2, True)
        res[on] = (y, dx, dw)
    ops.set_conv_split(True)
    for name, i in (('fwd', 0), ('dgrad', 1), ('wgrad', 2)):
        a, r = res[True][i], res[False][i]
        sc = r.pow(2).mean().sqrt().item() + 1e-30
        e = (a - r).abs().max().item() / sc
        if not torch.isfinite(a).all() or e > 2e-5:
            print('BAD', name, (N, C, K, D, H, W, ks), 'supported', sup, 'rel diff %.3e' % e, flush=True)
        worst[name] = max(worst[name], e)
print('cases', n_cases, 'worst relative differences', worst)
