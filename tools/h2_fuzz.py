"""Random shapes through the two-term convolutions (forward, data gradient, weight gradient) against fp64: every case must be finite,
bit-identical on a second run, and no further from fp64 (rms relative to the output rms) than 1.3 x the worse of the three-term form and the
fp32 MFMA kernels on the same case (weight gradients of long reductions sit at 4-6e-7 in ALL three: fp32 accumulation).
usage: python tools/h2_fuzz.py [cases] [seed]"""
import sys
import random
import torch
import torch.nn.functional as F
sys.path.insert(0, '.')
from neuroclear_amd import ops
from neuroclear_amd._lib import lib

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
lib().nc_set_split_terms(2)
worst = [0.0, 0.0, 0.0]
bad = 0
for case in range(n_cases):
    ks = rng.choice([3, 3, 3, 5])
    C, K = rng.choice([64, 128, 192, 256]), rng.choice([64, 128, 256])
    if ks == 5:
        C, K = 64, rng.choice([64, 128])
    N = rng.choice([1, 1, 2])
    D, H, W = rng.randint(2, 22), rng.randint(3, 40), rng.randint(3, 60)
    if N * C * D * H * W > 60e6 or N * K * D * H * W > 60e6:
        D = max(2, D // 3)
    g = torch.Generator(device='cuda').manual_seed(1000 + case)
    kind = rng.choice(['randn', 'relu', 'grad'])
    x = torch.randn(N, C, D, H, W, device='cuda', generator=g)
    if kind == 'relu':
        x = x.clamp_min(0)
    if kind == 'grad':
        x = x * 1e-6 * torch.exp(2 * torch.randn(x.shape, device='cuda', generator=g))
    w = torch.randn(K, C, ks, ks, ks, device='cuda', generator=g) * (2.0 / (C * ks ** 3)) ** 0.5
    dy = torch.randn(N, K, D, H, W, device='cuda', generator=g) * rng.choice([1.0, 1e-4, 1e3])
    y = ops.conv_fwd_raw(x, w, None, 1, ks // 2)
    dx = ops.conv_dgrad_raw(dy, w, x.shape, 1, ks // 2)
    dw = ops.conv_wgrad_raw(x, dy, w.shape, 1, ks // 2, False)[0]
    same = torch.equal(y, ops.conv_fwd_raw(x, w, None, 1, ks // 2)) and torch.equal(dx, ops.conv_dgrad_raw(dy, w, x.shape, 1, ks // 2)) and \
        torch.equal(dw, ops.conv_wgrad_raw(x, dy, w.shape, 1, ks // 2, False)[0])
    ry = F.conv3d(x.double(), w.double(), padding=ks // 2)
    rdx = torch.nn.grad.conv3d_input(x.shape, w.double(), dy.double(), padding=ks // 2)
    rdw = torch.nn.grad.conv3d_weight(x.double(), w.shape, dy.double(), padding=ks // 2)
    def errs(a3):
        out = []
        for a, r in zip(a3, (ry, rdx, rdw)):
            s_ = r.pow(2).mean().sqrt().clamp_min(1e-300)
            out.append(float((a.double() - r).pow(2).mean().sqrt() / s_))
        return out
    es = errs((y, dx, dw))
    others = []
    for split, terms in ((True, 3), (False, 3)):
        ops.set_conv_split(split); lib().nc_set_split_terms(terms)
        others.append(errs((ops.conv_fwd_raw(x, w, None, 1, ks // 2), ops.conv_dgrad_raw(dy, w, x.shape, 1, ks // 2),
                            ops.conv_wgrad_raw(x, dy, w.shape, 1, ks // 2, False)[0])))
    ops.set_conv_split(True); lib().nc_set_split_terms(2)
    lim = [1.3 * max(o[i] for o in others) + 2e-8 for i in range(3)]
    ok = same and all(e <= l for e, l in zip(es, lim)) and all(torch.isfinite(t).all() for t in (y, dx, dw))
    bad += not ok
    worst = [max(a, b) for a, b in zip(worst, es)]
    print('%3d %s N%d %3d->%3d %2dx%2dx%2d k%d %-5s fwd %.2e dgrad %.2e wgrad %.2e (three-term %.2e fp32 %.2e) %s' % (
        case, 'ok ' if ok else 'BAD', N, C, K, D, H, W, ks, kind, es[0], es[1], es[2], others[0][2], others[1][2], '' if same else 'NOT DETERMINISTIC'), flush=True)
print('cases %d bad %d worst rms error fwd %.2e dgrad %.2e wgrad %.2e' % (n_cases, bad, worst[0], worst[1], worst[2]))
