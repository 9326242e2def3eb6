"""Gradient accuracy against an fp64 run of the oracle (the loosest bound of the suite used to be the L2-relative 2e-2 of
tests/test_gpu_nets.py::check_grads, justified by a measurement that lived in tools/debug_unet.py).

The networks' input- and weight-gradients (models/networks.py:512-538 unet_deconv, :1030-1066 PatchGAN, :913-917 deep_linear_gen) from
the HIP kernels are compared with oracle/nets.py evaluated in float64 on the same weights and inputs.  The yardstick is the fp32
ORACLE's own distance to fp64 (torch-CPU fp32, i.e. the reference's arithmetic): a correct fp32 implementation lands at about that
distance, a wrong-by-1 % weight gradient in one layer lands 10-100 x beyond it.

Through ReLU / max-pool an activation within ~1e-7 of zero (or two pool candidates within 1e-7 of each other) takes the other branch
under another summation order and moves the gradients of everything upstream by a visible amount: the test COUNTS those decisions
(stage outputs of the layer-by-layer path vs the fp64 run) and runs on a seed at which no side has one (round 5: no widened bounds)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from neuroclear_amd.models import networks  # noqa: E402
from neuroclear_amd.util import seed as S  # noqa: E402
from oracle import nets as onets  # noqa: E402

DEV = 'cuda'


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


def _oracle(fn, sd_np, x_np, r_np, dtype, **kw):
    sd = {k: torch.from_numpy(v).to(dtype).requires_grad_(True) for k, v in sd_np.items()}
    x = torch.from_numpy(x_np).to(dtype).requires_grad_(True)
    taps = {'acts': {}} if fn is onets.unet_deconv else {}
    y = fn(sd, x, **kw) if fn is not onets.unet_deconv else fn(sd, x, taps)
    (y * torch.from_numpy(r_np).to(dtype)).mean().backward()
    acts = taps.pop('acts', {})  # all ten norm + ReLU outputs (the five stage outputs are among them)
    return y.detach(), x.grad, {k: v.grad for k, v in sd.items()}, {k: v.detach() for k, v in (acts or taps).items()}


_POOLED = ('double_conv1.convolution.4', 'double_conv2.convolution.4')  # the two activations a MaxPool3d(2) reads (networks.py:491,494)


def _flips(a, ref, key=None):
    """ReLU decisions (zero pattern) that differ between two norm + ReLU outputs, and -- for the tensors a pool reads (key None: any) -- 2x2x2
    max-pool winners that differ."""
    relu = int(((a > 0) != (ref > 0)).sum())
    pool = 0
    if min(a.shape[2:]) >= 2 and (key is None or key in _POOLED):
        ia = F.max_pool3d(a.float(), 2, return_indices=True)[1]
        ib = F.max_pool3d(ref.float(), 2, return_indices=True)[1]
        pool = int((ia != ib).sum())
    return relu, pool


@pytest.fixture
def _terms():
    from neuroclear_amd._lib import lib
    t = lib().nc_get_split_terms()
    yield lib()
    lib().nc_set_split_terms(t)


_CLEAN = {}


def _hook_acts(net, into):
    for nm, mod in net.named_modules():
        if isinstance(mod, networks.InstanceNormAct):
            mod.register_forward_hook(lambda m, i, o, nm=nm: into.__setitem__(nm, o.detach().cpu()))


def _clean_case(terms, monkeypatch, size=16):
    """A (weight seed, input seed) at which NEITHER the fp32 oracle NOR the product under `terms` takes a ReLU / max-pool decision the fp64
    run does not take: a single such flip moves the gradients of everything upstream by 1e-3 .. 1e-2 (round 4 widened every tensor's bound
    to the oracle's worst tensor for that reason, which let a 0.9 % error in any one layer pass).  An activation lands within ~1e-7 of zero
    a few times per evaluation of the 4.6 M activations at 32^3 (1-4 decisions at every seed tried) and a few times in ten at 16^3, where a
    seed free of flips on all three sides turns up within a few tries; ALL ten norm + ReLU outputs are compared, not only the stage outputs; the decisions are counted on the layer-by-layer path, to which the fused path is bit-identical (tests/test_gpu_nets.py)."""
    from neuroclear_amd._lib import lib
    if terms in _CLEAN:
        return _CLEAN[terms]
    lib().nc_set_split_terms(terms)
    monkeypatch.setattr(networks, '_FUSED_GEN', False)
    spec = S.unet_deconv_spec()
    tried = []
    for seed in range(2, 40):
        sd_np = S.weights_from_seed(spec, seed)
        x_np = np.random.default_rng(100 + seed).random((1, 1, size, size, size), dtype=np.float32)
        r_np = np.random.default_rng(200 + seed).random((1, 1, size, size, size), dtype=np.float32)
        o64 = _oracle(onets.unet_deconv, sd_np, x_np, r_np, torch.float64)
        o32 = _oracle(onets.unet_deconv, sd_np, x_np, r_np, torch.float32)
        n32 = sum(sum(_flips(o32[3][k], o64[3][k], k)) for k in o64[3])
        net = networks.define_G(1, 1, 64, 'unet_deconv', 'instance', False, 'kaiming', 0.02, [0])
        net.load_state_dict(S.state_dict_from_seed(spec, seed, DEV))
        stages = {}
        _hook_acts(net, stages)
        net(torch.from_numpy(x_np).to(DEV).requires_grad_(True))  # (the training forward, layer by layer: no_grad is the whole-network inference call)
        npd = sum(sum(_flips(stages[k], o64[3][k], k)) for k in o64[3])
        tried.append((seed, n32, npd))
        if n32 == 0 and npd == 0:
            _CLEAN[terms] = (seed, sd_np, x_np, r_np, o64, o32)
            print('terms %d: flip-free seed %d after %s' % (terms, seed, tried))
            return _CLEAN[terms]
    raise AssertionError('no flip-free seed among %s' % tried)


@pytest.mark.parametrize('terms', [3, 2])
@pytest.mark.parametrize('fused', [True, False])
def test_unet_deconv_gradients_against_fp64(fused, terms, monkeypatch, _terms):
    """Every weight tensor of unet_deconv within 3 x the fp32 oracle's OWN distance to the fp64 run (+ 2e-5), for the three-term AND the
    two-term (default) arithmetic alike, on a case where no side flips a ReLU / pool decision (see _clean_case) -- no widening."""
    size = 16  # (0.6 M activations: a flip-free seed exists within a few tries; at 32^3 every seed has one to four)
    seed, sd_np, x_np, r_np, (y64, dx64, g64, t64), (y32, dx32, g32, t32) = _clean_case(terms, monkeypatch, size)
    _terms.nc_set_split_terms(terms)   # 3: the three-term bf16 form of the split-operand kernels, 2: the two-term fp16 form (the default)
    monkeypatch.setattr(networks, '_FUSED_GEN', fused)
    spec = S.unet_deconv_spec()
    net = networks.define_G(1, 1, 64, 'unet_deconv', 'instance', False, 'kaiming', 0.02, [0])
    net.load_state_dict(S.state_dict_from_seed(spec, seed, DEV))
    stages = {}
    if not fused:  # every norm + ReLU output of the layer-by-layer path, named as the oracle's taps
        _hook_acts(net, stages)
    x = torch.from_numpy(x_np).to(DEV).requires_grad_(True)
    y = net(x)
    (y * torch.from_numpy(r_np).to(DEV)).mean().backward()
    assert float((y.detach().cpu().double() - y64).abs().max()) < 2e-6
    if stages:
        assert sum(sum(_flips(stages[k], t64[k], k)) for k in t64) == 0
    e32, ep = rel(dx32, dx64), rel(x.grad, dx64)
    print('dx: fp32 oracle %.2e product %.2e' % (e32, ep))
    assert ep <= 3 * e32 + 2e-5
    worst = 0.0
    for k, p in net.named_parameters():
        if p.dim() < 2:
            continue  # biases in front of InstanceNorm: true gradient 0, both sides hold rounding noise
        a, b = rel(p.grad, g64[k]), rel(g32[k], g64[k])
        print('  %-38s fp32 oracle %.2e  product %.2e' % (k, b, a))
        worst = max(worst, a / max(b, 1e-7))
        # a correct fp32 gradient sits at the oracle's own distance or below it (1e-6 .. 3e-6 for every tensor); a 1 % error in one layer
        # is 1e-2, a 0.1 % error 1e-3
        assert a <= 3 * b + 2e-5, (k, a, b)
    print('worst ratio product / fp32 oracle: %.2f' % worst)


@pytest.mark.parametrize('kind', ['patchgan', 'deep_linear', 'convT'])
def test_flip_free_networks_gradients_against_fp64(kind):
    """No ReLU upstream of these parameters (LeakyReLU has a kink but no dead branch; deep_linear_gen and ConvTranspose3d are linear):
    per-parameter relative L2 <= 1e-3 against fp64, and within 1.5 x of the fp32 oracle's own distance + 1e-4."""
    rng = np.random.default_rng(7)
    if kind == 'patchgan':
        spec, fn = S.patchgan_spec(2), onets.patchgan
        net = networks.define_D(1, 64, 'basic', 3, 'instance', 'kaiming', 0.02, False, [0], dimension=2)
        x_np = rng.random((2, 1, 108, 108), dtype=np.float32)
    elif kind == 'deep_linear':
        spec, fn = S.deep_linear_spec(), onets.deep_linear
        net = networks.define_G(1, 1, 64, 'deep_linear_gen', 'instance', False, 'kaiming', 0.02, [0])
        x_np = rng.random((1, 1, 24, 24, 24), dtype=np.float32)
    else:
        spec = {'t.weight': (128, 64, 2, 2, 2), 't.bias': (64,)}
        fn = lambda sd, x: onets._convT(x, sd, 't')  # noqa: E731
        net = networks.ConvTranspose(128, 64, dimension=3).to(DEV)
        x_np = (rng.random((1, 128, 9, 10, 11), dtype=np.float32) - 0.5)
    if kind == 'convT':
        sd_np = {'t.weight': (rng.standard_normal(spec['t.weight']) * 0.05).astype(np.float32), 't.bias': rng.standard_normal(64).astype(np.float32)}
        net.load_state_dict({'weight': torch.from_numpy(sd_np['t.weight']), 'bias': torch.from_numpy(sd_np['t.bias'])})
        names = {'weight': 't.weight', 'bias': 't.bias'}
    else:
        sd_np = S.weights_from_seed(spec, 3)
        net.load_state_dict(S.state_dict_from_seed(spec, 3, DEV))
        names = {k: k for k in sd_np}
    x = torch.from_numpy(x_np).to(DEV).requires_grad_(True)
    y = net(x)
    r_np = rng.random(tuple(y.shape), dtype=np.float32)
    (y * torch.from_numpy(r_np).to(DEV)).mean().backward()
    y64, dx64, g64, _ = _oracle(fn, sd_np, x_np, r_np, torch.float64)
    y32, dx32, g32, _ = _oracle(fn, sd_np, x_np, r_np, torch.float32)
    print('%s dx: fp32 oracle %.2e product %.2e' % (kind, rel(dx32, dx64), rel(x.grad, dx64)))
    assert rel(x.grad, dx64) <= max(1.5 * rel(dx32, dx64) + 1e-4, 0) and rel(x.grad, dx64) < 1e-3
    for k, p in net.named_parameters():
        if kind == 'patchgan' and p.dim() < 2 and k not in ('model.0.bias', 'model.11.bias'):
            continue  # biases in front of InstanceNorm2d: true gradient 0
        a, b = rel(p.grad, g64[names[k]]), rel(g32[names[k]], g64[names[k]])
        print('  %-24s fp32 oracle %.2e  product %.2e' % (k, b, a))
        assert a < 1e-3 and a <= 1.5 * b + 1e-4, (k, a, b)
