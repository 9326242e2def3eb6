"""Conv3d(1, 64, 7, padding 3) -- deep_linear_gen's first layer, reference models/networks.py:899 -- on the two-term 16-bit matrix kernels
(round 6, csrc/conv_s3x.hip: k_conv_s3x PC = 1 forward in pseudo-channel form, PC = 2 + k_fold_c1k7 data gradient), through the C ABI
(nc_conv_fwd / nc_conv_dgrad) against fp64 and against the fp32 matrix kernels they replace (nc_set_split_terms(3) sends the layer back
to those).  Criteria as tests/test_gpu_h2.py: error against fp64 not above 1.3 x (rms) / 2 x (max) of the fp32 kernel's, run-to-run identical,
per-tensor scales 1e-30 .. 1e30 exact, a non-finite input reaches the outputs it touches."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from neuroclear_amd import ops  # noqa: E402
from neuroclear_amd._lib import I, lib  # noqa: E402

DEV = 'cuda'
SHAPES = [(1, 40, 40, 40), (2, 30, 44, 52), (1, 16, 16, 16), (1, 20, 36, 108), (3, 12, 24, 28)]


@pytest.fixture(autouse=True)
def _two_terms():
    t = lib().nc_get_split_terms()
    lib().nc_set_split_terms(I(2))
    yield
    lib().nc_set_split_terms(I(t))


def _err(a, ref):
    e = a.double() - ref
    sc = ref.pow(2).mean().sqrt()
    return float(e.abs().max() / sc), float(e.pow(2).mean().sqrt() / sc)


def _path(C, K):
    return lib().nc_conv_fwd_path(I(C), I(K), I(7), I(7), I(7), I(1), I(3))


@pytest.mark.parametrize('shape', SHAPES)
def test_forward_against_fp64_and_the_fp32_kernel(shape):
    N, D, H, W = shape
    g = torch.Generator(device=DEV).manual_seed(sum(shape))
    x = torch.rand(N, 1, D, H, W, device=DEV, generator=g)
    w = torch.randn(64, 1, 7, 7, 7, device=DEV, generator=g) * 0.05
    b = torch.randn(64, device=DEV, generator=g) * 0.1
    ref = F.conv3d(x.double(), w.double(), b.double(), padding=3)
    y = ops.conv_fwd_raw(x, w, b, 1, 3)
    assert torch.equal(y, ops.conv_fwd_raw(x, w, b, 1, 3))
    lib().nc_set_split_terms(I(3))
    y32 = ops.conv_fwd_raw(x, w, b, 1, 3)
    lib().nc_set_split_terms(I(2))
    (m2, r2), (m32, r32) = _err(y, ref), _err(y32, ref)
    print(shape, 'two-term max %.2e rms %.2e | fp32 kernel max %.2e rms %.2e' % (m2, r2, m32, r32))
    assert not torch.equal(y, y32)  # (the two paths really are different kernels)
    assert r2 <= 1.3 * r32 and m2 <= 2.0 * m32
    yn = ops.conv_fwd_raw(x, w, None, 1, 3)  # deep_linear_gen's layer has no bias
    assert _err(yn, F.conv3d(x.double(), w.double(), padding=3))[1] <= 1.3 * r32


@pytest.mark.parametrize('shape', SHAPES)
def test_data_gradient_against_fp64_and_the_fp32_kernel(shape):
    N, D, H, W = shape
    g = torch.Generator(device=DEV).manual_seed(7 + sum(shape))
    dy = torch.randn(N, 64, D, H, W, device=DEV, generator=g)
    w = torch.randn(64, 1, 7, 7, 7, device=DEV, generator=g) * 0.05
    ref = F.conv_transpose3d(dy.double(), w.double(), padding=3)
    dx = ops.conv_dgrad_raw(dy, w, (N, 1, D, H, W), 1, 3)
    assert torch.equal(dx, ops.conv_dgrad_raw(dy, w, (N, 1, D, H, W), 1, 3))
    lib().nc_set_split_terms(I(3))
    dx32 = ops.conv_dgrad_raw(dy, w, (N, 1, D, H, W), 1, 3)
    lib().nc_set_split_terms(I(2))
    (m2, r2), (m32, r32) = _err(dx, ref), _err(dx32, ref)
    print(shape, 'two-term max %.2e rms %.2e | fp32 kernel max %.2e rms %.2e' % (m2, r2, m32, r32))
    assert not torch.equal(dx, dx32)
    assert r2 <= 1.3 * r32 and m2 <= 2.0 * m32


def test_adjointness_at_108():
    """<conv(x), dy> = <x, dgrad(dy)> at the headline size (BASELINE configs[1]): the two new kernels against each other in fp64 sums."""
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.rand(1, 1, 108, 108, 108, device=DEV, generator=g)
    dy = torch.randn(1, 64, 108, 108, 108, device=DEV, generator=g)
    w = torch.randn(64, 1, 7, 7, 7, device=DEV, generator=g) * 0.05
    y = ops.conv_fwd_raw(x, w, None, 1, 3).double()
    dx = ops.conv_dgrad_raw(dy, w, (1, 1, 108, 108, 108), 1, 3).double()
    a, b = float((y * dy.double()).sum()), float((x.double() * dx).sum())
    # (both sides are heavily cancelling sums of 8e7 / 1.3e6 products: the yardstick is the size of the summands, not of the sum)
    tol = 1e-6 * float(y.norm()) * float(dy.double().norm())
    assert abs(a - b) <= tol, (a, b, tol)


@pytest.mark.parametrize('scale', [1e-30, 1.0, 1e30])
def test_scales_are_exact_powers_of_two_away(scale):
    """One power of two per tensor: a tensor scaled by 2^k gives the result scaled by 2^k bit for bit (no overflow / underflow inside)."""
    g = torch.Generator(device=DEV).manual_seed(11)
    x = torch.rand(1, 1, 24, 24, 24, device=DEV, generator=g)
    w = torch.randn(64, 1, 7, 7, 7, device=DEV, generator=g) * 0.05
    k = float(2.0 ** round(np.log2(scale)))
    y1 = ops.conv_fwd_raw(x, w, None, 1, 3)
    yk = ops.conv_fwd_raw(x * k, w, None, 1, 3)
    assert torch.equal(yk, y1 * k)
    dy = torch.randn(1, 64, 24, 24, 24, device=DEV, generator=g)
    d1 = ops.conv_dgrad_raw(dy, w, (1, 1, 24, 24, 24), 1, 3)
    dk = ops.conv_dgrad_raw(dy * k, w, (1, 1, 24, 24, 24), 1, 3)
    assert torch.equal(dk, d1 * k)


def test_non_finite_input_reaches_only_the_outputs_it_touches():
    g = torch.Generator(device=DEV).manual_seed(13)
    x = torch.rand(1, 1, 24, 24, 24, device=DEV, generator=g)
    w = torch.randn(64, 1, 7, 7, 7, device=DEV, generator=g) * 0.05
    x[0, 0, 12, 12, 12] = float('inf')
    y = ops.conv_fwd_raw(x, w, None, 1, 3)
    bad = ~torch.isfinite(y)
    assert bool(bad[0, :, 9:16, 9:16, 9:16].all())
    mask = torch.ones_like(bad)
    mask[0, :, 9:16, 9:16, 9:16] = False
    assert not bool((bad & mask).any())


def test_layer_is_on_the_new_path_and_falls_back_when_told():
    """nc_conv_fwd_path says where Conv3d(1, 64, 7) runs: the two-term kernels by default, the fp32 matrix kernel under nc_set_split_terms(3)
    and nc_set_conv_split(0).  (The range guard can only COUNT a flagged input here -- there is no three-term pseudo-channel kernel -- so the
    models' reaction to a counted flag, nc_set_split_terms(3), is also this layer's fallback.)"""
    assert _path(1, 64) == 11
    lib().nc_set_split_terms(I(3))
    assert _path(1, 64) == 1
    lib().nc_set_split_terms(I(2))
    ops.set_conv_split(False)
    try:
        assert _path(1, 64) == 1
    finally:
        ops.set_conv_split(True)
