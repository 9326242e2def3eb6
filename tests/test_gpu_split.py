"""GPU parity of the split-operand fp32 convolutions (csrc/conv_split.hip) through the C ABI.

The kernels compute an fp32 convolution (models/networks.py:420-425, 460-469, 900-902) on the bf16 matrix cores: every fp32
operand is the exact sum of three bf16 terms and six of the nine term products are accumulated in fp32.  The checks:
* the three terms reproduce the fp32 value bit for bit;
* forward, data gradient and weight gradient against an fp64 reference: the error must not exceed the fp32 MFMA kernel's own error against the
  same reference by more than a small factor (measured: it is smaller), with an absolute bound of 4e-7 of the output rms;
* the library switch: with the kernels off the same calls run on the fp32 MFMA kernels and agree to 2e-6 of the output rms;
* at the 108^3 size of the headline step: the adjoint identity <conv(x), g> == <x, dgrad(g)>, agreement with the fp32 MFMA
  kernels, and -- against fp64 on a slab -- an error not above theirs.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = 'cuda'


@pytest.fixture(autouse=True)
def _restore_switch():
    # this file is about the THREE-term bf16 form (nc_set_split_terms(3)); the two-term fp16 form, the default since round 4, has
    # tests/test_gpu_h2.py
    from neuroclear_amd import ops
    from neuroclear_amd._lib import lib
    prev = ops.set_conv_split(True)
    terms = lib().nc_get_split_terms()
    lib().nc_set_split_terms(3)
    yield
    lib().nc_set_split_terms(terms)
    ops.set_conv_split(prev)


def _ws(N, C, D, H, W, K, ks):
    from neuroclear_amd import ops
    from neuroclear_amd._lib import I, lib
    return ops.workspace(lib().nc_conv_split_ws_bytes(I(N), I(C), I(D), I(H), I(W), I(K), I(ks)), DEV, 'ws_split_test')


def to_s3(x):
    from neuroclear_amd import ops
    from neuroclear_amd._lib import I, L_, check, lib
    N, C = x.shape[:2]
    S = x.numel() // (N * C)
    assert lib().nc_s3_bytes(I(N), I(C), L_(S)) == N * C * S * 6
    out = torch.empty(N * C * S * 6, dtype=torch.uint8, device=x.device)
    check(lib().nc_to_s3(ops._ptr(x), ops._ptr(out), I(N), I(C), L_(S), ops._stream()), 'nc_to_s3')
    return out


def fwd_split(x, w, b, xs=None):
    from neuroclear_amd import ops
    from neuroclear_amd._lib import I, Z, check, lib
    N, C, D, H, W = x.shape
    K, ks = w.shape[0], w.shape[2]
    y = torch.empty(N, K, D, H, W, device=x.device)
    ws = _ws(N, C, D, H, W, K, ks)
    check(lib().nc_conv_fwd_split(ops._ptr(x), ops._ptr(xs), ops._ptr(w), ops._ptr(b), ops._ptr(y), I(N), I(C), I(D), I(H), I(W), I(K),
                                  I(ks), ops._ptr(ws), Z(ws.numel()), ops._stream()), 'nc_conv_fwd_split')
    return y


def dgrad_split(dy, w, dys=None):
    from neuroclear_amd import ops
    from neuroclear_amd._lib import I, Z, check, lib
    N, K, D, H, W = dy.shape
    C, ks = w.shape[1], w.shape[2]
    dx = torch.empty(N, C, D, H, W, device=dy.device)
    ws = _ws(N, C, D, H, W, K, ks)
    check(lib().nc_conv_dgrad_split(ops._ptr(dy), ops._ptr(dys), ops._ptr(w), ops._ptr(dx), I(N), I(C), I(D), I(H), I(W), I(K),
                                    I(ks), ops._ptr(ws), Z(ws.numel()), ops._stream()), 'nc_conv_dgrad_split')
    return dx


def wgrad_split(x, dy, ks, xs=None, dys=None):
    from neuroclear_amd import ops
    from neuroclear_amd._lib import I, Z, check, lib
    N, C, D, H, W = x.shape
    K = dy.shape[1]
    dw = torch.empty(K, C, ks, ks, ks, device=x.device)
    ws = _ws(N, C, D, H, W, K, ks)
    check(lib().nc_conv_wgrad_split(ops._ptr(x), ops._ptr(xs), ops._ptr(dy), ops._ptr(dys), ops._ptr(dw), I(N), I(C), I(D), I(H), I(W),
                                    I(K), I(ks), ops._ptr(ws), Z(ws.numel()), ops._stream()), 'nc_conv_wgrad_split')
    return dw


def test_three_terms_are_the_fp32_value():
    torch.manual_seed(3)
    N, C, S = 2, 16, 5000
    x = torch.randn(N, C, S, device=DEV) * torch.pow(10.0, torch.empty(N, C, S, device=DEV).uniform_(-6, 6))
    x[0, 0, :4] = torch.tensor([0.0, 1.0, -1.0, 3.0e-30], device=DEV)
    raw = to_s3(x)
    t = raw.view(torch.bfloat16).view(N, C // 8, 3, S, 8).float()  # [n][block][term][voxel][channel in block]
    back = ((t[:, :, 0] + t[:, :, 1]) + t[:, :, 2]).permute(0, 1, 3, 2).reshape(N, C, S)
    assert torch.equal(back, x)
    # the terms shrink by 2^-8 each: |a1| <= 2^-8 |a0|, |a2| <= 2^-16 |a0|
    a0, a1, a2 = t[:, :, 0].abs(), t[:, :, 1].abs(), t[:, :, 2].abs()
    assert bool((a1 <= a0 * 2.0 ** -8).all()) and bool((a2 <= a0 * 2.0 ** -16).all())


CASES = [  # N, C, K, (D, H, W), kernel size
    (1, 64, 64, (20, 22, 27), 3),
    (2, 16, 128, (9, 17, 30), 3),
    (1, 128, 64, (12, 12, 12), 3),
    (1, 8, 64, (1, 5, 3), 3),        # one plane, three columns
    (1, 256, 256, (6, 27, 27), 3),
    (2, 64, 64, (3, 54, 54), 3),
    (1, 64, 64, (11, 14, 19), 5),    # G_B's 5^3 feature layer (networks.py:900)
    (2, 8, 64, (3, 30, 40), 5),
    (1, 64, 128, (7, 9, 150), 3),    # wide rows: the 256-position tile
]


@pytest.mark.parametrize('case', CASES, ids=[str(c) for c in CASES])
def test_split_conv_against_fp64(case):
    from neuroclear_amd import ops
    N, C, K, n, ks = case
    pd = ks // 2
    torch.manual_seed(11)
    x = torch.randn(N, C, *n, device=DEV)
    w = torch.randn(K, C, ks, ks, ks, device=DEV) * 0.02
    b = torch.randn(K, device=DEV)
    ref = F.conv3d(x.double().cpu(), w.double().cpu(), b.double().cpu(), padding=pd)
    sc = ref.pow(2).mean().sqrt().item()
    prev = ops.set_conv_split(False)
    y32 = ops.conv_fwd_raw(x, w, b, 1, pd)
    ops.set_conv_split(prev)
    ys = fwd_split(x, w, b)
    ys_pre = fwd_split(x, w, b, to_s3(x))
    assert torch.equal(ys, ys_pre)  # operand converted by the call or by the caller: the same kernel, the same result

    def err(y, r, s):
        e = y.double().cpu() - r
        return e.abs().max().item() / s, e.pow(2).mean().sqrt().item() / s
    m32, r32 = err(y32, ref, sc)
    ms, rs = err(ys, ref, sc)
    print(case, 'fwd fp32 max %.2e rms %.2e | split max %.2e rms %.2e' % (m32, r32, ms, rs))
    assert rs <= 1.3 * r32 + 2e-8 and rs < 4e-7, (rs, r32)
    assert ms <= 2.0 * m32 + 2e-7, (ms, m32)
    if C % 32 == 0:  # weight gradient
        dy = torch.randn(N, K, *n, device=DEV)
        refw = torch.nn.grad.conv3d_weight(x.double().cpu(), w.shape, dy.double().cpu(), padding=pd)
        sw = refw.pow(2).mean().sqrt().item()
        prev = ops.set_conv_split(False)
        w32 = ops.conv_wgrad_raw(x, dy, w.shape, 1, pd, False)[0]
        ops.set_conv_split(prev)
        wsp = wgrad_split(x, dy, ks)
        assert torch.equal(wsp, wgrad_split(x, dy, ks, to_s3(x), to_s3(dy)))
        assert torch.equal(wsp, ops.conv_wgrad_raw(x, dy, w.shape, 1, pd, False)[0])  # nc_conv_wgrad takes the same kernel
        m32, r32 = err(w32, refw, sw)
        ms, rs = err(wsp, refw, sw)
        print(case, 'wgrad fp32 max %.2e rms %.2e | split max %.2e rms %.2e' % (m32, r32, ms, rs))
        assert rs <= 1.3 * r32 + 2e-8 and rs < 6e-7, (rs, r32)
        assert ms <= 2.0 * m32 + 3e-7, (ms, m32)
    if C % 64 == 0:  # data gradient: the "output" side is C
        dy = torch.randn(N, K, *n, device=DEV)
        refd = torch.nn.grad.conv3d_input(x.shape, w.double().cpu(), dy.double().cpu(), padding=pd)
        sd = refd.pow(2).mean().sqrt().item()
        prev = ops.set_conv_split(False)
        d32 = ops.conv_dgrad_raw(dy, w, x.shape, 1, pd)
        ops.set_conv_split(prev)
        dsp = dgrad_split(dy, w)
        m32, r32 = err(d32, refd, sd)
        ms, rs = err(dsp, refd, sd)
        print(case, 'dgrad fp32 max %.2e rms %.2e | split max %.2e rms %.2e' % (m32, r32, ms, rs))
        assert rs <= 1.3 * r32 + 2e-8 and rs < 4e-7, (rs, r32)
        assert ms <= 2.0 * m32 + 2e-7, (ms, m32)


def test_library_switch_selects_the_kernels():
    from neuroclear_amd import ops
    from neuroclear_amd._lib import I, lib
    L = lib()
    torch.manual_seed(5)
    x = torch.randn(1, 64, 10, 20, 30, device=DEV)
    w = torch.randn(64, 64, 3, 3, 3, device=DEV) * 0.03
    assert L.nc_get_conv_split() == 1
    assert L.nc_conv_fwd_path(I(64), I(64), I(3), I(3), I(3), I(1), I(1)) == 9
    y_on = ops.conv_fwd_raw(x, w, None, 1, 1)
    assert torch.equal(y_on, fwd_split(x, w, None))           # nc_conv_fwd took the split kernels
    g_on = ops.conv_dgrad_raw(y_on, w, x.shape, 1, 1)
    assert torch.equal(g_on, dgrad_split(y_on, w))
    ops.set_conv_split(False)
    assert L.nc_get_conv_split() == 0
    assert L.nc_conv_fwd_path(I(64), I(64), I(3), I(3), I(3), I(1), I(1)) == 1
    y_off = ops.conv_fwd_raw(x, w, None, 1, 1)
    g_off = ops.conv_dgrad_raw(y_on, w, x.shape, 1, 1)
    ops.set_conv_split(True)
    assert not torch.equal(y_on, y_off)                       # different kernels, different summation order ...
    sc = y_off.pow(2).mean().sqrt().item()
    assert (y_on - y_off).abs().max().item() < 2e-6 * sc      # ... the same fp32 result
    sg = g_off.pow(2).mean().sqrt().item()
    assert (g_on - g_off).abs().max().item() < 2e-6 * sg
    # shapes the kernels do not cover are refused by the explicit entry points and stay on the other kernels in nc_conv_fwd
    assert L.nc_conv_split_supported(I(0), I(1), I(4), I(8), I(8), I(8), I(64), I(3), I(3), I(3), I(1), I(1)) == 0   # C % 8
    assert L.nc_conv_split_supported(I(0), I(1), I(64), I(8), I(8), I(8), I(32), I(3), I(3), I(3), I(1), I(1)) == 0  # K % 64
    assert L.nc_conv_split_supported(I(0), I(1), I(64), I(8), I(8), I(8), I(64), I(3), I(3), I(3), I(2), I(1)) == 0  # stride
    x4 = torch.randn(1, 4, 8, 8, 8, device=DEV)
    w4 = torch.randn(64, 4, 3, 3, 3, device=DEV)
    with pytest.raises(Exception):
        fwd_split(x4, w4, None)


@pytest.mark.parametrize('split', [True, False])
def test_layer_backward_in_one_call(split):
    """nc_conv_bwd == nc_conv_dgrad + nc_conv_wgrad, bit for bit (on the split kernels dY is converted once for both)."""
    from neuroclear_amd import ops
    from neuroclear_amd._lib import I, Z, check, lib
    ops.set_conv_split(split)
    torch.manual_seed(9)
    N, C, K, n = 2, 64, 128, (6, 20, 33)
    x = torch.randn(N, C, *n, device=DEV)
    dy = torch.randn(N, K, *n, device=DEV)
    w = torch.randn(K, C, 3, 3, 3, device=DEV) * 0.05
    dx_ref = ops.conv_dgrad_raw(dy, w, x.shape, 1, 1)
    dw_ref, db_ref = ops.conv_wgrad_raw(x, dy, w.shape, 1, 1, True)
    L = lib()
    ws = ops.workspace(L.nc_conv_ws_bytes(I(N), I(C), I(n[0]), I(n[1]), I(n[2]), I(K), I(3), I(3), I(3), I(1), I(1)), DEV, 'ws_bwd_test')
    for with_dx in (True, False):
        dx, dw, db = torch.full_like(x, 7.0), torch.empty_like(w), torch.empty(K, device=DEV)
        check(L.nc_conv_bwd(ops._ptr(x), ops._ptr(dy), ops._ptr(w), ops._ptr(dx if with_dx else None), ops._ptr(dw), ops._ptr(db), I(N),
                            I(C), I(n[0]), I(n[1]), I(n[2]), I(K), I(3), I(3), I(3), I(1), I(1), ops._ptr(ws), Z(ws.numel()),
                            ops._stream()), 'nc_conv_bwd')
        assert torch.equal(dw, dw_ref) and torch.equal(db, db_ref)
        assert torch.equal(dx, dx_ref) if with_dx else bool((dx == 7.0).all())


@pytest.mark.parametrize('ks,C,K', [(3, 64, 64), (3, 128, 64), (5, 64, 64)])
def test_full_size_layer_adjoint_and_fp32_agreement(ks, C, K):
    """The layer shapes of the headline step at 108^3 (BASELINE configs[1])."""
    from neuroclear_amd import ops
    torch.manual_seed(2)
    E, pd = 108, ks // 2
    x = torch.randn(1, C, E, E, E, device=DEV)
    w = torch.randn(K, C, ks, ks, ks, device=DEV) * (1.0 / np.sqrt(C * ks ** 3))
    g = torch.randn(1, K, E, E, E, device=DEV)
    y = ops.conv_fwd_raw(x, w, None, 1, pd)
    dx = ops.conv_dgrad_raw(g, w, x.shape, 1, pd)
    lhs = torch.dot(y.double().flatten(), g.double().flatten()).item()
    rhs = torch.dot(x.double().flatten(), dx.double().flatten()).item()
    nrm = (y.double().norm() * g.double().norm()).item()
    assert abs(lhs - rhs) < 1e-6 * nrm, (lhs, rhs, nrm)
    ops.set_conv_split(False)
    y32 = ops.conv_fwd_raw(x, w, None, 1, pd)
    dx32 = ops.conv_dgrad_raw(g, w, x.shape, 1, pd)
    ops.set_conv_split(True)
    # the two kernel families agree to the fp32 MFMA kernel's own accuracy at this reduction length (C * ks^3 sequential
    # fp32 roundings: ~1e-5 of the rms at the worst voxel) ...
    assert (y - y32).abs().max().item() < 3e-5 * y32.pow(2).mean().sqrt().item()
    assert (dx - dx32).abs().max().item() < 3e-5 * dx32.pow(2).mean().sqrt().item()
    # weight gradient (a sum over 1.26 M voxels per weight): <dW, w> == <conv(x) , g> and agreement with the fp32 kernels
    dw = ops.conv_wgrad_raw(x, g, w.shape, 1, pd, False)[0]
    assert abs(torch.dot(dw.double().flatten(), w.double().flatten()).item() - lhs) < 1e-6 * nrm
    ops.set_conv_split(False)
    dw32 = ops.conv_wgrad_raw(x, g, w.shape, 1, pd, False)[0]
    ops.set_conv_split(True)
    assert (dw - dw32).abs().max().item() < 1e-4 * dw32.pow(2).mean().sqrt().item()
    # ... and against fp64 on the first 4 x 4 channel pairs the split kernel is not the worse one
    refw = torch.nn.grad.conv3d_weight(x[:, :4].double().cpu(), (4, 4, ks, ks, ks), g[:, :4].double().cpu(), padding=pd)
    sw = refw.pow(2).mean().sqrt().item()
    w_s = (dw[:4, :4].double().cpu() - refw).abs().max().item() / sw
    w_f = (dw32[:4, :4].double().cpu() - refw).abs().max().item() / sw
    print('108^3 ks=%d %d->%d: weight gradient max error vs fp64  split %.2e  fp32 MFMA %.2e' % (ks, C, K, w_s, w_f))
    assert w_s < 2e-5 and w_s <= 1.5 * w_f + 1e-6
    # ... and against fp64 on a slab of 6 planes the split kernel is the closer one
    z0, z1 = 40, 46
    ref = F.conv3d(x[:, :, z0 - pd:z1 + pd].double().cpu(), w.double().cpu(), padding=pd)[:, :, pd:pd + z1 - z0]
    sc = ref.pow(2).mean().sqrt().item()
    e_s = (y[:, :, z0:z1].double().cpu() - ref).abs().max().item() / sc
    e_f = (y32[:, :, z0:z1].double().cpu() - ref).abs().max().item() / sc
    print('108^3 ks=%d %d->%d: max error vs fp64  split %.2e  fp32 MFMA %.2e' % (ks, C, K, e_s, e_f))
    assert e_s < 3e-6 and e_s <= e_f


def test_whole_network_inference_with_and_without_the_split_kernels():
    """unet_deconv forward of one 60^3 cube (the TestModel path): both kernel families give the same image."""
    from neuroclear_amd import ops
    from neuroclear_amd.models import networks
    torch.manual_seed(7)
    net = networks.define_G(1, 1, 64, 'unet_deconv', 'instance', False, 'normal', 0.02, [0])
    x = torch.rand(1, 1, 60, 60, 60, device=DEV)
    with torch.no_grad():
        y_on = net(x).clone()
        ops.set_conv_split(False)
        y_off = net(x).clone()
        ops.set_conv_split(True)
    assert (y_on - y_off).abs().max().item() < 2e-6
    # the normalisation pass writing the three-term form itself or a separate conversion pass: the same bits.  (The inference forward
    # defaults to the TWO-term form since round 4 -- tests/test_gpu_h2.py; this is about the three-term one.)
    from neuroclear_amd._lib import I, lib
    terms = lib().nc_get_split_terms()
    lib().nc_set_split_terms(I(3))
    try:
        with torch.no_grad():
            y_on3 = net(x).clone()
        lib().nc_set_s3_fusion(I(0))
        with torch.no_grad():
            y_sep = net(x).clone()
    finally:
        lib().nc_set_s3_fusion(I(1))
        lib().nc_set_split_terms(I(terms))
    assert torch.equal(y_on3, y_sep)
    assert (y_on3 - y_on).abs().max().item() < 2e-6


@pytest.mark.parametrize('kind', ['unet_deconv', 'deep_linear_gen'])
def test_switch_toggled_between_forward_and_backward(kind):
    """The whole-network training calls keep the forward's three-term copies of the layer inputs for the weight gradient; which
    copies a `saved` buffer holds is host state.  A backward after the switch was flipped must neither use stale copies (forward with
    the kernels off, backward with them on) nor need them (forward on, backward off): in all four combinations the gradients are the
    fp32 gradients to rounding."""
    from neuroclear_amd import ops
    from neuroclear_amd.models import networks
    torch.manual_seed(21)
    net = networks.define_G(1, 1, 64, kind, 'instance', False, 'normal', 0.02, [0])
    x = torch.rand(1, 1, 24, 24, 24, device=DEV)
    g = torch.randn(1, 1, 24, 24, 24, device=DEV)
    grads = {}
    for fwd_on in (True, False):
        for bwd_on in (True, False):
            for p_ in net.parameters():
                p_.grad = None
            ops.set_conv_split(fwd_on)
            y = net(x)
            ops.set_conv_split(bwd_on)
            y.backward(g)
            grads[(fwd_on, bwd_on)] = torch.cat([p_.grad.reshape(-1).clone() for p_ in net.parameters() if p_.dim() > 1])
    ops.set_conv_split(True)
    ref = grads[(False, False)].double()
    for k, v in grads.items():
        assert torch.isfinite(v).all()
        rel = ((v.double() - ref).norm() / ref.norm()).item()
        print(kind, k, 'relative L2 difference of the weight gradients to the all-fp32-MFMA run: %.2e' % rel)
        # (two fp32 evaluations of this gradient differ by ~1e-4: ten InstanceNorm backward passes amplify rounding -- the golden
        #  tests allow 2e-2; a stale or missing operand copy would be an O(1) difference)
        assert rel < 5e-3, (k, rel)


# ---- adversarial inputs (every other comparison above uses randn operands of one scale) --------------------------------------
def _fp64_err(y, ref):
    e = y.double().cpu() - ref
    sc = ref.pow(2).mean().sqrt().item()
    return e.abs().max().item() / sc, e.pow(2).mean().sqrt().item() / sc


def _both_kernels(x, w, b, pd):
    from neuroclear_amd import ops
    prev = ops.set_conv_split(False)
    y32 = ops.conv_fwd_raw(x, w, b, 1, pd)
    ops.set_conv_split(prev)
    return y32, fwd_split(x, w, b)


@pytest.mark.parametrize('C,K,n,ks', [(64, 64, (12, 20, 27), 3), (16, 64, (6, 17, 30), 3), (64, 64, (8, 14, 19), 5)])
def test_split_conv_large_mean_input(C, K, n, ks):
    """Inputs 1000 +- 1 (a reduction that mixes a large common part with small differences: what an un-normalised bright volume
    or a big-mean activation looks like): forward error against fp64 not above the fp32 MFMA kernel's; the weight gradient against a
    zero-mean dY is a CANCELLING sum (1000 * sum(dY) + sum(dY * delta)) and is judged the same way."""
    torch.manual_seed(21)
    pd = ks // 2
    x = 1000.0 + torch.randn(1, C, *n, device=DEV)
    w = torch.randn(K, C, ks, ks, ks, device=DEV) * 0.02
    b = torch.randn(K, device=DEV)
    ref = F.conv3d(x.double().cpu(), w.double().cpu(), b.double().cpu(), padding=pd)
    y32, ys = _both_kernels(x, w, b, pd)
    (m32, r32), (ms, rs) = _fp64_err(y32, ref), _fp64_err(ys, ref)
    print('big mean fwd fp32 max %.2e rms %.2e | split max %.2e rms %.2e' % (m32, r32, ms, rs))
    assert rs <= 1.3 * r32 + 2e-8 and ms <= 2.0 * m32 + 2e-7, (rs, r32, ms, m32)
    if C % 32 == 0:
        from neuroclear_amd import ops
        dy = torch.randn(1, K, *n, device=DEV)
        refw = torch.nn.grad.conv3d_weight(x.double().cpu(), w.shape, dy.double().cpu(), padding=pd)
        prev = ops.set_conv_split(False)
        w32 = ops.conv_wgrad_raw(x, dy, w.shape, 1, pd, False)[0]
        ops.set_conv_split(prev)
        wsp = wgrad_split(x, dy, ks)
        (m32, r32), (ms, rs) = _fp64_err(w32, refw), _fp64_err(wsp, refw)
        print('big mean wgrad fp32 max %.2e rms %.2e | split max %.2e rms %.2e' % (m32, r32, ms, rs))
        assert rs <= 1.3 * r32 + 2e-8 and ms <= 2.0 * m32 + 3e-7, (rs, r32, ms, m32)


def test_split_conv_zero_channels_and_zero_bricks():
    """Channels that are exactly zero over the whole volume, whole planes of zeros, and an all-zero input: zeros contribute exact
    zeros (every term of 0 is 0), so the all-zero input gives the bias bit for bit and the rest stays within the usual bounds."""
    torch.manual_seed(22)
    x = torch.randn(1, 64, 10, 18, 27, device=DEV)
    x[:, ::2] = 0.0          # every other channel: half of each 8-channel unit is zero
    x[:, 8:24] = 0.0         # two whole 8-channel blocks
    x[:, :, 3:6] = 0.0       # three whole planes: bricks of zeros in the middle of the volume
    w = torch.randn(64, 64, 3, 3, 3, device=DEV) * 0.02
    b = torch.randn(64, device=DEV)
    ref = F.conv3d(x.double().cpu(), w.double().cpu(), b.double().cpu(), padding=1)
    y32, ys = _both_kernels(x, w, b, 1)
    (m32, r32), (ms, rs) = _fp64_err(y32, ref), _fp64_err(ys, ref)
    assert rs <= 1.3 * r32 + 2e-8 and ms <= 2.0 * m32 + 2e-7, (rs, r32, ms, m32)
    assert torch.equal(ys[:, :, 4], b.view(1, 64, 1, 1).expand(1, 64, 18, 27))  # plane 4 sees only zero planes: exactly the bias
    z = fwd_split(torch.zeros_like(x), w, b)
    assert torch.equal(z, b.view(1, 64, 1, 1, 1).expand_as(z))
    assert torch.equal(dgrad_split(torch.zeros(1, 64, 10, 18, 27, device=DEV), w), torch.zeros_like(x))


@pytest.mark.parametrize('scale,bound', [(1e-30, 4e-7), (1e-34, 2e-4), (1e-36, 5e-2)])
def test_split_conv_tiny_magnitudes(scale, bound):
    """The three terms of a value v are ~v, ~2^-8 v, ~2^-16 v: below |v| ~ 2^-110 the later terms fall under bf16's normal range
    (2^-126) and the matrix core treats them as zero, so the result degrades from 24 to 16 to 8 significant bits -- it does not blow
    up and stays finite.  (fp32 arithmetic on such values is itself at the edge of its range: |w x| ~ 1e-38.)  The bounds are the
    measured behaviour with a margin; at 1e-30 (all terms normal) the ordinary bound holds."""
    torch.manual_seed(23)
    x = torch.randn(1, 64, 8, 14, 27, device=DEV) * scale
    w = torch.randn(64, 64, 3, 3, 3, device=DEV) * 0.02
    ref = F.conv3d(x.double().cpu(), w.double().cpu(), None, padding=1)
    ys = fwd_split(x, w, None)
    ms, rs = _fp64_err(ys, ref)
    print('scale %g: split max %.2e rms %.2e' % (scale, ms, rs))
    assert bool(torch.isfinite(ys).all()) and rs < bound, (scale, rs)


def test_split_conv_values_beyond_the_bf16_range_and_non_finite_inputs():
    """(i) A finite fp32 value above the largest finite bf16 (3.3895e38 < |v| <= 3.4028e38) still splits exactly (its first term is the
    largest finite bf16, not infinity) and convolves to a finite, accurate result.  (ii) THE PROPAGATION RULE for non-finite inputs: an
    inf or a NaN input element makes every output it touches (its 3^3 neighbourhood, all output channels) non-finite -- on the
    split-operand kernels always NaN (inf - inf inside the split; the fp32 MFMA kernels give inf or NaN there) -- and leaves every other
    output bit-identical to the result without it."""
    torch.manual_seed(24)
    big = torch.tensor([3.4e38, -3.4028234e38, 3.39e38, 1.0], device=DEV).repeat(2, 8, 1)[:, :, :4].contiguous()  # [2][8][4]
    t = to_s3(big).view(torch.bfloat16).view(2, 1, 3, 4, 8).double()  # (summed in fp64: for -FLT_MAX the first two terms alone are -2^128)
    assert torch.equal(((t[:, :, 0] + t[:, :, 1]) + t[:, :, 2]).permute(0, 1, 3, 2).reshape(2, 8, 4), big.double())
    assert bool(torch.isfinite(t).all())
    x = torch.randn(1, 64, 8, 12, 20, device=DEV)
    w = torch.randn(64, 64, 3, 3, 3, device=DEV) * 1e-3
    clean = fwd_split(x, w, None)
    xb = x.clone()
    xb[0, 5, 4, 6, 7] = 3.4e38
    yb = fwd_split(xb, w, None)
    refb = F.conv3d(xb.double().cpu(), w.double().cpu(), None, padding=1)
    assert bool(torch.isfinite(yb).all())
    e = (yb.double().cpu() - refb).abs()
    assert float((e / refb.abs().clamp_min(1e30)).max()) < 1e-6  # relative to the huge contributions where they dominate
    for bad in (float('inf'), float('-inf'), float('nan')):
        xn = x.clone()
        xn[0, 5, 4, 6, 7] = bad
        for y in _both_kernels(xn, w, None, 1):
            touched = torch.zeros_like(y, dtype=torch.bool)
            touched[:, :, 3:6, 5:8, 6:9] = True
            assert not bool(torch.isfinite(y[touched]).any()), bad
        ys = fwd_split(xn, w, None)
        assert bool(torch.isnan(ys[touched]).all()), bad
        assert torch.equal(ys[~touched], clean[~touched]), bad
