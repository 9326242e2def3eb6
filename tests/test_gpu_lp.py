"""GPU parity of the 16-bit (bf16 / fp16) MFMA convolutions (BASELINE.json configs[3]) through the C ABI.

Two kinds of checks:
* exactness of the kernels: against torch fp32 convolutions of the SAME 16-bit-rounded operands the difference is only
  the fp32 summation order (tolerance 5e-5 of the largest magnitude -- the fp32 kernels' own tolerance is 2e-5; the sums
  here are over up to 4.6e4 voxels x 27 taps per output);
* closeness to the fp32 path (the reference's arithmetic): generator output after the sigmoid within 3e-3 (bf16) /
  5e-4 (fp16) absolute, Apollo losses of the first step within 1e-2 / 2e-3 relative -- the "looser stated tolerance" of
  SURVEY 8(d) config 4.  InstanceNorm statistics, losses and Adam are fp32 in both.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rnd(t, prec):
    return t.to(torch.bfloat16 if prec == 'bf16' else torch.float16).float()


CASES = [  # N, C, K, (D, H, W), kernel size
    (1, 32, 64, (8, 8, 8), 3),
    (2, 32, 64, (5, 9, 13), 3),
    (1, 64, 128, (20, 20, 20), 3),
    (1, 128, 64, (7, 30, 37), 3),
    (1, 64, 64, (36, 36, 36), 3),
    (1, 256, 128, (6, 27, 27), 3),
    (2, 64, 64, (3, 54, 54), 3),
    (1, 64, 64, (20, 20, 20), 5),   # G_B's feature block (networks.py:900)
    (2, 32, 64, (6, 11, 23), 5),
    (1, 64, 64, (4, 40, 52), 5),
]


@pytest.fixture(autouse=True)
def _restore_precision():
    from neuroclear_amd import ops
    yield
    ops.set_conv_precision('fp32')


@pytest.mark.parametrize('prec', ['bf16', 'fp16'])
@pytest.mark.parametrize('case', CASES)
def test_lp_conv_matches_rounded_operands(case, prec):
    from neuroclear_amd import ops
    N, C, K, (D, H, W), ks = case
    pad, k3 = ks // 2, (ks, ks, ks)
    g = torch.Generator(device='cuda').manual_seed(3)
    x = torch.randn(N, C, D, H, W, device='cuda', generator=g)
    w = torch.randn(K, C, ks, ks, ks, device='cuda', generator=g) / (C * ks ** 3) ** 0.5
    b = torch.randn(K, device='cuda', generator=g)
    dy = torch.randn(N, K, D, H, W, device='cuda', generator=g)
    xr, wr, dyr = _rnd(x, prec), _rnd(w, prec), _rnd(dy, prec)
    ops.set_conv_precision(prec)
    assert ops._lp(0, (N, C, D, H, W), K, k3, 1, pad), 'forward must take the 16-bit kernel'
    y = ops.conv_fwd_raw(x, w, b, 1, pad)
    dx = ops.conv_dgrad_raw(dy, w, x.shape, 1, pad) if ops._lp(1, (N, C, D, H, W), K, k3, 1, pad) else None
    dw, db = ops.conv_wgrad_raw(x, dy, w.shape, 1, pad, True)
    assert ops._lp(2, (N, C, D, H, W), K, k3, 1, pad)
    if prec == 'fp16':  # backward operands are bf16 under --precision fp16 (ops._lp)
        dyr, wr_b, xr_b = _rnd(dy, 'bf16'), _rnd(w, 'bf16'), _rnd(x, 'bf16')
    else:
        wr_b, xr_b = wr, xr
    ops.set_conv_precision('fp32')

    def close(a, ref, tol=5e-5):
        assert (a - ref).abs().max().item() <= tol * ref.abs().max().item()

    close(y, F.conv3d(xr, wr, b, padding=pad))
    if dx is not None:
        close(dx, F.conv_transpose3d(dyr, wr_b, padding=pad))
    wz = torch.zeros_like(w, requires_grad=True)
    F.conv3d(xr_b, wz, None, padding=pad).backward(dyr)
    close(dw, wz.grad, 1e-4)
    close(db, dy.sum((0, 2, 3, 4)), 1e-5)  # bias gradient: fp32 sum of the fp32 dy


def test_lp_preconverted_operands_give_identical_results():
    """xh / dyh (nc_to_c8 once, reused by forward + weight gradient / data + weight gradient) vs per-call conversion."""
    from neuroclear_amd import ops
    g = torch.Generator(device='cuda').manual_seed(9)
    x = torch.randn(2, 64, 6, 20, 28, device='cuda', generator=g)
    w = torch.randn(64, 64, 3, 3, 3, device='cuda', generator=g) * 0.02
    dy = torch.randn(2, 64, 6, 20, 28, device='cuda', generator=g)
    ops.set_conv_precision('bf16')
    xh, dyh = ops.to_c8(x, 2), ops.to_c8(dy, 2)
    assert xh.numel() == x.numel() * 2
    assert torch.equal(ops.conv_fwd_raw(x, w, None, 1, 1), ops.conv_fwd_raw(x, w, None, 1, 1, xh=xh))
    assert torch.equal(ops.conv_dgrad_raw(dy, w, x.shape, 1, 1), ops.conv_dgrad_raw(dy, w, x.shape, 1, 1, dyh=dyh))
    a, _ = ops.conv_wgrad_raw(x, dy, w.shape, 1, 1, False)
    b, _ = ops.conv_wgrad_raw(None, dy, w.shape, 1, 1, False, xh=xh, dyh=dyh, x_shape=x.shape)
    assert torch.equal(a, b)
    # and through autograd (which keeps the C8 copy of x instead of x)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ops.conv(xr, wr, None, 1, 1).backward(dy)
    assert torch.equal(wr.grad, a) and torch.equal(xr.grad, ops.conv_dgrad_raw(dy, w, x.shape, 1, 1))


def test_lp_unsupported_shape_is_refused():
    """The 16-bit entry points never fall back: a shape they do not cover is an error at the C ABI."""
    from neuroclear_amd import _lib, ops
    x = torch.randn(1, 1, 8, 8, 8, device='cuda')
    w = torch.randn(64, 1, 3, 3, 3, device='cuda')
    y = torch.empty(1, 64, 8, 8, 8, device='cuda')
    ws = torch.empty(1 << 20, dtype=torch.uint8, device='cuda')
    L = _lib.lib()
    rc = L.nc_conv_fwd_lp(ops._ptr(x), None, ops._ptr(w), None, ops._ptr(y), 1, 1, 8, 8, 8, 64, 3, 3, 3, 1, 1, 2, ops._ptr(ws),
                          _lib.Z(ws.numel()), None)
    assert rc == -1 and b'not covered' in L.nc_last_error()
    rc = L.nc_conv_fwd_lp(ops._ptr(x), None, ops._ptr(w), None, ops._ptr(y), 1, 1, 8, 8, 8, 64, 3, 3, 3, 1, 1, 7, ops._ptr(ws),
                          _lib.Z(ws.numel()), None)
    assert rc == -4


@pytest.mark.parametrize('prec,tol', [('bf16', 1.5e-2), ('fp16', 2e-3)])
def test_lp_unet_close_to_fp32(prec, tol):
    """Measured (tools/lp_err.py, 48^3): output max |err| 7.6e-3 bf16 / 1.0e-3 fp16, mean 1.0e-3 / 1.3e-4.  Weight gradients
    of this randomly weighted net under a random cotangent are chaotic (ReLU / max-pool flips): a 1e-3 relative input
    perturbation in pure fp32 moves them by 6-17 %, so the 16-bit path is held to 2.5 x that noise floor + 2 %."""
    from neuroclear_amd import ops
    from neuroclear_amd.models import networks
    from neuroclear_amd.util import seed as S
    net = networks.define_G(1, 1, 64, 'unet_deconv', 'instance', False, 'kaiming', 0.02, [0])
    net.load_state_dict(S.state_dict_from_seed(S.unet_deconv_spec(), 11, device='cuda'))
    gen = torch.Generator(device='cuda').manual_seed(5)
    x = torch.rand(1, 1, 32, 32, 32, device='cuda', generator=gen)
    r = torch.randn(1, 1, 32, 32, 32, device='cuda', generator=gen)
    noise = 1 + 1e-3 * torch.randn(1, 1, 32, 32, 32, device='cuda', generator=gen)

    def run(p, xi):
        ops.set_conv_precision(p)
        for q in net.parameters():
            q.grad = None
        y = net(xi)
        (y * r).mean().backward()
        ops.set_conv_precision('fp32')
        return y.detach().clone(), {n: q.grad.clone() for n, q in net.named_parameters()}

    y32, g32 = run('fp32', x)
    _, gp = run('fp32', x * noise)
    y16, g16 = run(prec, x)
    assert (y16 - y32).abs().max().item() <= tol
    assert (y16 - y32).abs().mean().item() <= tol / 5
    for n, g in g32.items():
        if g.dim() == 5 and g.shape[1] >= 32 and g.shape[2] == 3:
            floor = (gp[n] - g).norm().item() / g.norm().item()
            rel = (g16[n] - g).norm().item() / g.norm().item()
            assert rel <= 2.5 * floor + 0.02, (n, rel, floor)


def test_lp_apollo_step_close_to_fp32():
    """One Apollo step at 36^3 in bf16 vs fp32 from the same seeds: the 11 losses of the first step within 2e-2 relative
    (losses are means over whole volumes / planes, so the 1e-3 mean output error averages out further)."""
    from argparse import Namespace
    import contextlib
    import io
    from neuroclear_amd.models import create_model
    from neuroclear_amd.util import seed as S

    def one(prec):
        o = Namespace(gpu_ids=[0], isTrain=True, image_dimension=3, checkpoints_dir='/tmp/nc_ckpt', name='lp',
                      preprocess='none', gan_mode='lsgan', randomize_projection_depth=True, projection_depth=10,
                      min_projection_depth=2, lambda_plane=[1, 1, 1], lambda_A=5.0, input_nc=1, output_nc=1, ngf=64,
                      ndf=64, netG='unet_deconv', netG_B='deep_linear_gen', netD='basic', n_layers_D=3,
                      norm='instance', no_dropout=True, init_type='kaiming', init_gain=0.02, lr=1e-4, beta1=0.1,
                      direction='AtoB', model='axial_to_lateral_gan_apollo', precision=prec)
        torch.manual_seed(7)
        np.random.seed(7)
        with contextlib.redirect_stdout(io.StringIO()):
            m = create_model(o)
        v = S.random_volume(3, 36)
        real = torch.from_numpy((v.astype(np.float64) / 65535.0).astype(np.float32))[None, None].cuda()
        m.set_input({'A': real, 'A_paths': 'synthetic'})
        m.optimize_parameters()
        return m.get_current_losses()

    l32, l16 = one('fp32'), one('bf16')
    assert set(l32) == set(l16) and len(l32) == 11
    for k in l32:
        assert abs(l16[k] - l32[k]) <= 2e-2 * max(abs(l32[k]), 1e-3), (k, l32[k], l16[k])


@pytest.mark.parametrize('prec', ['bf16', 'fp16'])
def test_lp_links_hand_over_identical_operands(prec):
    """The InstanceNorm kernels that emit the C8 copy of their result (forward: input of the next convolution; backward:
    dy of the previous one) round the same fp32 values the separate conversion pass would: a block run with the links
    gives bit-identical activations and weight / data gradients to the same block with per-call conversions."""
    from neuroclear_amd import ops
    from neuroclear_amd.models import networks
    blk = networks.triple_conv(64, 64, 3, 1, 1, networks.get_norm_layer('instance'), 3).cuda()
    gen = torch.Generator(device='cuda').manual_seed(12)
    with torch.no_grad():
        for p_ in blk.parameters():
            p_.copy_(torch.randn(p_.shape, device='cuda', generator=gen) * (0.05 if p_.dim() > 1 else 0.1))
    x = torch.randn(2, 64, 9, 20, 22, device='cuda', generator=gen)
    r = torch.randn(2, 64, 9, 20, 22, device='cuda', generator=gen)

    def run(linked):
        old = networks._BIAS_LINK
        networks._BIAS_LINK = linked
        try:
            ops.set_conv_precision(prec)
            for p_ in blk.parameters():
                p_.grad = None
            xi = x.clone().requires_grad_(True)
            y = blk(xi)
            (y * r).sum().backward()
            return y.detach().clone(), xi.grad.clone(), {n: p_.grad.clone() for n, p_ in blk.named_parameters()}
        finally:
            networks._BIAS_LINK = old
            ops.set_conv_precision('fp32')

    y0, gx0, g0 = run(False)
    y1, gx1, g1 = run(True)
    assert torch.equal(y0, y1) and torch.equal(gx0, gx1)
    for n in g0:
        if g0[n].dim() > 1:
            assert torch.equal(g0[n], g1[n]), n
        else:  # bias gradients in front of InstanceNorm: rounding noise on both sides
            scale = max(v.abs().max().item() for k, v in g0.items() if v.dim() > 1)
            assert g0[n].abs().max().item() <= 1e-4 * scale and g1[n].abs().max().item() <= 1e-4 * scale


def test_lp_shape_sweep():
    """36 random small shapes (odd sizes, single rows / planes, batches, both kernel sizes, every supported channel
    multiple): forward, data and weight gradient of the 16-bit kernels against torch on the same rounded operands."""
    from neuroclear_amd import ops
    rng = np.random.default_rng(2024)
    done = 0
    while done < 36:
        ks = int(rng.choice([3, 3, 5]))
        C = int(rng.choice([16, 32, 48, 64, 96])) if ks == 3 else int(rng.choice([8, 32, 64]))
        K = int(rng.choice([64, 128]))
        N = int(rng.integers(1, 4))
        D, H, W = (int(rng.integers(1, 12)), int(rng.integers(1, 24)), int(rng.integers(1, 40)))
        dims = (N, C, D, H, W)
        k3, pad = (ks, ks, ks), ks // 2
        ops.set_conv_precision('bf16')
        sup = [ops._lp(w_, dims, K, k3, 1, pad) for w_ in (0, 1, 2)]
        ops.set_conv_precision('fp32')
        if not sup[0]:
            continue
        g = torch.Generator(device='cuda').manual_seed(done)
        x = torch.randn(dims, device='cuda', generator=g)
        w = torch.randn(K, C, ks, ks, ks, device='cuda', generator=g) / (C * ks ** 3) ** 0.5
        dy = torch.randn(N, K, D, H, W, device='cuda', generator=g)
        xr, wr, dyr = _rnd(x, 'bf16'), _rnd(w, 'bf16'), _rnd(dy, 'bf16')
        ops.set_conv_precision('bf16')
        y = ops.conv_fwd_raw(x, w, None, 1, pad)
        dx = ops.conv_dgrad_raw(dy, w, x.shape, 1, pad) if sup[1] else None
        dw = ops.conv_wgrad_raw(x, dy, w.shape, 1, pad, False)[0] if sup[2] else None
        ops.set_conv_precision('fp32')

        def close(a, ref, tol, what):
            err = (a - ref).abs().max().item()
            assert err <= tol * max(ref.abs().max().item(), 1e-6), (what, dims, K, ks, err)

        close(y, F.conv3d(xr, wr, None, padding=pad), 5e-5, 'fwd')
        if dx is not None:
            close(dx, F.conv_transpose3d(dyr, wr, padding=pad), 5e-5, 'dgrad')
        if dw is not None:
            wz = torch.zeros_like(w, requires_grad=True)
            F.conv3d(xr, wz, None, padding=pad).backward(dyr)
            close(dw, wz.grad, 1e-4, 'wgrad')
        done += 1
