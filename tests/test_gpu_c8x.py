"""GPU parity of the tap-stream 16-bit convolution kernel k_conv_c8x (csrc/conv_c8x.hip; BASELINE.json configs[3]: the nn.Conv3d
layers of models/networks.py:420-425, 460-469, 900-902 with 16-bit operands) through the C ABI, with the kernel FORCED on
(nc_set_c8x_mode(2): by default small launches stay on k_conv_h, and the sizes an oracle finishes in seconds are small):

* against torch fp32 convolutions of the SAME 16-bit-rounded operands: only the fp32 summation order differs (5e-5 of the largest
  magnitude, the tolerance of tests/test_gpu_lp.py) -- forward and data gradient, bf16 and fp16, 3^3 and 5^3, ragged planes
  (widths that are no multiple of anything, planes smaller and larger than one 512-position tile, z extents of 1 and 2: every
  brick of a tile then has a plane outside the volume), batches, every supported channel count;
* C8 output bit-identical to nc_to_c8 of the fp32 output of the same kernel, written into a channel range of a wider buffer without
  touching the rest;
* k_conv_c8x against k_conv_h (mode 0) on the same call: same rounded operands, so the two agree to the summation order;
* the whole-network 16-bit calls and the tap-diffused weight rounding give the same answers on either kernel."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = 'cuda'
BF, FP = 2, 1
P = ctypes.c_void_p


def L():
    from neuroclear_amd._lib import lib
    return lib()


def ptr(t):
    return P(t.data_ptr()) if t is not None else P(0)


def ok(code):
    assert code == 0, L().nc_last_error().decode()


def rnd(t, dt):
    return t.to(torch.bfloat16 if dt == BF else torch.float16).float()


@pytest.fixture(autouse=True)
def _mode():
    from neuroclear_amd import ops
    prev = L().nc_get_c8x_mode()
    L().nc_set_c8x_mode(2)
    yield
    L().nc_set_c8x_mode(prev)
    ops.set_conv_precision('fp32')


CASES = [  # N, C, K, (D, H, W), kernel size
    (1, 64, 64, (8, 8, 8), 3),
    (2, 64, 64, (5, 9, 13), 3),
    (1, 64, 128, (20, 20, 20), 3),
    (1, 128, 64, (7, 30, 37), 3),      # 30 x 39 = 1170 positions: 2.3 tiles per plane
    (1, 64, 64, (36, 36, 36), 3),
    (1, 256, 128, (6, 27, 27), 3),
    (2, 64, 64, (3, 54, 54), 3),
    (1, 64, 64, (1, 23, 75), 3),       # one plane: two of three bricks of every chunk are outside the volume
    (3, 64, 192, (2, 17, 19), 3),      # three output-channel tiles, three samples
    (1, 64, 64, (20, 20, 20), 5),      # G_B's feature block (networks.py:900)
    (2, 64, 128, (6, 11, 23), 5),
    (1, 64, 64, (4, 40, 52), 5),
    (1, 128, 128, (2, 9, 150), 5),     # the row pitch of configs[3]'s 148-wide rows and beyond
]


def _run(case, dt, mode):
    from neuroclear_amd import ops
    N, C, K, (D, H, W), ks = case
    pad = ks // 2
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(N, C, D, H, W, device=DEV, generator=g)
    w = torch.randn(K, C, ks, ks, ks, device=DEV, generator=g) / (C * ks ** 3) ** 0.5
    b = torch.randn(K, device=DEV, generator=g)
    dy = torch.randn(N, K, D, H, W, device=DEV, generator=g)
    L().nc_set_c8x_mode(mode)
    ops.set_conv_precision('bf16' if dt == BF else 'fp16')
    y = ops.conv_fwd_raw(x, w, b, 1, pad)
    dx = None
    if L().nc_conv_lp_supported(1, N, C, D, H, W, K, ks, ks, ks, 1, pad):
        ops.set_conv_precision('bf16')  # backward operands are bf16 in both modes (ops._lp)
        dx = ops.conv_dgrad_raw(dy, w, x.shape, 1, pad)
    ops.set_conv_precision('fp32')
    return x, w, b, dy, y, dx


@pytest.mark.parametrize('dt', [BF, FP])
@pytest.mark.parametrize('case', CASES)
def test_c8x_matches_rounded_operands(case, dt):
    N, C, K, (D, H, W), ks = case
    pad = ks // 2
    x, w, b, dy, y, dx = _run(case, dt, 2)

    def close(a, ref, tol=5e-5):
        assert (a - ref).abs().max().item() <= tol * ref.abs().max().item()
    close(y, F.conv3d(rnd(x, dt), rnd(w, dt), b, padding=pad))
    if dx is not None:
        close(dx, F.conv_transpose3d(rnd(dy, BF), rnd(w, BF), padding=pad))
    # the same call again: bit for bit (a fragment consumed before it arrived shows up as a run-to-run difference long before it shows
    # up against the tolerance)
    for _ in range(2):
        _, _, _, _, y1, dx1 = _run(case, dt, 2)
        assert torch.equal(y, y1) and (dx is None or torch.equal(dx, dx1))
    # the other kernel on the same call: same operands, another summation order
    _, _, _, _, y0, dx0 = _run(case, dt, 0)
    close(y, y0, 2e-5)
    if dx is not None:
        close(dx, dx0, 2e-5)
    L().nc_set_c8x_mode(2)
    assert L().nc_conv_lp_uses_c8x(0, 1, N, C, D, H, W, K, ks) == 1 and (dx is None or L().nc_conv_lp_uses_c8x(1, 1, N, C, D, H, W, K, ks) == 1)
    L().nc_set_c8x_mode(0)
    assert L().nc_conv_lp_uses_c8x(0, 1, N, C, D, H, W, K, ks) == 0


@pytest.mark.parametrize('dt', [BF, FP])
@pytest.mark.parametrize('N,C,K,dims,ks', [(2, 64, 64, (5, 9, 13), 3), (1, 64, 128, (12, 12, 12), 3), (2, 64, 64, (6, 11, 23), 5),
                                           (1, 64, 64, (3, 40, 44), 3)])
def test_c8x_c8_output_is_the_rounded_fp32_output(N, C, K, dims, ks, dt):
    from neuroclear_amd import ops
    D, H, W = dims
    S = D * H * W
    pad = ks // 2
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn(N, C, D, H, W, device=DEV, generator=g)
    w = torch.randn(K, C, ks, ks, ks, device=DEV, generator=g) / (C * ks ** 3) ** 0.5
    b = torch.randn(K, device=DEV, generator=g)
    dy = torch.randn(N, K, D, H, W, device=DEV, generator=g)
    ops.set_conv_precision('bf16' if dt == BF else 'fp16')
    xh = ops.to_c8(x, dt)
    y32 = ops.conv_fwd_raw(x, w, b, 1, pad, xh=xh)
    nb = L().nc_conv_lp_ws_bytes(N, C, D, H, W, K, ks, ks, ks, 1, pad)
    wsb = torch.empty(int(nb) + 256, dtype=torch.uint8, device=DEV)
    yh = torch.empty(N * K * S * 2, dtype=torch.uint8, device=DEV)
    ok(L().nc_conv_fwd_c8(ptr(xh), ptr(w), ptr(b), ptr(yh), K, 0, N, C, D, H, W, K, ks, ks, ks, 1, pad, dt, ptr(wsb), ctypes.c_size_t(wsb.numel()), None))
    assert torch.equal(yh, ops.to_c8(y32, dt))
    # into channels [K, 2K) of a 2K + 8 channel buffer; the rest must stay untouched
    ctot = 2 * K + 8
    buf = torch.full((N * ctot * S * 2,), 0x5a, dtype=torch.uint8, device=DEV)
    ok(L().nc_conv_fwd_c8(ptr(xh), ptr(w), ptr(b), ptr(buf), ctot, K, N, C, D, H, W, K, ks, ks, ks, 1, pad, dt, ptr(wsb), ctypes.c_size_t(wsb.numel()),
                          None))
    v = buf.view(N, ctot // 8, S * 16)
    assert torch.equal(v[:, K // 8:2 * K // 8].reshape(-1), yh)
    assert bool((v[:, :K // 8] == 0x5a).all()) and bool((v[:, 2 * K // 8:] == 0x5a).all())
    ops.set_conv_precision('bf16')
    dyh = ops.to_c8(dy, BF)
    dx32 = ops.conv_dgrad_raw(dy, w, x.shape, 1, pad, dyh=dyh)
    dxh = torch.empty(N * C * S * 2, dtype=torch.uint8, device=DEV)
    ok(L().nc_conv_dgrad_c8(ptr(dyh), ptr(w), ptr(dxh), N, C, D, H, W, K, ks, ks, ks, 1, pad, BF, ptr(wsb), ctypes.c_size_t(wsb.numel()), None))
    assert torch.equal(dxh, ops.to_c8(dx32, BF))


def test_c8x_default_mode_picks_by_launch_quantisation():
    """Mode 1 (the default): the tap-stream kernel wherever its 512-position tiles fill >= 60 % of the launch's rounds of 512 workgroups --
    every layer of configs[1] / configs[3] -- and k_conv_h for launches of a few planes.  Checked through the query AND through the
    results (the two kernels sum 3^3 layers in different orders)."""
    from neuroclear_amd import ops
    ops.set_conv_precision('bf16')

    def run(shape, K, mode):
        L().nc_set_c8x_mode(mode)
        gg = torch.Generator(device=DEV).manual_seed(7)
        x = torch.randn(*shape, device=DEV, generator=gg)
        w = torch.randn(K, shape[1], 3, 3, 3, device=DEV, generator=gg) * 0.03
        return ops.conv_fwd_raw(x, w, None, 1, 1)

    def uses(shape, K):
        L().nc_set_c8x_mode(1)
        N, C, D, H, W = shape
        return L().nc_conv_lp_uses_c8x(0, 1, N, C, D, H, W, K, 3)
    small = (1, 64, 8, 20, 20)      # 8 planes x 1 tile: 8 of 512 workgroups
    assert uses(small, 64) == 0
    assert torch.equal(run(small, 64, 1), run(small, 64, 0)) and not torch.equal(run(small, 64, 1), run(small, 64, 2))
    for shape, K in (((4, 256, 37, 37, 37), 256), ((1, 64, 108, 108, 108), 64), ((4, 64, 74, 74, 74), 128)):  # configs[3] bottom, configs[1], 1/2 res
        assert uses(shape, K) == 1, shape
    full = (2, 64, 64, 148, 148)    # 2 x 64 planes x 44 tiles = 5632 tiles = 11 rounds
    assert uses(full, 64) == 1
    assert torch.equal(run(full, 64, 1), run(full, 64, 2)) and not torch.equal(run(full, 64, 1), run(full, 64, 0))


@pytest.mark.parametrize('kind,shape', [('unet', (1, 1, 32, 36, 40)), ('glin', (2, 1, 24, 28, 36))])
def test_whole_network_16bit_calls_agree_between_the_two_kernels(kind, shape):
    """nc_unet_deconv_lp_fwd / _bwd and nc_deep_linear_lp_fwd / _bwd (tap-diffused weight rounding included) with every 3^3 / 5^3
    layer on k_conv_c8x vs on k_conv_h: outputs within the 16-bit rounding of the stored activations (a different summation order can
    move a stored bf16 value by one ulp, 2^-8 relative), parameter gradients within 2 % in L2."""
    from neuroclear_amd import ops
    from neuroclear_amd.models import networks
    from neuroclear_amd.util import seed as S

    def run(mode):
        L().nc_set_c8x_mode(mode)
        ops.set_conv_precision('bf16')
        if kind == 'unet':
            net = networks.define_G(1, 1, 64, 'unet_deconv', 'instance', False, 'kaiming', 0.02, [0])
            net.load_state_dict(S.state_dict_from_seed(S.unet_deconv_spec(), 5, DEV))
        else:
            net = networks.define_G(1, 1, 64, 'deep_linear_gen', 'instance', False, 'kaiming', 0.02, [0])
            net.load_state_dict(S.state_dict_from_seed(S.deep_linear_spec(), 5, DEV))
        g = torch.Generator(device=DEV).manual_seed(11)
        x = torch.rand(*shape, device=DEV, generator=g).requires_grad_(True)
        y = net(x)
        (y * torch.linspace(0, 1, y.numel(), device=DEV).view_as(y)).sum().backward()
        ops.set_conv_precision('fp32')
        return y.detach(), x.grad, [p.grad.clone() for p in net.parameters()]
    y2, dx2, g2 = run(2)
    y0, dx0, g0 = run(0)
    assert (y2 - y0).abs().max().item() <= 2e-2 * max(y0.abs().max().item(), 1e-6)
    # unet_deconv with random weights: a stored activation that moves by one bf16 ulp flips ReLU / max-pool decisions downstream (the same
    # 6-17 % a 1e-3 input perturbation causes in pure fp32, tests/test_gpu_c8.py); deep_linear_gen is linear: only rounding
    tol = 0.3 if kind == 'unet' else 2e-2
    assert ((dx2 - dx0).norm() / dx0.norm()).item() < tol
    for a, b in zip(g2, g0):
        if b.dim() > 1:
            assert ((a - b).norm() / b.norm().clamp_min(1e-20)).item() < tol


@pytest.mark.parametrize('shape,K,ks', [((8, 64, 40, 40, 52), 64, 5), ((8, 64, 40, 40, 52), 128, 3), ((6, 192, 20, 40, 52), 64, 5), ((6, 128, 30, 40, 52), 64, 5)])
def test_c8x_many_tiles_per_workgroup(shape, K, ks):
    """1,200-3,200 tiles on 512 workgroups: every workgroup walks several tiles, so the ring, the tap state and the three rotating
    weight-fragment sets carry over tile boundaries -- with 250 Cin / 64 k-steps per 5^3 tile (no multiple of three) through the
    set-exchange path.  5^3: bit-identical to k_conv_h (same tap order, same 16-product accumulation steps); 3^3: to 2e-5."""
    from neuroclear_amd import ops
    ops.set_conv_precision('bf16')
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(*shape, device=DEV, generator=g)
    w = torch.randn(K, shape[1], ks, ks, ks, device=DEV, generator=g) / (shape[1] * ks ** 3) ** 0.5
    out = {}
    for mode in (0, 2, 2):
        L().nc_set_c8x_mode(mode)
        y = ops.conv_fwd_raw(x, w, None, 1, ks // 2)
        assert mode == 0 or 2 not in out or torch.equal(out[2], y)  # run to run
        out[mode] = y
    if ks == 5:
        assert torch.equal(out[0], out[2])
    else:
        assert (out[0] - out[2]).abs().max().item() <= 2e-5 * out[0].abs().max().item()
