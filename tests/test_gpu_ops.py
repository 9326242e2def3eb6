"""GPU parity tests, op level (run with -m gpu on the MI355X box).  Every op of libnc_hip.so is called through the
C ABI (neuroclear_amd.ops -> ctypes) and compared with
  * torch's own fp32 implementation of the same op on the GPU (a plain fp32 reference, any size), and
  * the MFMA path against the direct path of this library (nc_set_force_direct).
Tolerances are relative to the largest reference magnitude: fp32 accumulation order differs from MIOpen's."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from neuroclear_amd import ops  # noqa: E402

DEV = 'cuda'


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def rel2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def T(seed, shape, scale=1.0, shift=0.0):
    g = torch.Generator().manual_seed(seed)
    return ((torch.rand(shape, generator=g) - 0.5) * 2 * scale + shift).to(DEV)


CONV_CASES = [
    # N, C, K, spatial, k, stride, pad
    (1, 1, 64, (12, 12, 12), 3, 1, 1),
    (1, 64, 64, (12, 14, 20), 3, 1, 1),      # MFMA
    (2, 64, 128, (9, 10, 27), 3, 1, 1),      # MFMA, odd sizes, batch 2
    (1, 128, 256, (7, 7, 7), 3, 1, 1),       # MFMA small
    (1, 256, 128, (8, 12, 12), 3, 1, 1),     # MFMA
    (1, 128, 64, (6, 20, 36), 3, 1, 1),      # MFMA (ex_conv1_1 shape class)
    (1, 64, 64, (10, 12, 16), 5, 1, 2),      # MFMA 5^3
    (1, 64, 64, (6, 7, 60), 5, 1, 2),        # 5^3, W > 56, W % 4 == 0: k_wgrad_dma<5,1>
    (1, 64, 64, (5, 6, 72), 3, 1, 1),        # 3^3, W > 56, W % 4 == 0: k_wgrad_dma<3,2>, two column blocks
    (1, 32, 64, (5, 6, 70), 3, 1, 1),        # 3^3, W > 56, W % 4 != 0: register-staged k_wgrad_mfma, TAIL fwd kernel
    (2, 32, 64, (6, 5, 54), 3, 1, 1),        # W = 54: k_wgrad_rows with 2 rows per step, row tails
    (1, 1, 64, (12, 12, 12), 7, 1, 3),
    (1, 1, 64, (10, 9, 20), 7, 1, 3),        # tap-axis MFMA wgrad (W % 4 == 0), ragged D/H
    (2, 1, 64, (7, 11, 36), 3, 1, 1),        # same, 3^3, batch 2
    (2, 1, 64, (5, 30, 24), 7, 1, 3),        # MFMA 64 -> 1 data gradient: row ranges that split planes (halo rows)
    (2, 1, 64, (9, 10, 70), 7, 1, 3),        # 7^3, ragged: exercises tile edges of the many->one dgrad kernel
    (1, 1, 64, (8, 9, 148), 7, 1, 3),        # W > 112: the 64 -> 1 data gradient runs as two column segments
    (1, 1, 64, (7, 8, 116), 7, 1, 3),        # same, narrowest two-segment case
    (1, 64, 1, (10, 10, 10), 1, 1, 0),
    (1, 1, 1, (8, 8, 8), 1, 1, 0),
    (1, 64, 32, (6, 6, 6), 1, 1, 0),
    (1, 64, 32, (28, 28, 24), 1, 1, 0),      # 1x1 wgrad on the flat-voxel MFMA kernel (18816 voxels, ragged last chunk)
    (2, 16, 1, (26, 26, 28), 1, 1, 0),       # same, channel padding on both sides, batch 2
    (1, 1, 1, (32, 32, 32), 1, 1, 0),
    (3, 1, 64, (36, 36), 4, 2, 1),           # PatchGAN 2-D
    (1, 64, 128, (18, 18), 4, 2, 1),
    (1, 256, 512, (4, 4), 4, 1, 1),
    (1, 512, 1, (3, 3), 4, 1, 1),
    (32, 128, 256, (40, 40), 4, 1, 1),       # Athena-sized batch: 128-row gather-GEMM tiles (fwd, dgrad)
    (32, 128, 256, (40, 40), 4, 2, 1),       # same, strided: parity-class dgrad on 128-row tiles
    (24, 160, 192, (33, 35), 4, 1, 1),
    (20, 512, 1, (12, 12), 4, 1, 1),         # PatchGAN head at a large batch: K = 1 reduction kernels (fwd, wgrad)
    (3, 100, 1, (9, 10, 11), 4, 1, 1),       # same in 3-D, ragged channel groups
       # 128-row tiles with a ragged last row tile (M = 192 / 160) and ragged N
    (1, 1, 64, (12, 12, 12), 4, 2, 1),       # PatchGAN 3-D
]


@pytest.mark.parametrize('case', CONV_CASES, ids=lambda c: 'N%dC%dK%d_%s_k%ds%dp%d' % (c[0], c[1], c[2], 'x'.join(map(str, c[3])), c[4], c[5], c[6]))
@pytest.mark.parametrize('force_direct', [False, True], ids=['auto', 'direct'])
def test_conv(case, force_direct):
    N, C, K, sp, k, s, p = case
    nd = len(sp)
    x = T(1, (N, C) + sp).requires_grad_(True)
    w = T(2, (K, C) + (k,) * nd, scale=(2.0 / (C * k ** nd)) ** 0.5 * 1.7).requires_grad_(True)
    b = T(3, (K,), scale=0.1).requires_grad_(True)
    fn = F.conv3d if nd == 3 else F.conv2d
    yr = fn(x, w, b, stride=s, padding=p)
    r = T(4, yr.shape)
    gx, gw, gb = torch.autograd.grad((yr * r).sum(), (x, w, b))
    ops.set_force_direct(force_direct)
    try:
        x2, w2, b2 = (t.detach().clone().requires_grad_(True) for t in (x, w, b))
        y = ops.conv(x2, w2, b2, s, p)
        hx, hw, hb = torch.autograd.grad((y * r).sum(), (x2, w2, b2))
    finally:
        ops.set_force_direct(False)
    torch.cuda.synchronize()
    assert y.shape == yr.shape
    errs = dict(y=rel(y, yr), dx=rel(hx, gx), dw=rel(hw, gw), db=rel(hb, gb))
    print(case, 'direct' if force_direct else 'auto', errs)
    assert all(e < 2e-5 for e in errs.values()), errs


def test_conv_paths_reported():
    from neuroclear_amd._lib import I, lib
    L = lib()
    assert L.nc_conv_fwd_path(I(64), I(64), I(3), I(3), I(3), I(1), I(1)) in (1, 9)  # 9: split-operand kernel (default on)
    assert L.nc_conv_fwd_path(I(64), I(64), I(5), I(5), I(5), I(1), I(2)) in (1, 9)
    assert L.nc_conv_wgrad_path(I(64), I(64), I(3), I(3), I(3), I(1), I(1)) in (1, 9)
    assert L.nc_conv_fwd_path(I(1), I(64), I(3), I(3), I(3), I(1), I(1)) == 1  # single-channel mode of the brick kernel
    assert L.nc_conv_wgrad_path(I(1), I(64), I(7), I(7), I(7), I(1), I(3)) == 4
    assert L.nc_conv_fwd_path(I(64), I(128), I(1), I(4), I(4), I(2), I(1)) == 2
    assert L.nc_conv_fwd_path(I(64), I(1), I(1), I(1), I(1), I(1), I(0)) == 3  # flat-voxel pointwise kernel


@pytest.mark.parametrize('shape', [(1, 256, 4, 5, 6), (2, 128, 3, 4, 4), (1, 16, 3, 4, 5), (2, 128, 9, 20, 21)])
def test_convT(shape):  # C >= 64: data gradient on the MFMA gather GEMM; C = 16: VALU kernel
    N, C = shape[:2]
    K = C // 2
    x = T(5, shape).requires_grad_(True)
    w = T(6, (C, K, 2, 2, 2), scale=0.1).requires_grad_(True)
    b = T(7, (K,), scale=0.1).requires_grad_(True)
    yr = F.conv_transpose3d(x, w, b, stride=2)
    r = T(8, yr.shape)
    g = torch.autograd.grad((yr * r).sum(), (x, w, b))
    x2, w2, b2 = (t.detach().clone().requires_grad_(True) for t in (x, w, b))
    y = ops.conv_transpose_k2s2(x2, w2, b2)
    h = torch.autograd.grad((y * r).sum(), (x2, w2, b2))
    errs = [rel(y, yr)] + [rel(a, c) for a, c in zip(h, g)]
    print(errs)
    assert all(e < 1e-5 for e in errs), errs


@pytest.mark.parametrize('shape,slope,shift', [((1, 64, 20, 20, 20), 0.0, 0.0), ((2, 16, 9, 9, 9), 0.0, 0.0),
                                               ((3, 128, 17, 17), 0.2, 0.0), ((1, 16, 20, 20, 20), 0.0, 100.0)])
def test_instnorm_act(shape, slope, shift):
    x = (T(9, shape, scale=0.05 if shift else 1.0, shift=shift)).requires_grad_(True)
    yr = F.leaky_relu(F.instance_norm(x.double(), eps=1e-5), slope) if slope else F.relu(F.instance_norm(x.double(), eps=1e-5))
    r = T(10, shape)
    (gx,) = torch.autograd.grad((yr * r.double()).sum(), x)
    x2 = x.detach().clone().requires_grad_(True)
    y = ops.instance_norm_act(x2, slope)
    (hx,) = torch.autograd.grad((y * r).sum(), x2)
    # with a mean of 100 the fp32 inputs themselves carry ~1e-4 sigma of representation noise: a ReLU mask flip at
    # xhat ~ 0 moves single elements of dx by O(1), so that case is judged in the L2 norm
    e = (rel(y, yr), rel2(hx, gx) if shift else rel(hx, gx))
    print(shape, e)
    assert e[0] < (2e-3 if shift else 2e-6) and e[1] < (2e-2 if shift else 2e-5), e


def test_pool_sigmoid_lrelu():
    for shape in [(1, 64, 12, 12, 12), (1, 8, 13, 11, 9), (2, 4, 10, 10)]:
        x = T(11, shape).requires_grad_(True)
        pool = F.max_pool3d if len(shape) == 5 else F.max_pool2d
        yr = pool(x, 2)
        r = T(12, yr.shape)
        (gx,) = torch.autograd.grad((yr * r).sum(), x)
        x2 = x.detach().clone().requires_grad_(True)
        y = ops.maxpool2(x2)
        (hx,) = torch.autograd.grad((y * r).sum(), x2)
        assert torch.equal(y, yr) and torch.equal(hx, gx), shape
    x = T(13, (5, 7, 33), scale=6).requires_grad_(True)
    for mine, ref in ((ops.sigmoid, torch.sigmoid), (lambda t: ops.leaky_relu(t, 0.2), lambda t: F.leaky_relu(t, 0.2))):
        yr = ref(x)
        (gx,) = torch.autograd.grad(yr.sum(), x)
        x2 = x.detach().clone().requires_grad_(True)
        y = mine(x2)
        (hx,) = torch.autograd.grad(y.sum(), x2)
        assert rel(y, yr) < 1e-6 and rel(hx, gx) < 1e-6


def test_slice_mip():
    vol = T(14, (2, 1, 10, 12, 14)).requires_grad_(True)
    for axis in range(3):
        idx = [3, 7, 13][axis]
        ref = vol.select(axis + 2, idx)
        r = T(15, ref.shape)
        (g,) = torch.autograd.grad((ref * r).sum(), vol)
        v2 = vol.detach().clone().requires_grad_(True)
        out = ops.volume_slice(v2, axis, idx)
        (h,) = torch.autograd.grad((out * r).sum(), v2)
        assert torch.equal(out, ref) and torch.equal(h, g)
        start, depth = 2, 5
        roi = vol.narrow(axis + 2, start, depth)
        ref = torch.max(roi, axis + 2)[0]
        (g,) = torch.autograd.grad((ref * r).sum(), vol)
        v2 = vol.detach().clone().requires_grad_(True)
        out = ops.volume_mip(v2, axis, start, depth)
        (h,) = torch.autograd.grad((out * r).sum(), v2)
        assert torch.equal(out, ref) and torch.equal(h, g)


def test_losses_and_adam():
    p = T(16, (1, 1, 11, 11)).requires_grad_(True)
    for tgt in (1.0, 0.0):
        ref = F.mse_loss(p, torch.full_like(p, tgt)) * 0.37
        (g,) = torch.autograd.grad(ref, p)
        p2 = p.detach().clone().requires_grad_(True)
        out = ops.mse_const(p2, tgt) * 0.37
        (h,) = torch.autograd.grad(out, p2)
        assert abs(float(out) - float(ref)) < 1e-6 * abs(float(ref)) + 1e-9 and rel(h, g) < 1e-6
    a = T(17, (1, 1, 20, 20, 20)).requires_grad_(True)
    b = T(18, (1, 1, 20, 20, 20))
    ref = F.l1_loss(a, b) * 5.0
    (g,) = torch.autograd.grad(ref, a)
    a2 = a.detach().clone().requires_grad_(True)
    out = ops.l1_loss(a2, b) * 5.0
    (h,) = torch.autograd.grad(out, a2)
    assert abs(float(out) - float(ref)) < 1e-6 * abs(float(ref)) and rel(h, g) < 1e-6
    # Adam: 3 steps against torch.optim.Adam (lr 1e-4, betas (0.1, 0.999) -- options/train_options.py:35-36)
    w = T(19, (1000,))
    wr = w.clone().requires_grad_(True)
    opt = torch.optim.Adam([wr], lr=1e-4, betas=(0.1, 0.999))
    m = torch.zeros_like(w)
    v = torch.zeros_like(w)
    for step in range(1, 4):
        g = T(20 + step, (1000,))
        wr.grad = g.clone()
        opt.step()
        ops.adam_step(w, g, m, v, 1e-4, 0.1, 0.999, 1e-8, step)
    assert rel(w, wr.detach()) < 1e-6
    assert float((w - wr.detach()).abs().max()) < 1e-7


def _sweep_cases():
    """Seeded sweep over the shape space of nc_conv_*: every width class the planners distinguish (W % 4, W <= 56,
    W > 56, W > 112, W > 192), channel counts on and off the MFMA grid, every kernel size the path dispatch knows."""
    rng = np.random.default_rng(2024)
    cases = []
    widths = [5, 8, 12, 16, 20, 27, 28, 36, 54, 57, 60, 70, 72, 108, 116, 130, 148, 196]
    for i in range(44):
        k, s, p = [(3, 1, 1), (3, 1, 1), (5, 1, 2), (7, 1, 3), (1, 1, 0), (4, 2, 1), (4, 1, 1)][int(rng.integers(0, 7))]
        W = int(widths[int(rng.integers(0, len(widths)))])
        if k == 4 and W < 8:
            W = 8
        C = int([1, 16, 32, 64, 128][int(rng.integers(0, 5))])
        K = int([1, 32, 64, 64, 128][int(rng.integers(0, 5))])
        if k == 7:
            C, K = (1, 64) if rng.integers(0, 2) else (int(C), 64)
        N = int(rng.integers(1, 3))
        nd = 2 if (k == 4 and rng.integers(0, 2)) else 3
        D = int(rng.integers(max(2, k), 7)) if k < 7 else int(rng.integers(7, 9))
        H = int(rng.integers(max(3, k), 10)) if k < 7 else int(rng.integers(7, 10))
        if C * K * k ** nd * W * D * H > 6e9:  # keep the torch reference quick
            C, K = min(C, 32), min(K, 64)
        sp = (D, H, W) if nd == 3 else (H * 3, W)
        cases.append((N, C, K, sp, k, s, p))
    return cases


@pytest.mark.parametrize('case', _sweep_cases(), ids=lambda c: 'N%dC%dK%d_%s_k%ds%dp%d' % (c[0], c[1], c[2], 'x'.join(map(str, c[3])), c[4], c[5], c[6]))
def test_conv_shape_sweep(case):
    test_conv(case, False)


def test_instnorm_bwd_dbias_is_the_channel_sum_of_dx():
    """nc_instnorm_act_bwd_dbias: dbias[c] = sum over samples and voxels of dx.  With the true statistics that sum is
    rounding noise (InstanceNorm's backward has zero mean per instance), so the plumbing is checked with deliberately
    wrong statistics, where the sums are O(1)."""
    from neuroclear_amd import _lib
    from neuroclear_amd._lib import F as CF, I, L_, Z
    L = _lib.lib()
    N, C, S = 3, 10, 5000
    g = torch.Generator(device='cuda').manual_seed(4)
    x = torch.randn(N, C, S, device='cuda', generator=g)
    dy = torch.randn(N, C, S, device='cuda', generator=g)
    mean = x.mean(2).reshape(-1) + 0.3 * torch.randn(N * C, device='cuda', generator=g)
    rstd = 1.0 / (x.var(2, unbiased=False).reshape(-1) + 1e-5).sqrt() * 1.2
    dx1, dx2 = torch.empty_like(x), torch.empty_like(x)
    db = torch.empty(C, device='cuda')
    nb = L.nc_instnorm_bwd_dbias_ws_bytes(I(N * C), L_(S))
    ws = torch.empty(nb, dtype=torch.uint8, device='cuda')
    P = ops._ptr
    assert L.nc_instnorm_act_bwd(P(dy), P(x), P(mean), P(rstd), CF(0.2), P(dx1), I(N * C), L_(S), P(ws), Z(nb), None) == 0
    assert L.nc_instnorm_act_bwd_dbias(P(dy), P(x), P(mean), P(rstd), CF(0.2), P(dx2), P(db), I(N), I(C), L_(S), P(ws), Z(nb),
                                       None) == 0
    assert torch.equal(dx1, dx2)
    want = dx1.double().sum((0, 2))
    assert want.abs().min().item() > 1e-2  # the sums are not noise in this set-up
    assert (db.double() - want).abs().max().item() <= 1e-6 * dx1.abs().double().sum((0, 2)).max().item()


@pytest.mark.parametrize('N,C,S', [(5, 24, 144), (3, 16, 169), (2, 32, 729), (2, 8, 2048), (7, 3, 7), (1, 2, 1),
                                   (300, 256, 16)])
def test_instnorm_short_instances(N, C, S):
    """Instances of S <= 2048 elements (the 2-D PatchGAN layers) run in the one-kernel group-per-instance form: statistics
    against fp64, stats / stats + fwd / fwd-with-given-stats bit-equal, backward against autograd, the instance sums of dx
    (bias gradient hand-over) with deliberately wrong statistics; N * C = 76800 > the 65535 grid.y limit of the long path."""
    from neuroclear_amd import _lib
    from neuroclear_amd._lib import F as CF, I, L_, Z
    L = _lib.lib()
    P = ops._ptr
    g = torch.Generator(device='cuda').manual_seed(N * 1000 + S)
    x = torch.randn(N, C, S, device='cuda', generator=g) * 2 + 0.5
    dy = torch.randn(N, C, S, device='cuda', generator=g)
    NC = N * C
    nb = L.nc_instnorm_bwd_dbias_ws_bytes(I(NC), L_(S))
    ws = torch.empty(max(nb, 256), dtype=torch.uint8, device='cuda')
    m0, r0, m1, r1 = (torch.empty(NC, device='cuda') for _ in range(4))
    y1, y2 = torch.empty_like(x), torch.empty_like(x)
    assert L.nc_instnorm_stats(P(x), I(NC), L_(S), CF(1e-5), P(m0), P(r0), P(ws), Z(nb), None) == 0
    assert L.nc_instnorm_fwd(P(x), CF(1e-5), CF(0.2), P(m1), P(r1), P(y1), I(NC), L_(S), P(ws), Z(nb), None) == 0
    assert L.nc_instnorm_act_fwd(P(x), P(m0), P(r0), CF(0.2), P(y2), I(NC), L_(S), None) == 0
    assert torch.equal(m0, m1) and torch.equal(r0, r1) and torch.equal(y1, y2)
    xd = x.double().requires_grad_(True)
    mean = xd.mean(2)
    var = xd.var(2, unbiased=False)
    assert (m0.double() - mean.reshape(-1)).abs().max().item() <= 1e-6
    assert rel(r0, (1.0 / (var + 1e-5).sqrt()).reshape(-1).detach()) <= 1e-6
    yr = F.leaky_relu((xd - mean[..., None]) / (var[..., None] + 1e-5).sqrt(), 0.2)
    assert rel(y1, yr.detach()) <= 2e-6
    (gx,) = torch.autograd.grad((yr * dy.double()).sum(), xd)
    dx1, dx2 = torch.empty_like(x), torch.empty_like(x)
    db = torch.empty(C, device='cuda')
    assert L.nc_instnorm_act_bwd(P(dy), P(x), P(m0), P(r0), CF(0.2), P(dx1), I(NC), L_(S), P(ws), Z(nb), None) == 0
    assert L.nc_instnorm_act_bwd_dbias(P(dy), P(x), P(m0), P(r0), CF(0.2), P(dx2), P(db), I(N), I(C), L_(S), P(ws), Z(nb),
                                       None) == 0
    assert torch.equal(dx1, dx2)
    if S > 1:
        assert rel(dx1, gx) <= 2e-5, rel(dx1, gx)
    # wrong statistics: the channel sums of dx are O(1) and must come out of the hand-over
    mw = m0 + 0.3 * torch.randn(NC, device='cuda', generator=g)
    rw = r0 * 1.2
    assert L.nc_instnorm_act_bwd_dbias(P(dy), P(x), P(mw), P(rw), CF(0.2), P(dx2), P(db), I(N), I(C), L_(S), P(ws), Z(nb),
                                       None) == 0
    want = dx2.double().sum((0, 2))
    assert (db.double() - want).abs().max().item() <= 1e-6 * dx2.abs().double().sum((0, 2)).max().item()


def test_bias_link_gives_the_same_gradients():
    """(Conv, InstanceNormAct) pairs of the U-Net blocks hand the bias gradient over through ops.BiasLink: same dx / dw as
    the unlinked ops, bias gradient = the (noise-level) channel sum of the same dx."""
    x = T(21, (2, 8, 6, 7, 9)).requires_grad_(True)
    w = T(22, (12, 8, 3, 3, 3), scale=0.2).requires_grad_(True)
    b = T(23, (12,), scale=0.1).requires_grad_(True)
    r = T(24, (2, 12, 6, 7, 9))
    y0 = ops.instance_norm_act(ops.conv(x, w, b, 1, 1), 0.0)
    g0 = torch.autograd.grad((y0 * r).sum(), (x, w, b))
    link = ops.BiasLink()
    y1 = ops.instance_norm_act(ops.conv(x, w, b, 1, 1, link), 0.0, 1e-5, link)
    g1 = torch.autograd.grad((y1 * r).sum(), (x, w, b))
    assert torch.equal(y0, y1) and torch.equal(g0[0], g1[0]) and torch.equal(g0[1], g1[1])
    scale = g0[1].abs().max().item()
    assert g0[2].abs().max().item() <= 1e-4 * scale and g1[2].abs().max().item() <= 1e-4 * scale  # both are noise
    assert link.dbias is None  # consumed


@pytest.mark.parametrize('B,C,K,H,W,s', [(54, 64, 128, 54, 54, 2), (108, 128, 256, 27, 27, 2), (108, 256, 512, 13, 13, 1),
                                         (70, 32, 64, 40, 36, 2), (90, 16, 64, 21, 30, 1), (130, 64, 64, 23, 19, 2),
                                         (9, 64, 128, 108, 108, 2), (216, 256, 512, 13, 13, 1), (3, 64, 128, 54, 54, 2),
                                         (24, 64, 128, 74, 74, 2), (40, 32, 128, 31, 45, 2), (30, 64, 128, 20, 70, 1),
                                         (50, 8, 256, 17, 33, 1)])
def test_image_staged_patchgan_convs(B, C, K, H, W, s):
    """conv2d_img.hip (k_sconv): the 4 x 4 / padding 1 PatchGAN layers at batches that fill the chip (smaller ones stay on
    the gather GEMM: last case), forward and data gradient (stride 2: four parity classes in one launch), against torch
    fp32 -- odd sizes, tiles that span three images, Athena's real + fake batch; the last four: weight-gradient stages whose
    input rows are wider than one 64-lane copy (W = 74 / 70), odd non-square planes, 8 input channels."""
    import torch.nn.functional as F
    from neuroclear_amd._lib import lib
    g = torch.Generator(device=DEV).manual_seed(B * 1000 + H)
    x = torch.randn(B, C, H, W, device=DEV, generator=g)
    w = torch.randn(K, C, 4, 4, device=DEV, generator=g) / (C * 16) ** 0.5
    b = torch.randn(K, device=DEV, generator=g)
    assert lib().nc_conv_fwd_path(C, K, 1, 4, 4, s, 1) in (2, 7)  # (the path query assumes a 32^2 image)
    y = ops.conv_fwd_raw(x, w, b, s, 1)
    ref = F.conv2d(x, w, b, stride=s, padding=1)
    assert y.shape == ref.shape
    assert float((y - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    dy = torch.randn(ref.shape, device=DEV, generator=g)
    dx = ops.conv_dgrad_raw(dy, w, x.shape, s, 1)
    refd = F.conv_transpose2d(dy, w, stride=s, padding=1, output_padding=(H + 2 - 4) % s if s > 1 else 0)
    if refd.shape != x.shape:  # odd input sizes: the transposed convolution's natural size is one short
        refd = F.pad(refd, (0, x.shape[3] - refd.shape[3], 0, x.shape[2] - refd.shape[2]))
    assert float((dx - refd).abs().max()) <= 2e-5 * float(refd.abs().max())
    # weight gradient (k_swgrad: the reduction runs over the flat (image, u, v) axis in chunks, summed in a fixed order)
    dw, db = ops.conv_wgrad_raw(x, dy, w.shape, s, 1, True)
    refw = torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double(), stride=s, padding=1)
    assert float((dw.double() - refw).abs().max()) <= 2e-5 * float(refw.abs().max())
    assert float((db.double() - dy.double().sum((0, 2, 3))).abs().max()) <= 1e-5 * float(dy.double().sum((0, 2, 3)).abs().max() + 1)
    dw2, _ = ops.conv_wgrad_raw(x, dy, w.shape, s, 1, False)
    assert torch.equal(dw, dw2)  # deterministic


@pytest.mark.parametrize('B,C,K,H,W,s', [(108, 128, 256, 27, 27, 2), (60, 256, 128, 13, 13, 1), (40, 64, 128, 31, 22, 2)])
def test_image_staged_configs_agree(B, C, K, H, W, s):
    """The tile shape of k_sconv is picked per problem by timing (conv2d_img.hip, run_tuned).  That is only admissible because
    every shape accumulates an output element in the same order: forward and data gradient of every applicable shape must be
    BIT-equal to the automatically chosen one."""
    from neuroclear_amd._lib import lib, I
    L = lib()
    g = torch.Generator(device=DEV).manual_seed(B + H)
    x = torch.randn(B, C, H, W, device=DEV, generator=g)
    w = torch.randn(K, C, 4, 4, device=DEV, generator=g) / (C * 16) ** 0.5
    b = torch.randn(K, device=DEV, generator=g)
    y0 = ops.conv_fwd_raw(x, w, b, s, 1)
    dy = torch.randn(y0.shape, device=DEV, generator=g)
    dx0 = ops.conv_dgrad_raw(dy, w, x.shape, s, 1)
    n = [0, 0]
    try:
        for cfg in range(14):
            L.nc_sconv_set_cfg(I(cfg))
            try:
                y = ops.conv_fwd_raw(x, w, b, s, 1)
                assert torch.equal(y, y0), cfg
                n[0] += 1
            except Exception as e:
                assert 'does not apply' in str(e), e
            try:
                dx = ops.conv_dgrad_raw(dy, w, x.shape, s, 1)
                assert torch.equal(dx, dx0), cfg
                n[1] += 1
            except Exception as e:
                assert 'does not apply' in str(e), e
    finally:
        L.nc_sconv_set_cfg(I(-1))
    assert n[0] >= 4 and n[1] >= 4, n


@pytest.mark.parametrize('B,K,H,W', [(108, 64, 108, 108), (7, 64, 37, 53), (3, 48, 20, 21), (216, 64, 36, 36), (1, 8, 5, 4), (4, 64, 108, 108), (2, 20, 9, 31)])
def test_patchgan_first_layer_kernels(B, K, H, W):
    """patchgan_edge.hip: Conv2d(1 -> K, k 4, s 2, p 1) forward, weight + bias gradient (MFMA over the pixel axis, the bias as
    a column of ones) and data gradient (2 x 2 input blocks; below 128 workgroups of pixel blocks the few-planes kernel: 8 channel groups per
    workgroup, K not a multiple of 8 included), against torch fp32 -- Athena's batch, Apollo's one to four planes, odd planes, K < 64."""
    import torch.nn.functional as F
    from neuroclear_amd._lib import lib
    g = torch.Generator(device=DEV).manual_seed(B * 100 + H)
    x = torch.randn(B, 1, H, W, device=DEV, generator=g)
    w = torch.randn(K, 1, 4, 4, device=DEV, generator=g) * 0.25
    b = torch.randn(K, device=DEV, generator=g)
    assert lib().nc_conv_fwd_path(1, K, 1, 4, 4, 2, 1) == 8
    y = ops.conv_fwd_raw(x, w, b, 2, 1)
    ref = F.conv2d(x, w, b, stride=2, padding=1)
    assert y.shape == ref.shape
    assert float((y - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    dy = torch.randn(ref.shape, device=DEV, generator=g)
    dx = ops.conv_dgrad_raw(dy, w, x.shape, 2, 1)
    refd = torch.nn.grad.conv2d_input(x.shape, w, dy, stride=2, padding=1)
    assert float((dx - refd).abs().max()) <= 1e-5 * float(refd.abs().max())
    dw, db = ops.conv_wgrad_raw(x, dy, w.shape, 2, 1, True)
    refw = torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double(), stride=2, padding=1)
    assert float((dw.double() - refw).abs().max()) <= 2e-5 * float(refw.abs().max())
    refb = dy.double().sum((0, 2, 3))
    assert float((db.double() - refb).abs().max()) <= 2e-5 * float(refb.abs().max() + 1)
    dw2, _ = ops.conv_wgrad_raw(x, dy, w.shape, 2, 1, False)
    assert torch.equal(dw, dw2)


@pytest.mark.parametrize('N,C,S,slope', [(1, 64, 4096, 0.0), (2, 16, 2500, 0.2), (1, 128, 27 * 27 * 27, 0.0)])
def test_instnorm_backward_in_s3_form_is_the_fp32_backward_split(N, C, S, slope):
    """nc_instnorm_act_bwd_dbias_s3 (the dY of a split-operand convolution written by the norm's backward, no fp32 tensor in between)
    must store exactly the three-term form of what nc_instnorm_act_bwd_dbias stores in fp32 -- the whole-network backward relies on
    it for bit-equality with the layer-by-layer path -- and the same bias gradient."""
    from neuroclear_amd import ops
    from neuroclear_amd._lib import F, I, L_, Z, check, lib
    torch.manual_seed(6)
    x = torch.randn(N, C, S, device=DEV) * 2 + 0.5
    dy = torch.randn(N, C, S, device=DEV)
    mean = x.mean(2).reshape(-1).contiguous()
    rstd = (1.0 / (x.var(2, unbiased=False) + 1e-5).sqrt()).reshape(-1).contiguous()
    L = lib()
    wsb = L.nc_instnorm_bwd_dbias_ws_bytes(I(N * C), L_(S))
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    dx, db = torch.empty_like(x), torch.empty(C, device=DEV)
    check(L.nc_instnorm_act_bwd_dbias(ops._ptr(dy), ops._ptr(x), ops._ptr(mean), ops._ptr(rstd), F(slope), ops._ptr(dx), ops._ptr(db),
                                      I(N), I(C), L_(S), ops._ptr(ws), Z(wsb), ops._stream()), 'nc_instnorm_act_bwd_dbias')
    ref = torch.empty(N * C * S * 6, dtype=torch.uint8, device=DEV)
    check(L.nc_to_s3(ops._ptr(dx), ops._ptr(ref), I(N), I(C), L_(S), ops._stream()), 'nc_to_s3')
    out, db2 = torch.empty_like(ref), torch.empty(C, device=DEV)
    check(L.nc_instnorm_act_bwd_dbias_s3(ops._ptr(dy), ops._ptr(x), ops._ptr(mean), ops._ptr(rstd), F(slope), ops._ptr(out), ops._ptr(db2),
                                         I(N), I(C), L_(S), ops._ptr(ws), Z(wsb), ops._stream()), 'nc_instnorm_act_bwd_dbias_s3')
    assert torch.equal(out, ref) and torch.equal(db, db2)
