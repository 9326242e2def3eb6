"""GPU parity of the split-operand PatchGAN kernels k_conv_p2d (csrc/conv_p2d.hip): Conv2d 4 x 4, padding 1, stride 1 (the 256 -> 512 layer of
NLayerDiscriminator, models/networks.py:1049-1055) and stride 2 (the 64 -> 128 and 128 -> 256 layers, :1037-1046; forward over a
space-to-depth image, data gradient as four output-parity classes) -- forward and data gradient at Athena's batches, fp32 operands as exact
three-term bf16 sums, six bf16 MFMA products per fp32 product.  Criteria of tests/test_gpu_split.py: against an fp64 convolution the error
must be no worse than 1.3 x rms / 2 x max of the fp32 MFMA kernel's own (the image-staged k_sconv, reached with ops.set_conv_split(False));
plus determinism, ragged / odd planes, partial last tiles, several output-channel tiles, the bias, the batch threshold, non-finite inputs.
The stride-1 layer's weight gradient (k_wgrad_p2d, csrc/wgrad_p2d.hip) is held to the same criteria against k_swgrad."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = 'cuda'


def L():
    from neuroclear_amd._lib import lib
    return lib()


CASES = [  # B, C, K, H, W, stride
    (216, 256, 512, 13, 13, 1),   # the stride-1 layer at Athena's discriminator-loss batch (real + fake planes of a 108^3 cube)
    (108, 256, 512, 13, 13, 1),   # ... at its generator-loss batch
    (70, 128, 64, 17, 12, 1),     # ragged plane, one output-channel tile, 12,320 positions = 24.06 tiles of 512
    (33, 64, 192, 19, 21, 1),     # three output-channel tiles, planes larger than a tile
    (300, 64, 64, 7, 8, 1),       # tiny planes: a tile spans ten of them
    (216, 64, 128, 54, 54, 2),    # the stride-2 layers at Athena's batches: even planes (28 x 28 space-to-depth) ...
    (108, 128, 256, 27, 27, 2),   # ... and odd ones (the padded plane is extended to 30 x 30; the parity classes of the data gradient differ in size)
    (90, 64, 64, 21, 30, 2),      # odd x even
    (128, 128, 128, 19, 17, 2),
]


@pytest.fixture(autouse=True)
def _restore_terms():
    a, b = L().nc_get_p2d_terms(), L().nc_get_split_terms()
    yield
    L().nc_set_p2d_terms(a)
    L().nc_set_split_terms(b)


@pytest.mark.parametrize('mode', [3, 1, 2])
@pytest.mark.parametrize('case', CASES, ids=[str(c) for c in CASES])
def test_p2d_against_fp64(case, mode):
    """mode (nc_set_p2d_terms): 3 = three bf16 terms, six products (rounds 3-4); 1 (default) = the stride-1 layer on two fp16 terms of the tensor
    times a measured power of two, three products; 2 = every layer on the two-term form.  Same criteria for all."""
    from neuroclear_amd import ops
    B, C, K, H, W, st = case
    if mode == 1 and st != 1:
        pytest.skip('mode 1 changes the stride-1 layer only')
    L().nc_set_split_terms(2)
    L().nc_set_p2d_terms(mode)
    assert L().nc_get_p2d_terms() == mode
    g = torch.Generator(device=DEV).manual_seed(11)
    x = torch.randn(B, C, H, W, device=DEV, generator=g)
    w = torch.randn(K, C, 4, 4, device=DEV, generator=g) * 0.02
    b = torch.randn(K, device=DEV, generator=g)
    assert L().nc_conv2d_split_active(0, B, C, H, W, K, 4, st, 1) == 1 and L().nc_conv2d_split_active(1, B, C, H, W, K, 4, st, 1) == 1
    ys = ops.conv_fwd_raw(x, w, b, st, 1)
    dy = torch.randn(ys.shape, device=DEV, generator=g)
    ds = ops.conv_dgrad_raw(dy, w, x.shape, st, 1)
    for _ in range(2):  # run to run: bit for bit
        assert torch.equal(ys, ops.conv_fwd_raw(x, w, b, st, 1)) and torch.equal(ds, ops.conv_dgrad_raw(dy, w, x.shape, st, 1))
    prev = ops.set_conv_split(False)
    try:
        assert L().nc_conv2d_split_active(0, B, C, H, W, K, 4, st, 1) == 0
        y32 = ops.conv_fwd_raw(x, w, b, st, 1)
        d32 = ops.conv_dgrad_raw(dy, w, x.shape, st, 1)
    finally:
        ops.set_conv_split(prev)
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=st, padding=1)
    refd = torch.nn.grad.conv2d_input(x.shape, w.double(), dy.double(), stride=st, padding=1)

    def err(y, r):
        s = r.pow(2).mean().sqrt().item()
        e = y.double() - r
        return e.abs().max().item() / s, e.pow(2).mean().sqrt().item() / s
    for name, got, got32, r in (('fwd', ys, y32, ref), ('dgrad', ds, d32, refd)):
        m32, r32 = err(got32, r)
        ms, rs = err(got, r)
        print(case, name, 'fp32 max %.2e rms %.2e | split max %.2e rms %.2e' % (m32, r32, ms, rs))
        assert rs <= 1.3 * r32 + 2e-8 and rs < 4e-7, (name, rs, r32)
        assert ms <= 2.0 * m32 + 2e-7, (name, ms, m32)


WCASES = [  # B, C, K, H, W (stride 1)
    (216, 256, 512, 13, 13),   # the 256 -> 512 layer at Athena's discriminator-loss batch: 64 (k, c) pairs x 4 workgroups, 44 tiles of 5 planes
    (108, 256, 512, 13, 13),
    (70, 128, 64, 17, 12),     # ragged plane; 4 pairs x 64 workgroups: a workgroup's share starts in the middle of a tile
    (33, 64, 192, 19, 21),     # 33 planes: the last tile is partly beyond the batch (zero planes of the converted tensors)
    (300, 64, 64, 7, 8),       # tiny planes
    (128, 32, 64, 13, 13),     # one c-tile
    (16, 32, 128, 40, 41),     # rows longer than a k-step
]


@pytest.mark.parametrize('case', WCASES, ids=[str(c) for c in WCASES])
def test_p2d_wgrad_against_fp64(case):
    from neuroclear_amd import ops
    B, C, K, H, W = case
    g = torch.Generator(device=DEV).manual_seed(17)
    x = torch.randn(B, C, H, W, device=DEV, generator=g)
    dy = torch.randn(B, K, H - 1, W - 1, device=DEV, generator=g)
    assert L().nc_conv2d_split_active(2, B, C, H, W, K, 4, 1, 1) == 1
    dw, db = ops.conv_wgrad_raw(x, dy, (K, C, 4, 4), 1, 1, True)
    for _ in range(2):
        assert torch.equal(dw, ops.conv_wgrad_raw(x, dy, (K, C, 4, 4), 1, 1, False)[0])
    prev = ops.set_conv_split(False)
    try:
        assert L().nc_conv2d_split_active(2, B, C, H, W, K, 4, 1, 1) == 0
        dw32 = ops.conv_wgrad_raw(x, dy, (K, C, 4, 4), 1, 1, False)[0]
    finally:
        ops.set_conv_split(prev)
    ref = torch.nn.grad.conv2d_weight(x.double(), (K, C, 4, 4), dy.double(), stride=1, padding=1)

    def err(y, r):
        s = r.pow(2).mean().sqrt().item()
        e = y.double() - r
        return e.abs().max().item() / s, e.pow(2).mean().sqrt().item() / s
    m32, r32 = err(dw32, ref)
    ms, rs = err(dw, ref)
    print(case, 'wgrad fp32 max %.2e rms %.2e | split max %.2e rms %.2e' % (m32, r32, ms, rs))
    assert rs <= 1.3 * r32 + 2e-8 and ms <= 2.0 * m32 + 2e-7, (ms, rs, m32, r32)
    assert torch.allclose(db, dy.sum((0, 2, 3)), rtol=1e-4, atol=1e-3)


def test_p2d_wgrad_zero_dy_planes_and_scope():
    """Planes whose dy is zero contribute nothing, whatever the input holds there (finite); the stride-2 layers, small batches and other
    channel counts stay on k_swgrad."""
    from neuroclear_amd import ops
    B, C, K, H, W = 96, 64, 64, 13, 13
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn(B, C, H, W, device=DEV, generator=g)
    dy = torch.randn(B, K, H - 1, W - 1, device=DEV, generator=g)
    dy[40:] = 0
    x2 = x.clone()
    x2[40:] = 1e30
    a = ops.conv_wgrad_raw(x, dy, (K, C, 4, 4), 1, 1, False)[0]
    b = ops.conv_wgrad_raw(x2, dy, (K, C, 4, 4), 1, 1, False)[0]
    assert torch.equal(a, b)
    ref = torch.nn.grad.conv2d_weight(x[:40].double(), (K, C, 4, 4), dy[:40].double(), stride=1, padding=1)
    assert float((a.double() - ref).abs().max() / ref.abs().max()) < 2e-6
    q = L().nc_conv2d_split_active
    assert q(2, 216, 256, 13, 13, 512, 4, 1, 1) == 1
    assert q(2, 216, 128, 27, 27, 256, 4, 2, 1) == 0 and q(2, 4, 256, 13, 13, 512, 4, 1, 1) == 0
    assert q(2, 216, 48, 13, 13, 512, 4, 1, 1) == 0 and q(2, 216, 256, 13, 13, 96, 4, 1, 1) == 0


def test_p2d_threshold_and_scope():
    """Small batches (Apollo's 1-4 planes per discriminator), the one-channel first layer, the one-channel head and other kernel sizes stay
    where they were."""
    q = L().nc_conv2d_split_active
    assert q(0, 216, 256, 13, 13, 512, 4, 1, 1) == 1 and q(1, 216, 256, 13, 13, 512, 4, 1, 1) == 1
    assert q(0, 216, 128, 27, 27, 256, 4, 2, 1) == 1 and q(1, 216, 64, 54, 54, 128, 4, 2, 1) == 1
    assert q(0, 4, 256, 13, 13, 512, 4, 1, 1) == 0          # 576 positions
    assert q(0, 4, 64, 54, 54, 128, 4, 2, 1) == 0
    assert q(0, 216, 1, 108, 108, 64, 4, 2, 1) == 0         # the first layer: k_pg1
    assert q(0, 216, 512, 12, 12, 1, 4, 1, 1) == 0          # the head
    assert q(0, 216, 256, 13, 13, 512, 3, 1, 1) == 0
    assert q(0, 216, 48, 13, 13, 512, 4, 1, 1) == 0 and q(1, 216, 256, 13, 13, 96, 4, 1, 1) == 0   # channels % 64


def test_p2d_nonfinite_inputs_follow_the_split_rule():
    """An inf / NaN input element makes every output it touches NaN and leaves every other output bit-identical (the propagation rule of
    the split-operand kernels, tests/test_gpu_split.py); a value beyond the largest finite bf16 still splits exactly."""
    from neuroclear_amd import ops
    B, C, K, H, W = 80, 64, 64, 13, 13
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(B, C, H, W, device=DEV, generator=g)
    w = torch.randn(K, C, 4, 4, device=DEV, generator=g) * 0.02
    y = ops.conv_fwd_raw(x, w, None, 1, 1)
    x2 = x.clone()
    x2[5, 7, 6, 6] = float('inf')
    x2[40, 3, 0, 12] = float('nan')
    y2 = ops.conv_fwd_raw(x2, w, None, 1, 1)
    touched = torch.zeros_like(y, dtype=torch.bool)
    # output (y, x) reads input rows y - 1 .. y + 2: input row i is touched by outputs i - 2 .. i + 1 (inside the 12 x 12 output plane)
    touched[5, :, 4:8, 4:8] = True      # input (6, 6)
    touched[40, :, 0:2, 10:12] = True   # input (0, 12): rows 0..1, columns 10..11
    assert bool(torch.isnan(y2[touched]).all()) and torch.equal(y2[~touched], y[~touched])
    x3 = x.clone()
    x3[9, 1, 2, 2] = 3.39e38  # finite, above the largest finite bf16
    y3 = ops.conv_fwd_raw(x3, w, None, 1, 1)
    ref = F.conv2d(x3.double(), w.double(), padding=1)
    assert bool(torch.isfinite(y3).all()) and float((y3.double() - ref).abs().max() / ref.abs().max()) < 1e-6
