"""GPU parity tests, network level (-m gpu): the HIP path (through the C ABI) against
  * the golden vectors generated from the reference itself (tests/golden/*.npz), and
  * the oracle (oracle/*.py, CPU) on the same seeded inputs.
Stated fp32 tolerances: generator outputs (after sigmoid) |err| <= 2e-5; linear-chain outputs 2e-4 relative.
Gradients through ReLU / MaxPool are piecewise: an activation within ~1e-7 of zero takes the other branch under a
different fp32 summation order and moves ONE element of the upstream gradient by O(1) -- relative L2 effect
~1/sqrt(elements) per flip (measured 2e-3..7e-3 at 32^3 against an fp64 reference, for the direct AND the MFMA path,
tools/debug_unet.py).  Gradients are therefore judged in the L2 norm at 2e-2; Apollo losses at 2e-5 relative on the
first step and 5e-3 after one Adam update (Adam's first step moves every weight by +-lr whatever the gradient's size,
so noise-level gradients pick their sign at random on any two implementations)."""
import hashlib
import os
from argparse import Namespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from neuroclear_amd import ops  # noqa: E402
from neuroclear_amd.models import networks  # noqa: E402
from neuroclear_amd.util import seed as S  # noqa: E402

DEV = 'cuda'


def G(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def rnd(seed, shape):
    return np.random.default_rng(int(seed)).random(tuple(int(s) for s in shape), dtype=np.float32)


def load(net, spec, seed):
    net.load_state_dict(S.state_dict_from_seed(spec, seed, DEV))
    return net


def rel2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.sqrt(((a - b) ** 2).sum()) / max(np.sqrt((b ** 2).sum()), 1e-30))


def relmax(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / max(np.abs(b).max(), 1e-30))


def check_grads(g, net, tol):
    for i, (k, p) in enumerate(net.named_parameters()):
        gr = p.grad.detach().cpu().numpy().ravel()
        l2 = np.sqrt((gr.astype(np.float64) ** 2).sum())
        # biases in front of InstanceNorm(affine=False) have an exactly-zero true gradient: both sides
        # hold rounding noise of ~1e-8 there, hence the absolute floor
        assert abs(l2 - g['g_l2'][i]) <= tol * g['g_l2'][i] + 1e-6, (k, l2, g['g_l2'][i])
        idx = np.random.default_rng([77, i]).integers(0, gr.size, size=8)
        np.testing.assert_allclose(gr[idx], g['g_samp'][i], rtol=2e-2, atol=5 * tol * l2 / np.sqrt(gr.size) + 1e-7,
                                   err_msg=k)


@pytest.mark.parametrize('size', [16, 32])
def test_unet_deconv(golden_dir, size):
    g = G(golden_dir, 'unet_deconv_%d.npz' % size)
    net = load(networks.define_G(1, 1, 64, 'unet_deconv', 'instance', False, 'kaiming', 0.02, [0]),
               S.unet_deconv_spec(), int(g['seed']))
    x = torch.from_numpy(rnd(g['x_seed'], (1, 1, size, size, size))).to(DEV).requires_grad_(True)
    y = net(x)
    assert float(np.abs(y.detach().cpu().numpy() - g['y']).max()) < 2e-5
    with torch.no_grad():  # whole-network C entry point (nc_unet_deconv_fwd)
        yf = net(x.detach())
    assert float((yf - y.detach()).abs().max()) < 1e-5
    r = torch.from_numpy(rnd(g['r_seed'], y.shape)).to(DEV)
    (y * r).mean().backward()
    # a ReLU/maxpool decision that flips on a 1e-7 difference moves isolated voxels: judge dx in the L2 norm
    assert rel2(x.grad.cpu().numpy(), g['dx']) < 2e-2
    check_grads(g, net, 2e-2)


@pytest.mark.parametrize('size', [16, 24])
def test_deep_linear(golden_dir, size):
    g = G(golden_dir, 'deep_linear_%d.npz' % size)
    net = load(networks.define_G(1, 1, 64, 'deep_linear_gen', 'instance', False, 'kaiming', 0.02, [0]),
               S.deep_linear_spec(), int(g['seed']))
    x = torch.from_numpy(rnd(g['x_seed'], (1, 1, size, size, size))).to(DEV).requires_grad_(True)
    y = net(x)
    assert relmax(y.detach().cpu().numpy(), g['y']) < 2e-4
    r = torch.from_numpy(rnd(g['r_seed'], y.shape)).to(DEV)
    (y * r).mean().backward()
    assert relmax(x.grad.cpu().numpy(), g['dx']) < 1e-3
    check_grads(g, net, 1e-3)


@pytest.mark.parametrize('tag', ['2d_36', '2d_108', '2d_36_b3', '3d_36'])
def test_patchgan(golden_dir, tag):
    g = G(golden_dir, 'patchgan_%s.npz' % tag)
    dim = int(g['dim'])
    net = load(networks.define_D(1, 64, 'basic', 3, 'instance', 'kaiming', 0.02, False, [0], dimension=dim),
               S.patchgan_spec(dim), int(g['seed']))
    x = torch.from_numpy(rnd(g['x_seed'], g['shape'])).to(DEV).requires_grad_(True)
    y = net(x)
    assert relmax(y.detach().cpu().numpy(), g['y']) < 5e-4
    r = torch.from_numpy(rnd(g['r_seed'], y.shape)).to(DEV)
    (y * r).mean().backward()
    # LeakyReLU has a kink but no dead branch: a decision that differs changes one element's slope 0.2 <-> 1, not the whole path --
    # measured 6e-7 .. 1e-6 against fp64 (tests/test_gpu_grad_fp64.py); the bound for ReLU / max-pool networks stays 2e-2
    assert rel2(x.grad.cpu().numpy(), g['dx']) < 1e-3
    check_grads(g, net, 1e-3)


def test_blocks(golden_dir):
    norm = networks.get_norm_layer('instance', 3)
    for tag in ('dc_1_64', 'dc_64_128'):
        g = G(golden_dir, 'block_%s.npz' % tag)
        cin, cout, size, seed = int(g['cin']), int(g['cout']), int(g['size']), int(g['seed'])
        blk = networks.double_conv(cin, cout, 3, 1, 1, norm, 3).to(DEV)
        spec = [('convolution.0.weight', (cout, cin, 3, 3, 3)), ('convolution.0.bias', (cout,)),
                ('convolution.3.weight', (cout, cout, 3, 3, 3)), ('convolution.3.bias', (cout,))]
        load(blk, spec, seed)
        x = torch.from_numpy(rnd(100 + seed, (1, cin, size, size, size))).to(DEV).requires_grad_(True)
        y = blk(x)
        assert relmax(y.detach().cpu().numpy(), g['y']) < 1e-4, tag
        r = torch.from_numpy(rnd(200 + seed, y.shape)).to(DEV)
        (y * r).mean().backward()
        assert relmax(x.grad.cpu().numpy(), g['dx']) < 1e-3, tag
        for key, conv in (('dw0', blk.convolution[0]), ('dw3', blk.convolution[3])):
            a = conv.weight.grad.cpu().numpy().ravel()
            idx = np.random.default_rng(55).integers(0, a.size, size=min(4096, a.size))
            got = np.concatenate([[np.sqrt((a.astype(np.float64) ** 2).sum()), a.astype(np.float64).sum()], a[idx]])
            assert abs(got[0] - g[key][0]) < 1e-3 * g[key][0], (tag, key)
            assert relmax(got[2:], g[key][2:]) < 2e-3, (tag, key)
    g = G(golden_dir, 'block_in_relu_bigmean.npz')
    x = torch.from_numpy(rnd(114, (1, 16, 20, 20, 20)) * 0.05 + 100.0).to(DEV).requires_grad_(True)
    y = ops.instance_norm_act(x, 0.0)
    # inputs of magnitude 100 carry an fp32 representation error of ~4e-6 = 3e-4 sigma: both sides are noisy
    assert float(np.abs(y.detach().cpu().numpy() - g['y']).max()) < 5e-3


def test_convT_block_golden(golden_dir):
    """ConvTranspose3d(128, 64, 2, 2) as built at networks.py:500,503, against the reference's own forward / input
    gradient / weight + bias gradients (block_convT_128_64.npz)."""
    g = G(golden_dir, 'block_convT_128_64.npz')
    ct = networks.ConvTranspose(128, 64, 2, 2, 3).to(DEV)
    load(ct, [('weight', (128, 64, 2, 2, 2)), ('bias', (64,))], int(g['seed']))
    x = torch.from_numpy(rnd(113, (1, 128, 6, 6, 6))).to(DEV).requires_grad_(True)
    y = ct(x)
    assert y.shape == (1, 64, 12, 12, 12)
    assert relmax(y.detach().cpu().numpy(), g['y']) < 2e-5
    r = torch.from_numpy(rnd(213, y.shape)).to(DEV)
    (y * r).mean().backward()
    assert relmax(x.grad.cpu().numpy(), g['dx']) < 2e-5
    assert relmax(ct.bias.grad.cpu().numpy(), g['db']) < 2e-5
    a = ct.weight.grad.cpu().numpy().ravel()
    idx = np.random.default_rng(55).integers(0, a.size, size=min(4096, a.size))
    got = np.concatenate([[np.sqrt((a.astype(np.float64) ** 2).sum()), a.astype(np.float64).sum()], a[idx]])
    assert abs(got[0] - g['dw'][0]) < 1e-4 * g['dw'][0]
    assert relmax(got[2:], g['dw'][2:]) < 1e-4


def test_init_net_kaiming():
    """init_net / init_weights('kaiming') (networks.py:88-137): every module whose class name contains 'Conv' --
    ConvTranspose included -- gets kaiming_normal_(a=0, fan_in) weights (std = sqrt(2 / fan_in), fan_in = the torch rule
    size(1) * receptive field) and a zero bias; the net lands on gpu_ids[0]; unknown init types raise."""
    torch.manual_seed(3)
    net = networks.define_G(1, 1, 64, 'unet_deconv', 'instance', False, 'kaiming', 0.02, [0])
    seen = 0
    for name, m in net.named_modules():
        if not hasattr(m, 'weight') or 'Conv' not in m.__class__.__name__:
            continue
        w = m.weight.detach()
        assert w.is_cuda and w.device.index == 0
        fan_in = w.shape[1] * w[0, 0].numel()
        if w.numel() >= 4096:
            std = float(w.double().std())
            assert abs(std / (2.0 / fan_in) ** 0.5 - 1.0) < 0.05, (name, std, fan_in)
            assert abs(float(w.double().mean())) < 0.1 * (2.0 / fan_in) ** 0.5
        assert m.bias is None or float(m.bias.abs().max()) == 0.0, name
        seen += 1
    assert seen == 14  # 10 3^3 convs + 2 transposed + 2 pointwise
    d = networks.define_D(1, 64, 'basic', 3, 'instance', 'normal', 0.02, False, [0], dimension=2)
    w = d.model[2].weight.detach()
    assert abs(float(w.double().std()) / 0.02 - 1.0) < 0.05
    with pytest.raises(NotImplementedError):
        networks.define_G(1, 1, 64, 'unet_deconv', 'instance', False, 'no_such_init', 0.02, [0])


def _apollo_opt():
    return Namespace(gpu_ids=[0], isTrain=True, image_dimension=3, checkpoints_dir='/tmp/nc_ckpt', name='t',
                     preprocess='none', gan_mode='lsgan', randomize_projection_depth=True, projection_depth=10,
                     min_projection_depth=2, lambda_plane=[1, 1, 1], lambda_A=5.0, input_nc=1, output_nc=1, ngf=64,
                     ndf=64, netG='unet_deconv', netG_B='deep_linear_gen', netD='basic', n_layers_D=3,
                     norm='instance', no_dropout=True, init_type='kaiming', init_gain=0.02, lr=1e-4, beta1=0.1,
                     direction='AtoB', model='axial_to_lateral_gan_apollo')


APOLLO_NETS = ['G_A', 'G_B', 'D_A_axial', 'D_A_lateral', 'D_B_axial', 'D_B_lateral']


@pytest.fixture
def three_term():
    """The three-term bf16 form of the split-operand convolutions (rounds 2-3) for one test; the default is the two-term fp16 form."""
    from neuroclear_amd._lib import lib
    t = lib().nc_get_split_terms()
    lib().nc_set_split_terms(3)
    yield
    lib().nc_set_split_terms(t)


def test_apollo_step_three_term(golden_dir, three_term, monkeypatch):
    test_apollo_step(golden_dir, 'apollo_step_36.npz', True, monkeypatch)


@pytest.mark.parametrize('fname,d_streams', [('apollo_step_36.npz', True), ('apollo_step_24_b2.npz', True),
                                             ('apollo_step_24_b2.npz', False), ('apollo_step_24_vanilla.npz', True),
                                             ('apollo_step_24_wgangp.npz', True)])
def test_apollo_step(golden_dir, fname, d_streams, monkeypatch):
    """One + one optimize_parameters() against the reference's own losses; the batch-2 fixture pins the per-plane batch
    split of the batched discriminator passes (each LSGAN mean runs over the whole batch of ONE plane)."""
    from neuroclear_amd.models import create_model
    from neuroclear_amd.models.axial_to_lateral_gan_apollo_model import AxialToLateralGANApolloModel
    monkeypatch.setattr(AxialToLateralGANApolloModel, '_d_streams_on', d_streams)
    g = G(golden_dir, fname)
    size, batch = int(g['size']), int(g['batch']) if 'batch' in g else 1
    opt = _apollo_opt()
    opt.gan_mode = str(g['gan_mode']) if 'gan_mode' in g else 'lsgan'  # (--gan_mode vanilla | wgangp: networks.py:252-319)
    model = create_model(opt)
    specs = [S.unet_deconv_spec(), S.deep_linear_spec()] + [S.patchgan_spec(2)] * 4
    for i, (n, sp) in enumerate(zip(APOLLO_NETS, specs)):
        load(getattr(model, 'net' + n), sp, int(g['net_seed0']) + i)
    before = {n: [p.detach().clone() for p in getattr(model, 'net' + n).parameters()] for n in APOLLO_NETS}
    real = torch.from_numpy(rnd(g['real_seed'], (batch, 1, size, size, size)))
    np.random.seed(int(g['step_seed']))
    names = [str(s) for s in g['loss_names']]
    for it in range(2):
        model.set_input({'A': real, 'A_paths': 'x'})
        model.optimize_parameters()
        L = model.get_current_losses()
        got = np.array([L[k] for k in names])
        print(it, dict(zip(names, got)), g['losses'][it])
        # ('wgangp': every discriminator loss is a DIFFERENCE of two prediction means of size ~0.1: an absolute floor next to the relative bound)
        np.testing.assert_allclose(got, g['losses'][it], rtol=2e-5 if it == 0 else 5e-3, atol=1e-5 if 'wgan' in opt.gan_mode else 0,
                                   err_msg='step %d' % it)
        if it == 0:
            assert float(np.abs(model.fake.detach().cpu().numpy() - g['fake0']).max()) < 2e-5
            assert relmax(model.rec.detach().cpu().numpy(), g['rec0']) < 2e-4
    for n in APOLLO_NETS:
        ps = list(getattr(model, 'net' + n).parameters())
        upd = np.array([float((a.detach() - b).double().norm()) for a, b in zip(ps, before[n])])
        # 1-D parameters are mostly biases in front of InstanceNorm (true gradient 0): Adam turns their rounding
        # noise into +-lr-sized moves that no two implementations share -- only weight tensors are compared
        sel = np.array([a.dim() > 1 for a in ps])
        np.testing.assert_allclose(upd[sel], g['upd_' + n][sel], rtol=5e-2, err_msg=n)


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize('tag', ['100_32_4_4', 'ragged_70_50_61', 'u8_64_24_4_2'])
def test_dice_assemble_bit_exact(golden_dir, tag):
    from neuroclear_amd.data.diceImage_dataset import DiceImageDataSet
    from neuroclear_amd.util.assemble_dice import Assemble_Dice
    g = G(golden_dir, 'dice_%s.npz' % tag)
    shape, R, ov, b = tuple(int(v) for v in g['shape']), int(g['roi']), int(g['overlap']), int(g['border'])
    dtype = np.dtype(str(g['dtype']))
    vol = S.random_volume(int(g['vol_seed']), shape, dtype)
    opt = Namespace(dice_size=[R, R, R], overlap=ov, border_cut=b, gpu_ids=[0], skip_real=False,
                    data_type='uint8' if dtype == np.uint8 else 'uint16', histogram_match=False,
                    normalize_intensity=False)
    ds = DiceImageDataSet(opt, volume=vol)
    assert ds.size() == tuple(g['padded']) and ds.shape() == tuple(g['steps']) and len(ds) == int(g['n'])
    n = len(ds)
    np.testing.assert_array_equal(ds[0]['A'].cpu().numpy(), g['first'])
    np.testing.assert_array_equal(ds[n // 2]['A'].cpu().numpy(), g['mid'])
    np.testing.assert_array_equal(ds[n - 1]['A'].cpu().numpy(), g['last'])
    asm_id = Assemble_Dice(opt, shape)
    asm_pos = Assemble_Dice(opt, shape)
    for i in range(n):
        a = ds[i]['A'].unsqueeze(0)
        asm_id.addToStack(dict(real=a, fake=a))
        # same arithmetic as the generator script: fp32 mul, then add of a double constant rounded to fp32 tensor op
        asm_pos.addToStack(dict(real=a, fake=a * 0.5 + (i % 7) * 1e-3))
    asm_id.assemble_all()
    asm_pos.assemble_all()
    assert _sha(asm_id.getDict()['fake']) == str(g['sha_identity'])
    assert _sha(asm_pos.getDict()['real']) == str(g['sha_real'])
    assert int(np.abs(asm_id.getDict()['fake'].astype(np.int64) - vol).max()) <= 1
    assert _sha(asm_pos.getDict()['fake']) == str(g['sha_pos'])
    if g['out_pos'].size:
        np.testing.assert_array_equal(asm_pos.getDict()['fake'], g['out_pos'])


def test_error_behaviour():
    from neuroclear_amd._lib import NcError
    with pytest.raises(NotImplementedError):
        networks.define_G(1, 1, 64, 'no_such_net', 'instance')
    with pytest.raises(NotImplementedError):
        networks.define_D(1, 64, 'no_such_net', norm='instance')
    with pytest.raises(NotImplementedError):
        networks.GANLoss('bogus')
    net = networks.define_G(1, 1, 64, 'unet_deconv', 'instance', False, 'kaiming', 0.02, [0])
    with pytest.raises(ValueError):
        net(torch.zeros(1, 1, 10, 12, 12, device=DEV))
    with pytest.raises(NcError):  # CPU tensors are refused: no fallback
        ops.conv(torch.zeros(1, 1, 4, 4, 4), torch.zeros(1, 1, 3, 3, 3), None, 1, 1)


ATHENA_NETS = ['G_A', 'G_B', 'D_A_yz', 'D_A_xy', 'D_A_xz', 'D_B_yz', 'D_B_xy', 'D_B_xz']


def test_all_slices_op():
    vol = torch.rand(2, 3, 5, 6, 7, device=DEV).requires_grad_(True)
    for axis in range(3):
        ref = vol.movedim(axis + 2, 1).reshape((2 * vol.shape[axis + 2], 3) + tuple(
            s for i, s in enumerate(vol.shape[2:]) if i != axis))
        r = torch.rand_like(ref)
        (gr,) = torch.autograd.grad((ref * r).sum(), vol)
        v2 = vol.detach().clone().requires_grad_(True)
        out = ops.volume_all_slices(v2, axis)
        (h,) = torch.autograd.grad((out * r).sum(), v2)
        assert torch.equal(out, ref.contiguous()) and torch.equal(h, gr), axis


def test_athena_step_three_term(golden_dir, three_term):
    test_athena_step(golden_dir)


def test_athena_step(golden_dir):
    from neuroclear_amd.models import create_model
    g = G(golden_dir, 'athena_step_36.npz')
    size = int(g['size'])
    opt = _apollo_opt()
    opt.model = 'axial_to_lateral_gan_athena'
    opt.conversion_plane = ['yz', 'xy']
    opt.pool_size = 50
    model = create_model(opt)
    specs = [S.unet_deconv_spec(), S.deep_linear_spec()] + [S.patchgan_spec(2)] * 6
    for i, (n, sp) in enumerate(zip(ATHENA_NETS, specs)):
        load(getattr(model, 'net' + n), sp, int(g['net_seed0']) + i)
    before = {n: [p.detach().clone() for p in getattr(model, 'net' + n).parameters()] for n in ATHENA_NETS}
    real = torch.from_numpy(rnd(g['real_seed'], (1, 1, size, size, size)))
    names = [str(s) for s in g['loss_names']]
    for it in range(2):
        model.set_input({'A': real, 'A_paths': 'x'})
        model.optimize_parameters()
        L = model.get_current_losses()
        got = np.array([L[k] for k in names])
        print(it, got, g['losses'][it])
        np.testing.assert_allclose(got, g['losses'][it], rtol=2e-5 if it == 0 else 5e-3, err_msg='step %d' % it)
    for n in ATHENA_NETS:
        ps = list(getattr(model, 'net' + n).parameters())
        upd = np.array([float((a.detach() - b).double().norm()) for a, b in zip(ps, before[n])])
        sel = np.array([a.dim() > 1 for a in ps])
        np.testing.assert_allclose(upd[sel], g['upd_' + n][sel], rtol=5e-2, err_msg=n)


@pytest.mark.parametrize('size,batch', [(36, 1), (24, 2)])
def test_athena_shared_fake_pass_matches_two_passes(size, batch, monkeypatch):
    """Athena's discriminator loss re-evaluates every discriminator on the planes of `fake` / `rec` that the generator loss
    just sent through the same weights (athena:240-260, :190-238).  The product path keeps that pass (ops.PatchGANShare) and
    runs only the real planes the second time; with NC_D_REUSE=0 both passes run.  Same arithmetic per plane either way:
    the losses of the first step must agree to fp32 rounding, and the share must actually be used."""
    from neuroclear_amd.models import create_model
    from neuroclear_amd.models.axial_to_lateral_gan_athena_model import AxialToLateralGANAthenaModel as M
    used = []
    orig = ops.patchgan_join_real
    monkeypatch.setattr(ops, 'patchgan_join_real', lambda *a, **k: (used.append(1), orig(*a, **k))[1])
    res = {}
    for reuse in (True, False):
        monkeypatch.setattr(M, '_reuse_on', reuse)
        opt = _apollo_opt()
        opt.model = 'axial_to_lateral_gan_athena'
        opt.conversion_plane = ['yz', 'xy']
        opt.pool_size = 50
        model = create_model(opt)
        specs = [S.unet_deconv_spec(), S.deep_linear_spec()] + [S.patchgan_spec(2)] * 6
        for i, (n, sp) in enumerate(zip(ATHENA_NETS, specs)):
            load(getattr(model, 'net' + n), sp, 500 + i)
        real = torch.from_numpy(rnd(77, (batch, 1, size, size, size)))
        losses = []
        for it in range(2):
            model.set_input({'A': real, 'A_paths': 'x'})
            model.optimize_parameters()
            losses.append(dict(model.get_current_losses()))
        params = torch.cat([p.detach().reshape(-1) for n in ATHENA_NETS for p in getattr(model, 'net' + n).parameters()])
        res[reuse] = (losses, params.clone())
        if reuse:
            assert len(used) == 12, len(used)  # six discriminators x two steps
    assert len(used) == 12
    for it in range(2):
        for k in res[True][0][it]:
            # step 2 runs on Adam-updated weights: Adam turns rounding-level differences of near-zero gradients into
            # lr-sized steps (the same allowance as the golden test of the step, test_athena_step)
            np.testing.assert_allclose(res[True][0][it][k], res[False][0][it][k], rtol=1e-5 if it == 0 else 2e-3,
                                       err_msg='%d %s' % (it, k))
    a, b = res[True][1].double(), res[False][1].double()
    assert float((a - b).norm() / b.norm()) < 1e-3


@pytest.mark.parametrize('model_name', ['axial_to_lateral_gan_apollo', 'axial_to_lateral_gan_athena'])
def test_step_with_spectral_norm_discriminators(model_name):
    """--netD basic_SN through a whole optimisation step of both models: every plane goes through such a discriminator in its
    own call (each call moves the power-iteration vectors, as in the reference), so neither the batched pass nor Athena's shared
    fake pass applies -- two steps must run, with finite losses and moving discriminator weights."""
    from neuroclear_amd.models import create_model
    opt = _apollo_opt()
    opt.model = model_name
    opt.netD = 'basic_SN'
    if 'athena' in model_name:
        opt.conversion_plane = ['yz', 'xy']
        opt.pool_size = 50
    torch.manual_seed(3)
    np.random.seed(3)
    model = create_model(opt)
    dname = [n for n in model.model_names if n.startswith('D_')][0]
    before = [p.detach().clone() for p in getattr(model, 'net' + dname).parameters()]
    real = torch.from_numpy(rnd(5, (1, 1, 24, 24, 24)))
    for _ in range(2):
        model.set_input({'A': real, 'A_paths': 'x'})
        model.optimize_parameters()
    L = model.get_current_losses()
    assert all(np.isfinite(v) for v in L.values()), L
    after = list(getattr(model, 'net' + dname).parameters())
    assert any(float((a.detach() - b).abs().max()) > 0 for a, b in zip(after, before))
    if 'athena' in model_name:
        assert model._shares == {}


def test_patchgan_share_guards():
    """ops.PatchGANShare is only honoured for exactly the planes it holds and the weights that produced them: another source
    tensor, another slicing axis, an in-place change of the source or a parameter update (generation bump) make
    share_matches false (the model then runs the ordinary two-half batch), and a discriminator whose parameters require
    grad is never shared."""
    net = load(networks.define_D(1, 64, 'n_layers', 3, 'instance', 'kaiming', 0.02, False, [0], dimension=2),
               S.patchgan_spec(2), 5)
    from neuroclear_amd.models.axial_to_lateral_gan_apollo_model import FlatAdam
    FlatAdam(net.parameters(), lr=1e-4, betas=(0.5, 0.999))  # the parameters become views of one flat buffer, as in the models
    vol = torch.from_numpy(rnd(3, (1, 1, 24, 24, 24))).to(DEV).requires_grad_(True)
    fake = vol * 1.0
    planes = ops.volume_all_slices(fake, 1)
    assert not net.can_share(planes)  # parameters still require grad
    for p in net.parameters():
        p.requires_grad_(False)
    assert net.can_share(planes)
    share = ops.PatchGANShare()
    y = net.forward_fake_half(planes, share, fake, 1)
    assert y.shape[0] == planes.shape[0]
    assert net.share_matches(share, fake, 1)
    assert not net.share_matches(share, fake, 2)          # other axis
    assert not net.share_matches(share, fake.clone(), 1)  # other tensor
    ref = net(planes)                                     # the ordinary pass gives the same predictions
    assert torch.equal(ref, y)
    ops.bump_param_generation(ops._pack_params(list(net.parameters())))
    assert not net.share_matches(share, fake, 1)          # "optimizer step" since the pass
    share.release()
    assert not net.share_matches(share, fake, 1)


def test_train_onecube_and_checkpoint_roundtrip(tmp_path):
    """Entry-script level: two iterations of train_onecube on a synthetic volume (option parsing, on-device crops,
    schedulers, checkpoint files with the reference's names), then the generator is reloaded through TestModel /
    load_networks and must reproduce the training model's output bit for bit."""
    from neuroclear_amd import train_onecube
    from neuroclear_amd.models import create_model
    from neuroclear_amd.options import TestOptions
    d = tmp_path / 'data'
    d.mkdir()
    np.save(str(d / 'vol.npy'), S.random_volume(21, 48))
    ck = str(tmp_path / 'ckpt')
    argv = ['--dataroot', str(d), '--checkpoints_dir', ck, '--name', 'm', '--model', 'axial_to_lateral_gan_apollo',
            '--preprocess', 'randomcrop_randomflip_addColorChannel_addBatchChannel', '--crop_size', '36', '36', '36',
            '--gan_mode', 'lsgan', '--init_type', 'kaiming', '--norm', 'instance', '--lambda_A', '5',
            '--lambda_plane', '1', '1', '1', '--lr_policy', 'constant', '--randomize_projection_depth',
            '--projection_depth', '10', '--save_by_iter', '--save_latest_freq', '2', '--print_freq', '1',
            '--max_iters', '2', '--gpu_ids', '0']
    np.random.seed(3)
    import random
    random.seed(3)
    model = train_onecube.main(argv)
    assert os.path.exists(os.path.join(ck, 'm', 'iter_2_net_G_A.pth'))
    assert os.path.exists(os.path.join(ck, 'm', 'iter_2_net_D_B_axial.pth'))
    sd = torch.load(os.path.join(ck, 'm', 'iter_2_net_G_A.pth'))
    assert list(sd.keys()) == [k for k, _ in S.unet_deconv_spec()]
    topt = TestOptions().parse(['--dataroot', str(d), '--checkpoints_dir', ck, '--name', 'm', '--model_suffix', '_A',
                                '--load_iter', '2', '--gpu_ids', '0', '--init_type', 'kaiming', '--no_dropout'])
    topt.continue_train = False
    tm = create_model(topt)
    tm.setup(topt)
    x = torch.rand(1, 1, 36, 36, 36, device=DEV)
    tm.set_input({'A': x, 'A_paths': 'x'})
    tm.test()
    with torch.no_grad():
        ref = model.netG_A(x)
    assert torch.equal(tm.fake, ref)


# ---- widening row: unet_vanilla, pixel / n_layers discriminators, the Dryops step

@pytest.mark.parametrize('size', [16, 24])
def test_unet_vanilla(golden_dir, size):
    g = G(golden_dir, 'unet_vanilla_%d.npz' % size)
    net = load(networks.define_G(1, 1, 64, 'unet_vanilla', 'instance', False, 'kaiming', 0.02, [0]),
               S.unet_vanilla_spec(), int(g['seed']))
    x = torch.from_numpy(rnd(g['x_seed'], (1, 1, size, size, size))).to(DEV).requires_grad_(True)
    y = net(x)
    assert float(np.abs(y.detach().cpu().numpy() - g['y']).max()) < 2e-5
    with torch.no_grad():
        assert float((net(x.detach()) - y.detach()).abs().max()) < 1e-5
    r = torch.from_numpy(rnd(g['r_seed'], y.shape)).to(DEV)
    (y * r).mean().backward()
    assert rel2(x.grad.cpu().numpy(), g['dx']) < 2e-2
    check_grads(g, net, 2e-2)
    with pytest.raises(ValueError):
        net(torch.zeros(1, 1, 12, 16, 16, device=DEV))


@pytest.mark.parametrize('name,kind', [('pixel_2d_36', 'pixel'), ('pixel_3d_12', 'pixel'),
                                       ('patchgan_n2_2d_36', 'n_layers'), ('patchgan_n4_2d_72', 'n_layers')])
def test_discriminators_wide(golden_dir, name, kind):
    g = G(golden_dir, name + '.npz')
    dim = int(g['dim'])
    nl = int(g['n_layers']) if kind == 'n_layers' else 3
    spec = S.pixel_spec(dim) if kind == 'pixel' else S.patchgan_spec(dim, n_layers=nl)
    net = load(networks.define_D(1, 64, kind, nl, 'instance', 'kaiming', 0.02, False, [0], dimension=dim), spec,
               int(g['seed']))
    x = torch.from_numpy(rnd(g['x_seed'], g['shape'])).to(DEV).requires_grad_(True)
    y = net(x)
    assert relmax(y.detach().cpu().numpy(), g['y']) < 5e-4
    r = torch.from_numpy(rnd(g['r_seed'], y.shape)).to(DEV)
    (y * r).mean().backward()
    # (pixel: 1 x 1 convolutions -- every pixel is its own network, so one LeakyReLU decision that differs moves that pixel's whole input
    #  gradient; measured 1.3e-3 against the reference's fp32 run)
    tol = 5e-3 if kind == 'pixel' else 1e-3
    assert rel2(x.grad.cpu().numpy(), g['dx']) < tol
    check_grads(g, net, tol)


@pytest.mark.parametrize('tag', ['deconv_basic_36', 'vanilla_pixel_32', 'deconv_basic_24_b2'])
def test_dryops_step(golden_dir, tag):
    from neuroclear_amd.models import create_model
    g = G(golden_dir, 'dryops_step_%s.npz' % tag)
    size, netG, netD = int(g['size']), str(g['netG']), str(g['netD'])
    batch = int(g['batch']) if 'batch' in g else 1
    opt = _apollo_opt()
    opt.model, opt.netG, opt.netD = 'axial_to_lateral_gan_dryops', netG, netD
    model = create_model(opt)
    assert model.model_names == ['G_A', 'D_A_lateral', 'D_A_axial']
    gs = {'unet_deconv': S.unet_deconv_spec, 'unet_vanilla': S.unet_vanilla_spec}[netG]()
    ds = {'basic': S.patchgan_spec, 'pixel': S.pixel_spec}[netD](2)
    nets_ = ['G_A', 'D_A_axial', 'D_A_lateral']
    for i, (n, sp) in enumerate(zip(nets_, [gs, ds, ds])):
        load(getattr(model, 'net' + n), sp, int(g['net_seed0']) + i)
    before = {n: [p.detach().clone() for p in getattr(model, 'net' + n).parameters()] for n in nets_}
    real = torch.from_numpy(rnd(g['real_seed'], (batch, 1, size, size, size)))
    np.random.seed(int(g['step_seed']))
    names = [str(s) for s in g['loss_names']]
    for it in range(2):
        model.set_input({'A': real, 'A_paths': 'x'})
        model.optimize_parameters()
        L = model.get_current_losses()
        got = np.array([L[k] for k in names])
        print(it, dict(zip(names, got)), g['losses'][it])
        np.testing.assert_allclose(got, g['losses'][it], rtol=2e-5 if it == 0 else 5e-3, err_msg='step %d' % it)
        if it == 0:
            assert float(np.abs(model.fake.detach().cpu().numpy() - g['fake0']).max()) < 2e-5
    for n in nets_:
        ps = list(getattr(model, 'net' + n).parameters())
        upd = np.array([float((a.detach() - b).double().norm()) for a, b in zip(ps, before[n])])
        sel = np.array([a.dim() > 1 for a in ps])
        np.testing.assert_allclose(upd[sel], g['upd_' + n][sel], rtol=5e-2, err_msg=n)


@pytest.mark.parametrize('dim,shape,nl', [(2, (3, 1, 36, 36), 3), (2, (1, 1, 108, 108), 3), (3, (1, 1, 36, 36, 36), 3),
                                          (2, (2, 1, 72, 72), 4), (2, (1, 1, 20, 20), 2)])
def test_patchgan_whole_net_entry_matches_op_by_op(dim, shape, nl, monkeypatch):
    """nc_patchgan_fwd / nc_patchgan_bwd (one C call per direction) against the op-by-op path of the same module:
    same kernels in the same order, so outputs and gradients agree to rounding of the accumulation order only."""
    net = load(networks.define_D(1, 64, 'n_layers', nl, 'instance', 'kaiming', 0.02, False, [0], dimension=dim),
               S.patchgan_spec(dim, n_layers=nl), 31)
    x = torch.from_numpy(rnd(41, shape)).to(DEV)
    r = None
    res = {}
    for mode in ('1', '0'):
        monkeypatch.setenv('NC_FUSED_PATCHGAN', mode)
        net.zero_grad()
        xi = x.clone().requires_grad_(True)
        y = net(xi)
        if r is None:
            r = torch.from_numpy(rnd(42, y.shape)).to(DEV)
        (y * r).mean().backward()
        res[mode] = (y.detach().clone(), xi.grad.clone(), [p.grad.clone() for p in net.parameters()])
    ya, xa, pa = res['1']
    yb, xb, pb = res['0']
    assert torch.equal(ya, yb)
    assert torch.equal(xa, xb)
    for a, b in zip(pa, pb):
        assert torch.equal(a, b)
    # frozen parameters (generator phase): input gradient only
    monkeypatch.setenv('NC_FUSED_PATCHGAN', '1')
    for p in net.parameters():
        p.requires_grad_(False)
    xi = x.clone().requires_grad_(True)
    (net(xi) * r).mean().backward()
    assert torch.equal(xi.grad, xa)


@pytest.mark.parametrize('shape', [(1, 1, 16, 16, 16), (2, 1, 9, 14, 21), (1, 1, 12, 20, 24), (1, 1, 36, 36, 36), (1, 1, 72, 64, 80), (3, 1, 24, 28, 32)])
@pytest.mark.parametrize('want_dx', [False, True])
@pytest.mark.parametrize('terms', [3, 2])
def test_deep_linear_collapsed_tail_equals_the_layered_chain(shape, want_dx, terms):
    """nc_set_dl_collapse(1): layers 2 .. 5 of deep_linear_gen (reference networks.py:902-911: Conv3d 3^3 64 -> 64, then 1 x 1
    64 -> 32 -> 16 -> 1, no bias, nothing in between) as ONE 64 -> 1 convolution forward, and backward every parameter gradient of the four
    layers plus dL/dact1 from dy, act1 and the weights (csrc/gen_nets.hip).  In the two-term arithmetic the 5^3 layer's backward then runs
    from 27 shifted copies of the one-channel dy (a 32 x 64 weight gradient and a forward 32 -> 64 convolution with composed weights: half
    the matrix work each; the 32-channel launches carry one zero-weight padding k-step).  Exact algebra: output, input gradient and all six
    parameter gradients equal the layered evaluation to fp32 rounding -- shapes the one-channel kernels cover and shapes they do not, batches,
    odd sizes, several tiles per plane."""
    from neuroclear_amd._lib import lib
    prev = lib().nc_get_split_terms(), lib().nc_get_dl_collapse()
    lib().nc_set_split_terms(terms)
    net = load(networks.define_G(1, 1, 64, 'deep_linear_gen', 'instance', False, 'kaiming', 0.02, [0]), S.deep_linear_spec(), 32)
    x0 = torch.from_numpy(rnd(41, shape)).to(DEV)
    r = torch.from_numpy(rnd(42, shape)).to(DEV)

    def run(collapse):
        lib().nc_set_dl_collapse(collapse)
        assert lib().nc_get_dl_collapse() == collapse
        for p in net.parameters():
            p.grad = None
        x = x0.clone().requires_grad_(want_dx)
        y = net(x)
        (y * r).sum().backward()
        with torch.no_grad():
            yi = net(x0)  # the inference form (saved == NULL)
        return y.detach().clone(), yi, (x.grad.clone() if want_dx else None), [p.grad.clone() for p in net.parameters()]

    try:
        ya, ia, xa, ga = run(0)
        scale = float(ya.abs().max())
        assert torch.equal(ya, ia)
        # mode 1: round 5's collapsed tail + rank forms; mode 2 (default, round 6): layers 1 .. 5 as one position-typed 7^3 kernel where the shape
        # admits it (extents >= 8, W % 4 == 0, two-term arithmetic: csrc/dl_typed.hip), else as mode 1
        for mode in (1, 2):
            yb, ib, xb, gb = run(mode)
            # (the layered fp32 chain itself is ~2e-6 of the largest output away from fp64; the collapsed forms are closer: tests/test_gpu_grad_fp64.py)
            assert float((ya - yb).abs().max()) <= 1e-5 * scale, (mode, float((ya - yb).abs().max()) / scale)
            assert float((ib - yb).abs().max()) <= 1e-5 * scale  # (the inference form: saved == NULL takes the collapsed tail of mode 1)
            if mode == 1:
                assert torch.equal(yb, ib)
            if want_dx:
                assert rel2(xb.cpu().numpy(), xa.cpu().numpy()) < 1e-5, mode
            for (n, _), a, b in zip(net.named_parameters(), ga, gb):
                assert rel2(b.cpu().numpy(), a.cpu().numpy()) < 1e-5, (mode, n, rel2(b.cpu().numpy(), a.cpu().numpy()))
            # the switch at backward time does not matter: the forward's choice travels with its saved tensors
            lib().nc_set_dl_collapse(mode)
            for p in net.parameters():
                p.grad = None
            y = net(x0.clone().requires_grad_(False))
            lib().nc_set_dl_collapse(0)
            (y * r).sum().backward()
            for (n, _), b, c in zip(net.named_parameters(), gb, [p.grad for p in net.parameters()]):
                assert torch.equal(b, c), (mode, n)
    finally:
        lib().nc_set_split_terms(prev[0])
        lib().nc_set_dl_collapse(prev[1])


def test_deep_linear_backward_survives_a_switch_of_arithmetic_after_the_forward():
    """The default forward of deep_linear_gen never writes act1 (csrc/gen_nets.hip, "the forward without act1") and its backward takes q from the
    32 x 64 weight gradient P -- in the two-term arithmetic.  A caller who switches to the three-term form BETWEEN a forward and its backward
    (include/nc_hip.h advises against it) must still get the right gradients: the backward sees in `kept` that act1 is missing, forms it in a
    gradient buffer and walks the first-stage collapsed path."""
    from neuroclear_amd._lib import lib
    prev = lib().nc_get_split_terms()
    net = load(networks.define_G(1, 1, 64, 'deep_linear_gen', 'instance', False, 'kaiming', 0.02, [0]), S.deep_linear_spec(), 32)
    shape = (1, 1, 36, 36, 36)
    x0 = torch.from_numpy(rnd(41, shape)).to(DEV)
    r = torch.from_numpy(rnd(42, shape)).to(DEV)

    def run(switch):
        lib().nc_set_split_terms(2)
        for p in net.parameters():
            p.grad = None
        x = x0.clone().requires_grad_(True)
        y = net(x)
        if switch:
            lib().nc_set_split_terms(3)
        (y * r).sum().backward()
        return x.grad.clone(), [p.grad.clone() for p in net.parameters()]
    try:
        xa, ga = run(False)
        xb, gb = run(True)
    finally:
        lib().nc_set_split_terms(prev)
    assert rel2(xb.cpu().numpy(), xa.cpu().numpy()) < 1e-5
    for (n, _), a, b in zip(net.named_parameters(), ga, gb):
        assert rel2(b.cpu().numpy(), a.cpu().numpy()) < 1e-5, n


@pytest.mark.parametrize('shape', [(1, 1, 48, 48, 48), (1, 1, 80, 72, 64)])
def test_deep_linear_default_path_is_bit_identical_run_to_run(shape):
    """The default evaluation of deep_linear_gen launches k_conv_s3x in two shapes nothing else uses -- 32-channel output tiles (K32) and
    32-channel inputs with a zero-weight padding k-step that multiplies whatever the LDS ring holds -- next to hand-counted vmcnt waits: twelve
    forward + backward passes, every other one after an idle gap, must give the same bits (a timing dependence would show here)."""
    import time
    net = load(networks.define_G(1, 1, 64, 'deep_linear_gen', 'instance', False, 'kaiming', 0.02, [0]), S.deep_linear_spec(), 32)
    x0 = torch.from_numpy(rnd(41, shape)).to(DEV)
    r = torch.from_numpy(rnd(42, shape)).to(DEV)
    ref = None
    for it in range(12):
        torch.cuda.synchronize()
        if it % 2:
            time.sleep(0.03)
        for p in net.parameters():
            p.grad = None
        x = x0.clone().requires_grad_(True)
        y = net(x)
        (y * r).sum().backward()
        got = [y.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in net.parameters()]
        assert all(bool(torch.isfinite(t).all()) for t in got)
        if ref is None:
            ref = got
        else:
            assert all(torch.equal(a, b) for a, b in zip(ref, got)), it


@pytest.mark.parametrize('kind,shape', [('unet', (1, 1, 16, 16, 16)), ('unet', (2, 1, 12, 20, 24)), ('unet', (3, 1, 8, 8, 36)),
                                        ('linear', (1, 1, 16, 16, 16)), ('linear', (2, 1, 9, 14, 21))])
@pytest.mark.parametrize('want_dx', [False, True])
@pytest.mark.parametrize('terms', [3, 2])
def test_generator_whole_net_entries_match_layer_by_layer(kind, shape, want_dx, terms, monkeypatch):
    """nc_unet_deconv_train_fwd / nc_unet_deconv_bwd and nc_deep_linear_fwd / _bwd (one C call per direction, concat
    halves written in place, skip gradients merged inside the max-pool backward) against the layer-by-layer autograd path
    of the same module: same kernels in the same order, so output, input gradient and all parameter gradients are
    BIT-identical -- batches included (per-sample sub-ranges of the concat buffers)."""
    from neuroclear_amd._lib import lib
    prev_terms = lib().nc_get_split_terms()
    prev_collapse = lib().nc_get_dl_collapse()
    lib().nc_set_split_terms(terms)
    lib().nc_set_dl_collapse(0)  # (the collapsed tail of deep_linear_gen is the whole-network call's own arithmetic: its test is below)
    request_restore = lambda: (lib().nc_set_split_terms(prev_terms), lib().nc_set_dl_collapse(prev_collapse))
    if kind == 'unet':
        net = load(networks.define_G(1, 1, 64, 'unet_deconv', 'instance', False, 'kaiming', 0.02, [0]),
                   S.unet_deconv_spec(), 31)
    else:
        net = load(networks.define_G(1, 1, 64, 'deep_linear_gen', 'instance', False, 'kaiming', 0.02, [0]),
                   S.deep_linear_spec(), 32)
    x0 = torch.from_numpy(rnd(41, shape)).to(DEV)
    r = torch.from_numpy(rnd(42, shape)).to(DEV)

    def run(fused):
        monkeypatch.setattr(networks, '_FUSED_GEN', fused)
        for p in net.parameters():
            p.grad = None
        x = x0.clone().requires_grad_(want_dx)
        y = net(x)
        (y * r).sum().backward()
        return y.detach().clone(), (x.grad.clone() if want_dx else None), [p.grad.clone() for p in net.parameters()]

    try:
        ya, xa, ga = run(False)
        yb, xb, gb = run(True)
        # The two-term form of the U-Net is the one exception to "bit-identical": the whole-network call converts an InstanceNorm output with
        # the power of two its bound sqrt(voxels) allows (no pass over the tensor), the layer-by-layer path measures the tensor -- two valid
        # scalings of the same operands, results equal to fp32 rounding (and a ReLU / max-pool decision may fall the other way: L2 bounds).
        exact = terms == 3 or kind == 'linear'
        if exact:
            assert torch.equal(ya, yb)
            if want_dx:
                assert torch.equal(xa, xb)
            for (n, _), a, b in zip(net.named_parameters(), ga, gb):
                assert torch.equal(a, b), n
        else:
            assert float((ya - yb).abs().max()) < 2e-6
            if want_dx:
                assert rel2(xa.cpu().numpy(), xb.cpu().numpy()) < 2e-2
            for (n, p_), a, b in zip(net.named_parameters(), ga, gb):
                if p_.dim() > 1:
                    assert rel2(a.cpu().numpy(), b.cpu().numpy()) < 2e-2, n
        if kind == 'linear':  # inference form (saved == NULL: activations ping-pong through the workspace)
            with torch.no_grad():
                assert torch.equal(net(x0), ya)
    finally:
        request_restore()


def test_whole_network_backward_writes_gradients_in_place():
    """The whole-network backward calls write parameter gradients straight into the optimizer's flat gradient buffer and
    autograd adopts those views (no per-parameter add kernels): after an Apollo step's backward every generator / PatchGAN
    parameter's .grad aliases its slice of FlatAdam.grad, and NC_DIRECT_GRADS=0 (separate tensors + accumulate) gives
    bit-identical updates."""
    from neuroclear_amd.models import create_model

    def run(direct):
        os.environ['NC_DIRECT_GRADS'] = '1' if direct else '0'
        try:
            torch.manual_seed(5)
            np.random.seed(5)
            model = create_model(_apollo_opt())
            real = torch.from_numpy(rnd(77, (1, 1, 36, 36, 36))).to(DEV)
            model.set_input({'A': real, 'A_paths': 'x'})
            model.optimize_parameters()
            alias = []
            for opt in (model.optimizer_G, model.optimizer_D):
                off = 0
                for p in opt.params:
                    alias.append(p.grad is not None and p.grad.data_ptr() == opt.grad.data_ptr() + 4 * off)
                    off += p.numel()
            return model.optimizer_G.flat.clone(), model.optimizer_D.flat.clone(), alias, dict(model.get_current_losses())
        finally:
            os.environ.pop('NC_DIRECT_GRADS', None)

    g1, d1, alias, l1 = run(True)
    g0, d0, _, l0 = run(False)
    assert all(alias)
    assert l1 == l0 and torch.equal(g1, g0) and torch.equal(d1, d0)


@pytest.mark.parametrize('tag', ['2d_36', '3d_28'])
def test_patchgan_spectral_norm(golden_dir, tag):
    """--netD basic_SN (NLayerDiscriminatorSN, networks.py:1069-1111) against the reference's own outputs: two training-mode
    forwards (the power-iteration vectors move in between), backward of the second, the final u of every convolution; the
    state dict round-trips with torch's spectral_norm key names."""
    g = G(golden_dir, 'patchgan_sn_%s.npz' % tag)
    dim = int(g['dim'])
    net = networks.define_D(1, 64, 'basic_SN', 3, 'instance', 'kaiming', 0.02, False, [0], dimension=dim)
    spec = S.patchgan_sn_spec(dim)
    assert list(net.state_dict().keys()) == [k for k, _ in spec]
    load(net, spec, int(g['seed']))
    net.train()
    x = torch.from_numpy(rnd(g['x_seed'], g['shape'])).to(DEV).requires_grad_(True)
    y1 = net(x)
    assert relmax(y1.detach().cpu().numpy(), g['y1']) < 5e-4
    y = net(x)
    assert relmax(y.detach().cpu().numpy(), g['y']) < 5e-4
    r = torch.from_numpy(rnd(g['r_seed'], y.shape)).to(DEV)
    (y * r).mean().backward()
    assert rel2(x.grad.cpu().numpy(), g['dx']) < 2e-3
    params = dict(net.named_parameters())
    for i, k in enumerate(str(n) for n in g['g_names']):
        gr = params[k].grad.detach().cpu().numpy().ravel()
        l2 = np.sqrt((gr.astype(np.float64) ** 2).sum())
        assert abs(l2 - g['g_l2'][i]) <= 2e-3 * g['g_l2'][i] + 1e-9, (k, l2, g['g_l2'][i])
    us = np.concatenate([b.detach().cpu().numpy().ravel() for k, b in net.named_buffers() if k.endswith('weight_u')])
    assert float(np.abs(us - g['u_final']).max()) < 1e-4
    # eval mode: no power iteration -- u stays put
    net.eval()
    with torch.no_grad():
        net(x)
    us2 = np.concatenate([b.detach().cpu().numpy().ravel() for k, b in net.named_buffers() if k.endswith('weight_u')])
    assert np.array_equal(us, us2)


@pytest.mark.parametrize('pattern', ['every_third', 'first_only'])
@pytest.mark.parametrize('which', ['unet_deconv', 'deep_linear_gen', 'patchgan'])
def test_partly_frozen_parameters_are_not_updated(which, pattern):
    """Some (not all) parameters of a network with requires_grad off: the whole-network backward must not leave gradients for them
    in the optimizer's flat buffer (FlatAdam.step is one launch over the whole buffer), the others get exactly the gradient of the
    unfrozen run, and a step moves only those.  (base_model.py:221-232 set_requires_grad freezes whole networks; a user freezing an
    encoder is the partial case.)  'first_only' freezes the FIRST parameter alone (model.0.weight of the PatchGAN): the case an off-by-one
    in the autograd node's input index misses (_PatchGAN.backward looked at needs_input_grad[3:])."""
    from neuroclear_amd.models.axial_to_lateral_gan_apollo_model import FlatAdam

    def build():
        if which == 'patchgan':
            net = load(networks.define_D(1, 64, 'basic', 3, 'instance', 'kaiming', 0.02, False, [0], dimension=2), S.patchgan_spec(2), 5)
            x = torch.from_numpy(rnd(3, (4, 1, 36, 36))).to(DEV)
        elif which == 'unet_deconv':
            net = load(networks.define_G(1, 1, 64, 'unet_deconv', 'instance', False, 'kaiming', 0.02, [0]), S.unet_deconv_spec(), 5)
            x = torch.from_numpy(rnd(3, (1, 1, 16, 16, 16))).to(DEV)
        else:
            net = load(networks.define_G(1, 1, 64, 'deep_linear_gen', 'instance', False, 'kaiming', 0.02, [0]), S.deep_linear_spec(), 5)
            x = torch.from_numpy(rnd(3, (1, 1, 16, 16, 16))).to(DEV)
        return net, FlatAdam(net.parameters(), lr=1e-3, betas=(0.5, 0.999)), x

    net, opt, x = build()
    opt.zero_grad()
    (net(x) ** 2).mean().backward()
    opt._collect()
    full = opt.grad.clone()
    net2, opt2, _ = build()
    ps = list(net2.parameters())
    frozen = [0] if pattern == 'first_only' else [i for i in range(len(ps)) if i % 3 == 1]
    for i in frozen:
        ps[i].requires_grad_(False)
    before = opt2.flat.clone()
    opt2.zero_grad()
    (net2(x) ** 2).mean().backward()
    opt2.step()
    off = 0
    for i, p in enumerate(ps):
        n = p.numel()
        g, moved = opt2.grad[off:off + n], (opt2.flat[off:off + n] - before[off:off + n]).abs().max()
        if i in frozen:
            assert p.grad is None and float(g.abs().max()) == 0.0 and float(moved) == 0.0, i
        else:
            assert torch.equal(g, full[off:off + n]), i
            assert float(full[off:off + n].abs().max()) == 0.0 or float(moved) > 0.0, i
        off += n


def test_gan_loss_modes(golden_dir):
    """networks.GANLoss for 'lsgan' | 'vanilla' | 'wgangp' against the reference's own values and gradients (models/networks.py:252-319),
    and the NotImplementedError of an unknown objective."""
    g = G(golden_dir, 'ganloss_modes.npz')
    for mode in ('lsgan', 'vanilla', 'wgangp'):
        crit = networks.GANLoss(mode).to(DEV)
        for tag, shape, seed in (('a', (4, 1, 11, 11), 61), ('b', (2, 1, 2, 2), 62), ('c', (1, 1, 5, 6, 7), 63)):
            for flag in (True, False):
                p = torch.from_numpy(rnd(seed, shape) * 6 - 3).to(DEV).requires_grad_(True)
                loss = crit(p, flag)
                (loss * 1.0).backward()
                key = '%s_%s_%d' % (mode, tag, int(flag))
                np.testing.assert_allclose(float(loss), g[key + '_loss'], rtol=2e-6, atol=1e-7)
                np.testing.assert_allclose(p.grad.cpu().numpy(), g[key + '_grad'], rtol=2e-5, atol=1e-8)
    with pytest.raises(NotImplementedError):
        networks.GANLoss('hinge')


@pytest.mark.parametrize('tag', ['unet_deconv_bn_16', 'patchgan_bn_2d_36'])
def test_batch_norm_networks(golden_dir, tag):
    """--norm batch (models/networks.py:30-31, nn.BatchNorm{2,3}d with affine parameters and running statistics) behind the same
    factories: training-mode forward + backward against the reference's values, the running statistics and the step counter it leaves,
    an evaluation-mode forward on them, and the state-dict keys (a reference checkpoint loads as it is)."""
    g = G(golden_dir, tag + '.npz')
    unet = tag.startswith('unet')
    spec = S.unet_deconv_bn_spec() if unet else S.patchgan_bn_spec(2)
    net = (networks.define_G(1, 1, 64, 'unet_deconv', 'batch', False, 'kaiming', 0.02, [0]) if unet else
           networks.define_D(1, 64, 'basic', 3, 'batch', 'kaiming', 0.02, False, [0], dimension=2))
    assert list(net.state_dict().keys()) == [k for k, _ in spec]
    load(net, spec, int(g['seed']))
    net.train()
    shape = tuple(int(v) for v in g['shape'])
    x = torch.from_numpy(rnd(g['x_seed'], shape)).to(DEV).requires_grad_(True)
    y = net(x)
    np.testing.assert_allclose(y.detach().cpu().numpy(), g['y'], atol=2e-5, rtol=5e-4)
    r = torch.from_numpy(rnd(g['r_seed'], y.shape)).to(DEV)
    (y * r).mean().backward()
    assert rel2(x.grad.cpu().numpy(), g['dx']) < (2e-2 if unet else 1e-3)
    names = [str(n) for n in g['g_names']]
    params = dict(net.named_parameters())
    for i, k in enumerate(names):
        l2 = float(params[k].grad.double().norm())
        assert abs(l2 - g['g_l2'][i]) <= (2e-2 if unet else 1e-3) * g['g_l2'][i] + 1e-6, (k, l2, g['g_l2'][i])
    sd = net.state_dict()
    for k in g.files:
        if k.startswith('buf_'):
            np.testing.assert_allclose(sd[k[4:]].cpu().numpy(), g[k], rtol=2e-5, atol=1e-6, err_msg=k)
    net.eval()
    with torch.no_grad():
        ye = net(torch.from_numpy(rnd(g['xe_seed'], shape)).to(DEV))
    np.testing.assert_allclose(ye.cpu().numpy(), g['y_eval'], atol=2e-5, rtol=5e-4)
    with pytest.raises(NotImplementedError):
        networks.get_norm_layer('layer', 3)


def test_batch_norm_more_than_65535_instances():
    """--norm batch on a 2-D PatchGAN over a slice batch (148 slices x 512 channels = 75,776 (sample, channel) instances: more than one
    launch's gridDim.y): BatchNormAct forward + backward against F.batch_norm + leaky_relu in fp64 (models/networks.py:30-31)."""
    import torch.nn.functional as F
    N, C, H, W = 148, 512, 5, 6
    g = torch.Generator(device=DEV).manual_seed(4)
    x = (torch.randn(N, C, H, W, device=DEV, generator=g) * 2 + 0.5).requires_grad_(True)
    m = networks.BatchNormAct(C, slope=0.2, dimension=2).to(DEV)
    with torch.no_grad():
        m.weight.copy_(torch.rand(C, device=DEV, generator=g) + 0.5)
        m.bias.copy_(torch.randn(C, device=DEV, generator=g) * 0.1)
    m.train()
    y = m(x)
    r = torch.randn(y.shape, device=DEV, generator=g)
    (y * r).sum().backward()
    xd = x.detach().double().requires_grad_(True)
    wd, bd = m.weight.detach().double().requires_grad_(True), m.bias.detach().double().requires_grad_(True)
    yd = F.leaky_relu(F.batch_norm(xd, None, None, wd, bd, True, 0.1, 1e-5), 0.2)
    (yd * r.double()).sum().backward()
    assert float((y.detach().double() - yd.detach()).abs().max()) < 2e-5
    for a, b in ((x.grad, xd.grad), (m.weight.grad, wd.grad), (m.bias.grad, bd.grad)):
        assert float((a.double() - b).norm() / b.norm()) < 1e-5
