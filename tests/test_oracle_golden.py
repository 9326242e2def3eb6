"""CPU: pin the oracle (oracle/*.py) against the golden vectors produced by the reference itself
(oracle/gen_golden.py, run in the build container against /root/reference)."""
import hashlib
import os

import numpy as np
import pytest
import torch

from neuroclear_amd.util import seed as S
from oracle import apollo, dice, nets

torch.set_num_threads(8)


def G(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def rnd(seed, shape):
    return np.random.default_rng(int(seed)).random(tuple(shape), dtype=np.float32)


def _check_grads(g, sd, tol=2e-4):
    for i, (k, p) in enumerate(sd.items()):
        gr = p.grad.detach().numpy().ravel()
        l2 = np.sqrt((gr.astype(np.float64) ** 2).sum())
        assert abs(l2 - g['g_l2'][i]) <= tol * max(1e-6, g['g_l2'][i]), k
        idx = np.random.default_rng([77, i]).integers(0, gr.size, size=8)
        np.testing.assert_allclose(gr[idx], g['g_samp'][i], rtol=1e-3, atol=tol * l2 / np.sqrt(gr.size) + 1e-9)


@pytest.mark.parametrize('size', [16, 32])
def test_unet_deconv(golden_dir, size):
    g = G(golden_dir, 'unet_deconv_%d.npz' % size)
    sd = nets.to_torch(S.weights_from_seed(S.unet_deconv_spec(), int(g['seed'])), requires_grad=True)
    x = torch.from_numpy(rnd(g['x_seed'], (1, 1, size, size, size))).requires_grad_(True)
    taps = {}
    y = nets.unet_deconv(sd, x, taps)
    np.testing.assert_allclose(y.detach().numpy(), g['y'], atol=2e-6)
    for k, v in taps.items():
        ref = g['stage_' + k]
        got = np.array([v.mean().item(), v.abs().max().item(), v.norm().item()])
        np.testing.assert_allclose(got, ref, rtol=1e-4)
    r = torch.from_numpy(rnd(g['r_seed'], y.shape))
    (y * r).mean().backward()
    np.testing.assert_allclose(x.grad.numpy(), g['dx'], atol=1e-4 * np.abs(g['dx']).max())
    _check_grads(g, sd)


@pytest.mark.parametrize('size', [16, 24])
def test_deep_linear(golden_dir, size):
    g = G(golden_dir, 'deep_linear_%d.npz' % size)
    sd = nets.to_torch(S.weights_from_seed(S.deep_linear_spec(), int(g['seed'])), requires_grad=True)
    x = torch.from_numpy(rnd(g['x_seed'], (1, 1, size, size, size))).requires_grad_(True)
    y = nets.deep_linear(sd, x)
    np.testing.assert_allclose(y.detach().numpy(), g['y'], rtol=1e-4, atol=1e-4)
    r = torch.from_numpy(rnd(g['r_seed'], y.shape))
    (y * r).mean().backward()
    np.testing.assert_allclose(x.grad.numpy(), g['dx'], atol=1e-4 * np.abs(g['dx']).max())
    _check_grads(g, sd)


@pytest.mark.parametrize('tag', ['2d_36', '2d_108', '2d_36_b3', '3d_36'])
def test_patchgan(golden_dir, tag):
    g = G(golden_dir, 'patchgan_%s.npz' % tag)
    sd = nets.to_torch(S.weights_from_seed(S.patchgan_spec(int(g['dim'])), int(g['seed'])), requires_grad=True)
    x = torch.from_numpy(rnd(g['x_seed'], g['shape'])).requires_grad_(True)
    y = nets.patchgan(sd, x)
    np.testing.assert_allclose(y.detach().numpy(), g['y'], rtol=1e-4, atol=1e-5)
    r = torch.from_numpy(rnd(g['r_seed'], y.shape))
    (y * r).mean().backward()
    np.testing.assert_allclose(x.grad.numpy(), g['dx'], atol=2e-4 * np.abs(g['dx']).max())
    _check_grads(g, sd, tol=1e-3)


@pytest.mark.parametrize('fname', ['apollo_step_36.npz', 'apollo_step_24_b2.npz', 'apollo_step_24_vanilla.npz', 'apollo_step_24_wgangp.npz'])
def test_apollo_step(golden_dir, fname):
    g = G(golden_dir, fname)
    size, batch = int(g['size']), int(g['batch']) if 'batch' in g else 1
    specs = [('G_A', S.unet_deconv_spec()), ('G_B', S.deep_linear_spec())] + \
        [(n, S.patchgan_spec(2)) for n in apollo.APOLLO_D]
    sds = {n: S.weights_from_seed(sp, int(g['net_seed0']) + i) for i, (n, sp) in enumerate(specs)}
    model = apollo.ApolloOracle(sds, gan_mode=str(g['gan_mode']) if 'gan_mode' in g else 'lsgan')
    before = {n: [p.detach().clone() for p in model.n.sd[n].values()] for n in sds}
    real = torch.from_numpy(rnd(g['real_seed'], (batch, 1, size, size, size)))
    np.random.seed(int(g['step_seed']))
    names = [str(s) for s in g['loss_names']]
    for it in range(2):
        L = model.step(real)
        got = np.array([L[k] for k in names])
        np.testing.assert_allclose(got, g['losses'][it], rtol=2e-4, err_msg='step %d' % it)
        if it == 0:
            np.testing.assert_allclose(model.fake.detach().numpy(), g['fake0'], atol=2e-6)
            np.testing.assert_allclose(model.rec.detach().numpy(), g['rec0'], rtol=1e-4, atol=1e-4)
    for n in sds:
        upd = np.array([float((a.detach() - b).double().norm()) for a, b in zip(model.n.sd[n].values(), before[n])])
        np.testing.assert_allclose(upd, g['upd_' + n], rtol=2e-3, err_msg=n)


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize('tag', ['100_32_4_4', 'ragged_70_50_61', 'u8_64_24_4_2'])
def test_dice_assemble(golden_dir, tag):
    g = G(golden_dir, 'dice_%s.npz' % tag)
    shape, R, ov, b = tuple(int(v) for v in g['shape']), int(g['roi']), int(g['overlap']), int(g['border'])
    dtype = np.dtype(str(g['dtype']))
    vol = S.random_volume(int(g['vol_seed']), shape, dtype)
    padded = dice.pad_for_dicing(vol, R, ov)
    assert padded.shape == tuple(g['padded'])
    steps = dice.grid_steps(padded.shape, R, ov)
    assert steps == tuple(g['steps'])
    n = steps[0] * steps[1] * steps[2]
    assert n == int(g['n'])
    refl = dice.reflect_pad(padded, b)
    cubes = [dice.normalize(dice.cut_cube(refl, i, steps, R, ov, b)) for i in range(n)]
    np.testing.assert_array_equal(cubes[0][None], g['first'])
    np.testing.assert_array_equal(cubes[n // 2][None], g['mid'])
    np.testing.assert_array_equal(cubes[-1][None], g['last'])
    dt = 'uint8' if dtype == np.uint8 else 'uint16'
    ident = dice.assemble(cubes, padded.shape, shape, R, ov, b, dt)
    assert _sha(ident) == str(g['sha_identity'])
    assert int(np.abs(ident.astype(np.int64) - vol).max()) <= 1  # identity round trip within 1 LSB
    pos = dice.assemble([c * 0.5 + (i % 7) * 1e-3 for i, c in enumerate(cubes)], padded.shape, shape, R, ov, b, dt)
    assert _sha(pos) == str(g['sha_pos'])
    if g['out_pos'].size:
        np.testing.assert_array_equal(pos, g['out_pos'])


def test_dice_geometry(golden_dir):
    for L, R, ov, p, steps in G(golden_dir, 'dice_geometry.npz')['rows']:
        pad = dice.pad_amounts((int(L),) * 3, int(R), int(ov))[0]
        assert L + pad == p
        assert dice.grid_steps((int(p),) * 3, int(R), int(ov))[0] == steps
    # the reference screenshot: 900^3 / 120 / 15 -> 960^3, (9,9,9), 729 cubes of 140^3 with border_cut 10
    assert dice.pad_amounts((900,) * 3, 120, 15) == (60, 60, 60)
    assert dice.grid_steps((960,) * 3, 120, 15) == (9, 9, 9)


def test_border_zero_rejected():
    with pytest.raises(ValueError):
        dice.assemble([], (8, 8, 8), (8, 8, 8), 4, 0, 0)


def test_athena_step(golden_dir):
    g = G(golden_dir, 'athena_step_36.npz')
    size = int(g['size'])
    names_n = ['G_A', 'G_B'] + apollo.ATHENA_D
    specs = [S.unet_deconv_spec(), S.deep_linear_spec()] + [S.patchgan_spec(2)] * 6
    sds = {n: S.weights_from_seed(sp, int(g['net_seed0']) + i) for i, (n, sp) in enumerate(zip(names_n, specs))}
    model = apollo.AthenaOracle(sds)
    real = torch.from_numpy(rnd(g['real_seed'], (1, 1, size, size, size)))
    names = [str(s) for s in g['loss_names']]
    for it in range(2):
        L = model.step(real)
        np.testing.assert_allclose(np.array([L[k] for k in names]), g['losses'][it], rtol=2e-4, err_msg='step %d' % it)


# ---- widening row: unet_vanilla, pixel / n_layers discriminators, the Dryops step

@pytest.mark.parametrize('size', [16, 24])
def test_unet_vanilla(golden_dir, size):
    g = G(golden_dir, 'unet_vanilla_%d.npz' % size)
    sd = nets.to_torch(S.weights_from_seed(S.unet_vanilla_spec(), int(g['seed'])), requires_grad=True)
    x = torch.from_numpy(rnd(g['x_seed'], (1, 1, size, size, size))).requires_grad_(True)
    y = nets.unet_vanilla(sd, x)
    np.testing.assert_allclose(y.detach().numpy(), g['y'], atol=2e-6)
    r = torch.from_numpy(rnd(g['r_seed'], y.shape))
    (y * r).mean().backward()
    np.testing.assert_allclose(x.grad.numpy(), g['dx'], atol=1e-4 * np.abs(g['dx']).max())
    _check_grads(g, sd)


@pytest.mark.parametrize('name,fn', [('pixel_2d_36', 'pixel'), ('pixel_3d_12', 'pixel'),
                                     ('patchgan_n2_2d_36', 'patchgan'), ('patchgan_n4_2d_72', 'patchgan')])
def test_discriminators_wide(golden_dir, name, fn):
    g = G(golden_dir, name + '.npz')
    dim = int(g['dim'])
    if fn == 'pixel':
        spec, f = S.pixel_spec(dim), nets.pixel
    else:
        nl = int(g['n_layers'])
        spec, f = S.patchgan_spec(dim, n_layers=nl), lambda sd, x: nets.patchgan(sd, x, n_layers=nl)
    sd = nets.to_torch(S.weights_from_seed(spec, int(g['seed'])), requires_grad=True)
    x = torch.from_numpy(rnd(g['x_seed'], g['shape'])).requires_grad_(True)
    y = f(sd, x)
    np.testing.assert_allclose(y.detach().numpy(), g['y'], rtol=1e-4, atol=1e-5)
    r = torch.from_numpy(rnd(g['r_seed'], y.shape))
    (y * r).mean().backward()
    np.testing.assert_allclose(x.grad.numpy(), g['dx'], atol=2e-4 * np.abs(g['dx']).max())
    _check_grads(g, sd, tol=1e-3)


def dryops_specs(netG, netD):
    gs = {'unet_deconv': S.unet_deconv_spec, 'unet_vanilla': S.unet_vanilla_spec}[netG]()
    ds = {'basic': S.patchgan_spec, 'pixel': S.pixel_spec}[netD](2)
    return [('G_A', gs), ('D_A_axial', ds), ('D_A_lateral', ds)]


@pytest.mark.parametrize('tag', ['deconv_basic_36', 'vanilla_pixel_32', 'deconv_basic_24_b2'])
def test_dryops_step(golden_dir, tag):
    g = G(golden_dir, 'dryops_step_%s.npz' % tag)
    size, netG, netD = int(g['size']), str(g['netG']), str(g['netD'])
    batch = int(g['batch']) if 'batch' in g else 1
    sds = {n: S.weights_from_seed(sp, int(g['net_seed0']) + i) for i, (n, sp) in enumerate(dryops_specs(netG, netD))}
    model = apollo.DryopsOracle(sds, netG=netG, netD=netD)
    before = {n: [p.detach().clone() for p in model.n.sd[n].values()] for n in sds}
    real = torch.from_numpy(rnd(g['real_seed'], (batch, 1, size, size, size)))
    np.random.seed(int(g['step_seed']))
    names = [str(s) for s in g['loss_names']]
    for it in range(2):
        L = model.step(real)
        np.testing.assert_allclose(np.array([L[k] for k in names]), g['losses'][it], rtol=2e-4, err_msg='step %d' % it)
        if it == 0:
            np.testing.assert_allclose(model.fake.detach().numpy(), g['fake0'], atol=2e-6)
    # update norms of the weight tensors only: a bias in front of an InstanceNorm has a true gradient of zero, what
    # autograd leaves there is summation-order noise, and Adam's first steps turn any non-zero noise into +-lr
    for n in sds:
        w = [i for i, p in enumerate(before[n]) if p.dim() > 1]
        upd = np.array([float((a.detach() - b).double().norm()) for a, b in zip(model.n.sd[n].values(), before[n])])
        np.testing.assert_allclose(upd[w], g['upd_' + n][w], rtol=2e-3, err_msg=n)


@pytest.mark.parametrize('tag', ['2d_36', '3d_28'])
def test_patchgan_sn(golden_dir, tag):
    """Spectral-norm PatchGAN (--netD basic_SN): two training-mode forwards (u, v move), backward of the second."""
    g = G(golden_dir, 'patchgan_sn_%s.npz' % tag)
    dim = int(g['dim'])
    sd = nets.to_torch(S.weights_from_seed(S.patchgan_sn_spec(dim), int(g['seed'])))
    names = [str(n) for n in g['g_names']]
    for k in names:
        sd[k].requires_grad_(True)
    x = torch.from_numpy(rnd(g['x_seed'], g['shape'])).requires_grad_(True)
    y1 = nets.patchgan_sn(sd, x)
    np.testing.assert_allclose(y1.detach().numpy(), g['y1'], rtol=1e-4, atol=1e-6)
    y = nets.patchgan_sn(sd, x)
    np.testing.assert_allclose(y.detach().numpy(), g['y'], rtol=1e-4, atol=1e-6)
    r = torch.from_numpy(rnd(g['r_seed'], y.shape))
    (y * r).mean().backward()
    np.testing.assert_allclose(x.grad.numpy(), g['dx'], atol=2e-4 * np.abs(g['dx']).max())
    for i, k in enumerate(names):
        gr = sd[k].grad.numpy().ravel()
        l2 = np.sqrt((gr.astype(np.float64) ** 2).sum())
        assert abs(l2 - g['g_l2'][i]) <= 1e-3 * max(1e-9, g['g_l2'][i]), k
    us = np.concatenate([sd[k].detach().numpy().ravel() for k in sd if k.endswith('weight_u')])
    np.testing.assert_allclose(us, g['u_final'], atol=1e-5)


def test_gan_loss_modes(golden_dir):
    """oracle nets.gan_loss against the reference's GANLoss('lsgan' | 'vanilla' | 'wgangp') (models/networks.py:252-319): values and
    gradients on PatchGAN-shaped predictions."""
    g = G(golden_dir, 'ganloss_modes.npz')
    for mode in ('lsgan', 'vanilla', 'wgangp'):
        for tag, shape, seed in (('a', (4, 1, 11, 11), 61), ('b', (2, 1, 2, 2), 62), ('c', (1, 1, 5, 6, 7), 63)):
            for flag in (True, False):
                p = torch.from_numpy(rnd(seed, shape) * 6 - 3).requires_grad_(True)
                loss = nets.gan_loss(p, flag, mode)
                loss.backward()
                key = '%s_%s_%d' % (mode, tag, int(flag))
                np.testing.assert_allclose(loss.item(), g[key + '_loss'], rtol=1e-6, atol=1e-7)
                np.testing.assert_allclose(p.grad.numpy(), g[key + '_grad'], rtol=1e-6, atol=1e-9)
    with pytest.raises(NotImplementedError):
        nets.gan_loss(torch.zeros(1), True, 'hinge')


@pytest.mark.parametrize('tag', ['unet_deconv_bn_16', 'patchgan_bn_2d_36'])
def test_batch_norm_networks(golden_dir, tag):
    """--norm batch (networks.py:30-31): the oracle's unet_deconv / patchgan with BatchNorm against the reference's own factories --
    training-mode forward + backward, the running statistics it leaves, and an evaluation-mode forward on them."""
    g = G(golden_dir, tag + '.npz')
    unet = tag.startswith('unet')
    spec = S.unet_deconv_bn_spec() if unet else S.patchgan_bn_spec(2)
    sd = nets.to_torch(S.weights_from_seed(spec, int(g['seed'])), requires_grad=True)
    shape = tuple(int(v) for v in g['shape'])
    x = torch.from_numpy(rnd(g['x_seed'], shape)).requires_grad_(True)
    fn = (lambda xx, tr: nets.unet_deconv(sd, xx, None, 'batch', tr)) if unet else (lambda xx, tr: nets.patchgan(sd, xx, 3, 'batch', tr))
    y = fn(x, True)
    np.testing.assert_allclose(y.detach().numpy(), g['y'], atol=2e-6, rtol=2e-5)
    (y * torch.from_numpy(rnd(g['r_seed'], y.shape))).mean().backward()
    np.testing.assert_allclose(x.grad.numpy(), g['dx'], rtol=1e-3, atol=1e-3 * float(np.abs(g['dx']).max()))
    for i, k in enumerate(str(n) for n in g['g_names']):
        l2 = float(sd[k].grad.double().norm())
        assert abs(l2 - g['g_l2'][i]) <= 2e-3 * g['g_l2'][i] + 1e-6, (k, l2, g['g_l2'][i])
    for k in g.files:
        if k.startswith('buf_'):
            np.testing.assert_allclose(sd[k[4:]].detach().numpy(), g[k], rtol=1e-5, atol=1e-6, err_msg=k)
    with torch.no_grad():
        ye = fn(torch.from_numpy(rnd(g['xe_seed'], shape)), False)
    np.testing.assert_allclose(ye.numpy(), g['y_eval'], atol=2e-6, rtol=2e-5)
