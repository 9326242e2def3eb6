"""Row h (north_star "the synthetic volume from the Jupyter generator"): the structured synthetic volume -- sparse beads and
tubes, blurred far more along z than along x / y, Poisson + Gaussian noise, uint16 -- that stands in for the reference's missing notebook
generator (README.md:116).  CPU tests: the generator is pinned (bytes, sparsity, anisotropy) and the oracle reproduces what the REFERENCE
computes on structured crops (fixtures written by oracle/gen_golden.py `structured`, which runs the reference): unet_deconv forward +
backward, one Apollo and one Athena step.  The GPU side of the same fixtures: tests/test_gpu_structured.py."""
import hashlib
import os

import numpy as np
import torch

from neuroclear_amd.util import seed as S
from oracle import apollo, nets


def G(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def struct_crop(seed, size, vol=96):
    """oracle/gen_golden.py struct_crop, restated: the crop around the brightest voxel as the float32 [0, 1] network input."""
    v = S.structured_volume(seed, vol)
    c = np.array(np.unravel_index(int(np.argmax(v)), v.shape))
    lo = np.clip(c - size // 2, 0, vol - size)
    return (v[lo[0]:lo[0] + size, lo[1]:lo[1] + size, lo[2]:lo[2] + size].astype(np.float32) / np.float32(65535.0))[None, None]


def test_generator_is_pinned_sparse_and_anisotropic():
    v, truth = S.structured_volume(3, 96, with_truth=True)
    assert v.dtype == np.uint16 and v.shape == (96, 96, 96) and truth.shape == v.shape
    assert hashlib.sha256(v.tobytes()).hexdigest().startswith('119685fd69da3d88')  # elementwise numpy only: same bytes everywhere
    assert np.array_equal(v, S.structured_volume(3, 96))
    assert not np.array_equal(v, S.structured_volume(4, 96))
    med = float(np.median(v))
    assert 0.001 < float((v > 1.5 * med).mean()) < 0.03  # a dark volume with a few bright structures
    # the blur is much wider along z than along x: compare the autocorrelation of the (noise-free part of the) signal at lag 3
    f = v.astype(np.float64) - med
    f[f < 0.25 * med] = 0.0

    def ac(a, axis, lag):
        n = a.shape[axis]
        s0 = [slice(None)] * 3
        s1 = [slice(None)] * 3
        s0[axis] = slice(0, n - lag)
        s1[axis] = slice(lag, n)
        return float((a[tuple(s0)] * a[tuple(s1)]).sum() / (a * a).sum())
    assert ac(f, 0, 3) > 2.0 * ac(f, 2, 3)
    # uint8 form and the ragged form exist too
    assert S.structured_volume(1, (20, 33, 41), np.uint8).dtype == np.uint8


def test_fixture_inputs_come_from_the_generator(golden_dir):
    assert np.array_equal(G(golden_dir, 'unet_deconv_struct_32.npz')['x'], struct_crop(5, 32))
    assert np.array_equal(G(golden_dir, 'apollo_step_struct_36.npz')['real'], struct_crop(6, 36))
    assert np.array_equal(G(golden_dir, 'athena_step_struct_36.npz')['real'], struct_crop(7, 36))


def test_oracle_unet_deconv_on_a_structured_crop(golden_dir):
    g = G(golden_dir, 'unet_deconv_struct_32.npz')
    sd = nets.to_torch(S.weights_from_seed(S.unet_deconv_spec(), int(g['seed'])), requires_grad=True)
    x = torch.from_numpy(g['x']).requires_grad_(True)
    y = nets.unet_deconv(sd, x)
    np.testing.assert_allclose(y.detach().numpy(), g['y'], atol=2e-6)
    r = torch.from_numpy(np.random.default_rng(int(g['r_seed'])).random(tuple(y.shape), dtype=np.float32))
    (y * r).mean().backward()
    np.testing.assert_allclose(x.grad.numpy(), g['dx'], atol=1e-4 * np.abs(g['dx']).max())
    for i, (k, p) in enumerate(sd.items()):
        l2 = float(np.sqrt((p.grad.numpy().astype(np.float64) ** 2).sum()))
        assert abs(l2 - g['g_l2'][i]) <= 2e-4 * max(1e-6, g['g_l2'][i]), k


def test_oracle_apollo_step_on_a_structured_crop(golden_dir):
    g = G(golden_dir, 'apollo_step_struct_36.npz')
    specs = [('G_A', S.unet_deconv_spec()), ('G_B', S.deep_linear_spec())] + [(n, S.patchgan_spec(2)) for n in apollo.APOLLO_D]
    sds = {n: S.weights_from_seed(sp, int(g['net_seed0']) + i) for i, (n, sp) in enumerate(specs)}
    model = apollo.ApolloOracle(sds)
    np.random.seed(int(g['step_seed']))
    names = [str(s) for s in g['loss_names']]
    for it in range(2):
        L = model.step(torch.from_numpy(g['real']))
        np.testing.assert_allclose(np.array([L[k] for k in names]), g['losses'][it], rtol=2e-4, err_msg='step %d' % it)
        if it == 0:
            np.testing.assert_allclose(model.fake.detach().numpy(), g['fake0'], atol=2e-6)


def test_oracle_athena_step_on_a_structured_crop(golden_dir):
    g = G(golden_dir, 'athena_step_struct_36.npz')
    names_n = ['G_A', 'G_B'] + apollo.ATHENA_D
    specs = [S.unet_deconv_spec(), S.deep_linear_spec()] + [S.patchgan_spec(2)] * 6
    sds = {n: S.weights_from_seed(sp, int(g['net_seed0']) + i) for i, (n, sp) in enumerate(zip(names_n, specs))}
    model = apollo.AthenaOracle(sds)
    names = [str(s) for s in g['loss_names']]
    for it in range(2):
        L = model.step(torch.from_numpy(g['real']))
        np.testing.assert_allclose(np.array([L[k] for k in names]), g['losses'][it], rtol=2e-4, err_msg='step %d' % it)
