"""GPU parity at BASELINE.json's full sizes through size-independent properties (the oracle cannot run these sizes in
seconds): adjointness of fwd / dgrad / wgrad of the MFMA convolutions at 108^3, linearity of deep_linear_gen at 108^3,
whole-network C entry point vs the layer-by-layer path at 140^3, and the dice -> identity -> assemble round trip on a
900^3 uint16 volume (729 cubes of 140^3, the reference screenshot's geometry)."""
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


def _dot(a, b):
    return float((a.double() * b.double()).sum())


@pytest.mark.parametrize('C,K,k,E,N', [(64, 64, 3, 108, 1), (128, 64, 3, 108, 1), (64, 64, 5, 108, 1),
                                       (256, 128, 3, 54, 1), (256, 256, 3, 27, 1), (1, 64, 7, 108, 1),
                                       (64, 64, 3, 148, 2), (128, 128, 3, 74, 2), (64, 32, 1, 108, 1)])
def test_conv_adjoint_identities(C, K, k, E, N):
    """<conv(x, w), r> = <x, dgrad(r, w)> = <w, wgrad(x, r)>: the three kernels compute one bilinear form.  Sizes: the
    108^3 / 54^3 / 27^3 levels of BASELINE configs[1], the single-channel 7^3 and pointwise layers of G_B, and the
    148^3 / 74^3 levels of configs[3] with a batch of 2."""
    g = torch.Generator().manual_seed(1)
    x = (torch.rand((N, C, E, E, E), generator=g) - 0.5).to(DEV)
    w = ((torch.rand((K, C, k, k, k), generator=g) - 0.5) * 0.1).to(DEV)
    y = ops.conv_fwd_raw(x, w, None, 1, k // 2)
    r = (torch.rand(y.shape, generator=g) - 0.5).to(DEV)
    a = _dot(y, r)
    b = _dot(x, ops.conv_dgrad_raw(r, w, x.shape, 1, k // 2))
    dw, _ = ops.conv_wgrad_raw(x, r, w.shape, 1, k // 2, False)
    c = _dot(w, dw)
    scale = float(y.double().norm() * r.double().norm())
    print(C, K, k, E, a, b, c, scale)
    assert abs(a - b) < 1e-5 * scale and abs(a - c) < 1e-5 * scale


def test_deep_linear_is_linear_at_108():
    net = networks.define_G(1, 1, 64, 'deep_linear_gen', 'instance', False, 'kaiming', 0.02, [0])
    net.load_state_dict(S.state_dict_from_seed(S.deep_linear_spec(), 9, DEV))
    g = torch.Generator().manual_seed(2)
    x1 = torch.rand((1, 1, 108, 108, 108), generator=g).to(DEV)
    x2 = torch.rand((1, 1, 108, 108, 108), generator=g).to(DEV)
    with torch.no_grad():
        y1, y2, y12 = net(x1), net(x2), net(0.25 * x1 - 1.5 * x2)
    err = float((y12 - (0.25 * y1 - 1.5 * y2)).abs().max() / y12.abs().max())
    assert err < 2e-5, err


def test_unet_fused_entry_matches_layerwise_at_140():
    net = networks.define_G(1, 1, 64, 'unet_deconv', 'instance', False, 'kaiming', 0.02, [0])
    net.load_state_dict(S.state_dict_from_seed(S.unet_deconv_spec(), 4, DEV))
    x = torch.rand((1, 1, 140, 140, 140), generator=torch.Generator().manual_seed(3)).to(DEV)
    with torch.no_grad():
        fused = net(x)                    # nc_unet_deconv_fwd
    with torch.enable_grad():
        layerwise = net(x.clone().requires_grad_(True)).detach()
    assert float((fused - layerwise).abs().max()) < 1e-5
    assert 0.0 < float(fused.min()) and float(fused.max()) < 1.0


def test_dice_identity_roundtrip_900():
    from neuroclear_amd.data.diceImage_dataset import DiceImageDataSet
    from neuroclear_amd.util.assemble_dice import Assemble_Dice
    vol = S.random_volume(0, 900)
    opt = Namespace(dice_size=[120] * 3, overlap=15, border_cut=10, gpu_ids=[0], skip_real=True, data_type='uint16',
                    histogram_match=False, normalize_intensity=False)
    ds = DiceImageDataSet(opt, volume=vol)
    assert ds.size() == (960, 960, 960) and ds.shape() == (9, 9, 9) and len(ds) == 729
    asm = Assemble_Dice(opt, vol.shape)
    for i in range(len(ds)):
        c = ds[i]['A']
        assert c.shape == (1, 140, 140, 140)
        asm.addToStack(dict(fake=c.unsqueeze(0)))
    asm.assemble_all()
    out = asm.getDict()['fake']
    assert out.shape == vol.shape and out.dtype == np.uint16
    d = np.abs(out.astype(np.int32) - vol.astype(np.int32))
    assert int(d.max()) <= 1  # identity network: the reference's own round trip is within 1 LSB


def test_apollo_step_148_batch2():
    """BASELINE configs[3] shape class (148^3 crops, batch > 1) in fp32: one full optimisation step runs through
    every kernel variant that a 148-wide, batched volume selects (two column blocks in k_wgrad_dma, 74 / 37-wide
    levels with row tails), losses and every parameter update stay finite, and the batch really is processed (the
    cycle loss of the batch lies between the per-sample cycle losses)."""
    from neuroclear_amd.models import create_model
    opt = Namespace(gpu_ids=[0], isTrain=True, image_dimension=3, checkpoints_dir='/tmp/nc_ckpt', name='t148',
                    preprocess='none', gan_mode='lsgan', randomize_projection_depth=True, projection_depth=10,
                    min_projection_depth=2, lambda_plane=[1, 1, 1], lambda_A=5.0, input_nc=1, output_nc=1, ngf=64,
                    ndf=64, netG='unet_deconv', netG_B='deep_linear_gen', netD='basic', n_layers_D=3,
                    norm='instance', no_dropout=True, init_type='kaiming', init_gain=0.02, lr=1e-4, beta1=0.1,
                    direction='AtoB', model='axial_to_lateral_gan_apollo')
    torch.manual_seed(11)
    np.random.seed(12)
    model = create_model(opt)
    g = torch.Generator().manual_seed(13)
    real = torch.rand((2, 1, 148, 148, 148), generator=g)
    before = [p.detach().clone() for p in model.netG_A.parameters()]
    model.set_input({'A': real, 'A_paths': 'x'})
    model.optimize_parameters()
    losses = model.get_current_losses()
    assert all(np.isfinite(v) for v in losses.values()), losses
    moved = [float((a.detach() - b).abs().max()) for a, b in zip(model.netG_A.parameters(), before)]
    assert all(np.isfinite(m) for m in moved) and max(moved) > 0
    with torch.no_grad():
        per = [float((model.rec[i] - model.real[i]).abs().mean()) * 5.0 for i in range(2)]
    assert min(per) - 1e-4 <= losses['cycle'] <= max(per) + 1e-4, (per, losses['cycle'])


def test_config0_diced_inference_256():
    """BASELINE configs[0] on the product path: 256^3 volume, dice 64^3, overlap 8 (border_cut 8 -> 125 cubes of 80^3,
    padded 288^3).  Every cube goes through the HIP network; the oracle checks (i) three cubes end to
    end on the CPU (cut + normalise + Unet_deconv forward), (ii) the assembler: the oracle's overlap-add of the
    product's own cube outputs must give the product's uint16 volume bit for bit."""
    from neuroclear_amd.data.diceImage_dataset import DiceImageDataSet
    from neuroclear_amd.util.assemble_dice import Assemble_Dice
    from oracle import dice as odice
    from oracle import nets as onets
    vol = S.random_volume(7, 256)
    R, ov, b = 64, 8, 8
    opt = Namespace(dice_size=[R] * 3, overlap=ov, border_cut=b, gpu_ids=[0], skip_real=True, data_type='uint16',
                    histogram_match=False, normalize_intensity=False)
    sd_np = S.weights_from_seed(S.unet_deconv_spec(), 21)
    net = networks.define_G(1, 1, 64, 'unet_deconv', 'instance', False, 'kaiming', 0.02, [0])
    net.load_state_dict({k: torch.from_numpy(v).to(DEV) for k, v in sd_np.items()})
    ds = DiceImageDataSet(opt, volume=vol)
    assert ds.size() == (288, 288, 288) and ds.shape() == (5, 5, 5) and len(ds) == 125
    asm = Assemble_Dice(opt, vol.shape)
    outs = []
    with torch.no_grad():
        for i in range(len(ds)):
            y = net(ds[i]['A'].unsqueeze(0))
            outs.append(y.reshape(80, 80, 80).cpu().numpy())
            asm.addToStack(dict(fake=y))
    asm.assemble_all()
    got = asm.getDict()['fake']
    assert got.shape == vol.shape and got.dtype == np.uint16
    # (i) three cubes against the CPU oracle
    padded = odice.pad_for_dicing(vol, R, ov)
    steps = odice.grid_steps(padded.shape, R, ov)
    refl = odice.reflect_pad(padded, b)
    sd_t = onets.to_torch(sd_np)
    torch.set_num_threads(max(1, min(32, torch.get_num_threads())))
    for i in (0, 62, 124):
        cube = odice.normalize(odice.cut_cube(refl, i, steps, R, ov, b))
        with torch.no_grad():
            ref = onets.unet_deconv(sd_t, torch.from_numpy(cube)[None, None]).numpy()[0, 0]
        assert float(np.abs(outs[i] - ref).max()) < 2e-5, i
    # (ii) the oracle's assembler on the same cube values
    ref_vol = odice.assemble(outs, padded.shape, vol.shape, R, ov, b, 'uint16')
    assert np.array_equal(got, ref_vol)


def test_diced_inference_cubes_in_flight_is_bit_identical(monkeypatch):
    """Three cubes in flight on three HIP streams (the default) against one cube at a time: the overlap-adds stay in cube
    order, so the assembled volume is the same array; also with the input volume assembled beside it (with_real)."""
    from neuroclear_amd.test_dice import diced_inference
    vol = S.random_volume(23, (96, 110, 75))
    opt = Namespace(dice_size=[32] * 3, overlap=4, border_cut=4, gpu_ids=[0], skip_real=True, data_type='uint16',
                    histogram_match=False, normalize_intensity=False)
    net = networks.define_G(1, 1, 64, 'unet_deconv', 'instance', False, 'kaiming', 0.02, [0])
    net.load_state_dict(S.state_dict_from_seed(S.unet_deconv_spec(), 5, DEV))
    out = {}
    for n in ('1', '3', '4'):
        monkeypatch.setenv('NC_INFER_STREAMS', n)
        out[n] = diced_inference(net, vol, opt, with_real=True)
    for n in ('3', '4'):
        assert np.array_equal(out['1'][0], out[n][0]) and np.array_equal(out['1'][1], out[n][1])
    assert float(np.abs(out['3'][1].astype(np.int64) - vol.astype(np.int64)).max()) <= 1  # the dice -> assemble round trip of the input


def test_diced_inference_reduce_mode_matches_in_order():
    """assemble='reduce' (each rank overlap-adds its own cubes, one reduce(sum)) on the device: at world 1 it is the
    in-order path bit for bit; two emulated ranks (cubes i % 2, accumulators summed as RCCL's reduce would) stay within
    1 LSB of it."""
    from neuroclear_amd.data.diceImage_dataset import DiceImageDataSet
    from neuroclear_amd.test_dice import diced_inference
    from neuroclear_amd.util.assemble_dice import Assemble_Dice
    vol = S.random_volume(17, (100, 90, 80))
    opt = Namespace(dice_size=[32] * 3, overlap=4, border_cut=4, gpu_ids=[0], skip_real=True, data_type='uint16',
                    histogram_match=False, normalize_intensity=False)
    net = networks.define_G(1, 1, 64, 'unet_deconv', 'instance', False, 'kaiming', 0.02, [0])
    net.load_state_dict(S.state_dict_from_seed(S.unet_deconv_spec(), 4, DEV))
    a = diced_inference(net, vol, opt, assemble='gather')
    b = diced_inference(net, vol, opt, assemble='reduce')
    assert np.array_equal(a, b)
    ds = DiceImageDataSet(opt, volume=vol)
    asms = [Assemble_Dice(opt, vol.shape) for _ in range(2)]
    with torch.no_grad():
        for i in range(len(ds)):
            asms[i % 2].add_cube('fake', net(ds[i]['A'].unsqueeze(0)).reshape(40, 40, 40), i)
    asms[0].acc['fake'] += asms[1].acc['fake']
    asms[0].count['fake'] = asms[0].len_cube_queue
    asms[0].assemble_all()
    c = asms[0].getDict()['fake']
    assert int(np.abs(c.astype(np.int64) - a.astype(np.int64)).max()) <= 1


def _apollo_108(monkeypatch, d_streams, fused_patchgan, seed=31):
    import contextlib
    import io
    from neuroclear_amd.models import create_model
    from neuroclear_amd.models.axial_to_lateral_gan_apollo_model import AxialToLateralGANApolloModel
    monkeypatch.setattr(AxialToLateralGANApolloModel, '_d_streams_on', d_streams)
    monkeypatch.setenv('NC_FUSED_PATCHGAN', '1' if fused_patchgan else '0')
    opt = Namespace(gpu_ids=[0], isTrain=True, image_dimension=3, checkpoints_dir='/tmp/nc_ckpt', name='t108',
                    preprocess='none', gan_mode='lsgan', randomize_projection_depth=True, projection_depth=10,
                    min_projection_depth=2, lambda_plane=[1, 1, 1], lambda_A=5.0, input_nc=1, output_nc=1, ngf=64,
                    ndf=64, netG='unet_deconv', netG_B='deep_linear_gen', netD='basic', n_layers_D=3,
                    norm='instance', no_dropout=True, init_type='kaiming', init_gain=0.02, lr=1e-4, beta1=0.1,
                    direction='AtoB', model='axial_to_lateral_gan_apollo')
    torch.manual_seed(seed)
    np.random.seed(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        model = create_model(opt)
    v = S.random_volume(77, 108)
    real = torch.from_numpy((v.astype(np.float64) / 65535.0).astype(np.float32))[None, None].to(DEV)
    out = []
    for it in range(2):  # the second step runs on updated weights and with the early-start discriminator phase primed
        model.set_input({'A': real, 'A_paths': 'x'})
        model.optimize_parameters()
        out.append(dict(model.get_current_losses()))
    with torch.no_grad():
        fused_fake = model.netG_A(real)  # whole-network C entry point (nc_unet_deconv_fwd) on the updated weights
        with torch.enable_grad():
            layer_fake = model.netG_A(real)  # layer-by-layer path
    params = torch.cat([p.detach().reshape(-1) for p in model.optimizer_G.params + model.optimizer_D.params]).clone()
    return out, params, float((fused_fake - layer_fake.detach()).abs().max())


def test_apollo_step_108_streams_and_fused_paths_agree(monkeypatch):
    """BASELINE configs[1] at its size: two full optimisation steps at 108^3 with (a) the four discriminator chains on four
    HIP streams, started under the generators' backward pass, vs everything in sequence on one stream, and (b) the
    whole-network PatchGAN entry points vs the op-by-op path.  The kernels and their order per network are the same, and
    no kernel uses atomics, so every loss and every updated parameter must be BIT-equal -- a missing event or a scratch
    buffer shared between streams shows up here, where the kernels actually overlap; plus `fake` from the fused
    whole-network forward vs the layer-wise path <= 1e-5."""
    base, p0, dfake = _apollo_108(monkeypatch, True, True)
    assert dfake <= 1e-5, dfake
    for ds, fp in ((False, True), (True, False)):
        other, p1, _ = _apollo_108(monkeypatch, ds, fp)
        for it in range(2):
            for k in base[it]:
                assert base[it][k] == other[it][k], (ds, fp, it, k, base[it][k], other[it][k])
        assert torch.equal(p0, p1), (ds, fp)


def test_apollo_step_108_two_term_and_three_term_forms_agree(monkeypatch):
    """The same two optimisation steps at 108^3 on the two-term fp16 form of the split-operand convolutions (the default) and on the
    three-term bf16 form: two fp32-accurate evaluations of the same step -- every loss of the first step within 2e-5, of the second (one Adam
    update apart, ReLU / max-pool decisions may differ) within 5e-3, the generator's output on the updated weights within 1e-4."""
    from neuroclear_amd._lib import lib
    assert lib().nc_get_split_terms() == 2
    two, _, d2 = _apollo_108(monkeypatch, True, True)
    lib().nc_set_split_terms(3)
    try:
        three, _, d3 = _apollo_108(monkeypatch, True, True)
    finally:
        lib().nc_set_split_terms(2)
    assert d2 <= 1e-5 and d3 <= 1e-5
    for it, tol in ((0, 2e-5), (1, 5e-3)):
        for k in two[it]:
            assert abs(two[it][k] - three[it][k]) <= tol * max(abs(three[it][k]), 1e-3), (it, k, two[it][k], three[it][k])


def _athena_108(monkeypatch, d_streams, reuse, tune, seed=41):
    import contextlib
    import io
    from neuroclear_amd._lib import I, lib
    from neuroclear_amd.models import create_model
    from neuroclear_amd.models.axial_to_lateral_gan_athena_model import AxialToLateralGANAthenaModel
    monkeypatch.setattr(AxialToLateralGANAthenaModel, '_d_streams_on', d_streams)
    monkeypatch.setattr(AxialToLateralGANAthenaModel, '_reuse_on', reuse)
    lib().nc_sconv_set_tune(I(1 if tune else 0))
    opt = Namespace(gpu_ids=[0], isTrain=True, image_dimension=3, checkpoints_dir='/tmp/nc_ckpt', name='t108a', preprocess='none',
                    gan_mode='lsgan', randomize_projection_depth=True, projection_depth=10, min_projection_depth=2,
                    lambda_plane=[1, 1, 1], lambda_A=5.0, input_nc=1, output_nc=1, ngf=64, ndf=64, netG='unet_deconv',
                    netG_B='deep_linear_gen', netD='basic', n_layers_D=3, norm='instance', no_dropout=True, init_type='kaiming',
                    init_gain=0.02, lr=1e-4, beta1=0.1, direction='AtoB', model='axial_to_lateral_gan_athena',
                    conversion_plane=['yz', 'xy'], pool_size=50)
    torch.manual_seed(seed)
    np.random.seed(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        model = create_model(opt)
    v = S.structured_volume(78, 108)
    real = torch.from_numpy((v.astype(np.float64) / 65535.0).astype(np.float32))[None, None].to(DEV)
    out = []
    try:
        for it in range(2):  # the second step runs on updated weights, with the tile shapes of the first one cached
            model.set_input({'A': real, 'A_paths': 'x'})
            model.optimize_parameters()
            out.append(dict(model.get_current_losses()))
    finally:
        lib().nc_sconv_set_tune(I(-1))
    params = torch.cat([p.detach().reshape(-1) for p in model.optimizer_G.params + model.optimizer_D.params]).clone()
    return out, params


def test_athena_step_108_streams_tuner_and_shared_pass_agree(monkeypatch):
    """BASELINE configs[4] at its size (every one of the 108 slices of a 108^3 crop through six 2-D discriminators: batches of
    108-216 planes of 108^2, where the image-staged kernels, their first-call tile-shape tuner, the shared fake-plane pass and the
    six-stream overlap actually engage): two full optimisation steps on a structured crop with (a) the discriminator work on six HIP
    streams vs one, (b) the tile-shape tuner on vs the fixed heuristic, (c) the shared pass over the fake planes vs both passes run.
    Every tile shape accumulates in the same order, no kernel uses atomics and the shared pass evaluates the same planes through the
    same weights, so every loss and every updated parameter must be BIT-equal -- a missing stream event, a scratch buffer shared
    between streams or a tuner launch that leaks into results shows up here, where it cannot at the 36^3 of the golden step.
    (a) and (b) hold in the default arithmetic.  (c) holds bit for bit with the PatchGAN layers on the three-term form (nc_set_p2d_terms(3));
    the default since round 5 -- the stride-1 layer on two fp16 terms of the tensor times a power of two MEASURED per call -- sees another
    power of two when 216 planes share the call than when 108 do: same planes, results equal to fp32 rounding."""
    from neuroclear_amd._lib import lib
    prev = lib().nc_get_p2d_terms()
    try:
        for mode in (prev, 3):
            lib().nc_set_p2d_terms(mode)
            base, p0 = _athena_108(monkeypatch, True, True, True)
            assert all(np.isfinite(list(s.values())).all() for s in base)
            cases = ((False, True, True), (True, True, False), (True, False, True)) if mode == prev else ((True, False, True),)
            for ds, ru, tu in cases:
                other, p1 = _athena_108(monkeypatch, ds, ru, tu)
                exact = mode == 3 or ru
                for it in range(2):
                    for k in base[it]:
                        if exact:
                            assert base[it][k] == other[it][k], (mode, ds, ru, tu, it, k, base[it][k], other[it][k])
                        else:
                            assert abs(base[it][k] - other[it][k]) <= 2e-5 * max(abs(base[it][k]), 1e-3), (mode, it, k, base[it][k], other[it][k])
                if exact:
                    assert torch.equal(p0, p1), (mode, ds, ru, tu)
                else:  # (Adam's first steps move every weight by ~lr whatever the gradient's size: compare in the L2 norm of the update)
                    assert float((p0 - p1).norm()) <= 2e-2 * 1e-4 * float(p0.numel()) ** 0.5, float((p0 - p1).norm())
    finally:
        lib().nc_set_p2d_terms(prev)


def test_diced_inference_slab_mode_single_rank_is_bit_identical():
    """assemble='slab' with one rank walks the same code as the sharded run (slab accumulator addressed through a shifted base pointer,
    owner-side nc_assemble_finalize_slab, integer slabs concatenated) and adds the cubes in index order, so it must reproduce the
    in-order assembler bit for bit; the multi-rank schedule itself is covered on CPU (tests/test_dist_cpu.py, gloo, world 2-4)."""
    from neuroclear_amd.test_dice import diced_inference, slab_plan
    from neuroclear_amd.util import util as U
    vol = S.structured_volume(13, (150, 96, 110))
    opt = Namespace(dice_size=[48] * 3, overlap=8, border_cut=4, gpu_ids=[0], skip_real=True, data_type='uint16', histogram_match=False,
                    normalize_intensity=False)
    net = networks.define_G(1, 1, 64, 'unet_deconv', 'instance', False, 'kaiming', 0.02, [0])
    net.load_state_dict({k: torch.from_numpy(v).to(DEV) for k, v in S.weights_from_seed(S.unet_deconv_spec(), 22).items()})
    a = diced_inference(net, vol, opt, assemble='gather')
    b = diced_inference(net, vol, opt, assemble='slab')
    assert a.dtype == np.uint16 and a.shape == vol.shape and np.array_equal(a, b)
    # the plan for 8 ranks on the 900^3 geometry: 91 or 92 cubes each, at most three z-layers (330 of 960 planes) per local accumulator
    padded = U.padded_shape((900, 900, 900), 120, 15)
    plan = slab_plan(U.grid_steps(padded, 120, 15), 105, 120, padded[0], 8)
    assert sorted({c[1] - c[0] for c in plan['cubes']}) == [91, 92] and max(z[1] - z[0] for z in plan['local']) <= 330
    assert [o[1] - o[0] for o in plan['own']] == [120] * 8


# ---- full-size comparisons WITH the oracle (round 6): the same functions bench.py prints as `parity_vs_cpu_oracle`, same bounds

def _bench():
    import importlib
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    return importlib.import_module('bench')


def test_apollo_step_108_against_the_oracle():
    """The first optimize_parameters() at 108^3 (BASELINE configs[1]) from the same seeded weights, crop and np.random draws as
    oracle/apollo.py (reference axial_to_lateral_gan_apollo_model.py:285-307): all 11 losses to 2e-5, `fake` after the sigmoid to 2e-5
    absolute, `rec` to 2e-4 of its largest magnitude.  The oracle's step takes ~11 s on the GPU box's host cores."""
    B = _bench()
    dev = torch.device('cuda', 0)
    gpu = B.gpu_parity_train(dev, 108)
    model, real = B._oracle_apollo(108, min(os.cpu_count() or 1, 64))
    ref_losses = model.step(real)
    ref = dict(losses=dict(ref_losses), fake=model.fake.detach().numpy(), rec=model.rec.detach().numpy())
    r = B.parity_train(gpu, ref)
    print(r)
    assert r['n_losses'] == 11
    assert r['first_step_losses_max_rel_diff'] <= 2e-5
    assert r['fake_max_abs_diff'] <= 2e-5
    assert r['rec_max_rel_diff'] <= 2e-4
    assert r['ok']


def test_two_cubes_140_and_their_slab_against_the_oracle():
    """Two 140^3 cubes (BASELINE configs[2]'s cube: dice 120 / overlap 15 / border 10) of a small uint16 volume through the product's
    diced inference and through oracle/dice.py + oracle/nets.py (reference test_dice.py:107-118, util/assemble_dice.py:130-213): network
    outputs after the sigmoid to 2e-5, the assembled uint16 slab to 2 LSB (the truncating cast turns 1e-5 into one count)."""
    B = _bench()
    dev = torch.device('cuda', 0)
    gpu = B.gpu_parity_infer(dev)
    ref = B.cpu_baseline_infer(budget_s=0.0, max_cubes=2)['_parity_ref']
    r = B.parity_infer(gpu, ref)
    print(r)
    assert r['cubes'] == 2
    assert r['cube_max_abs_diff'] <= 2e-5
    assert r['slab_max_lsb_diff'] <= 2
    assert r['slab_equal_share'] > 0.9
    assert r['ok']


def test_batched_cubes_per_call_match_one_cube_per_call(monkeypatch):
    """Round 6: diced inference hands NC_INFER_BATCH cubes to each nc_unet_deconv_fwd call (the two-term mode runs the whole batch through every
    launch).  A cube's output may differ from its one-cube-per-call value only by the fp32 rounding of the InstanceNorm partial sums (their
    grouping follows the launch's tile plan), carried through ten normalised layers: <= 1e-5 after the sigmoid (the stated bound against the reference is 2e-5), the assembled uint16 volume within 1 LSB, and run-to-run identical.
    Also the raw C entry point: N = 3 in one call against three N = 1 calls."""
    from neuroclear_amd.test_dice import diced_inference
    net = networks.define_G(1, 1, 64, 'unet_deconv', 'instance', False, 'kaiming', 0.02, [0])
    net.load_state_dict(S.state_dict_from_seed(S.unet_deconv_spec(), 6, DEV))
    x = torch.rand((3, 1, 80, 72, 64), generator=torch.Generator().manual_seed(9)).to(DEV)
    with torch.no_grad():
        y3, y3b = net(x), net(x)
        y1 = torch.cat([net(x[i:i + 1]) for i in range(3)])
    assert torch.equal(y3, y3b)
    assert float((y3 - y1).abs().max()) <= 1e-5
    vol = S.random_volume(29, (200, 150, 130))
    opt = Namespace(dice_size=[64] * 3, overlap=8, border_cut=8, gpu_ids=[0], skip_real=True, data_type='uint16',
                    histogram_match=False, normalize_intensity=False)
    out = {}
    for b, st in (('1', '1'), ('5', '2'), ('3', '3'), ('5', '2')):
        monkeypatch.setenv('NC_INFER_BATCH', b)
        monkeypatch.setenv('NC_INFER_STREAMS', st)
        out.setdefault((b, st), []).append(diced_inference(net, vol, opt))
    assert np.array_equal(out[('5', '2')][0], out[('5', '2')][1])
    for k in (('5', '2'), ('3', '3')):
        d = np.abs(out[k][0].astype(np.int64) - out[('1', '1')][0].astype(np.int64))
        assert int(d.max()) <= 1 and float((d > 0).mean()) < 0.05, (k, int(d.max()), float((d > 0).mean()))
