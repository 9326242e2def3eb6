"""bench.py -- headline benchmark of the MI355X hot path (contract: see the task statement / DESIGN.md 'Measurement').

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

BASELINE.json's metric has two halves -- "voxels/sec: unet_deconv 108^3 train step + 900^3 diced inference" -- and the
default run (`--workload headline`) measures both in one process: the train step is the top-level `value` (it is the
half BASELINE.json quotes on configs[1]), the 900^3 diced inference (configs[2]) is the nested `inference` object with
its own `roofline` and `cpu_baseline`.

Train leg at every N (BASELINE.json configs[1]): one *step* = one `optimize_parameters()` of the Apollo model
(axial_to_lateral_gan_apollo: unet_deconv G_A + deep_linear_gen G_B + four 2-D PatchGANs, LSGAN + InstanceNorm,
fp32) on one 108^3 crop, batch 1, per GPU.  Inputs are synthetic random uint16 crops already resident in HBM; weights
are random-init (kaiming), as the reference would start.  Scaling is WEAK: every rank trains on its own crop and the
gradients are all-reduced (RCCL) once per optimizer phase.  value = voxels fed to G_A per second over all ranks.

Extra objects on the JSON line:
  roofline     -- for the kernel class that took most of the timed region: algorithmic FLOP / launch, measured with HIP
                  events on the launch stream inside the timed steps (neuroclear_amd.ops.prof), against the dense MFMA
                  peak of /opt/skills/guides/MI355X_MICROARCH.md for the instruction the class runs on: 157.3 TFLOP/s
                  (fp32 MFMA), or -- for the split-operand kernels -- the 16-bit dense peak over the MFMA products one fp32
                  product costs: 2500 / 3 = 833.3 TFLOP/s of fp32 products for the two-term fp16 form (the default since
                  round 4, csrc/conv_s3x.hip NT = 2), 2500 / 6 = 416.7 for the three-term bf16 form (`frac_of_six_product_roof`
                  keeps the round-3 yardstick next to `frac`).
  arithmetic / three_term_split / fp32_mfma_kernels -- what "f32" is computed with, and the same step with the split-operand
                  layers on the three-term form and on the fp32 MFMA kernels (3 steps each, same process, same seeds).
  cpu_baseline -- the oracle's CPU restatement of the same step (oracle/apollo.py, torch-CPU fp32) on the host cores of
                  the GPU box, rank 0 only (at every N, after the last GPU leg): full 108^3 step, 1 warm-up + median of 3 on all cores, plus a
                  1-thread figure on a bounded 36^3 sample; inference: 140^3 cubes through
                  oracle/nets.py + oracle/dice.py for a bounded time, extrapolated to 729 cubes (stated).
The inference leg keeps NC_INFER_STREAMS (3) cubes in flight on as many HIP streams; its `roofline.achieved` is therefore all cubes'
FLOP over the wall time of the timed region (`event_ms_per_cube` = the mean HIP-event interval of one whole-network call, which now
overlaps its neighbours').
`--workload train` / `--workload infer` run one leg only (infer = BASELINE.json configs[2], cubes sharded over ranks);
`--crop 148 --batch 4` is the shape of configs[3] (in fp32), `--model athena` the step of configs[4].
"""
import argparse
import json
import os
import sys
import time
from argparse import Namespace

import numpy as np

os.environ.setdefault('GPU_MAX_HW_QUEUES', '4')  # (neuroclear_amd/__init__.py says why; here too, in front of the first HIP call of this process)

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: "Peak FP32 (matrix) 157.3 TFLOPS"
MFMA_16BIT_PEAK_TFLOPS = 2500.0  # same guide: "Peak BF16/FP16 MFMA ~2.5 PF dense"
# The fp32 3^3 / 5^3 convolutions run on the bf16 matrix cores (csrc/conv_split.hip): each fp32 operand is the exact sum of three
# bf16 terms and an fp32 product costs six bf16 MFMA products.  Roofline of those kernels: algorithmic fp32 FLOP against the
# bf16 dense peak / 6 (what the matrix cores can deliver of THIS arithmetic); the fp32 MFMA peak is printed next to it.
SPLIT_PRODUCTS = 6
MFMA_SPLIT_PEAK_TFLOPS = MFMA_16BIT_PEAK_TFLOPS / SPLIT_PRODUCTS
ARITHMETIC = ('fp32 operands, fp32 accumulation; 3^3/5^3 convolution products as 6 bf16 MFMA products of an exact 3-term operand '
              'split (csrc/conv_s3x.hip, conv_split.hip; error vs fp64 <= the fp32 MFMA kernels\': tests/test_gpu_split.py); NC_CONV_SPLIT=0 = fp32 MFMA kernels')
# Round 4, the default: the TWO-term fp16 form -- operands as two fp16 terms of the tensor times a per-tensor power of two, three MFMA products
# per fp32 product (csrc/conv_s3x.hip NT = 2, csrc/h2.hip, k_wgrad_s3x<KS, 2, f16>).  Its roofline is the 16-bit dense peak / 3; the
# three-term step is timed next to it (`three_term_split`), as the fp32 MFMA kernels' is (`fp32_mfma_kernels`).
ARITHMETIC_H2 = ('fp32 operands, fp32 accumulation; 3^3/5^3 convolution products as 3 fp16 MFMA products of a 2-term operand split of the tensor times '
                 'a per-tensor power of two (csrc/conv_s3x.hip NT = 2, csrc/h2.hip; error vs fp64: the 3^3 forward / data-gradient tiles keep one running fp32 '
                 'accumulator (k_conv_s3w, NC_S3X_W64=1): 2^-24 sqrt(k-steps) rms, the fp32 MFMA kernels\' level; 5^3 layers, weight gradients and '
                 'NC_S3X_W64=0 restart their accumulators: <= the three-term form\'s, below the fp32 kernels\'; tests/test_gpu_h2.py); '
                 'NC_SPLIT_TERMS=3 = the three-term bf16 form (6 products), NC_CONV_SPLIT=0 = fp32 MFMA kernels')
KERNEL_OF = {  # profiler tag -> HIP kernel name to look for in profiles/*.csv
    'fwd_mfma_k3': 'k_conv_mfma<3,*>', 'dgrad_mfma_k3': 'k_conv_mfma<3,*>', 'fwd_mfma_k5': 'k_conv_mfma<5,*>',
    'dgrad_mfma_k5': 'k_conv_mfma<5,*>', 'wgrad_mfma_k3': 'k_wgrad_dma<3,2> (108^3) + k_wgrad_rows<3> (54^3, 27^3)',
    'wgrad_mfma_k5': 'k_wgrad_dma<5,1>',
    'fwd_split_k3': 'k_conv_s3x<3,*>', 'dgrad_split_k3': 'k_conv_s3x<3,*>',  # (two-term, whole 512-position tiles: k_conv_s3w, see below) 'fwd_split_k5': 'k_conv_s3x<5,*>',
    'dgrad_split_k5': 'k_conv_s3x<5,*>', 'wgrad_split_k3': 'k_wgrad_s3x<3>', 'wgrad_split_k5': 'k_wgrad_s3x<5>',
    'fwd_lp_k3': 'k_conv_c8x<*,3> (k_conv_h<*,3,3,3,*> for launches of a few planes)', 'dgrad_lp_k3': 'k_conv_c8x<*,3>',
    'fwd_lp_k5': 'k_conv_h<*,5,5,5,*>', 'dgrad_lp_k5': 'k_conv_h<*,5,5,5,*>', 'wgrad_lp_k3': 'k_wgrad_s3x<3,1,*>', 'wgrad_lp_k5': 'k_wgrad_s3x<5,1,*>',
    'fwd_lp_k7': 'k_conv_h<*,7,7,1,*> (pseudo-channel form)', 'dgrad_lp_k7': 'k_conv_h<*,7,7,1,*,1> + k_fold_x8',
}


# `roofline.traffic` is NOT measured in this process (counters cannot be read from inside it): it is the figure of the committed offline PMC
# passes of the same command, on whatever box those ran on
TRAFFIC_FILE = 'r06_pmc_traffic.json'
TRAFFIC_SOURCE = 'profiles/%s (offline rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; not measured in this run)' % TRAFFIC_FILE

PMC_CLASS = {'fwd_mfma_k3': 'conv_mfma_k3', 'dgrad_mfma_k3': 'conv_mfma_k3', 'fwd_mfma_k5': 'conv_mfma_k5',
             'dgrad_mfma_k5': 'conv_mfma_k5', 'wgrad_mfma_k3': 'wgrad_mfma_k3', 'wgrad_mfma_k5': 'wgrad_mfma_k5',
             # 16-bit classes: measured on ONE shape (64 -> 64, 4 x 148^3), so only reported for that workload
             'fwd_lp_k3': 'conv_h_k3', 'dgrad_lp_k3': 'conv_h_k3', 'fwd_lp_k5': 'conv_h_k5', 'dgrad_lp_k5': 'conv_h_k5',
             'wgrad_lp_k3': 'wgrad_h_k3', 'wgrad_lp_k5': 'wgrad_h_k5',
             'fwd_split_k3': 'conv_split_k3', 'dgrad_split_k3': 'conv_split_k3', 'fwd_split_k5': 'conv_split_k5',
             'dgrad_split_k5': 'conv_split_k5', 'wgrad_split_k3': 'wgrad_split_k3', 'wgrad_split_k5': 'wgrad_split_k5'}


def pmc_traffic(tag, crop=108, batch=1, three_term=False):
    """HBM bytes per launch of the kernel class, from the committed PMC passes of this same command
    (profiles/r06_pmc_traffic.json: FETCH_SIZE x 2 + WRITE_SIZE, separate rocprofv3 --pmc runs; counters cannot be
    read from inside the process).  None when the file or the class is missing."""
    if '_lp_' in tag and not (tag.endswith('k5') and crop == 148 and batch == 4):
        return None  # the 16-bit classes were counted on one shape only; a class of mixed shapes gets no figure
    if '_lp_' not in tag and (crop != 108 or batch != 1):
        return None
    try:
        with open(os.path.join(ROOT, 'profiles', TRAFFIC_FILE)) as f:
            key = ('split_classes_three_term' if three_term else 'split_classes') if '_split_' in tag else 'classes'
            return round(json.load(f)[key][PMC_CLASS[tag]]['hbm_bytes_per_launch'])
    except Exception:
        return None


def apollo_opt(gpu, model='apollo'):
    """The README training command (reference README.md:123-133) reduced to what the step uses.  model='athena':
    the artifact-correction variant of BASELINE configs[4] (--conversion_plane yz xy)."""
    o = _apollo_opt(gpu)
    if model == 'athena':
        o.model = 'axial_to_lateral_gan_athena'
        o.conversion_plane = ['yz', 'xy']
        o.pool_size = 50
    return o


def _apollo_opt(gpu):
    return Namespace(gpu_ids=[gpu], isTrain=True, image_dimension=3, checkpoints_dir='/tmp/nc_ckpt', name='bench',
                     preprocess='none', gan_mode='lsgan', randomize_projection_depth=True, projection_depth=10,
                     min_projection_depth=2, lambda_plane=[1, 1, 1], lambda_A=5.0, input_nc=1, output_nc=1, ngf=64,
                     ndf=64, netG='unet_deconv', netG_B='deep_linear_gen', netD='basic', n_layers_D=3,
                     norm='instance', no_dropout=True, init_type='kaiming', init_gain=0.02, lr=1e-4, beta1=0.1,
                     direction='AtoB', model='axial_to_lateral_gan_apollo')


def _oracle_apollo(crop, threads):
    import torch
    from neuroclear_amd.util import seed as S
    from oracle import apollo as oapollo
    torch.set_num_threads(threads)
    specs = [('G_A', S.unet_deconv_spec()), ('G_B', S.deep_linear_spec())] + \
        [(n, S.patchgan_spec(2)) for n in oapollo.APOLLO_D]
    sds = {n: S.weights_from_seed(sp, 7 + i, bias_scale=0.0) for i, (n, sp) in enumerate(specs)}
    model = oapollo.ApolloOracle(sds)
    real = torch.from_numpy((S.random_volume(11, crop).astype(np.float64) / 65535.0).astype(np.float32))[None, None]
    np.random.seed(0)
    return model, real


def cpu_baseline_train(crop=108, reps=3, budget_s=75.0):
    """Oracle Apollo step on the host cores (SURVEY.md 8d / BASELINE.md 3): full crop^3 step, 1 warm-up + median of
    `reps` on all cores (thread count reported); plus one 1-thread step on a 36^3 crop (a full-size 1-thread step takes
    minutes).  Bounded: if the warm-up step shows that `reps` more would pass budget_s, fewer are timed (stated)."""
    import torch
    ncores = min(os.cpu_count() or 1, 64)
    model, real = _oracle_apollo(crop, ncores)
    t0 = time.time()
    first = model.step(real)  # the FIRST step from the seeded weights: what gpu_parity_train() is compared with (parity_vs_cpu_oracle)
    warm = time.time() - t0
    parity_ref = dict(losses=dict(first), fake=model.fake.detach().numpy().copy(), rec=model.rec.detach().numpy().copy())
    reps = max(1, min(reps, int((budget_s - warm) / max(warm, 1e-3))))
    ts = []
    for _ in range(reps):
        t0 = time.time()
        model.step(real)
        ts.append(time.time() - t0)
    med = float(np.median(ts))
    del model, real
    m1, r1 = _oracle_apollo(36, 1)
    m1.step(r1)  # warm-up
    t0 = time.time()
    m1.step(r1)
    t1 = time.time() - t0
    torch.set_num_threads(ncores)
    return dict(value=crop ** 3 / med, unit='voxels/s', cores=ncores, kind='port', _parity_ref=parity_ref,
                sample='Apollo optimize_parameters() on a %d^3 crop (oracle/apollo.py, torch-CPU fp32): 1 warm-up (%.1f s) + '
                       'median of %d steps = %.2f s/step on %d threads' % (crop, warm, reps, med, ncores),
                one_thread=dict(value=36 ** 3 / t1, unit='voxels/s', cores=1, workload='apollo_train_step_36cube_bs1',
                                extrapolated=True,
                                sample='EXTRAPOLATED from a 36^3 crop: 1 warm-up + 1 timed step on 1 thread = %.1f s (a 108^3 step on one '
                                       'thread takes minutes); the per-voxel rate of the small crop stands in for the 108^3 one' % t1))


PARITY_SLAB_SHAPE = (100, 80, 80)  # pads to (225, 120, 120) at dice 120 / overlap 15: TWO 140^3 cubes stacked in z, overlapping by 15 planes
PARITY_NET_SEED, PARITY_VOL_SEED = 3, 21


def cpu_baseline_infer(budget_s=25.0, max_cubes=8):
    """configs[2] on the host cores: 140^3 cubes of the 900^3 / dice 120 / overlap 15 / border 10 geometry through the
    oracle's unet_deconv (oracle/nets.py) and overlap-add (oracle/dice.py) until max_cubes or budget_s, then scaled
    linearly to the 729 cubes of the volume (EXTRAPOLATED -- stated in `sample`).  The first two cubes are the two cubes of the
    parity slab (PARITY_SLAB_SHAPE) through the oracle's whole dice pipeline -- pad, reflect, cut, normalise, network, overlap-add,
    finalise (oracle/dice.py) -- and are what gpu_parity_infer() is compared with; the others are random cubes."""
    import torch
    from neuroclear_amd.util import seed as S
    from oracle import dice as odice
    from oracle import nets as onets
    ncores = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(ncores)
    sd = onets.to_torch(S.weights_from_seed(S.unet_deconv_spec(), PARITY_NET_SEED))
    E, R, ov, b, n_total = 140, 120, 15, 10, 729
    vol = S.random_volume(PARITY_VOL_SEED, PARITY_SLAB_SHAPE)
    padded = odice.pad_for_dicing(vol, R, ov)
    steps = odice.grid_steps(padded.shape, R, ov)
    refl = odice.reflect_pad(padded, b)
    assert steps == (2, 1, 1)
    acc = np.zeros((R + 105, R, R), np.float32)
    rng = np.random.default_rng(0)
    done, t_net, t_asm = 0, 0.0, 0.0
    slab_cubes = []
    with torch.no_grad():
        onets.unet_deconv(sd, torch.zeros(1, 1, 32, 32, 32))  # thread-pool warm-up
        t_start = time.time()
        while done < max_cubes and (done < 2 or (time.time() - t_start) * (done + 1) / done < budget_s):
            if done < 2:
                cube = odice.normalize(odice.cut_cube(refl, done, steps, R, ov, b))
            else:
                cube = (rng.integers(0, 65536, (E, E, E), dtype=np.uint16).astype(np.float64) / 65535.0).astype(np.float32)
            t0 = time.time()
            y = onets.unet_deconv(sd, torch.from_numpy(np.ascontiguousarray(cube))[None, None]).numpy()[0, 0]
            t_net += time.time() - t0
            t0 = time.time()
            z0 = 105 * (done % 2)
            acc[z0:z0 + R] += y[b:-b, b:-b, b:-b] / 8  # the assembler's per-cube work (util/assemble_dice.py:167-173)
            t_asm += time.time() - t0
            if done < 2:
                slab_cubes.append(y.copy())
            done += 1
    slab = odice.assemble(slab_cubes, padded.shape, vol.shape, R, ov, b, 'uint16')
    per_cube = (t_net + t_asm) / done
    total = per_cube * n_total
    return dict(value=900 ** 3 / total, unit='useful voxels/s', computed_voxels_per_s=n_total * E ** 3 / total,
                cores=ncores, kind='port', _parity_ref=dict(cubes=slab_cubes, slab=slab),
                sample='%d cubes of 140^3 through oracle/nets.py::unet_deconv + overlap-add on %d threads: %.2f s/cube '
                       '(network %.2f, assemble %.3f), EXTRAPOLATED linearly to 729 cubes = %.0f s per 900^3 volume'
                       % (done, ncores, per_cube, t_net / done, t_asm / done, total))


# ---- parity_vs_cpu_oracle: the HIP path on the SAME weights and inputs the CPU legs above run on, at the sizes the metric is quoted on
PARITY_BOUNDS = dict(first_step_losses_max_rel=2e-5, fake_max_abs=2e-5, rec_max_rel=2e-4, cube_max_abs=2e-5, slab_max_lsb=2)


def gpu_parity_train(dev, crop=108):
    """One optimize_parameters() of the Apollo model on the GPU from the weights, crop and np.random draws of _oracle_apollo()
    (reference axial_to_lateral_gan_apollo_model.py:285-307): unrounded first-step losses, fake and rec."""
    import contextlib
    import io
    import torch
    from neuroclear_amd.models import create_model
    from neuroclear_amd.util import seed as S
    from oracle import apollo as oapollo  # (names of the discriminators only)
    with contextlib.redirect_stdout(io.StringIO()):
        model = create_model(apollo_opt(dev.index))
    specs = [('G_A', S.unet_deconv_spec()), ('G_B', S.deep_linear_spec())] + [(n, S.patchgan_spec(2)) for n in oapollo.APOLLO_D]
    for i, (n, sp) in enumerate(specs):
        getattr(model, 'net' + n).load_state_dict(S.state_dict_from_seed(sp, 7 + i, dev, bias_scale=0.0))
    real = torch.from_numpy((S.random_volume(11, crop).astype(np.float64) / 65535.0).astype(np.float32))[None, None].to(dev)
    np.random.seed(0)
    model.set_input({'A': real, 'A_paths': 'synthetic'})
    model.optimize_parameters()
    out = dict(losses={k: float(v) for k, v in model.get_current_losses().items()},
               fake=model.fake.detach().float().cpu().numpy(), rec=model.rec.detach().float().cpu().numpy())
    del model
    return out


def gpu_parity_infer(dev):
    """The parity slab (two 140^3 cubes) through the product's diced inference (reference test_dice.py:107-118): the two raw network
    outputs and the assembled uint16 volume."""
    import torch
    from neuroclear_amd.models import networks
    from neuroclear_amd.test_dice import diced_inference
    from neuroclear_amd.util import seed as S
    net = networks.define_G(1, 1, 64, 'unet_deconv', 'instance', False, 'kaiming', 0.02, [dev.index])
    net.load_state_dict(S.state_dict_from_seed(S.unet_deconv_spec(), PARITY_NET_SEED, dev))
    vol = S.random_volume(PARITY_VOL_SEED, PARITY_SLAB_SHAPE)
    opt = Namespace(dice_size=[120] * 3, overlap=15, border_cut=10, gpu_ids=[dev.index], skip_real=True,
                    data_type='uint16', histogram_match=False, normalize_intensity=False)
    cubes = []

    def on_cube(fn):
        y = fn()
        cubes.append(y)
        return y
    slab = diced_inference(net, vol, opt, 0, 1, assemble='gather', broadcast=False, on_cube=on_cube)
    torch.cuda.synchronize()
    outs = [c for y in cubes for c in y.detach().float().reshape(-1, 140, 140, 140).cpu().numpy()]  # (a call may carry several cubes: NC_INFER_BATCH)
    return dict(cubes=outs, slab=np.asarray(slab))


def parity_train(gpu, ref):
    rel = {k: abs(gpu['losses'][k] - v) / max(abs(v), 1e-12) for k, v in ref['losses'].items()}
    f = float(np.abs(gpu['fake'].astype(np.float64) - ref['fake']).max())
    r = float(np.abs(gpu['rec'].astype(np.float64) - ref['rec']).max() / max(float(np.abs(ref['rec']).max()), 1e-30))
    B = PARITY_BOUNDS
    return dict(workload='apollo_train_step_108cube_bs1: first optimize_parameters() from the same seeded weights, crop and np.random draws',
                first_step_losses_max_rel_diff=max(rel.values()), n_losses=len(rel), fake_max_abs_diff=f, rec_max_rel_diff=r,
                bounds=dict(first_step_losses_max_rel_diff=B['first_step_losses_max_rel'], fake_max_abs_diff=B['fake_max_abs'], rec_max_rel_diff=B['rec_max_rel']),
                ok=bool(max(rel.values()) <= B['first_step_losses_max_rel'] and f <= B['fake_max_abs'] and r <= B['rec_max_rel']))


def parity_infer(gpu, ref):
    d = [float(np.abs(a.astype(np.float64) - b).max()) for a, b in zip(gpu['cubes'], ref['cubes'])]
    lsb = int(np.abs(gpu['slab'].astype(np.int64) - ref['slab'].astype(np.int64)).max())
    B = PARITY_BOUNDS
    return dict(workload='two 140^3 cubes of a %dx%dx%d uint16 volume (dice 120, overlap 15, border 10): network output after the sigmoid per cube, '
                         'assembled uint16 slab' % PARITY_SLAB_SHAPE, cube_max_abs_diff=max(d), cubes=len(d),
                slab_max_lsb_diff=lsb, slab_equal_share=float((gpu['slab'] == ref['slab']).mean()),
                bounds=dict(cube_max_abs_diff=B['cube_max_abs'], slab_max_lsb_diff=B['slab_max_lsb']),
                ok=bool(len(d) == 2 and max(d) <= B['cube_max_abs'] and lsb <= B['slab_max_lsb']))


def run_train(args, rank, world, dev):
    import torch
    import torch.distributed as dist
    from neuroclear_amd import ops
    from neuroclear_amd.models import create_model
    from neuroclear_amd.util import seed as S

    crop = args.crop
    torch.manual_seed(1234 + rank)
    np.random.seed(1234 + rank)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        o = apollo_opt(dev.index, args.model)
        o.precision = args.precision
        model = create_model(o)
    if world > 1:  # identical replicas: broadcast rank 0's parameters (one flat buffer per optimizer)
        for opt in model.optimizers:
            dist.broadcast(opt.flat, 0)
    # synthetic uint16 crops (one per batch element), normalised as data/base_dataset.py:134-143
    if args.data == 'structured':  # configs[4]: crops of the structured ("OT-LSM-style") volume, one region per rank and batch element
        big = S.structured_volume(9, 300)
        rs = np.random.default_rng(100 + rank)
        vols = []
        for b in range(args.batch):
            z, y, x = (int(rs.integers(0, 300 - crop + 1)) for _ in range(3))
            vols.append(big[z:z + crop, y:y + crop, x:x + crop])
    else:
        vols = [S.random_volume(100 + rank + 1000 * b, crop) for b in range(args.batch)]
    real = torch.stack([torch.from_numpy((v.astype(np.float64) / 65535.0).astype(np.float32))[None] for v in vols]).to(dev)
    data = {'A': real, 'A_paths': 'synthetic'}

    first = {}

    def step():
        model.set_input(data)
        model.optimize_parameters()
        if not first:  # losses of the very first step from the seeded initial weights: comparable between runs of different length
            first.update({k: float(v) for k, v in model.get_current_losses().items()})  # unrounded: `first_step_max_rel_diff` compares these

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    if not args.no_prof:
        ops.prof_start(0.0 if args.prof_all else 1e9)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stats, whole = ({}, []) if args.no_prof else ops.prof_stop()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # ---- per-kernel-class event statistics of the timed region (rank-local): HIP events recorded inside the library
    #      around every convolution launch of >= 1 GFLOP, on the stream it was launched on
    # (the PatchGAN launches -- gather-GEMM / k1 classes -- run on side streams underneath the generators' kernels: their
    #  event durations overlap the main stream's and are left out of the main-stream sum)
    conv_ms = sum(s[1] for t, s in stats.items() if '_gemm_' not in t and '_k1_' not in t)
    calls = {}
    for tag, fl, ms in whole:
        c = calls.setdefault(tag, [0, 0.0, 0.0])
        c[0] += 1
        c[1] += ms
        c[2] += fl
    top = max((t for t in stats if 'mfma' in t or '_lp_' in t or '_split_' in t), key=lambda t: stats[t][1], default=None)
    roof = None
    if top:
        n, ms, flop, abytes = stats[top]
        ach = flop / ms / 1e9
        from neuroclear_amd._lib import lib as _l
        two = '_split_' in top and int(_l().nc_get_split_terms()) == 2   # the two-term fp16 form: 3 MFMA products per fp32 product
        products = 3 if two else SPLIT_PRODUCTS
        peak = MFMA_16BIT_PEAK_TFLOPS if '_lp_' in top else MFMA_16BIT_PEAK_TFLOPS / products if '_split_' in top else MFMA_F32_PEAK_TFLOPS
        extra = {}
        if '_split_' in top:
            extra = dict(peak_is='%s dense MFMA peak %.0f / %d MFMA products per fp32 product' % ('fp16' if two else 'bf16', MFMA_16BIT_PEAK_TFLOPS, products),
                         split_terms=2 if two else 3, mfma_tflops=round(ach * products, 1),
                         frac_of_six_product_roof=round(ach / MFMA_SPLIT_PEAK_TFLOPS, 4), vs_fp32_mfma_peak=round(ach / MFMA_F32_PEAK_TFLOPS, 4))
        kname = KERNEL_OF.get(top, top)
        if two:
            w64 = bool(_l().nc_get_s3x_w64())
            kname = kname.replace('k_conv_s3x<3,*>', 'k_conv_s3w + k_conv_s3x<3,*,2> (fractional tiles, 54^3 / 27^3)' if w64 else 'k_conv_s3x<3,*,2>').replace('k_conv_s3x<5,*>', 'k_conv_s3x<5,*,2>').replace(
                'k_wgrad_s3x<3>', 'k_wgrad_s3x<3,2,f16>').replace('k_wgrad_s3x<5>', 'k_wgrad_s3x<5,2,f16>')
        traffic = pmc_traffic(top, crop, args.batch, three_term='_split_' in top and not two)
        roof = dict(bound='mfma', kernel=kname, kernel_class=top, achieved=round(ach, 2),
                    peak=round(peak, 2), unit='TFLOP/s', frac=round(ach / peak, 4), **extra,
                    traffic=traffic, traffic_source=TRAFFIC_SOURCE,
                    algorithmic_bytes=round(abytes / n),  # per launch, measured on THIS run's launches: operands once + result + weights
                    traffic_vs_algorithmic=round(traffic / (abytes / n), 3) if traffic else None,
                    launches=n, avg_launch_ms=round(ms / n, 4),
                    gflop_per_launch=round(flop / n / 1e9, 2),
                    share_of_step=round(ms / (dt * 1e3), 4),
                    classes={t: dict(n=s[0], ms_per_step=round(s[1] / args.steps, 3),
                                     tflops=round(s[2] / s[1] / 1e9, 2), algorithmic_gb_per_s=round(s[3] / s[1] / 1e6, 1))
                             for t, s in sorted(stats.items())},
                    conv_ms_per_step=round(conv_ms / args.steps, 2),
                    whole_network_calls={t: dict(n=c[0], ms_per_step=round(c[1] / args.steps, 3),
                                                 tflops=round(c[2] / c[1] / 1e9, 2), **_dl_note(t, c[2] / c[1] / 1e9, args))
                                         for t, c in sorted(calls.items())})
    losses = {k: round(v, 5) for k, v in model.get_current_losses().items()}
    return dt, crop ** 3 * args.batch * args.steps * world, roof, dict(
        workload='%s_train_step_%dcube_bs%d' % (args.model, crop, args.batch), crop=crop, batch_size=args.batch,
        parallelism='dp%d' % world, gan_mode='lsgan', norm='instance', data=args.data, first_step_losses=dict(first), losses=losses)


def _dl_note(tag, tflops, args):
    """deep_linear_gen's whole-network calls are counted with the REFERENCE's FLOPs (layer by layer: 647,120 MAC per voxel forward, twice that
    backward); the default evaluation executes 277,979 / 555,904 of them where the two-term kernels cover the shape."""
    from neuroclear_amd._lib import lib
    if not tag.startswith('deep_linear') or args.precision != 'fp32' or not lib().nc_get_dl_collapse() or lib().nc_get_split_terms() != 2:
        return {}
    if args.crop != 108 or args.batch != 1:
        return {}  # (the shares below are the headline shape's, where every collapsed / rank-structured plan applies; other shapes may take fewer of them)
    share = 277979.0 / 647120.0 if tag.endswith('fwd') else 555904.0 / 1294240.0
    return dict(tflops_is='on the reference\'s layer-by-layer FLOP count', tflops_executed=round(tflops * share, 2), executed_share_of_reference_flops=round(share, 4))


GA_FWD_FLOP_PER_VOXEL = 1.327618e6  # unet_deconv forward, dense count (BASELINE.md 2 / SURVEY.md 8d)


def pmc_traffic_cube(split=False, three_term=False):
    """HBM bytes of ONE 140^3 cube forward (all its kernels), from the committed PMC passes (tools/pmc_infer.sh, tools/pmc_split.sh)."""
    try:
        with open(os.path.join(ROOT, 'profiles', TRAFFIC_FILE)) as f:
            key = ('inference_cube_140_three_term' if three_term else 'inference_cube_140_split') if split else 'inference_cube_140'
            return round(json.load(f)[key]['hbm_bytes_per_cube'])
    except Exception:
        return None


def run_infer(args, rank, world, dev, steps=None, warmup=None):
    """configs[2]: 900^3 synthetic volume, dice 120 / overlap 15 / border_cut 10 -> 729 cubes of 140^3, a contiguous range of
    cubes per rank; rank 0's weights are broadcast (RCCL), every rank overlap-adds its own cubes into an accumulator of the two or
    three z-layers they touch, ONE point-to-point exchange gives every rank the z-slab it owns, which it finalises; the uint16
    slabs are gathered on rank 0 (neuroclear_amd/test_dice.py, assemble='slab')."""
    import torch
    import torch.distributed as dist
    from neuroclear_amd import test_dice as _td
    from neuroclear_amd.test_dice import diced_inference
    from neuroclear_amd.models import networks
    from neuroclear_amd.util import seed as S
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    L = args.volume
    torch.manual_seed(4321)  # same init on every rank; rank 0's copy is broadcast inside diced_inference anyway
    net = networks.define_G(1, 1, 64, 'unet_deconv', 'instance', False, 'kaiming', 0.02, [dev.index])
    vol = S.random_volume(5, L)
    opt = Namespace(dice_size=[120] * 3, overlap=15, border_cut=10, gpu_ids=[dev.index], skip_real=True,
                    data_type='uint16', histogram_match=False, normalize_intensity=False)
    in_flight = max(int(os.environ.get('NC_INFER_STREAMS', '2')), 1)
    nbatch = max(1, int(os.environ.get('NC_INFER_BATCH', '5')))
    for _ in range(max(warmup, 1)):  # every stream of the cubes in flight gets its workspace and its first launches (of a full batch) here
        diced_inference(net, vol, opt, rank, world, max_cubes=max(2, in_flight * nbatch) * world)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev = []

    def on_cube(fn):  # HIP events on the launch stream around every whole-network call
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y = fn()
        e1.record()
        ev.append((e0, e1))
        return y
    t0 = time.perf_counter()
    per_volume = []  # seconds of each volume (rank-local clock; diced_inference ends with the result on rank 0's host)
    for _ in range(steps):
        tv = time.perf_counter()
        out = diced_inference(net, vol, opt, rank, world, on_cube=None if args.no_prof else on_cube)
        per_volume.append(time.perf_counter() - tv)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt] + per_volume, dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, per_volume = float(t[0].item()), [float(v) for v in t[1:].tolist()]
    del out
    from neuroclear_amd.util import util as U
    padded = U.padded_shape((L, L, L), 120, 15)
    ncubes = int(np.prod(U.grid_steps(padded, 120, 15)))
    computed = ncubes * 140 ** 3  # voxels the network actually processes (overlap + border: 2.74 x the volume at 900^3)
    roof = None
    if ev:
        # Cubes overlap on NC_INFER_STREAMS HIP streams (test_dice.py), so the per-call event intervals overlap too: the rate is
        # taken over the wall time of the timed region (all cubes' FLOP / dt); avg_launch_ms = wall time per cube.
        in_flight = int(os.environ.get('NC_INFER_STREAMS', '2'))
        ms_events = sum(a.elapsed_time(b) for a, b in ev)
        ms = dt * 1e3 if in_flight > 1 else ms_events
        flop = GA_FWD_FLOP_PER_VOXEL * 140 ** 3
        batch = max(1, int(os.environ.get('NC_INFER_BATCH', '5')))  # cubes per network call (neuroclear_amd/test_dice.py)
        ncalls = len(ev)
        my_cubes = ncubes * steps if world == 1 else None  # cubes THIS rank ran in the timed region (sharded: its share, counted below)
        if my_cubes is None:
            from neuroclear_amd.test_dice import slab_plan
            a_, b_ = slab_plan(U.grid_steps(padded, 120, 15), 105, 120, padded[0], world)['cubes'][rank]
            my_cubes = (b_ - a_) * steps if _td.LAST['assemble'] == 'slab' else len(range(rank, ncubes, world)) * steps
        ach = flop * my_cubes / ms / 1e9
        from neuroclear_amd._lib import lib
        split = bool(lib().nc_get_conv_split())
        terms = int(lib().nc_unet_deconv_fwd_terms(140, 140, 140)) if split else 0  # 2: the two-term fp16 form (3 products), 3: three-term bf16 (6)
        products = 3 if terms == 2 else SPLIT_PRODUCTS
        peak = MFMA_16BIT_PEAK_TFLOPS / products if split else MFMA_F32_PEAK_TFLOPS
        extra = dict(peak_is='%s dense MFMA peak %.0f / %d MFMA products per fp32 product' % ('fp16' if terms == 2 else 'bf16', MFMA_16BIT_PEAK_TFLOPS, products),
                     split_terms=terms, frac_of_six_product_roof=round(ach / MFMA_SPLIT_PEAK_TFLOPS, 4),
                     vs_fp32_mfma_peak=round(ach / MFMA_F32_PEAK_TFLOPS, 4)) if split else {}
        cube_traffic = pmc_traffic_cube(split, three_term=split and terms != 2)
        roof = dict(bound='mfma', kernel='nc_unet_deconv_fwd: all kernels of one 140^3 cube forward (dominant: %s, '
                                        'profiles/r06_infer480_one_stream_kernel_stats.csv)' % (('k_conv_s3w<ST>' if terms == 2 and lib().nc_get_s3x_w64() else 'k_conv_s3x<3,*,%d>' % (2 if terms == 2 else 3)) if split else 'k_conv_mfma<3,*>'),
                    achieved=round(ach, 2), peak=round(peak, 2), unit='TFLOP/s',
                    frac=round(ach / peak, 4), **extra, traffic=cube_traffic, traffic_source=TRAFFIC_SOURCE,
                    algorithmic_bytes=round(3192 * 140 ** 3),  # 3,192 B per voxel for a perfectly fused fp32 G_A forward
                    traffic_vs_algorithmic=round(cube_traffic / (3192 * 140 ** 3), 3) if cube_traffic else None, launches=my_cubes,
                    network_calls=ncalls, cubes_per_call=batch, streams=in_flight,
                    cubes_in_flight=in_flight * batch, avg_launch_ms=round(ms / my_cubes, 3), gflop_per_launch=round(flop / 1e9, 1),
                    event_ms_per_cube=round(ms_events / my_cubes, 3),
                    whole_volume_tflops=round(GA_FWD_FLOP_PER_VOXEL * computed * steps / dt / 1e12, 2))
    return dt, L ** 3 * steps, roof, dict(workload='diced_inference_%dcube_dice120_ov15_b10' % L,
                                          parallelism=('contiguous cube ranges over %d ranks' % world) if world > 1 else 'cubes%1', cubes=ncubes,
                                          assemble={'slab': 'slab (owned z-slabs exchanged point to point, finalised per rank, uint16 slabs gathered)',
                                                    'reduce': 'reduce (per-rank accumulators, one reduce(sum) to rank 0; NC_ASSEMBLE or the point-to-point '
                                                              'self-check of neuroclear_amd/test_dice.py chose it)',
                                                    'gather': 'gather (lock-step rounds, tiles to rank 0)'}[_td.LAST['assemble']]
                                          if world > 1 else 'in-order',
                                          computed_voxels_per_s=round(computed * steps / dt),
                                          seconds_per_volume=dict(median=float(np.median(per_volume)), min=min(per_volume),
                                                                  max=max(per_volume), n=len(per_volume)))


class _LineGuard:
    """Keeps the measured train line printable while the inference leg runs.  rank 0 owns the line.  fail(msg): print the line with
    `inference: {error: msg}` and leave with code 3 (a failing rank > 0 only reports on stderr and leaves: the launcher then terminates
    the others).  SIGTERM (the launcher's reaction to another rank's death) and the timeout reach a helper THREAD through
    signal.set_wakeup_fd -- Python-level handlers only run between bytecodes of the main thread, which may be blocked inside RCCL."""

    def __init__(self, out, rank, world, timeout_s):
        import signal
        import threading
        self.out, self.rank, self.done = out, rank, threading.Event()
        self.r, self.w = os.pipe()
        os.set_blocking(self.w, False)
        self.old_fd = None
        if rank == 0 and world > 1:
            signal.signal(signal.SIGTERM, lambda *a: None)  # (a handler must exist for the wake-up fd to be written; the thread does the work)
            self.old_fd = signal.set_wakeup_fd(self.w, warn_on_full_buffer=False)
            self.thread = threading.Thread(target=self._watch, args=(timeout_s,), daemon=True)
            self.thread.start()

    def _watch(self, timeout_s):
        import select
        ready, _, _ = select.select([self.r], [], [], timeout_s)
        if self.done.is_set():
            return
        self._emit('terminated during the inference leg (another rank failed)' if ready else
                   'inference leg exceeded %.0f s (NC_BENCH_INFER_TIMEOUT): hang cut by the watchdog' % timeout_s)
        os._exit(3)

    def _emit(self, msg):
        self.out['inference'] = dict(error=msg)
        sys.stdout.write(json.dumps(self.out) + '\n')
        sys.stdout.flush()

    def fail(self, msg):
        self.done.set()
        print('bench.py: ' + msg, file=sys.stderr, flush=True)
        if self.rank == 0:
            self._emit(msg)
        os._exit(3)  # no destroy_process_group(): the other ranks may sit in a collective this rank will never join

    def disarm(self):
        import signal
        self.done.set()
        if self.old_fd is not None:
            signal.set_wakeup_fd(self.old_fd)
            signal.signal(signal.SIGTERM, signal.SIG_DFL)
            os.write(self.w, b'x')  # wakes the thread, which sees `done`


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--workload', default='headline', choices=['headline', 'train', 'infer'],
                    help='headline = the 108^3 train step (top-level value) + one 900^3 diced inference (nested)')
    ap.add_argument('--crop', type=int, default=108)
    ap.add_argument('--batch', type=int, default=1, help='crops per step and GPU (headline: 1; configs[3] shape: --crop 148 --batch 4)')
    ap.add_argument('--model', default='apollo', choices=['apollo', 'athena'], help='athena = configs[4]')
    ap.add_argument('--precision', default='fp32', choices=['fp32', 'bf16', 'fp16'],
                    help='arithmetic of the 3^3/5^3 convolutions; the headline is fp32 (the reference), bf16/fp16 = configs[3]')
    ap.add_argument('--volume', type=int, default=900)
    ap.add_argument('--data', default='random', choices=['random', 'structured'],
                    help='training crops: uniform uint16 noise (the north star\'s "synthetic random volumes") or crops of the structured '
                         'volume of SURVEY.md 8d (seed.structured_volume; configs[4])')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-prof', action='store_true', help='skip the per-launch HIP events (A/B runs)')
    ap.add_argument('--prof-all', action='store_true',
                    help='bracket every convolution launch with HIP events (default: launches of >= 1 GFLOP only -- the '
                         'events around the ~700 small PatchGAN launches of a step cost more than those kernels)')
    args = ap.parse_args()

    if args.gpus > 1 and 'RANK' not in os.environ:
        # `python bench.py --gpus N` with no launcher around it (the reference's own multi-GPU run needs none either:
        # nn.DataParallel inside `python train.py --gpu_ids 0,1,...`, models/networks.py:132-136): start the N ranks as a CHILD
        # torch.distributed.run and leave with its return code.  Nothing in this process has touched the GPU yet (a process that
        # has initialised HIP must never exec another program on this pool), and device_count() does not initialise it.
        import subprocess
        import torch
        env = dict(os.environ)
        ndev = torch.cuda.device_count()
        if ndev < args.gpus and 'NC_DIST_BACKEND' not in env:
            # fewer GPUs than ranks (a 1-GPU box): RCCL refuses two ranks on one device, so the ranks share GPUs and talk over gloo
            # -- a dry run of the sharded code path, labelled as such on the JSON line
            env['NC_DIST_BACKEND'] = 'gloo'
            print('note: --gpus %d on a box with %d GPU(s): ranks share devices, collectives over gloo (dry run of the sharded path)'
                  % (args.gpus, ndev), file=sys.stderr)
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--standalone', '--local-addr', '127.0.0.1', '--nnodes=1',
               '--nproc-per-node', str(args.gpus), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=env))

    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs the MI355X (no CPU fallback)')
    dev = torch.device('cuda', local % torch.cuda.device_count())  # (single-GPU dry runs of the N > 1 path)
    torch.cuda.set_device(dev)
    if world > 1:
        from neuroclear_amd.util.dist import init_process_group
        init_process_group(dev)  # 'nccl' IS RCCL on ROCm; NC_DIST_BACKEND=gloo = dry run with ranks sharing a GPU
    if args.gpus != world:
        if rank == 0:
            print('note: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE' % (args.gpus, world), file=sys.stderr)

    headline = args.workload == 'headline' and args.model == 'apollo' and args.crop == 108 and args.batch == 1 \
        and args.precision == 'fp32'
    train_like = args.workload in ('headline', 'train')
    run = run_train if train_like else run_infer
    dt, units, roof, cfg = run(args, rank, world, dev)
    out = dict(metric='voxels/sec', value=units / dt, unit='voxels/s', n_gpus=world, steps=args.steps,
               warmup=args.warmup, ms_per_step=dt / args.steps * 1e3, higher_is_better=True,
               scaling='weak' if train_like else 'strong',
               vs_baseline=None, dtype='f32' if args.precision == 'fp32' or not train_like else
               '%s (3^3/5^3 conv operands; fp32 accumulate, fp32 everywhere else)' % args.precision,
               data='synthetic', config=cfg)
    from neuroclear_amd import ops as _ops
    from neuroclear_amd._lib import lib as _lib
    split_on = bool(_lib().nc_get_conv_split()) and args.precision == 'fp32'
    if split_on:
        out['arithmetic'] = ARITHMETIC_H2 if (roof or {}).get('split_terms') == 2 else ARITHMETIC
    if roof:
        out['roofline'] = roof
    if split_on and int(_lib().nc_get_split_terms()) == 2:
        # the range guard of the two-term form (include/nc_hip.h): what it saw during warm-up + timed steps of the main run
        import ctypes
        gs = (ctypes.c_ulonglong * 4)()
        _lib().nc_h2_guard_stats(gs, 1)
        out['two_term_range_guard'] = dict(on=bool(_lib().nc_get_h2_guard()), tensors_measured=int(gs[0]), calls_fell_back_to_three_terms=int(gs[1]),
                                           flagged_without_switch=int(gs[2]), largest_low_chunk_share_ppm=int(gs[3]))
    if world > 1:
        out['dist'] = dict(backend=dist.get_backend(), devices=torch.cuda.device_count(),
                           ranks_share_devices=torch.cuda.device_count() < world)
    if headline and split_on and world == 1:
        # the same step with those layers on the fp32 MFMA kernels (v_mfma_f32_32x32x2_f32), for comparison
        import copy
        a2 = copy.copy(args)
        a2.steps, a2.warmup, a2.no_prof = 3, 1, True
        _ops.set_conv_split(False)
        try:
            dt2, units2, _, cfg2 = run_train(a2, rank, world, dev)
            f1, f2 = cfg['first_step_losses'], cfg2['first_step_losses']
            out['fp32_mfma_kernels'] = dict(ms_per_step=dt2 / a2.steps * 1e3, value=units2 / dt2, unit='voxels/s', steps=a2.steps,
                                            first_step_losses=f2,  # same seeds, same data: an in-line A/B of the two kernel sets
                                            first_step_max_rel_diff=max(abs(f1[k] - f2[k]) / max(abs(f2[k]), 1e-12) for k in f2),
                                            note='losses after %d steps differ from the main run\'s after %d: compare first_step_losses' % (
                                                a2.steps + a2.warmup, args.steps + args.warmup))
        finally:
            _ops.set_conv_split(True)
        if int(_lib().nc_get_split_terms()) == 2:
            # ... and on the THREE-term bf16 form of the split (six MFMA products per fp32 product: rounds 2-3), same seeds, same data
            _lib().nc_set_split_terms(3)
            try:
                dt3, units3, _, cfg3 = run_train(a2, rank, world, dev)
                f1, f3 = cfg['first_step_losses'], cfg3['first_step_losses']
                out['three_term_split'] = dict(ms_per_step=dt3 / a2.steps * 1e3, value=units3 / dt3, unit='voxels/s', steps=a2.steps,
                                               first_step_losses=f3,
                                               first_step_max_rel_diff=max(abs(f1[k] - f3[k]) / max(abs(f3[k]), 1e-12) for k in f3))
            finally:
                _lib().nc_set_split_terms(2)
    if train_like and args.precision == 'fp32' and bool(_lib().nc_get_dl_collapse()):
        # deep_linear_gen's layers 2 .. 5 run in collapsed form (include/nc_hip.h nc_set_dl_collapse: exact algebra of the bias-free linear
        # tail, same outputs and gradients); the same step with the layer-by-layer evaluation, for the record
        out['deep_linear_tail'] = dict(form='collapsed: the 3^3 + three 1x1 layers of deep_linear_gen as ONE 64->1 convolution, their parameter gradients in weight '
                                            'space, the 5^3 layer\'s backward from 27 shifted copies of the one-channel dy (32 x 64 problems); exact algebra, every '
                                            'output and gradient of the reference step is produced (DESIGN.md 4.6, tests/test_collapse_algebra.py); '
                                            'layer_by_layer = the same step with nc_set_dl_collapse(0)',
                                       macs_per_voxel_reference_fwd_bwd=3 * 647120)
        if args.crop == 108 and args.batch == 1 and args.model == 'apollo':  # (figures of the headline shape, where every plan of DESIGN.md 4.6 applies)
            out['deep_linear_tail'].update(macs_per_voxel_executed_fwd_bwd=833900,
                                           note='the reference-count 9.904 TFLOP per step include 2.79 TFLOP this evaluation does not execute')
        if headline and world == 1:
            import copy
            a4 = copy.copy(args)
            a4.steps, a4.warmup, a4.no_prof = 3, 1, True
            prev_mode = int(_lib().nc_get_dl_collapse())  # (2: the position-typed 7^3 form of round 6; 1: the collapsed tail + rank forms of round 5)
            _lib().nc_set_dl_collapse(0)
            try:
                dt4, units4, _, cfg4 = run_train(a4, rank, world, dev)
                f1, f4 = cfg['first_step_losses'], cfg4['first_step_losses']
                out['deep_linear_tail']['layer_by_layer'] = dict(ms_per_step=dt4 / a4.steps * 1e3, value=units4 / dt4, unit='voxels/s', steps=a4.steps,
                                                                 first_step_losses=f4,
                                                                 first_step_max_rel_diff=max(abs(f1[k] - f4[k]) / max(abs(f4[k]), 1e-12) for k in f4))
            finally:
                _lib().nc_set_dl_collapse(prev_mode)
    # CPU legs: rank 0 only, at every N (the other ranks have nothing to add to a host-core figure), and AFTER the last GPU leg and the
    # process group are done with -- no rank sits in an RCCL call while rank 0 spends a minute on its host cores
    cpu = rank == 0 and not args.no_cpu_baseline
    inf = None
    if headline:
        # second half of BASELINE.json's metric: one 900^3 diced inference (configs[2]), strong scaling over ranks
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        isteps = 3  # three whole volumes: the boxes of this pool differ by ~8 % and one volume has no spread to show
        # The train line above is measured; nothing the inference leg does may lose it (N > 1: the first real multi-GPU run of the slab
        # exchange happens on the driver's box).  An exception here prints the train line with `inference: {error}` and leaves non-zero; a
        # rank that dies elsewhere makes the launcher SIGTERM rank 0, and a hang is cut by the watchdog -- both print the same line
        # (_LineGuard: a helper thread, because the main thread may sit inside a collective when the signal arrives).
        guard = _LineGuard(out, rank, world, timeout_s=float(os.environ.get('NC_BENCH_INFER_TIMEOUT', '900')))
        try:
            if os.environ.get('NC_BENCH_FAIL_INFER') == '1':  # test hook: the injected failure of tests/test_bench_guard.py
                raise RuntimeError('injected inference failure (NC_BENCH_FAIL_INFER=1)')
            idt, iunits, iroof, icfg = run_infer(args, rank, world, dev, steps=isteps, warmup=1)
        except Exception as e:
            guard.fail('inference leg raised on rank %d: %r' % (rank, e))  # does not return
        guard.disarm()
        inf = dict(metric='voxels/sec (useful output voxels of the %d^3 volume, assemble included)' % args.volume,
                   value=iunits / idt, unit='voxels/s', seconds_per_volume=idt / isteps, steps=isteps,
                   seconds_per_volume_spread=icfg.pop('seconds_per_volume'), n_gpus=world, scaling='strong',
                   dtype='f32', config=icfg)
        if split_on:
            inf['arithmetic'] = ARITHMETIC_H2 if (iroof or {}).get('split_terms') == 2 else ARITHMETIC
        if iroof:
            inf['roofline'] = iroof
        out['inference'] = inf
    # parity_vs_cpu_oracle, GPU side: the product path on the weights / inputs / np.random draws the CPU legs below run on (rank 0 alone,
    # no collective: the other ranks wait at the barrier)
    gp_train = gp_infer = None
    if cpu and headline:
        try:
            gp_train = gpu_parity_train(dev)
        except Exception as e:
            gp_train = dict(error=repr(e))
    if cpu and (headline or args.workload == 'infer'):
        try:
            gp_infer = gpu_parity_infer(dev)
        except Exception as e:
            gp_infer = dict(error=repr(e))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    parity = {}
    if cpu and train_like:
        try:
            out['cpu_baseline'] = cpu_baseline_train()
            ref = out['cpu_baseline'].pop('_parity_ref')
            if gp_train is not None:
                parity['train'] = gp_train if 'error' in gp_train else parity_train(gp_train, ref)
        except Exception as e:  # the baseline is a report, never a reason to lose the GPU number
            out['cpu_baseline'] = dict(error=str(e))
    if cpu and (args.workload == 'infer' or inf is not None):
        tgt = out if args.workload == 'infer' else inf
        try:
            tgt['cpu_baseline'] = cpu_baseline_infer()
            ref = tgt['cpu_baseline'].pop('_parity_ref')
            if gp_infer is not None:
                parity['infer'] = gp_infer if 'error' in gp_infer else parity_infer(gp_infer, ref)
        except Exception as e:
            tgt['cpu_baseline'] = dict(error=str(e))
    if parity:
        # the HIP path against the CPU oracle at the sizes the metric is quoted on, SAME weights, inputs and random draws on both sides
        # (tests/test_gpu_fullsize.py asserts the same bounds)
        parity['ok'] = all(v.get('ok', False) for v in parity.values())
        out['parity_vs_cpu_oracle'] = parity
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
