"""BASELINE.json configs[3] AT ITS SIZE (-m gpu): "Batched training crop 148^3 bs=4, fp16 MFMA path with fp32
InstanceNorm accumulate".  The 16-bit kernels at the three levels of a 148^3 crop -- W = 148 (k_wgrad_h tiles of Tx = 37),
74 and 37 -- at batch 4, in bf16 and fp16:
  * adjoint identities <conv(x,w), r> = <x, dgrad(r,w)> = <w, wgrad(x,r)> on nc_conv_*_lp (size-independent property);
    each of the three kernels rounds ITS operands to 16 bits, so the three evaluations of the bilinear form agree to the
    operand rounding, not to fp32: 3 products of values rounded at 2^-9 (bf16) / 2^-12 (fp16), averaged over >= 1e7 terms
    of random sign -> stated tolerance 2e-3 (bf16) / 3e-4 (fp16) of |y| |r|;
  * rounded-operand comparison of one 64 -> 64 3^3 layer and one 5^3 layer at 4 x 148^3 against torch fp32 convolutions
    of the same 16-bit-rounded operands, chunked over samples and z-slabs (only the fp32 summation order differs: 5e-5 of
    the largest magnitude, 1e-4 for the weight gradient -- the tolerances of tests/test_gpu_lp.py);
  * one full Apollo optimisation step at 148^3 x 4 with --precision bf16 and fp16: the 11 losses against the fp32 step from
    the same seeds within the stated 2e-2 relative."""
import contextlib
import io
from argparse import Namespace

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = 'cuda'


@pytest.fixture(autouse=True)
def _restore_precision():
    from neuroclear_amd import ops
    yield
    ops.set_conv_precision('fp32')
    torch.cuda.empty_cache()


def _dot(a, b):
    return float((a.double() * b.double()).sum())


def _rnd(t, prec):
    return t.to(torch.bfloat16 if prec == 'bf16' else torch.float16).float()


@pytest.mark.parametrize('prec,tol', [('bf16', 2e-3), ('fp16', 3e-4)])
@pytest.mark.parametrize('C,K,k,E', [(64, 64, 3, 148), (128, 64, 3, 148), (64, 64, 5, 148), (64, 128, 3, 74),
                                     (256, 128, 3, 74), (128, 256, 3, 37), (256, 256, 3, 37)])
def test_lp_adjoint_identities_at_config3_size(C, K, k, E, prec, tol):
    from neuroclear_amd import ops
    N = 4
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.rand((N, C, E, E, E), device=DEV, generator=g) - 0.5
    w = (torch.rand((K, C, k, k, k), device=DEV, generator=g) - 0.5) * 0.1
    ops.set_conv_precision(prec)
    dims, k3 = (N, C, E, E, E), (k, k, k)
    assert ops._lp(0, dims, K, k3, 1, k // 2) and ops._lp(1, dims, K, k3, 1, k // 2) and ops._lp(2, dims, K, k3, 1, k // 2)
    y = ops.conv_fwd_raw(x, w, None, 1, k // 2)
    r = torch.rand(y.shape, device=DEV, generator=g) - 0.5
    a = _dot(y, r)
    dx = ops.conv_dgrad_raw(r, w, x.shape, 1, k // 2)
    b = _dot(x, dx)
    del dx
    dw, _ = ops.conv_wgrad_raw(x, r, w.shape, 1, k // 2, False)
    c = _dot(w, dw)
    scale = float(y.double().norm() * r.double().norm())
    print(C, K, k, E, prec, a, b, c, scale, abs(a - b) / scale, abs(a - c) / scale)
    assert abs(a - b) < tol * scale and abs(a - c) < tol * scale


@pytest.mark.parametrize('prec', ['bf16', 'fp16'])
@pytest.mark.parametrize('ks', [3, 5])
def test_lp_rounded_operands_at_4x148(ks, prec):
    """64 -> 64 at 4 x 148^3 (the full-resolution layers of configs[3]) vs torch on the same rounded operands; the torch
    reference runs per sample and per z-slab with the halo it needs."""
    from neuroclear_amd import ops
    N, C, K, E, pad = 4, 64, 64, 148, ks // 2
    g = torch.Generator(device=DEV).manual_seed(4)
    x = torch.randn((N, C, E, E, E), device=DEV, generator=g)
    w = torch.randn((K, C, ks, ks, ks), device=DEV, generator=g) / (C * ks ** 3) ** 0.5
    b = torch.randn(K, device=DEV, generator=g)
    dy = torch.randn((N, K, E, E, E), device=DEV, generator=g)
    ops.set_conv_precision(prec)
    y = ops.conv_fwd_raw(x, w, b, 1, pad)
    dx = ops.conv_dgrad_raw(dy, w, x.shape, 1, pad)
    dw, db = ops.conv_wgrad_raw(x, dy, w.shape, 1, pad, True)
    ops.set_conv_precision('fp32')
    bprec = 'bf16'  # backward operands are bf16 under both 16-bit precisions (ops._lp)
    wr_f, wr_b = _rnd(w, prec), _rnd(w, bprec)
    ymax, dxmax = float(y.abs().max()), float(dx.abs().max())
    dw_ref = torch.zeros_like(w, dtype=torch.float64)
    ZS = 37
    for n in range(N):
        for z0 in range(0, E, ZS):
            z1 = min(z0 + ZS, E)
            lo, hi = max(z0 - pad, 0), min(z1 + pad, E)
            # forward slab: x[lo:hi] with zero padding outside the volume only
            xs = _rnd(x[n:n + 1, :, lo:hi], prec)
            xs = F.pad(xs, (0, 0, 0, 0, pad - (z0 - lo), pad - (hi - z1)))
            ref = F.conv3d(xs, wr_f, b, padding=(0, pad, pad))
            assert float((y[n:n + 1, :, z0:z1] - ref).abs().max()) <= 5e-5 * ymax, ('fwd', n, z0)
            # data gradient slab
            ds = _rnd(dy[n:n + 1, :, lo:hi], bprec)
            ds = F.pad(ds, (0, 0, 0, 0, pad - (z0 - lo), pad - (hi - z1)))
            ref = F.conv3d(ds, wr_b.flip(2, 3, 4).transpose(0, 1).contiguous(), None, padding=(0, pad, pad))
            assert float((dx[n:n + 1, :, z0:z1] - ref).abs().max()) <= 5e-5 * dxmax, ('dgrad', n, z0)
            # weight gradient: contribution of the output slab z0:z1
            xs = _rnd(x[n:n + 1, :, lo:hi], bprec)
            xs = F.pad(xs, (0, 0, 0, 0, pad - (z0 - lo), pad - (hi - z1)))
            wz = torch.zeros_like(w, requires_grad=True)
            F.conv3d(xs, wz, None, padding=(0, pad, pad)).backward(_rnd(dy[n:n + 1, :, z0:z1], bprec))
            dw_ref += wz.grad.double()
            del xs, ds, ref, wz
    assert float((dw.double() - dw_ref).abs().max()) <= 1e-4 * float(dw_ref.abs().max())
    assert float((db - dy.sum((0, 2, 3, 4))).abs().max()) <= 1e-5 * float(dy.sum((0, 2, 3, 4)).abs().max()) + 1e-3


def _apollo_losses(prec, crop, batch, seed=21):
    from neuroclear_amd.models import create_model
    from neuroclear_amd.util import seed as S
    o = Namespace(gpu_ids=[0], isTrain=True, image_dimension=3, checkpoints_dir='/tmp/nc_ckpt', name='c3',
                  preprocess='none', gan_mode='lsgan', randomize_projection_depth=True, projection_depth=10,
                  min_projection_depth=2, lambda_plane=[1, 1, 1], lambda_A=5.0, input_nc=1, output_nc=1, ngf=64,
                  ndf=64, netG='unet_deconv', netG_B='deep_linear_gen', netD='basic', n_layers_D=3,
                  norm='instance', no_dropout=True, init_type='kaiming', init_gain=0.02, lr=1e-4, beta1=0.1,
                  direction='AtoB', model='axial_to_lateral_gan_apollo', precision=prec)
    torch.manual_seed(seed)
    np.random.seed(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        m = create_model(o)
    vols = [S.random_volume(100 + b, crop) for b in range(batch)]
    real = torch.stack([torch.from_numpy((v.astype(np.float64) / 65535.0).astype(np.float32))[None] for v in vols]).to(DEV)
    m.set_input({'A': real, 'A_paths': 'synthetic'})
    m.optimize_parameters()
    L = m.get_current_losses()
    upd_ok = all(bool(torch.isfinite(p).all()) for p in m.netG_A.parameters())
    del m
    torch.cuda.empty_cache()
    return L, upd_ok


def test_apollo_step_at_config3_size_16bit_vs_fp32():
    """One optimize_parameters() at 148^3 x 4 (BASELINE configs[3]) in fp32, bf16 and fp16 from the same seeds: the 11
    losses of the 16-bit steps within 2e-2 relative of the fp32 step, every updated parameter finite."""
    l32, ok32 = _apollo_losses('fp32', 148, 4)
    assert ok32 and len(l32) == 11
    for prec in ('bf16', 'fp16'):
        l16, ok16 = _apollo_losses(prec, 148, 4)
        assert ok16
        for k in l32:
            assert abs(l16[k] - l32[k]) <= 2e-2 * max(abs(l32[k]), 1e-3), (prec, k, l32[k], l16[k])
