"""Deterministic parameter filler shared by the product path, the oracle and the golden-vector generator.

The reference initialises networks with torch's global RNG (``models/networks.py:88-119`` ``init_weights``), which
cannot be replayed across library versions.  Parity work therefore needs parameters that BOTH sides can rebuild from
a seed without shipping tensors: ``weights_from_seed`` fills a state dict, in state-dict order, from
``numpy.random.default_rng`` streams keyed by ``(seed, tensor index)``.  Weights follow the scale of the reference's
``kaiming_normal_(a=0, mode='fan_in')`` (``networks.py:106-107``); biases are drawn non-zero on purpose (the reference
zeroes them, ``networks.py:112-113``) so that every bias path is exercised by the parity tests.

State-dict key names / shapes are the on-disk contract of ``models/base_model.py:146-162``.
"""
from collections import OrderedDict

import numpy as np


def unet_deconv_spec(dimension=3):
    """(key, shape) list of ``Unet_deconv`` (``models/networks.py:478-510``), input_nc forced to 1 (``:174``)."""
    k3 = (3,) * dimension
    k2 = (2,) * dimension
    k1 = (1,) * dimension
    spec = []

    def conv(name, cout, cin, k):
        spec.append((name + '.weight', (cout, cin) + k))
        spec.append((name + '.bias', (cout,)))

    conv('double_conv1.convolution.0', 64, 1, k3)
    conv('double_conv1.convolution.3', 64, 64, k3)
    conv('double_conv2.convolution.0', 128, 64, k3)
    conv('double_conv2.convolution.3', 128, 128, k3)
    conv('bottom_layer.convolution.0', 256, 128, k3)
    conv('bottom_layer.convolution.3', 256, 256, k3)
    conv('bottom_layer.convolution.6', 256, 256, k3)
    # ConvTranspose weight is (Cin, Cout, k, k, k)
    spec.append(('t_conv2.weight', (256, 128) + k2))
    spec.append(('t_conv2.bias', (128,)))
    conv('ex_double_conv2.convolution.0', 128, 256, k3)
    conv('ex_double_conv2.convolution.3', 128, 128, k3)
    spec.append(('t_conv1.weight', (128, 64) + k2))
    spec.append(('t_conv1.bias', (64,)))
    conv('ex_conv1_1.convolution.0', 64, 128, k3)
    conv('one_by_one', 1, 64, k1)
    conv('one_by_one_2', 1, 1, k1)
    return spec


def deep_linear_spec():
    """(key, shape) list of ``DeepLinearGenerator`` (``models/networks.py:893-911``); every conv is bias-free."""
    return [
        ('first_layer.weight', (64, 1, 7, 7, 7)),
        ('feature_block.0.weight', (64, 64, 5, 5, 5)),
        ('feature_block.1.weight', (64, 64, 3, 3, 3)),
        ('feature_block.2.weight', (32, 64, 1, 1, 1)),
        ('feature_block.3.weight', (16, 32, 1, 1, 1)),
        ('final_layer.weight', (1, 16, 1, 1, 1)),
    ]


def patchgan_spec(dimension=2, input_nc=1, ndf=64, n_layers=3):
    """(key, shape) list of ``NLayerDiscriminator`` (``models/networks.py:1009-1061``) with instance norm
    (=> every conv carries a bias, ``:1025-1028``).  Sequential indices: conv at 0, then 2+3*(n-1) for the middle
    blocks (conv, norm, lrelu), then the stride-1 block, then the 1-channel head."""
    k = (4,) * dimension
    spec = [('model.0.weight', (ndf, input_nc) + k), ('model.0.bias', (ndf,))]
    idx = 2
    nf_prev, nf = 1, 1
    for n in range(1, n_layers):
        nf_prev, nf = nf, min(2 ** n, 8)
        spec.append(('model.%d.weight' % idx, (ndf * nf, ndf * nf_prev) + k))
        spec.append(('model.%d.bias' % idx, (ndf * nf,)))
        idx += 3
    nf_prev, nf = nf, min(2 ** n_layers, 8)
    spec.append(('model.%d.weight' % idx, (ndf * nf, ndf * nf_prev) + k))
    spec.append(('model.%d.bias' % idx, (ndf * nf,)))
    idx += 3
    spec.append(('model.%d.weight' % idx, (1, ndf * nf) + k))
    spec.append(('model.%d.bias' % idx, (1,)))
    return spec


def patchgan_sn_spec(dimension=2, input_nc=1, ndf=64, n_layers=3):
    """(key, shape) list of ``NLayerDiscriminatorSN`` (``models/networks.py:1069-1111``) in state-dict order: per conv
    ``bias`` (first and last conv only), ``weight_orig``, ``weight_u`` [K], ``weight_v`` [C * taps]; Sequential indices
    0, 2, 4, ... (conv, LeakyReLU pairs)."""
    k = (4,) * dimension
    taps = 4 ** dimension
    chans = [(input_nc, ndf, True)]
    nf = 1
    for n in range(1, n_layers):
        nf_prev, nf = nf, min(2 ** n, 8)
        chans.append((ndf * nf_prev, ndf * nf, False))
    nf_prev, nf = nf, min(2 ** n_layers, 8)
    chans.append((ndf * nf_prev, ndf * nf, False))
    chans.append((ndf * nf, 1, True))
    spec = []
    for i, (c, kk, bias) in enumerate(chans):
        pre = 'model.%d.' % (2 * i)
        if bias:
            spec.append((pre + 'bias', (kk,)))
        spec += [(pre + 'weight_orig', (kk, c) + k), (pre + 'weight_u', (kk,)), (pre + 'weight_v', (c * taps,))]
    return spec


def unet_vanilla_spec(dimension=3):
    """(key, shape) list of ``Unet_vanilla`` (``models/networks.py:540-574``), input_nc forced to 1 (``:176``):
    three double_conv levels, a double_conv bottom at 512 channels, three transposed convs, one 1x1 head."""
    k3, k2, k1 = (3,) * dimension, (2,) * dimension, (1,) * dimension
    spec = []

    def dconv(name, cout, cin):
        for i, ci in ((0, cin), (3, cout)):
            spec.append(('%s.convolution.%d.weight' % (name, i), (cout, ci) + k3))
            spec.append(('%s.convolution.%d.bias' % (name, i), (cout,)))

    def tconv(name, cin, cout):
        spec.append((name + '.weight', (cin, cout) + k2))
        spec.append((name + '.bias', (cout,)))

    dconv('double_conv1', 64, 1)
    dconv('double_conv2', 128, 64)
    dconv('double_conv3', 256, 128)
    dconv('bottom_layer', 512, 256)
    tconv('t_conv3', 512, 256)
    dconv('ex_double_conv3', 256, 512)
    tconv('t_conv2', 256, 128)
    dconv('ex_double_conv2', 128, 256)
    tconv('t_conv1', 128, 64)
    dconv('ex_conv1_1', 64, 128)
    spec.append(('one_by_one.weight', (1, 64) + k1))
    spec.append(('one_by_one.bias', (1,)))
    return spec


def pixel_spec(dimension=2, input_nc=1, ndf=64):
    """(key, shape) list of ``PixelDiscriminator`` (``models/networks.py:1147-1179``) with instance norm (every 1x1
    conv carries a bias): net.0 (1->ndf), net.2 (ndf->2ndf) + norm + LeakyReLU, net.5 (2ndf->1)."""
    k = (1,) * dimension
    return [('net.0.weight', (ndf, input_nc) + k), ('net.0.bias', (ndf,)),
            ('net.2.weight', (ndf * 2, ndf) + k), ('net.2.bias', (ndf * 2,)),
            ('net.5.weight', (1, ndf * 2) + k), ('net.5.bias', (1,))]


def with_batch_norm(spec, norm_index=lambda conv_key: None):
    """The (key, shape) list of the same network built with --norm batch (networks.py:30-31): behind every convolution that is followed
    by a norm layer -- `norm_index(conv weight key)` gives that layer's key prefix, or None -- the five BatchNorm entries in state-dict
    order (weight, bias, running_mean, running_var, num_batches_tracked)."""
    out = []
    pending = None
    for key, shape in spec:
        if pending is not None and not key.startswith(pending[0]):
            out.extend(pending[1])
            pending = None
        out.append((key, shape))
        if key.endswith('.weight') and len(shape) > 1:
            nk = norm_index(key)
            if nk is not None:
                c = shape[0]
                pending = (key[:-len('weight')], [(nk + '.weight', (c,)), (nk + '.bias', (c,)), (nk + '.running_mean', (c,)),
                                                  (nk + '.running_var', (c,)), (nk + '.num_batches_tracked', ())])
    if pending is not None:
        out.extend(pending[1])
    return out


def _block_norm(conv_key):
    # '<block>.convolution.<i>.weight' -> '<block>.convolution.<i + 1>' (Sequential: conv, norm, relu)
    parts = conv_key.split('.')
    if len(parts) >= 4 and parts[-3] == 'convolution':
        return '.'.join(parts[:-2] + [str(int(parts[-2]) + 1)])
    return None


def unet_deconv_bn_spec(dimension=3):
    """Unet_deconv with BatchNorm layers (the convolutions keep their biases: double_conv does not look at the norm, networks.py:420)."""
    return with_batch_norm(unet_deconv_spec(dimension), _block_norm)


def patchgan_bn_spec(dimension=2, input_nc=1, ndf=64, n_layers=3):
    """NLayerDiscriminator with BatchNorm: the convolutions in front of a norm layer carry NO bias (use_bias is True for InstanceNorm
    only, networks.py:1025-1028); BatchNorm sits at Sequential index conv + 1."""
    base = patchgan_spec(dimension, input_nc, ndf, n_layers)
    last = max(int(k.split('.')[1]) for k, _ in base)
    normed = {k.split('.')[1] for k, _ in base if k.endswith('.weight') and k.split('.')[1] not in ('0', str(last))}
    spec = [(k, sh) for k, sh in base if not (k.endswith('.bias') and k.split('.')[1] in normed)]
    return with_batch_norm(spec, lambda key: 'model.%d' % (int(key.split('.')[1]) + 1) if key.split('.')[1] in normed else None)


def _fan_in(key, shape):
    # torch's _calculate_fan_in_and_fan_out: fan_in = size(1) * receptive field -- for ConvTranspose weights
    # (Cin, Cout, k..) that is Cout * k^d, which is what kaiming_normal_ in the reference ends up using.
    rf = 1
    for s in shape[2:]:
        rf *= s
    return shape[1] * rf


def weights_from_seed(spec, seed, bias_scale=0.1):
    """Return an OrderedDict {key: float32 ndarray} for a (key, shape) spec.  Pure numpy: torch-free on purpose."""
    out = OrderedDict()
    for idx, (key, shape) in enumerate(spec):
        rng = np.random.default_rng([int(seed), idx])
        if key.endswith('.num_batches_tracked'):
            out[key] = np.zeros(shape, np.int64)
        elif key.endswith('.running_var'):
            out[key] = rng.uniform(0.5, 1.5, size=shape).astype(np.float32)
        elif key.endswith('.weight') and len(shape) == 1:  # BatchNorm gamma: around one, away from zero
            out[key] = rng.uniform(0.8, 1.2, size=shape).astype(np.float32)
        elif key.endswith('.weight') or key.endswith('.weight_orig'):
            std = np.sqrt(2.0 / _fan_in(key, shape))
            out[key] = (rng.standard_normal(shape) * std).astype(np.float32)
        elif key.endswith('.weight_u') or key.endswith('.weight_v'):  # spectral-norm power-iteration vectors: unit length
            t = rng.standard_normal(shape)
            out[key] = (t / np.linalg.norm(t)).astype(np.float32)
        else:
            out[key] = rng.uniform(-bias_scale, bias_scale, size=shape).astype(np.float32)
    return out


def state_dict_from_seed(spec, seed, device='cpu', bias_scale=0.1):
    """Same as weights_from_seed, as torch tensors (importing torch lazily)."""
    import torch
    sd = OrderedDict()
    for k, v in weights_from_seed(spec, seed, bias_scale).items():
        sd[k] = torch.from_numpy(v).to(device)
    return sd


def random_volume(seed, size, dtype=np.uint16):
    """Seeded synthetic *random* volume of SURVEY.md §8(d): uniform integers over the dtype range."""
    if isinstance(size, int):
        size = (size, size, size)
    hi = 65536 if np.dtype(dtype) == np.uint16 else 256
    return np.random.default_rng(int(seed)).integers(0, hi, size=tuple(size), dtype=dtype)


def _blur_axis(a, sigma, axis):
    """Gaussian blur along one axis with explicit taps (radius 4 sigma, reflect-free zero padding, normalised taps) -- plain
    elementwise numpy on purpose: the volume is reproducible bit for bit wherever numpy runs."""
    r = max(1, int(np.ceil(4.0 * sigma)))
    k = np.exp(-0.5 * (np.arange(-r, r + 1, dtype=np.float64) / sigma) ** 2)
    k = (k / k.sum()).astype(np.float32)
    out = np.zeros_like(a)
    n = a.shape[axis]
    for i, w in enumerate(k):
        sh = i - r
        lo, hi = max(0, -sh), min(n, n - sh)
        if lo >= hi:
            continue
        dst = [slice(None)] * a.ndim
        src = [slice(None)] * a.ndim
        dst[axis] = slice(lo, hi)
        src[axis] = slice(lo + sh, hi + sh)
        out[tuple(dst)] += w * a[tuple(src)]
    return out


def structured_volume(seed, size, dtype=np.uint16, sigma_z=4.5, sigma_xy=1.0, with_truth=False):
    """Seeded *structured* synthetic volume of SURVEY.md 8(d) ("OT-LSM-style"): the stand-in for the reference's missing
    Jupyter generator (README.md:116, .MISSING_LARGE_BLOBS).  Sparse beads and random 3-D line segments ("tubes") are
    rendered into a float volume, blurred with an anisotropic Gaussian (sigma_z >> sigma_xy: the axial blur the network is
    trained to remove; the authors' run name `gaublur-std-4pt5` is the only hint at its size), given Poisson photon noise on
    a camera offset plus Gaussian read noise, and scaled to the integer range.  Most voxels are dark background: per-channel
    variances are small and sums mix magnitudes -- the input class uniform noise never exercises.
    Axis order (z, y, x).  with_truth: also return the isotropic (sigma_xy in z too), noise-free volume in the same scale."""
    if isinstance(size, int):
        size = (size, size, size)
    size = tuple(int(s) for s in size)
    rng = np.random.default_rng([int(seed), 7])
    vol = np.zeros(size, np.float32)
    nvox = float(np.prod(size))
    # beads: ~2 per 10^4 voxels, log-normal brightness
    nb = max(4, int(round(nvox * 2e-4)))
    bz, by, bx = (rng.integers(0, s, nb) for s in size)
    np.add.at(vol, (bz, by, bx), rng.lognormal(3.0, 0.6, nb).astype(np.float32))
    # tubes: segments of random direction, length up to half the volume, sampled every half voxel
    nt = max(2, int(round(nvox ** (1.0 / 3.0) / 6.0)))
    for _ in range(nt):
        p0 = rng.uniform(0, 1, 3) * np.array(size)
        d = rng.normal(0, 1, 3)
        d /= np.linalg.norm(d) + 1e-12
        length = rng.uniform(0.15, 0.5) * min(size)
        t = np.arange(0.0, length, 0.5)
        pts = np.rint(p0[None, :] + t[:, None] * d[None, :]).astype(np.int64)
        ok = np.all((pts >= 0) & (pts < np.array(size)[None, :]), axis=1)
        pts = pts[ok]
        np.add.at(vol, (pts[:, 0], pts[:, 1], pts[:, 2]), np.float32(rng.uniform(2.0, 6.0)))
    truth = _blur_axis(_blur_axis(_blur_axis(vol, sigma_xy, 0), sigma_xy, 1), sigma_xy, 2) if with_truth else None
    vol = _blur_axis(_blur_axis(_blur_axis(vol, sigma_z, 0), sigma_xy, 1), sigma_xy, 2)
    # photons: brightest structure ~400 photons above a 12-photon background; camera offset 100 counts, read noise 2 counts rms
    peak = float(vol.max()) or 1.0
    photons = vol * np.float32(400.0 / peak) + np.float32(12.0)
    noisy = rng.poisson(photons).astype(np.float32) + rng.normal(0.0, 2.0, size).astype(np.float32) + np.float32(100.0)
    hi = 65535.0 if np.dtype(dtype) == np.uint16 else 255.0
    gain = np.float32(0.85 * hi / (400.0 + 12.0 + 100.0))
    out = np.clip(np.rint(noisy * gain), 0, hi).astype(dtype)
    if with_truth:
        t = truth * np.float32(400.0 / peak) + np.float32(112.0)
        return out, np.clip(np.rint(t * gain), 0, hi).astype(dtype)
    return out
