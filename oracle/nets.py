"""ORACLE (test infrastructure, NOT product code) -- CPU restatement of the reference's network arithmetic.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module.  The
product path (``neuroclear_amd``) never does; it fails loudly when the HIP library is missing.

What is restated (plain ``torch.nn.functional`` calls on CPU fp32, parameters passed as a state dict with the
reference's key names):

* ``unet_deconv``   -- ``Unet_deconv.forward``           /root/reference/models/networks.py:512-538
                       built from ``double_conv`` :413-432, ``triple_conv`` :452-476, ``last_conv`` :434-450
* ``deep_linear``   -- ``DeepLinearGenerator.forward``   models/networks.py:913-917 (layers :899-911)
* ``patchgan``      -- ``NLayerDiscriminator.forward``   models/networks.py:1063-1066 (layers :1030-1061)
* ``unet_vanilla``  -- ``Unet_vanilla.forward``          models/networks.py:576-608 (widening row)
* ``pixel``         -- ``PixelDiscriminator.forward``    models/networks.py:1166-1179 (widening row)
* ``lsgan`` / L1    -- ``GANLoss('lsgan')`` :252-319 ; ``torch.nn.L1Loss`` apollo_model.py:128

Parity pin: ``tests/test_oracle_golden.py`` checks every function here against ``tests/golden/*.npz``, which were
produced by importing the reference itself in the build container (``oracle/gen_golden.py``).
"""
import numpy as np
import torch
import torch.nn.functional as F

EPS = 1e-5  # torch default of nn.InstanceNorm{2,3}d, used unchanged by networks.py:34


def _in_act(x, slope):
    # get_norm_layer('instance'): affine=False, track_running_stats=False (networks.py:33-34) => same in train/eval
    x = F.instance_norm(x, eps=EPS)
    return F.relu(x) if slope == 0.0 else F.leaky_relu(x, slope)


def _bn_act(x, sd, name, slope, training):
    # get_norm_layer('batch'): nn.BatchNorm{2,3}d(affine=True, track_running_stats=True) (networks.py:30-31); training mode normalises with
    # the batch statistics and updates sd[name + '.running_*'] IN PLACE (momentum 0.1, unbiased variance), evaluation uses them
    if training:
        sd[name + '.num_batches_tracked'] += 1
    x = F.batch_norm(x, sd[name + '.running_mean'], sd[name + '.running_var'], sd[name + '.weight'], sd[name + '.bias'], training, 0.1, EPS)
    return F.relu(x) if slope == 0.0 else F.leaky_relu(x, slope)


def _norm_act(x, sd, name, slope, norm, training):
    return _in_act(x, slope) if norm == 'instance' else _bn_act(x, sd, name, slope, training)


def _conv(x, sd, name, stride=1, padding=0):
    w = sd[name + '.weight']
    b = sd.get(name + '.bias')
    fn = F.conv3d if w.dim() == 5 else F.conv2d
    return fn(x, w, b, stride=stride, padding=padding)


def _convT(x, sd, name):
    w = sd[name + '.weight']
    fn = F.conv_transpose3d if w.dim() == 5 else F.conv_transpose2d
    return fn(x, w, sd[name + '.bias'], stride=2)


def _block(x, sd, prefix, idxs, norm='instance', training=True, acts=None):
    for i in idxs:
        x = _norm_act(_conv(x, sd, '%s.convolution.%d' % (prefix, i), padding=1), sd, '%s.convolution.%d' % (prefix, i + 1), 0.0, norm, training)
        if acts is not None:  # every norm + ReLU output, keyed by the module path of the norm layer (tests count ReLU / pool decisions on them)
            acts['%s.convolution.%d' % (prefix, i + 1)] = x
    return x


def unet_deconv(sd, x, taps=None, norm='instance', training=True):
    """networks.py:512-538.  ``taps`` (optional dict) receives the five stage outputs named as in the reference and, when the caller
    put an 'acts' dict into it, the output of every norm + ReLU layer there.  norm: 'instance'
    (the hot path) or 'batch' (--norm batch; `training` picks batch or running statistics)."""
    pool = F.max_pool3d if x.dim() == 5 else F.max_pool2d
    kw = dict(norm=norm, training=training, acts=taps.pop('acts', None) if taps is not None else None)
    conv1 = _block(x, sd, 'double_conv1', (0, 3), **kw)
    conv2 = _block(pool(conv1, 2), sd, 'double_conv2', (0, 3), **kw)
    bottom = _block(pool(conv2, 2), sd, 'bottom_layer', (0, 3, 6), **kw)
    cat2 = torch.cat([conv2, _convT(bottom, sd, 't_conv2')], 1)
    ex2 = _block(cat2, sd, 'ex_double_conv2', (0, 3), **kw)
    cat1 = torch.cat([conv1, _convT(ex2, sd, 't_conv1')], 1)
    ex1 = _block(cat1, sd, 'ex_conv1_1', (0,), **kw)
    y = _conv(_conv(ex1, sd, 'one_by_one'), sd, 'one_by_one_2')
    if taps is not None:
        taps.update(conv1=conv1, conv2=conv2, conv_bottom=bottom, ex_conv2=ex2, ex_conv1=ex1)
        if kw['acts'] is not None:
            taps['acts'] = kw['acts']
    return torch.sigmoid(y)


def deep_linear(sd, x):
    """networks.py:913-917: bias-free linear chain 7^3 -> 5^3 -> 3^3 -> 1x1 -> 1x1 -> 1x1, each layer zero-padding
    its own input (:899-902)."""
    x = _conv(x, sd, 'first_layer', padding=3)
    x = _conv(x, sd, 'feature_block.0', padding=2)
    x = _conv(x, sd, 'feature_block.1', padding=1)
    x = _conv(x, sd, 'feature_block.2')
    x = _conv(x, sd, 'feature_block.3')
    return _conv(x, sd, 'final_layer')


def patchgan(sd, x, n_layers=3, norm='instance', training=True):
    """networks.py:1030-1066: conv(s2)+LReLU, (conv(s2)+norm+LReLU)x(n_layers-1), conv(s1)+norm+LReLU, conv(s1) head.  kernel 4,
    padding 1 everywhere.  norm 'instance' (every conv biased) or 'batch' (the normed convs carry no bias: sd simply has none)."""
    x = F.leaky_relu(_conv(x, sd, 'model.0', stride=2, padding=1), 0.2)
    idx = 2
    for _ in range(1, n_layers):
        x = _norm_act(_conv(x, sd, 'model.%d' % idx, stride=2, padding=1), sd, 'model.%d' % (idx + 1), 0.2, norm, training)
        idx += 3
    x = _norm_act(_conv(x, sd, 'model.%d' % idx, stride=1, padding=1), sd, 'model.%d' % (idx + 1), 0.2, norm, training)
    idx += 3
    return _conv(x, sd, 'model.%d' % idx, stride=1, padding=1)


def spectral_weight(sd, name, training=True):
    """torch.nn.utils.spectral_norm's forward pre-hook (legacy hook, n_power_iterations = 1, eps = 1e-12, dim = 0) on the
    conv `name`: one power iteration updating sd[name + '.weight_u'/'_v'] IN PLACE (training), then weight_orig / sigma."""
    w = sd[name + '.weight_orig']
    u, v = sd[name + '.weight_u'], sd[name + '.weight_v']
    wm = w.reshape(w.shape[0], -1)
    if training:
        with torch.no_grad():
            v.copy_(F.normalize(torch.mv(wm.t(), u), dim=0, eps=1e-12))
            u.copy_(F.normalize(torch.mv(wm, v), dim=0, eps=1e-12))
    sigma = torch.dot(u.detach().clone(), torch.mv(wm, v.detach().clone()))
    return w / sigma


def patchgan_sn(sd, x, n_layers=3, training=True):
    """networks.py:1069-1111 (NLayerDiscriminatorSN): spectrally normalised convs + LeakyReLU(0.2), no norm layers; bias on
    the first and last conv only.  Each call performs one power iteration per conv, like a forward of the module."""
    conv = F.conv3d if x.dim() == 5 else F.conv2d
    n_conv = n_layers + 2
    for i in range(n_conv):
        name = 'model.%d' % (2 * i)
        stride = 2 if i < n_layers else 1
        x = conv(x, spectral_weight(sd, name, training), sd.get(name + '.bias'), stride=stride, padding=1)
        if i < n_conv - 1:
            x = F.leaky_relu(x, 0.2)
    return x


def unet_vanilla(sd, x):
    """networks.py:576-608: four-level U-Net, double_conv everywhere, single 1x1 head + sigmoid."""
    pool = F.max_pool3d if x.dim() == 5 else F.max_pool2d
    conv1 = _block(x, sd, 'double_conv1', (0, 3))
    conv2 = _block(pool(conv1, 2), sd, 'double_conv2', (0, 3))
    conv3 = _block(pool(conv2, 2), sd, 'double_conv3', (0, 3))
    bottom = _block(pool(conv3, 2), sd, 'bottom_layer', (0, 3))
    ex3 = _block(torch.cat([conv3, _convT(bottom, sd, 't_conv3')], 1), sd, 'ex_double_conv3', (0, 3))
    ex2 = _block(torch.cat([conv2, _convT(ex3, sd, 't_conv2')], 1), sd, 'ex_double_conv2', (0, 3))
    ex1 = _block(torch.cat([conv1, _convT(ex2, sd, 't_conv1')], 1), sd, 'ex_conv1_1', (0, 3))
    return torch.sigmoid(_conv(ex1, sd, 'one_by_one'))


def pixel(sd, x):
    """networks.py:1166-1179 with instance norm: 1x1 conv + LReLU, 1x1 conv + IN + LReLU, 1x1 conv head."""
    x = F.leaky_relu(_conv(x, sd, 'net.0'), 0.2)
    x = _in_act(_conv(x, sd, 'net.2'), 0.2)
    return _conv(x, sd, 'net.5')


def lsgan(pred, target_is_real):
    """GANLoss('lsgan').__call__ (networks.py:299-319): MSE against a constant 1.0 / 0.0 expanded to pred's shape."""
    t = torch.full_like(pred, 1.0 if target_is_real else 0.0)
    return F.mse_loss(pred, t)


def gan_loss(pred, target_is_real, mode='lsgan'):
    """GANLoss.__call__ (networks.py:299-319) for its three objectives: 'lsgan' MSE, 'vanilla' BCE-with-logits against the constant
    label, 'wgan*' -+mean(prediction)."""
    if mode == 'lsgan':
        return lsgan(pred, target_is_real)
    if mode == 'vanilla':
        return F.binary_cross_entropy_with_logits(pred, torch.full_like(pred, 1.0 if target_is_real else 0.0))
    if 'wgan' in mode:
        return -pred.mean() if target_is_real else pred.mean()
    raise NotImplementedError('gan mode %s not implemented' % mode)


def l1(a, b):
    return F.l1_loss(a, b)


def to_torch(sd_np, requires_grad=False):
    out = {}
    for k, v in sd_np.items():
        t = torch.from_numpy(np.asarray(v)).clone()
        # (BatchNorm buffers -- running statistics, the step counter -- are state, not parameters)
        t.requires_grad_(requires_grad and t.is_floating_point() and '.running_' not in k)
        out[k] = t
    return out
