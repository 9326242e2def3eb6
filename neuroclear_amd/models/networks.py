"""Host-side mirror of the reference's network zoo for the hot path (reference: models/networks.py).

Same factory names, argument meaning, state-dict keys and error behaviour as the reference for the networks that
BASELINE.json's north_star names -- `unet_deconv` (:478-538), `deep_linear_gen` (:893-917), `basic` / `n_layers`
PatchGAN (:1009-1067) -- plus, as the first widening row, `unet_vanilla` (:540-608) and the `pixel`
discriminator (:1147-1179), which are built from the same kernels; `get_norm_layer` (:20-44: instance -- the hot path -- and batch),
`GANLoss` (:252-319: lsgan -- the hot path -- vanilla, wgangp), `init_net` (:122-137), `get_scheduler` (:50-86).  Every forward/backward runs HIP kernels from libnc_hip.so through neuroclear_amd.ops;
there is no torch.nn.functional compute and no CPU fallback.  Networks outside the scope table (SURVEY.md 8a:
resnet, VGG, linear kernels, spectral-norm D, ...) raise NotImplementedError exactly like an unknown name does in
the reference (:196, :246).
"""
import functools
import os

import torch
import torch.nn as nn
from torch.nn import init
from torch.optim import lr_scheduler

from .. import ops
from .._lib import F, I, P, Z, check, lib


class Identity(nn.Module):
    def forward(self, x):
        return x


class InstanceNormAct(nn.Module):
    """InstanceNorm{2,3}d(affine=False, track_running_stats=False) fused with the activation that follows it in the
    reference's Sequential (ReLU at networks.py:422-423, LeakyReLU(0.2) at :1042-1046).  Holds no parameters, so the
    state-dict keys of the enclosing Sequential are unchanged."""

    def __init__(self, num_features, slope=0.0, eps=1e-5):
        super().__init__()
        self.num_features, self.slope, self.eps = num_features, slope, eps

    def forward(self, x, link=None, nxt=None):
        return ops.instance_norm_act(x, self.slope, self.eps, link, nxt)

    def extra_repr(self):
        return '%d, slope=%g (HIP fused IN+act)' % (self.num_features, self.slope)


class BatchNormAct(nn.Module):
    """nn.BatchNorm{2,3}d(num_features, affine=True, track_running_stats=True) (get_norm_layer('batch'), networks.py:30-31) fused with the
    activation behind it.  Parameters and buffers carry torch's names (weight, bias, running_mean, running_var, num_batches_tracked), so
    checkpoints of the reference load as they are."""

    def __init__(self, num_features, slope=0.0, eps=1e-5, momentum=0.1, dimension=3):
        super().__init__()
        self.num_features, self.slope, self.eps, self.momentum, self.dimension = num_features, slope, eps, momentum, dimension
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer('running_mean', torch.zeros(num_features))
        self.register_buffer('running_var', torch.ones(num_features))
        self.register_buffer('num_batches_tracked', torch.tensor(0, dtype=torch.long))

    def forward(self, x):
        if self.training:
            self.num_batches_tracked += 1
        return ops.batch_norm_act(x, self.weight, self.bias, self.running_mean, self.running_var, self.training, self.momentum, self.eps,
                                  self.slope)

    def extra_repr(self):
        return '%d, slope=%g (HIP fused BN%dd+act)' % (self.num_features, self.slope, self.dimension)


def _is_instance_norm(norm_layer):
    return norm_layer is not None and getattr(norm_layer, 'func', norm_layer) is InstanceNormAct


class FusedActivation(nn.Module):
    """Placeholder at the Sequential index where the reference has nn.ReLU()/nn.LeakyReLU(): the activation is
    applied by the preceding InstanceNormAct kernel, this module is the identity."""

    def forward(self, x):
        return x


class LeakyReLU(nn.Module):
    def __init__(self, slope):
        super().__init__()
        self.slope = slope

    def forward(self, x):
        return ops.leaky_relu(x, self.slope)


def get_norm_layer(norm_type='instance', dimension=3):
    """networks.py:20-44.  Returns a constructor norm_layer(num_features, slope) -> module."""
    if norm_type == 'instance':
        return functools.partial(InstanceNormAct)
    if norm_type in ('none', 'spectral'):
        return None
    if norm_type == 'batch':
        return functools.partial(BatchNormAct, dimension=dimension)
    raise NotImplementedError('normalization layer [%s] is not found' % norm_type)


def get_scheduler(optimizer, opt):
    """networks.py:50-86 (unknown policy RAISES here; the reference returns the exception object, a bug)."""
    if opt.lr_policy == 'linear':
        def lambda_rule(epoch):
            return 1.0 - max(0, epoch + opt.epoch_count - opt.n_epochs) / float(opt.n_epochs_decay + 1)
        return lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda_rule)
    if opt.lr_policy == 'constant':
        return lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda epoch: 1.0)
    if opt.lr_policy == 'step':
        return lr_scheduler.StepLR(optimizer, step_size=opt.lr_decay_iters, gamma=0.1)
    if opt.lr_policy == 'plateau':
        return lr_scheduler.ReduceLROnPlateau(optimizer, mode='min', factor=0.2, threshold=0.01, patience=5)
    if opt.lr_policy == 'cosine':
        return lr_scheduler.CosineAnnealingLR(optimizer, T_max=opt.n_epochs, eta_min=0)
    raise NotImplementedError('learning rate policy [%s] is not implemented' % opt.lr_policy)


def init_weights(net, init_type='normal', init_gain=0.02):
    """networks.py:88-119: every module whose class name contains 'Conv' (ConvTranspose included)."""
    def init_func(m):
        classname = m.__class__.__name__
        if hasattr(m, 'weight') and (classname.find('Conv') != -1 or classname.find('Linear') != -1):
            if init_type == 'normal':
                init.normal_(m.weight.data, 0.0, init_gain)
            elif init_type == 'xavier':
                init.xavier_normal_(m.weight.data, gain=init_gain)
            elif init_type == 'kaiming':
                init.kaiming_normal_(m.weight.data, a=0, mode='fan_in')
            elif init_type == 'orthogonal':
                init.orthogonal_(m.weight.data, gain=init_gain)
            else:
                raise NotImplementedError('initialization method [%s] is not implemented' % init_type)
            if hasattr(m, 'bias') and m.bias is not None:
                init.constant_(m.bias.data, 0.0)
        elif isinstance(m, BatchNormAct) and m.dimension == 3:  # the reference tests for 'BatchNorm3d' only (networks.py:115-117)
            init.normal_(m.weight.data, 1.0, init_gain)
            init.constant_(m.bias.data, 0.0)
    net.apply(init_func)


def init_net(net, init_type='normal', init_gain=0.02, gpu_ids=[]):
    """networks.py:122-137.  One process drives one GPU here (torch.distributed over RCCL does the scaling), so the
    reference's nn.DataParallel wrapper is not reproduced: only gpu_ids[0] is used."""
    if len(gpu_ids) > 0:
        assert torch.cuda.is_available()
        net.to(torch.device('cuda', gpu_ids[0]))
    init_weights(net, init_type, init_gain=init_gain)
    return net


class Conv(nn.Module):
    """nn.Conv2d / nn.Conv3d parameter layout (weight (K,C,k..), bias (K,)) on the HIP conv kernels."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True, dimension=3):
        super().__init__()
        self.stride, self.padding, self.dimension = stride, padding, dimension
        self.weight = nn.Parameter(torch.empty((out_channels, in_channels) + (kernel_size,) * dimension))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        init.kaiming_uniform_(self.weight, a=5 ** 0.5)
        if bias:
            init.zeros_(self.bias)

    def forward(self, x, link=None, prev=None):
        return ops.conv(x, self.weight, self.bias, self.stride, self.padding, link, prev)

    def extra_repr(self):
        return '%s, stride=%d, padding=%d' % (tuple(self.weight.shape), self.stride, self.padding)


class ConvTranspose(nn.Module):
    """nn.ConvTranspose3d(C, K, kernel 2, stride 2): weight (C,K,2,2,2), bias (K,)."""

    def __init__(self, in_channels, out_channels, kernel_size=2, stride=2, dimension=3):
        super().__init__()
        if kernel_size != 2 or stride != 2 or dimension != 3:
            raise NotImplementedError('only ConvTranspose3d(k=2, s=2) is on the hot path (networks.py:500,503)')
        self.weight = nn.Parameter(torch.empty((in_channels, out_channels, 2, 2, 2)))
        self.bias = nn.Parameter(torch.zeros(out_channels))
        init.kaiming_uniform_(self.weight, a=5 ** 0.5)

    def forward(self, x):
        return ops.conv_transpose_k2s2(x, self.weight, self.bias)


class SpectralConv(nn.Module):
    """nn.utils.spectral_norm(nn.Conv{2,3}d(...)) (networks.py:1079-1102): parameters `weight_orig` (+ `bias`), buffers
    `weight_u` [K] and `weight_v` [C * taps] -- the state-dict keys of torch's hook.  Every forward in training mode runs
    one power iteration on (u, v) and convolves with weight_orig / sigma (nc_spectral_norm_fwd + the conv kernels).
    `weight` is a plain attribute, as under torch's hook: the reference's init_weights writes into it, which the next
    forward overwrites -- weight_orig keeps the Conv default initialisation there, and so it does here."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True, dimension=3):
        super().__init__()
        self.stride, self.padding, self.dimension = stride, padding, dimension
        # registration order = torch's state-dict order under the hook: bias, weight_orig, then the buffers u, v
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        self.weight_orig = nn.Parameter(torch.empty((out_channels, in_channels) + (kernel_size,) * dimension))
        init.kaiming_uniform_(self.weight_orig, a=5 ** 0.5)
        m = self.weight_orig.numel() // out_channels
        self.register_buffer('weight_u', torch.nn.functional.normalize(torch.randn(out_channels), dim=0, eps=1e-12))
        self.register_buffer('weight_v', torch.nn.functional.normalize(torch.randn(m), dim=0, eps=1e-12))
        self.weight = self.weight_orig.detach().clone()

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self.weight = fn(self.weight)
        return r

    def forward(self, x):
        w = ops.spectral_norm_weight(self.weight_orig, self.weight_u, self.weight_v, self.training)
        self.weight = w.detach()
        return ops.conv(x, w, self.bias, self.stride, self.padding)


class NLayerDiscriminatorSN(nn.Module):
    """networks.py:1069-1111: the PatchGAN with every convolution spectrally normalised, no norm layers; only the first
    and the last convolution carry a bias."""

    def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer=None, use_sigmoid=False, dimension=3):
        super().__init__()
        if use_sigmoid:
            raise NotImplementedError('use_sigmoid=True is never set by the hot-path models (apollo:108-123)')
        kw, padw = 4, 1
        seq = [SpectralConv(input_nc, ndf, kw, 2, padw, dimension=dimension), LeakyReLU(0.2)]
        nf_mult = 1
        for n in range(1, n_layers):
            nf_prev, nf_mult = nf_mult, min(2 ** n, 8)
            seq += [SpectralConv(ndf * nf_prev, ndf * nf_mult, kw, 2, padw, bias=False, dimension=dimension), LeakyReLU(0.2)]
        nf_prev, nf_mult = nf_mult, min(2 ** n_layers, 8)
        seq += [SpectralConv(ndf * nf_prev, ndf * nf_mult, kw, 1, padw, bias=False, dimension=dimension), LeakyReLU(0.2)]
        seq += [SpectralConv(ndf * nf_mult, 1, kw, 1, padw, dimension=dimension)]
        self.model = nn.Sequential(*seq)
        self.one_plane_per_call = True  # a forward updates (u, v): planes must go through one by one, as in the reference

    def forward(self, input):
        return self.model(input)


def conv(dimension):
    if dimension not in (2, 3):
        raise Exception('Invalid image dimension.')
    return functools.partial(Conv, dimension=dimension)


_BIAS_LINK = os.environ.get('NC_BIAS_LINK', '1') != '0'  # A/B switch (timing experiments)
# whole-network C entry points for the generators' training passes (NC_FUSED_GEN=0: layer by layer through autograd)
_FUSED_GEN = os.environ.get('NC_FUSED_GEN', '1') != '0'


def _run_linked(seq, x):
    """nn.Sequential.forward, with every (Conv, InstanceNormAct) pair sharing an ops.BiasLink (the norm's backward hands
    the convolution its bias gradient and, on the 16-bit path, dy in the kernels' operand layout) and every
    (InstanceNormAct, next Conv) pair sharing one too (the norm's forward hands the next layer its input in that layout)."""
    mods = list(seq)
    i, prev = 0, None
    while i < len(mods):
        m = mods[i]
        if _BIAS_LINK and isinstance(m, Conv) and i + 1 < len(mods) and isinstance(mods[i + 1], InstanceNormAct) \
                and torch.is_grad_enabled():
            link = ops.BiasLink()
            raw = m(x, link, prev)
            j = i + 2
            while j < len(mods) and isinstance(mods[j], FusedActivation):
                j += 1
            nxt = None
            if j < len(mods) and isinstance(mods[j], Conv):
                dt = ops.lp_fwd_dtype(raw.shape, mods[j].weight.shape, mods[j].stride, mods[j].padding)
                if dt:
                    nxt = ops.BiasLink()
                    nxt.want_xh = dt
            x = mods[i + 1](raw, link, nxt)
            prev = nxt
            i += 2
        else:
            if isinstance(m, Conv):
                x = m(x, None, prev)
                prev = None
            else:
                x = m(x)
                if not isinstance(m, FusedActivation):
                    prev = None
            i += 1
    return x


def _conv_norm_relu(cin, cout, k, s, p, norm_layer, dimension):
    mods = [Conv(cin, cout, k, s, p, dimension=dimension)]
    if norm_layer is not None:
        mods += [norm_layer(cout, 0.0), FusedActivation()]
    else:
        mods += [Identity(), LeakyReLU(0.0)]
    return mods


class double_conv(nn.Module):
    """networks.py:413-432: Sequential indices 0 / 3 carry the convs."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=0, norm_layer=None, dimension=3):
        super().__init__()
        self.convolution = nn.Sequential(
            *(_conv_norm_relu(in_channels, out_channels, kernel_size, stride, padding, norm_layer, dimension) +
              _conv_norm_relu(out_channels, out_channels, kernel_size, stride, padding, norm_layer, dimension)))

    def forward(self, x):
        return _run_linked(self.convolution, x)


class last_conv(nn.Module):
    """networks.py:434-450."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=0, norm_layer=None, dimension=3):
        super().__init__()
        self.convolution = nn.Sequential(
            *_conv_norm_relu(in_channels, out_channels, kernel_size, stride, padding, norm_layer, dimension))

    def forward(self, x):
        return _run_linked(self.convolution, x)


class triple_conv(nn.Module):
    """networks.py:452-476: Sequential indices 0 / 3 / 6 carry the convs."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=0, norm_layer=None, dimension=3):
        super().__init__()
        self.convolution = nn.Sequential(
            *(_conv_norm_relu(in_channels, out_channels, kernel_size, stride, padding, norm_layer, dimension) +
              _conv_norm_relu(out_channels, out_channels, kernel_size, stride, padding, norm_layer, dimension) +
              _conv_norm_relu(out_channels, out_channels, kernel_size, stride, padding, norm_layer, dimension)))

    def forward(self, x):
        return _run_linked(self.convolution, x)


class Unet_deconv(nn.Module):
    """networks.py:478-538.  Training (grad enabled) runs layer by layer through neuroclear_amd.ops; inference
    (torch.no_grad, as BaseModel.test does at base_model.py:101-109) takes the whole-network C entry point
    nc_unet_deconv_fwd with the parameters packed in state-dict order."""

    def __init__(self, input_nc, output_nc, norm_layer=None, dimension=3):
        super().__init__()
        if dimension != 3:
            raise NotImplementedError('Unet_deconv: the hot path is the 3-D generator (apollo_model.py:66)')
        start_nc = input_nc * 64
        self.input_nc, self.output_nc = input_nc, output_nc
        self.double_conv1 = double_conv(input_nc, start_nc, 3, 1, 1, norm_layer, dimension)
        self.double_conv2 = double_conv(start_nc, start_nc * 2, 3, 1, 1, norm_layer, dimension)
        self.bottom_layer = triple_conv(start_nc * 2, start_nc * 4, 3, 1, 1, norm_layer, dimension)
        self.t_conv2 = ConvTranspose(start_nc * 4, start_nc * 2, 2, 2, dimension)
        self.ex_double_conv2 = double_conv(start_nc * 4, start_nc * 2, 3, 1, 1, norm_layer, dimension)
        self.t_conv1 = ConvTranspose(start_nc * 2, start_nc, 2, 2, dimension)
        self.ex_conv1_1 = last_conv(start_nc * 2, start_nc, 3, 1, 1, norm_layer, dimension)
        self.one_by_one = Conv(start_nc, output_nc, 1, 1, 0, dimension=dimension)
        self.one_by_one_2 = Conv(output_nc, output_nc, 1, 1, 0, dimension=dimension)
        self._fusable = _is_instance_norm(norm_layer) and input_nc == 1 and output_nc == 1

    def _packed_params(self):
        """The 28 tensors in state-dict order as one flat blob: a zero-copy view of FlatAdam's buffer when the
        parameters live there, otherwise a fresh concatenation per call (28 MB, ~10 us) -- never a cached copy, which
        nc_adam_step's raw-pointer update would leave stale (it bumps no tensor version)."""
        return ops._pack_params(list(self.parameters()))

    def _forward_fused(self, x):
        x = x.contiguous()
        N, _, S0, S1, S2 = x.shape
        nb = lib().nc_unet_deconv_fwd_ws_bytes(I(N), I(S0), I(S1), I(S2))
        if nb == 0:
            raise ValueError('Unet_deconv: every edge must be a positive multiple of 4, got %s '
                             '(MaxPool3d floors, torch.cat at networks.py:526,531 would fail)' % ((S0, S1, S2),))
        ws = ops.workspace(nb, x.device, 'unet')
        y = torch.empty_like(x)
        check(lib().nc_unet_deconv_fwd(P(self._packed_params().data_ptr()), P(x.data_ptr()), P(y.data_ptr()), I(N),
                                       I(S0), I(S1), I(S2), P(ws.data_ptr()), Z(ws.numel()),
                                       P(torch.cuda.current_stream().cuda_stream)), 'nc_unet_deconv_fwd')
        return y

    def forward(self, inputs):
        if any(s % 4 for s in inputs.shape[2:]):
            raise ValueError('Unet_deconv: every edge must be a multiple of 4, got %s' % (tuple(inputs.shape[2:]),))
        if self._fusable and not torch.is_grad_enabled() and inputs.is_cuda:
            return self._forward_fused(inputs)
        if self._fusable and inputs.is_cuda and ops.conv_precision == 'fp32' and _FUSED_GEN:
            # training: the whole forward (and, through autograd, the whole backward) as one C call
            return ops.unet_deconv_train(inputs, list(self.parameters()))
        if self._fusable and inputs.is_cuda and _FUSED_GEN and ops.gen_lp_supported('unet', inputs.shape):
            # --precision bf16: the 16-bit end-to-end whole-network call (bf16 activations in HBM, fp32 statistics)
            return ops.unet_deconv_lp(inputs, list(self.parameters()))
        conv1 = self.double_conv1(inputs)
        conv2 = self.double_conv2(ops.maxpool2(conv1))
        conv_bottom = self.bottom_layer(ops.maxpool2(conv2))
        cat2 = torch.cat([conv2, self.t_conv2(conv_bottom)], 1)
        ex_conv2 = self.ex_double_conv2(cat2)
        cat1 = torch.cat([conv1, self.t_conv1(ex_conv2)], 1)
        ex_conv1 = self.ex_conv1_1(cat1)
        return ops.sigmoid(self.one_by_one_2(self.one_by_one(ex_conv1)))


class Unet_vanilla(nn.Module):
    """networks.py:540-608: the four-level U-Net behind --netG unet_vanilla (double_conv on every level, 512-channel
    bottom, one 1x1 head + sigmoid).  Layer by layer through neuroclear_amd.ops in both modes (the whole-network C
    entry point covers unet_deconv only)."""

    def __init__(self, input_nc, output_nc, norm_layer=None, dimension=3):
        super().__init__()
        if dimension != 3:
            raise NotImplementedError('Unet_vanilla: 3-D only (ConvTranspose3d k2 s2 is the upsampling kernel)')
        c = input_nc * 64
        dc = functools.partial(double_conv, kernel_size=3, stride=1, padding=1, norm_layer=norm_layer,
                               dimension=dimension)
        self.double_conv1, self.double_conv2, self.double_conv3 = dc(input_nc, c), dc(c, c * 2), dc(c * 2, c * 4)
        self.bottom_layer = dc(c * 4, c * 8)
        self.t_conv3 = ConvTranspose(c * 8, c * 4, 2, 2, dimension)
        self.ex_double_conv3 = dc(c * 8, c * 4)
        self.t_conv2 = ConvTranspose(c * 4, c * 2, 2, 2, dimension)
        self.ex_double_conv2 = dc(c * 4, c * 2)
        self.t_conv1 = ConvTranspose(c * 2, c, 2, 2, dimension)
        self.ex_conv1_1 = dc(c * 2, c)
        self.one_by_one = Conv(c, output_nc, 1, 1, 0, dimension=dimension)

    def forward(self, inputs):
        if any(s % 8 for s in inputs.shape[2:]):
            raise ValueError('Unet_vanilla: every edge must be a multiple of 8, got %s (MaxPool3d floors, torch.cat '
                             'at networks.py:594,598,602 would fail)' % (tuple(inputs.shape[2:]),))
        conv1 = self.double_conv1(inputs)
        conv2 = self.double_conv2(ops.maxpool2(conv1))
        conv3 = self.double_conv3(ops.maxpool2(conv2))
        bottom = self.bottom_layer(ops.maxpool2(conv3))
        ex3 = self.ex_double_conv3(torch.cat([conv3, self.t_conv3(bottom)], 1))
        ex2 = self.ex_double_conv2(torch.cat([conv2, self.t_conv2(ex3)], 1))
        ex1 = self.ex_conv1_1(torch.cat([conv1, self.t_conv1(ex2)], 1))
        return ops.sigmoid(self.one_by_one(ex1))


class DeepLinearGenerator(nn.Module):
    """networks.py:893-917: bias-free linear chain, each layer zero-pads its own input."""

    def __init__(self, input_nc, output_nc):
        super().__init__()
        c = input_nc * 64
        self.first_layer = Conv(input_nc, c, 7, 1, 3, bias=False)
        self.feature_block = nn.Sequential(
            Conv(c, c, 5, 1, 2, bias=False), Conv(c, c, 3, 1, 1, bias=False),
            Conv(c, c // 2, 1, 1, 0, bias=False), Conv(c // 2, c // 4, 1, 1, 0, bias=False))
        self.final_layer = Conv(c // 4, output_nc, 1, 1, 0, bias=False)

    def forward(self, input):
        if input.is_cuda and input.dim() == 5 and input.shape[1] == 1 and self.final_layer.weight.shape[0] == 1 and \
                self.first_layer.weight.shape[0] == 64 and ops.conv_precision == 'fp32' and _FUSED_GEN:
            return ops.deep_linear(input, list(self.parameters()))
        if input.is_cuda and input.dim() == 5 and self.final_layer.weight.shape[0] == 1 and \
                self.first_layer.weight.shape[0] == 64 and _FUSED_GEN and torch.is_grad_enabled() and \
                ops.gen_lp_supported('linear', input.shape):
            return ops.deep_linear_lp(input, list(self.parameters()))
        return self.final_layer(self.feature_block(self.first_layer(input)))


class NLayerDiscriminator(nn.Module):
    """networks.py:1009-1067 (PatchGAN).  With instance norm every conv carries a bias (:1025-1028)."""

    def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer=None, use_sigmoid=False, dimension=3):
        super().__init__()
        if use_sigmoid:
            raise NotImplementedError('use_sigmoid=True is never set by the hot-path models (apollo:108-123)')
        use_bias = _is_instance_norm(norm_layer)  # reference networks.py:1025-1028: use_bias = (norm is InstanceNorm)
        kw, padw = 4, 1
        seq = [Conv(input_nc, ndf, kw, 2, padw, dimension=dimension), LeakyReLU(0.2)]
        nf_mult = 1
        for n in range(1, n_layers):
            nf_prev, nf_mult = nf_mult, min(2 ** n, 8)
            seq += [Conv(ndf * nf_prev, ndf * nf_mult, kw, 2, padw, bias=use_bias, dimension=dimension)]
            seq += [norm_layer(ndf * nf_mult, 0.2), FusedActivation()] if norm_layer else [Identity(), LeakyReLU(0.2)]
        nf_prev, nf_mult = nf_mult, min(2 ** n_layers, 8)
        seq += [Conv(ndf * nf_prev, ndf * nf_mult, kw, 1, padw, bias=use_bias, dimension=dimension)]
        seq += [norm_layer(ndf * nf_mult, 0.2), FusedActivation()] if norm_layer else [Identity(), LeakyReLU(0.2)]
        seq += [Conv(ndf * nf_mult, 1, kw, 1, padw, dimension=dimension)]
        self.model = nn.Sequential(*seq)
        # one C call per direction (nc_patchgan_fwd / _bwd) instead of ~25 Python-driven launches: the discriminator
        # chains are host-enqueue-bound otherwise.  Same kernels in the same order; NC_FUSED_PATCHGAN=0 or an
        # architecture the entry point does not cover (no norm, other input channels) takes the op-by-op path.
        self._cfg = (n_layers, ndf, dimension)
        self._fusable = _is_instance_norm(norm_layer) and input_nc == 1 and 1 <= n_layers <= 6

    def _fused_on(self, input):
        return self._fusable and input.is_cuda and os.environ.get('NC_FUSED_PATCHGAN', '1') != '0'

    def forward(self, input):
        if self._fused_on(input):
            return ops.patchgan(input, list(self.parameters()), *self._cfg)
        return _run_linked(self.model, input)

    # Athena evaluates every discriminator on the slices of `fake` twice with the same weights (generator loss, then --
    # detached, next to the slices of `real` -- discriminator loss, athena_model.py:240-260 / :190-238).  forward_fake_half
    # is the first of the two and keeps its activations in `share`; forward_join_real then runs only the real planes.
    def can_share(self, input):
        return self._fused_on(input) and not any(p.requires_grad for p in self.parameters())

    def forward_fake_half(self, input, share, src, axis):
        return ops.patchgan_fake_half(input, list(self.parameters()), *self._cfg, share, src=src, axis=axis)

    def share_matches(self, share, fake, axis):
        return share is not None and share.matches(list(self.parameters()), self._cfg, fake, axis)

    def forward_join_real(self, input, share):
        return ops.patchgan_join_real(input, list(self.parameters()), *self._cfg, share)


class PixelDiscriminator(nn.Module):
    """networks.py:1147-1179 (1x1 PatchGAN): conv1x1(1->ndf) + LeakyReLU, conv1x1(ndf->2ndf) + norm + LeakyReLU,
    conv1x1(2ndf->1); Sequential indices 0 / 2 / 5 carry the convs.  With instance norm all convs have a bias."""

    def __init__(self, input_nc, ndf=64, norm_layer=None, dimension=3):
        super().__init__()
        use_bias = _is_instance_norm(norm_layer)
        seq = [Conv(input_nc, ndf, 1, 1, 0, dimension=dimension), LeakyReLU(0.2),
               Conv(ndf, ndf * 2, 1, 1, 0, bias=use_bias, dimension=dimension)]
        seq += [norm_layer(ndf * 2, 0.2), FusedActivation()] if norm_layer else [Identity(), LeakyReLU(0.2)]
        seq += [Conv(ndf * 2, 1, 1, 1, 0, bias=use_bias, dimension=dimension)]
        self.net = nn.Sequential(*seq)

    def forward(self, input):
        return self.net(input)


def define_G(input_nc, output_nc, ngf, netG, norm='batch', use_dropout=False, init_type='normal', init_gain=0.02,
             gpu_ids=[], kernel_size=9, given_psf=None, noise_setting=None, dimension=3):
    """networks.py:140-197 (same signature)."""
    norm_layer = get_norm_layer(norm_type=norm, dimension=dimension)
    if netG == 'unet_deconv':
        net = Unet_deconv(1, output_nc, norm_layer=norm_layer, dimension=dimension)  # input_nc forced to 1 (:174)
    elif netG == 'deep_linear_gen':
        net = DeepLinearGenerator(input_nc, output_nc)
    elif netG == 'unet_vanilla':
        net = Unet_vanilla(1, output_nc, norm_layer=norm_layer, dimension=dimension)  # input_nc forced to 1 (:176)
    elif netG in ('unet_twoouts', 'resnet_9blocks', 'resnet_6blocks', 'VGG', 'linearkernel',
                  'linearkernel_double', 'linearkernel_LK31', 'linearkernel_NC', 'fixed_kernel'):
        raise NotImplementedError('Generator [%s] is outside the MI355X hot path (SURVEY.md 8a)' % netG)
    else:
        raise NotImplementedError('Generator model name [%s] is not recognized' % netG)
    return init_net(net, init_type, init_gain, gpu_ids)


def define_D(input_nc, ndf, netD, n_layers_D=3, norm='batch', init_type='normal', init_gain=0.02, use_sigmoid=False,
             gpu_ids=[], dimension=3):
    """networks.py:199-247 (same signature)."""
    norm_layer = get_norm_layer(norm_type=norm, dimension=dimension)
    if netD == 'basic':
        net = NLayerDiscriminator(input_nc, ndf, 3, norm_layer, use_sigmoid, dimension)
    elif netD == 'n_layers':
        net = NLayerDiscriminator(input_nc, ndf, n_layers_D, norm_layer, use_sigmoid, dimension)
    elif netD == 'pixel':
        net = PixelDiscriminator(input_nc, ndf, norm_layer, dimension)
    elif netD == 'basic_SN':
        net = NLayerDiscriminatorSN(input_nc, ndf, 3, norm_layer, use_sigmoid, dimension)
    elif netD == 'n_layers_SN':
        net = NLayerDiscriminatorSN(input_nc, ndf, n_layers_D, norm_layer, use_sigmoid, dimension)
    elif netD in ('kernelGAN',):
        raise NotImplementedError('Discriminator [%s] is outside the MI355X hot path (SURVEY.md 8a)' % netD)
    else:
        raise NotImplementedError('Discriminator model name [%s] is not recognized' % netD)
    return init_net(net, init_type, init_gain, gpu_ids)


class GANLoss(nn.Module):
    """networks.py:252-319.  'lsgan' (the README configuration) runs on the fused MSE-vs-constant kernel, 'vanilla' on the
    BCE-with-logits-vs-constant kernel, 'wgangp' (any name containing 'wgan', as in the reference) on the mean kernel."""

    def __init__(self, gan_mode, target_real_label=1.0, target_fake_label=0.0):
        super().__init__()
        self.register_buffer('real_label', torch.tensor(target_real_label))
        self.register_buffer('fake_label', torch.tensor(target_fake_label))
        self.gan_mode = gan_mode
        self._real, self._fake = float(target_real_label), float(target_fake_label)
        if gan_mode not in ('lsgan', 'vanilla') and 'wgan' not in gan_mode:
            raise NotImplementedError('gan mode %s not implemented' % gan_mode)

    def __call__(self, prediction, target_is_real):
        if self.gan_mode == 'lsgan':
            return ops.mse_const(prediction, self._real if target_is_real else self._fake)
        if self.gan_mode == 'vanilla':
            return ops.bce_logits_const(prediction, self._real if target_is_real else self._fake)
        m = ops.mean(prediction)  # 'wgan*': networks.py:314-318
        return -m if target_is_real else m
