"""ORACLE (test infrastructure, NOT product code) -- CPU restatement of one Apollo / Athena optimisation step.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module.

Restates /root/reference/models/axial_to_lateral_gan_apollo_model.py:
  set_input :142-160, forward :162-167, backward_G :255-283, backward_D_* :169-253, optimize_parameters :285-307,
  Volume.get_slice / get_projection :322-351
/root/reference/models/axial_to_lateral_gan_dryops_model.py (Apollo without G_B / D_B): optimize_parameters :234-258
and /root/reference/models/axial_to_lateral_gan_athena_model.py:
  __init__ plane wiring :93-148, backward_G :240-260, backward_D_* :190-238, iter_f :286-296, Volume :298-331

All random indices come from ``np.random`` in the reference's own draw order, so ``np.random.seed(s)`` before a step
replays the reference step exactly.  Autograd and Adam are torch's (the reference's dependency).

Parity pin: tests/golden/apollo_step_*.npz / athena_step_*.npz (oracle/gen_golden.py).
"""
import itertools
from collections import OrderedDict

import numpy as np
import torch

from . import nets

APOLLO_D = ['D_A_axial', 'D_A_lateral', 'D_B_axial', 'D_B_lateral']  # optimizer_D chain order, apollo:133-135


class Nets:
    """Parameter holder: {net name: {key: leaf tensor}}."""

    def __init__(self, sds):
        self.sd = OrderedDict((n, nets.to_torch(sd, requires_grad=True)) for n, sd in sds.items())

    def params(self, names):
        return list(itertools.chain(*[self.sd[n].values() for n in names]))

    def set_requires_grad(self, names, flag):
        for p in self.params(names):
            p.requires_grad_(flag)


def _slice(vol, axis):
    """Volume.get_slice apollo:328-337 -- num_slice is vol.shape[-1] for every axis (:325)."""
    i = np.random.randint(vol.shape[-1])
    return [vol[:, :, i, :, :], vol[:, :, :, i, :], vol[:, :, :, :, i]][axis]


def _mip(vol, depth, axis):
    """Volume.get_projection apollo:339-351."""
    s = np.random.randint(0, vol.shape[-1] - depth)
    roi = [vol[:, :, s:s + depth], vol[:, :, :, s:s + depth], vol[:, :, :, :, s:s + depth]][axis]
    return torch.max(roi, axis + 2)[0]


class ApolloOracle:
    def __init__(self, sds, lr=1e-4, beta1=0.1, lambda_A=5.0, lambda_plane=(1, 1, 1), projection_depth=10,
                 randomize_projection_depth=True, min_projection_depth=2, gan_mode='lsgan'):
        self.n = Nets(sds)
        self.gan = lambda pred, flag: nets.gan_loss(pred, flag, gan_mode)  # networks.GANLoss(opt.gan_mode), apollo:127
        s = float(sum(lambda_plane))
        self.w_target, self.w_slice, self.w_proj = [f / s for f in lambda_plane]  # apollo:81-82
        self.lambda_A = lambda_A
        self.randomize = randomize_projection_depth
        self.max_depth, self.min_depth = projection_depth, min_projection_depth
        self.opt_G = torch.optim.Adam(self.n.params(['G_A', 'G_B']), lr=lr, betas=(beta1, 0.999))
        self.opt_D = torch.optim.Adam(self.n.params(APOLLO_D), lr=lr, betas=(beta1, 0.999))
        self.losses = OrderedDict()

    def D(self, name, x):
        return nets.patchgan(self.n.sd[name], x)

    def step(self, real):
        if self.randomize:  # apollo:157-160
            self.depth = np.random.randint(max(2, self.min_depth), self.max_depth + 1)
        else:
            self.depth = self.max_depth
        L = self.losses
        fake = nets.unet_deconv(self.n.sd['G_A'], real)
        rec = nets.deep_linear(self.n.sd['G_B'], fake)
        self.fake, self.rec = fake, rec

        # ---- generators (apollo:255-283)
        self.n.set_requires_grad(APOLLO_D, False)
        self.opt_G.zero_grad()
        L['G_A_lateral'] = self.gan(self.D('D_A_lateral', _mip(fake, self.depth, 0)), True) * self.w_target
        L['G_A_axial'] = self.gan(self.D('D_A_axial', _mip(fake, self.depth, 1)), True) * self.w_slice + \
            self.gan(self.D('D_A_axial', _mip(fake, self.depth, 2)), True) * self.w_slice
        L['G_A'] = L['G_A_lateral'] + L['G_A_axial'] * 0.5
        L['G_B_lateral'] = self.gan(self.D('D_B_lateral', _slice(rec, 0)), True) * self.w_target
        L['G_B_axial'] = self.gan(self.D('D_B_axial', _slice(rec, 1)), True) * self.w_slice + \
            self.gan(self.D('D_B_axial', _slice(rec, 2)), True) * self.w_slice
        L['G_B'] = L['G_B_lateral'] + L['G_B_axial'] * 0.5
        L['cycle'] = nets.l1(rec, real) * self.lambda_A
        (L['G_A'] + L['G_B'] + L['cycle']).backward()
        self.grads_G = [p.grad.clone() for p in self.n.params(['G_A', 'G_B'])]
        self.opt_G.step()

        # ---- discriminators (apollo:297-307)
        self.n.set_requires_grad(APOLLO_D, True)
        self.opt_D.zero_grad()
        fd, rd = fake.detach(), rec.detach()

        def d_proj(name, ax_real, ax_fake):  # backward_D_projection apollo:195-223
            pr = self.D(name, _slice(real, ax_real))
            pf = self.D(name, _mip(fd, self.depth, ax_fake))
            loss = (self.gan(pr, True) + self.gan(pf, False)) * 0.5
            loss.backward()
            return loss

        def d_slice(name, ax_real, ax_fake):  # backward_D_slice apollo:169-193
            pr = self.D(name, _slice(real, ax_real))
            pf = self.D(name, _slice(rd, ax_fake))
            loss = (self.gan(pr, True) + self.gan(pf, False)) * 0.5
            loss.backward()
            return loss

        L['D_A_lateral'] = d_proj('D_A_lateral', 0, 0)
        L['D_A_axial'] = (d_proj('D_A_axial', 0, 1) + d_proj('D_A_axial', 0, 2)) * 0.5
        L['D_B_lateral'] = d_slice('D_B_lateral', 0, 0)
        L['D_B_axial'] = (d_slice('D_B_axial', 1, 1) + d_slice('D_B_axial', 2, 2)) * 0.5
        self.grads_D = [p.grad.clone() for p in self.n.params(APOLLO_D)]
        self.opt_D.step()
        return OrderedDict((k, float(v.detach())) for k, v in L.items())


DRYOPS_D = ['D_A_axial', 'D_A_lateral']  # optimizer_D chain order, dryops:103-104


class DryopsOracle:
    """One Dryops step (axial_to_lateral_gan_dryops_model.py:234-276): Apollo without the backward path -- G_A and
    the two D_A discriminators only.  ``netG`` / ``netD`` pick the restated network ('unet_deconv' | 'unet_vanilla',
    'basic' | 'pixel')."""

    def __init__(self, sds, lr=1e-4, beta1=0.1, lambda_plane=(1, 1, 1), projection_depth=10,
                 randomize_projection_depth=True, min_projection_depth=2, netG='unet_deconv', netD='basic'):
        self.n = Nets(sds)
        s = float(sum(lambda_plane))
        self.w_target, self.w_slice, self.w_proj = [f / s for f in lambda_plane]  # dryops:83-84
        self.randomize = randomize_projection_depth
        self.max_depth, self.min_depth = projection_depth, min_projection_depth
        self.G = {'unet_deconv': nets.unet_deconv, 'unet_vanilla': nets.unet_vanilla}[netG]
        self.Dfn = {'basic': nets.patchgan, 'pixel': nets.pixel}[netD]
        self.opt_G = torch.optim.Adam(self.n.params(['G_A']), lr=lr, betas=(beta1, 0.999))
        self.opt_D = torch.optim.Adam(self.n.params(DRYOPS_D), lr=lr, betas=(beta1, 0.999))
        self.losses = OrderedDict()

    def D(self, name, x):
        return self.Dfn(self.n.sd[name], x)

    def step(self, real):
        if self.randomize:  # dryops:127-130
            self.depth = np.random.randint(max(2, self.min_depth), self.max_depth + 1)
        else:
            self.depth = self.max_depth
        L = self.losses
        fake = self.G(self.n.sd['G_A'], real)
        self.fake = fake
        self.n.set_requires_grad(DRYOPS_D, False)
        self.opt_G.zero_grad()
        L['G_A_lateral'] = nets.lsgan(self.D('D_A_lateral', _mip(fake, self.depth, 0)), True) * self.w_target
        L['G_A_axial'] = nets.lsgan(self.D('D_A_axial', _mip(fake, self.depth, 1)), True) * self.w_slice + \
            nets.lsgan(self.D('D_A_axial', _mip(fake, self.depth, 2)), True) * self.w_slice
        L['G_A'] = L['G_A_lateral'] + L['G_A_axial'] * 0.5
        L['G_A'].backward()
        self.opt_G.step()
        self.n.set_requires_grad(DRYOPS_D, True)
        self.opt_D.zero_grad()
        fd = fake.detach()

        def d_proj(name, ax_real, ax_fake):  # backward_D_projection dryops:182-199
            loss = (nets.lsgan(self.D(name, _slice(real, ax_real)), True) +
                    nets.lsgan(self.D(name, _mip(fd, self.depth, ax_fake)), False)) * 0.5
            loss.backward()
            return loss

        L['D_A_lateral'] = d_proj('D_A_lateral', 0, 0)
        L['D_A_axial'] = (d_proj('D_A_axial', 0, 1) + d_proj('D_A_axial', 0, 2)) * 0.5
        self.opt_D.step()
        return OrderedDict((k, float(v.detach())) for k, v in L.items())


ATHENA_D = ['D_A_yz', 'D_A_xy', 'D_A_xz', 'D_B_yz', 'D_B_xy', 'D_B_xz']  # optimizer_D chain order, athena:163-164


class AthenaOracle:
    """One Athena step (axial_to_lateral_gan_athena_model.py:183-296).  iter_f is restated as the reference writes
    it: a loop over the S slices along the axis, discriminator per slice, outputs stacked along dim 2."""

    def __init__(self, sds, lr=1e-4, beta1=0.1, lambda_A=5.0, lambda_plane=(1, 1, 1), conversion_plane=('yz', 'xy')):
        self.n = Nets(sds)
        axis = {'xy': 0, 'xz': 1, 'yz': 2}
        remain = [a for a in axis if a not in conversion_plane][0]
        self.src, self.tgt, self.rem = axis[conversion_plane[0]], axis[conversion_plane[1]], axis[remain]
        s = float(sum(lambda_plane))
        self.w_target, self.w_source, self.w_ref = [f / s for f in lambda_plane]  # athena:110
        self.lambda_A = lambda_A
        self.opt_G = torch.optim.Adam(self.n.params(['G_A', 'G_B']), lr=lr, betas=(beta1, 0.999))
        self.opt_D = torch.optim.Adam(self.n.params(ATHENA_D), lr=lr, betas=(beta1, 0.999))
        self.losses = OrderedDict()

    def iter_f(self, vol, name, axis):
        S = vol.shape[-3]
        outs = []
        for i in range(S):
            sl = [vol[:, :, i, :, :], vol[:, :, :, i, :], vol[:, :, :, :, i]][axis]
            outs.append(nets.patchgan(self.n.sd[name], sl))
        return torch.stack(outs, dim=2)

    def step(self, real):
        L = self.losses
        fake = nets.unet_deconv(self.n.sd['G_A'], real)
        rec = nets.deep_linear(self.n.sd['G_B'], fake)
        self.fake, self.rec = fake, rec
        t, s, r = self.tgt, self.src, self.rem
        self.n.set_requires_grad(ATHENA_D, False)
        self.opt_G.zero_grad()
        L['G_A_xy'] = nets.lsgan(self.iter_f(fake, 'D_A_xy', t), True) * self.w_target
        L['G_A_yz'] = nets.lsgan(self.iter_f(fake, 'D_A_yz', s), True) * self.w_source
        L['G_A_xz'] = nets.lsgan(self.iter_f(fake, 'D_A_xz', r), True) * self.w_ref
        L['G_A'] = L['G_A_xy'] + L['G_A_yz'] + L['G_A_xz']
        L['G_B_xy'] = nets.lsgan(self.iter_f(rec, 'D_B_xy', t), True) * (1 / 3)
        L['G_B_yz'] = nets.lsgan(self.iter_f(rec, 'D_B_yz', s), True) * (1 / 3)
        L['G_B_xz'] = nets.lsgan(self.iter_f(rec, 'D_B_xz', r), True) * (1 / 3)
        L['G_B'] = L['G_B_xy'] + L['G_B_yz'] + L['G_B_xz']
        L['cycle_A'] = nets.l1(rec, real) * self.lambda_A
        (L['G_A'] + L['G_B'] + L['cycle_A']).backward()
        self.opt_G.step()
        self.n.set_requires_grad(ATHENA_D, True)
        self.opt_D.zero_grad()
        fd, rd = fake.detach(), rec.detach()

        def d_basic(name, other, ax_real, ax_fake):
            loss = (nets.lsgan(self.iter_f(real, name, ax_real), True) +
                    nets.lsgan(self.iter_f(other, name, ax_fake), False)) * 0.5
            loss.backward()
            return loss

        L['D_A_xy'] = d_basic('D_A_xy', fd, t, t)
        L['D_A_yz'] = d_basic('D_A_yz', fd, t, s)
        L['D_A_xz'] = d_basic('D_A_xz', fd, t, r)
        L['D_B_xy'] = d_basic('D_B_xy', rd, t, t)
        L['D_B_yz'] = d_basic('D_B_yz', rd, s, s)
        L['D_B_xz'] = d_basic('D_B_xz', rd, r, r)
        self.opt_D.step()
        return OrderedDict((k, float(v.detach())) for k, v in L.items())
