"""Athena (artifact-correction) training step on MI355X (reference: models/axial_to_lateral_gan_athena_model.py:6-331).

Same option surface, loss names and weights, plane wiring and optimizer grouping as the reference.  The one structural
change: the reference's `iter_f` runs the 2-D discriminator on every slice of the volume in a Python loop and stacks
the results (:286-296 -- 18*S discriminator passes per step); InstanceNorm is per instance, so the identical numbers
come out of ONE batched pass over all S slices (`ops.volume_all_slices` -> [S, C, A, B]), and the LSGAN mean over the
stacked volume equals the mean over the batch.
"""
import itertools
import os

import torch

from .. import ops
from . import networks
from .axial_to_lateral_gan_apollo_model import FlatAdam
from .base_model import BaseModel


class AxialToLateralGANAthenaModel(BaseModel):
    @staticmethod
    def modify_commandline_options(parser, is_train=True):
        parser.set_defaults(no_dropout=True)
        if is_train:
            parser.add_argument('--lambda_A', type=float, default=10.0, help='weight for cycle loss (A -> B -> A)')
            parser.add_argument('--pool_size', type=int, default=50)
            parser.add_argument('--gan_mode', type=str, default='vanilla', help='[vanilla| lsgan | wgangp]')
        parser.add_argument('--conversion_plane', type=str, nargs='+', default=['yz', 'xy'])
        parser.add_argument('--lambda_plane', type=int, nargs='+', default=[1, 1, 1])
        parser.add_argument('--netG_B', type=str, default='deep_linear_gen')
        return parser

    def __init__(self, opt):
        BaseModel.__init__(self, opt)
        self.loss_names = ['D_A_xy', 'D_A_xz', 'D_A_yz', 'G_A', 'G_A_xy', 'G_A_xz', 'G_A_yz', 'cycle_A', 'D_B_xy',
                           'D_B_xz', 'D_B_yz', 'G_B', 'G_B_xy', 'G_B_xz', 'G_B_yz']
        self.gan_mode = opt.gan_mode
        self.gen_dimension, self.dis_dimension = 3, 2  # athena:93-94
        plane_to_slice_axis = {'xy': 0, 'xz': 1, 'yz': 2}
        remain = [a for a in plane_to_slice_axis if a != opt.conversion_plane[0] and a != opt.conversion_plane[1]][0]
        self.source_sl_axis = plane_to_slice_axis[opt.conversion_plane[0]]
        self.target_sl_axis = plane_to_slice_axis[opt.conversion_plane[1]]
        self.remain_sl_axis = plane_to_slice_axis[remain]
        tot = float(opt.lambda_plane[0] + opt.lambda_plane[1] + opt.lambda_plane[2])
        self.lambda_plane_target, self.lambda_plane_source, self.lambda_plane_ref = [f / tot for f in opt.lambda_plane]
        self.visual_names = ['real', 'fake', 'rec'] * 2
        self.model_names = ['G_A', 'G_B', 'D_A_xy', 'D_A_xz', 'D_A_yz', 'D_B_xy', 'D_B_xz', 'D_B_yz'] \
            if self.isTrain else ['G_A', 'G_B']
        G = networks.define_G
        self.netG_A = G(opt.input_nc, opt.output_nc, opt.ngf, opt.netG, opt.norm, not opt.no_dropout, opt.init_type,
                        opt.init_gain, self.gpu_ids, dimension=self.gen_dimension)
        self.netG_B = G(opt.output_nc, opt.input_nc, opt.ngf, opt.netG_B, opt.norm, not opt.no_dropout, opt.init_type,
                        opt.init_gain, self.gpu_ids, dimension=self.gen_dimension)
        if self.isTrain:
            def D(nc):
                return networks.define_D(nc, opt.ndf, opt.netD, opt.n_layers_D, opt.norm, opt.init_type,
                                         opt.init_gain, False, self.gpu_ids, dimension=self.dis_dimension)
            self.netD_A_yz, self.netD_A_xy, self.netD_A_xz = D(opt.output_nc), D(opt.output_nc), D(opt.output_nc)
            self.netD_B_yz, self.netD_B_xy, self.netD_B_xz = D(opt.input_nc), D(opt.input_nc), D(opt.input_nc)
            self.criterionGAN = networks.GANLoss(opt.gan_mode).to(self.device)
            self.criterionCycle = ops.l1_loss
            self.optimizer_G = FlatAdam(itertools.chain(self.netG_A.parameters(), self.netG_B.parameters()),
                                        lr=opt.lr, betas=(opt.beta1, 0.999), overlap_all_reduce=True)
            self.optimizer_D = FlatAdam(  # chain order of athena:163-164
                itertools.chain(self.netD_A_yz.parameters(), self.netD_A_xy.parameters(), self.netD_A_xz.parameters(),
                                self.netD_B_yz.parameters(), self.netD_B_xy.parameters(), self.netD_B_xz.parameters()),
                lr=opt.lr, betas=(opt.beta1, 0.999))
            self.optimizers = [self.optimizer_G, self.optimizer_D]

    def set_input(self, input):
        AtoB = self.opt.direction == 'AtoB'
        self.real = input['A' if AtoB else 'B'].to(self.device)
        self.image_paths = input['A_paths' if AtoB else 'B_paths']
        self.cube_shape = self.real.shape
        self.num_slice = self.cube_shape[-3]

    def forward(self):
        self.fake = self.netG_A(self.real)
        self.rec = self.netG_B(self.fake)

    # The six discriminators are independent networks: their passes run on six HIP streams (autograd replays every op on
    # the stream of its forward, so the backward chains overlap too).  One batched PatchGAN pass keeps the matrix pipe
    # ~45 % busy (profiles/README.md); two or three side by side fill the gaps.  NC_D_STREAMS=0: one stream.
    _d_streams_on = os.environ.get('NC_D_STREAMS', '1') != '0'
    _d_nstreams = int(os.environ.get('NC_D_NSTREAMS', '6'))  # side streams the six jobs are dealt onto (job i -> stream i % n)
    # backward_G's discriminator passes over the slices of fake / rec are kept and reused by backward_D_basic, which runs
    # the same planes through the same (not yet updated) weights (ops.PatchGANShare).  NC_D_REUSE=0: both passes are run.
    _reuse_on = os.environ.get('NC_D_REUSE', '1') != '0'
    _shares = {}

    def _on_streams(self, fns, after=None):
        """fns[i]() on stream i, after everything queued on the calling stream so far (or after the event `after`); the
        calling stream waits for all of them.  Same numbers as running them in sequence: the jobs touch disjoint
        parameter / gradient sets."""
        if not self._d_streams_on or not self.real.is_cuda:
            return [f() for f in fns]
        if not hasattr(self, '_d_streams'):
            self._d_streams = []
        ns = min(len(fns), self._d_nstreams)
        while len(self._d_streams) < ns:
            self._d_streams.append(torch.cuda.Stream(device=self.device))
        main = torch.cuda.current_stream()
        out = []
        for i, f in enumerate(fns):
            st = self._d_streams[i % ns]
            if i < ns:
                if after is not None:
                    st.wait_event(after)
                else:
                    st.wait_stream(main)
            with torch.cuda.stream(st):
                out.append(f())
        for st in self._d_streams[:ns]:
            main.wait_stream(st)
        return out

    def iter_f(self, input, function, slice_axis):
        """athena:286-296, batched.  The reference iterates range(self.num_slice) = shape[-3] slices for every axis
        (cubes are cubic); a non-cubic crop is refused rather than silently sliced differently."""
        if not (input.shape[-1] == input.shape[-2] == input.shape[-3]):
            raise ValueError('Athena assumes cubic crops (num_slice = shape[-3] is used for every axis)')
        return function(ops.volume_all_slices(input, slice_axis))

    def backward_D_basic(self, netD, real, fake, slice_axis_real, slice_axis_fake):
        """athena:190-238.  The slices of `real` and of `fake` go through netD as ONE batch (InstanceNorm is per instance
        and each LSGAN mean runs over its own half, so every number is what two passes give; the GEMMs are twice as long
        and there are half as many launches).  A discriminator whose forward has side effects per call (spectral norm's
        power iteration) keeps the reference's two calls."""
        sr = ops.volume_all_slices(real, slice_axis_real)
        if not (real.shape[-1] == real.shape[-2] == real.shape[-3]):
            raise ValueError('Athena assumes cubic crops (num_slice = shape[-3] is used for every axis)')
        share = self._shares.pop(id(netD), None)
        if share is not None and netD.share_matches(share, fake, slice_axis_fake):
            # netD already saw exactly these fake planes with these weights in backward_G: only the real half runs
            pred = netD.forward_join_real(sr, share)
            pred_real, pred_fake = pred[:sr.shape[0]], pred[sr.shape[0]:]
        elif getattr(netD, 'one_plane_per_call', False):
            pred_real, pred_fake = netD(sr), netD(ops.volume_all_slices(fake.detach(), slice_axis_fake))
        else:
            sf = ops.volume_all_slices(fake.detach(), slice_axis_fake)
            pred = netD(torch.cat([sr, sf], 0))
            pred_real, pred_fake = pred[:sr.shape[0]], pred[sr.shape[0]:]
        loss_D = (self.criterionGAN(pred_real, True) + self.criterionGAN(pred_fake, False)) * 0.5
        loss_D.backward()
        return loss_D

    def backward_G(self):
        """athena:240-260"""
        g, f, r = self.criterionGAN, self.fake, self.rec
        jobs = [(f, self.netD_A_xy, self.target_sl_axis, self.lambda_plane_target),
                (f, self.netD_A_yz, self.source_sl_axis, self.lambda_plane_source),
                (f, self.netD_A_xz, self.remain_sl_axis, self.lambda_plane_ref),
                (r, self.netD_B_xy, self.target_sl_axis, 1 / 3),
                (r, self.netD_B_yz, self.source_sl_axis, 1 / 3),
                (r, self.netD_B_xz, self.remain_sl_axis, 1 / 3)]
        self._shares = {}

        def job(x, n, a, w):
            def fn(planes):
                if self._reuse_on and hasattr(n, 'can_share') and n.can_share(planes):
                    share = self._shares[id(n)] = ops.PatchGANShare()
                    return n.forward_fake_half(planes, share, x, a)
                return n(planes)
            return g(self.iter_f(x, fn, a), True) * w
        (self.loss_G_A_xy, self.loss_G_A_yz, self.loss_G_A_xz, self.loss_G_B_xy, self.loss_G_B_yz,
         self.loss_G_B_xz) = self._on_streams([lambda x=x, n=n, a=a, w=w: job(x, n, a, w) for x, n, a, w in jobs])
        self.loss_G_A = self.loss_G_A_xy + self.loss_G_A_yz + self.loss_G_A_xz
        self.loss_G_B = self.loss_G_B_xy + self.loss_G_B_yz + self.loss_G_B_xz
        self.loss_cycle_A = self.criterionCycle(self.rec, self.real) * self.opt.lambda_A
        self.loss_G = self.loss_G_A + self.loss_G_B + self.loss_cycle_A
        self.loss_G.backward()

    # The discriminators' own update only needs real / fake / rec (detached), the planes' activations kept by backward_G's passes and the
    # discriminators' parameters -- nothing the generators' backward or optimizer_G.step touch.  Its six jobs therefore wait for the END OF
    # forward() only (an event), not for everything the calling stream has queued: behind the data-gradient chains of the generator loss on
    # their streams they run UNDERNEATH the generators' backward (19 ms of the calling stream) instead of behind it.  Round 2 measured no gain
    # from this (167.3 vs 167.8 ms: every kernel filled the chip then); round 6: the update was 20.6 ms of a 50.9 ms step with the calling stream idle
    # (tools/athena_phases.py).  NC_ATHENA_D_EARLY=0: behind the generators' optimizer step, as the reference orders it.
    _d_early = os.environ.get('NC_ATHENA_D_EARLY', '1') != '0'

    def optimize_parameters(self):
        """athena:262-283"""
        Ds = [self.netD_A_xy, self.netD_A_yz, self.netD_A_xz, self.netD_B_xy, self.netD_B_yz, self.netD_B_xz]
        t, s, r = self.target_sl_axis, self.source_sl_axis, self.remain_sl_axis
        self.forward()
        fwd_done = None
        if self._d_early and self._d_streams_on and self.real.is_cuda:
            fwd_done = torch.cuda.Event()
            fwd_done.record()
        self.set_requires_grad(Ds, False)
        self.optimizer_G.zero_grad()
        self.backward_G()
        self.optimizer_G.all_reduce_mean()
        self.optimizer_G.step()
        self.set_requires_grad(Ds, True)
        # early start: the flat gradient buffer was zeroed right after the previous optimizer_D.step (below) -- zeroing it here, on the calling
        # stream, would run AFTER the early jobs have written their gradients
        self.optimizer_D.zero_grad(fill=fwd_done is None)
        bd = self.backward_D_basic
        (self.loss_D_A_xy, self.loss_D_A_yz, self.loss_D_A_xz, self.loss_D_B_xy, self.loss_D_B_yz,
         self.loss_D_B_xz) = self._on_streams([
             lambda: bd(self.netD_A_xy, self.real, self.fake, t, t), lambda: bd(self.netD_A_yz, self.real, self.fake, t, s),
             lambda: bd(self.netD_A_xz, self.real, self.fake, t, r), lambda: bd(self.netD_B_xy, self.real, self.rec, t, t),
             lambda: bd(self.netD_B_yz, self.real, self.rec, s, s), lambda: bd(self.netD_B_xz, self.real, self.rec, r, r)], after=fwd_done)
        for sh in self._shares.values():
            sh.release()
        self._shares = {}
        self.optimizer_D.all_reduce_mean()
        self.optimizer_D.step()
        if fwd_done is not None:
            self.optimizer_D.grad.zero_()
