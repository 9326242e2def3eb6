"""Apollo training step on MI355X (reference: models/axial_to_lateral_gan_apollo_model.py:7-353).

Same option surface, loss names, loss weights, np.random draw order and optimizer grouping as the
reference; what changes is underneath:
  * G_A / G_B / the four 2-D PatchGANs, slices, MIPs, LSGAN / L1 losses all run HIP kernels (neuroclear_amd.ops);
  * each optimizer owns ONE flat parameter buffer (+ flat grad, exp_avg, exp_avg_sq): Adam is a single fused launch
    and, with torch.distributed initialised, the gradient all-reduce is a single RCCL call per optimizer phase
    (G before optimizer_G.step at apollo:295, the Ds before optimizer_D.step at :307).
"""
import itertools
import os

import numpy as np
import torch

from .. import ops
from . import networks
from .base_model import BaseModel


class FlatAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (lr, betas, eps=1e-8, no weight decay / amsgrad) over parameters that are views
    into one flat fp32 buffer; step() is one nc_adam_step launch and the data-parallel gradient exchange is one
    all-reduce.  A real torch Optimizer subclass, so torch's lr schedulers (networks.get_scheduler) drive
    param_groups[0]['lr'] as usual."""

    def __init__(self, params, lr, betas, eps=1e-8, overlap_all_reduce=False):
        params = list(params)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.params = params
        dev = params[0].device
        n = sum(p.numel() for p in params)
        self.flat = torch.empty(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in params:
            k = p.numel()
            self.flat[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + k].view_as(p.data)
            # .grad stays None until a backward delivers one: with a view of self.grad here, a backward BEFORE the first zero_grad()
            # would have its whole-network call write the slice and AccumulateGrad then add the same memory to itself (2 x the gradient)
            p.grad = None
            off += k
        self.state_step = 0
        ops.register_flat_grad(self.flat, self.grad)
        self._buckets = []
        self._overlap_armed = True
        self._overlap = overlap_all_reduce  # only for parameter sets whose gradients are all produced on ONE stream

    def zero_grad(self, set_to_none=False, fill=True):
        if self._overlap:
            self.enable_overlapped_all_reduce()
            for b in self._buckets:
                b['seen'] = 0
        if fill:
            self.grad.zero_()
        ops.flat_grad_step_begin(self.flat)
        # .grad = None: the whole-network backward calls write into slices of self.grad and hand autograd views of them,
        # which AccumulateGrad adopts as they are; gradients that arrive any other way are collected by _collect()
        for p in self.params:
            p.grad = None

    def _collect(self):
        """Make self.grad hold every parameter's gradient and p.grad a view of it (gradients produced outside the direct
        path -- layer-by-layer modules, clones made by autograd -- are copied in; a parameter without a gradient keeps
        the zeros of zero_grad)."""
        off = 0
        for p in self.params:
            k = p.numel()
            g = p.grad
            if g is not None and g.data_ptr() != self.grad.data_ptr() + 4 * off:
                view = self.grad[off:off + k].view_as(p.data)
                view.copy_(g)
                p.grad = view
            off += k

    # -- data-parallel gradient exchange ---------------------------------------------------------------------------
    # Backward produces the gradients of the LAST parameters first, so the flat gradient buffer fills from its end:
    # it is cut into `n_buckets` contiguous ranges of about equal size and the all-reduce of a range is issued (async,
    # on the collective's own stream) the moment the last of its parameters has received its gradient -- the exchange of
    # G_B's and the U-Net decoder's gradients runs underneath the encoder's backward kernels, and only the last small
    # bucket is exposed.  xGMI is point-to-point (7 links x ~153 GB/s), a ring all-reduce is per-link bound: a few
    # buckets of >= 8 MB keep every link busy without paying the ~50 us launch latency too often.
    def enable_overlapped_all_reduce(self, n_buckets=3):
        """Arm the hooks (idempotent).  Without an initialised process group of size > 1 this does nothing (util.dist.exchange_active)."""
        from ..util.dist import exchange_active
        if not exchange_active() or self._buckets:
            return
        total = self.flat.numel()
        target = (total + n_buckets - 1) // n_buckets
        bounds, off, start = [], 0, 0
        members = []
        for p in self.params:
            members.append(p)
            off += p.numel()
            if off - start >= target or off == total:
                bounds.append((start, off, members))
                start, members = off, []
        self._buckets = [dict(lo=lo, hi=hi, n=len(m), seen=0, work=None, members=m) for lo, hi, m in bounds]
        for bi, (_, _, m) in enumerate(bounds):
            for p in m:
                p.register_post_accumulate_grad_hook(lambda _p, bi=bi: self._grad_ready(bi))

    def _grad_ready(self, bi):
        b = self._buckets[bi]
        b['seen'] += 1
        if b['seen'] > b['n'] or b['work'] is not None:
            # a second backward into this optimizer before all_reduce_mean(): its gradients are being ADDED into slices whose
            # exchange is already in flight (or done) -- the averaged result would silently miss them
            raise RuntimeError('FlatAdam: gradients arrived for a bucket whose all-reduce was already issued; call all_reduce_mean() / '
                               'zero_grad() between backward passes, or construct the optimizer with overlap_all_reduce=False')
        if b['seen'] == b['n'] and self._overlap_armed:
            # only when every gradient of the range already sits in the flat buffer (written there by the whole-network
            # backward calls); otherwise all_reduce_mean() collects them first and exchanges the range synchronously
            off = b['lo']
            for p in b['members']:
                if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * off:
                    return
                off += p.numel()
            # the collective is ordered behind everything queued on the current stream (autograd runs this hook on the
            # stream that produced the gradient)
            b['work'] = torch.distributed.all_reduce(self.grad[b['lo']:b['hi']], async_op=True)

    def all_reduce_mean(self):
        dist = torch.distributed
        from ..util.dist import exchange_active
        if not exchange_active():
            return
        ws = dist.get_world_size()
        self._collect()
        if self._buckets:
            for b in self._buckets:
                if b['work'] is not None:
                    b['work'].wait()  # the current stream waits for the collective's stream
                else:  # a bucket with a parameter that got no gradient this step (e.g. requires_grad off): do it now
                    dist.all_reduce(self.grad[b['lo']:b['hi']])
                b['work'], b['seen'] = None, 0
        else:
            dist.all_reduce(self.grad)
        self.grad.div_(ws)

    @torch.no_grad()
    def step(self, closure=None):
        self._collect()
        self.state_step += 1
        g = self.param_groups[0]
        ops.adam_step(self.flat, self.grad, self.exp_avg, self.exp_avg_sq, g['lr'], g['betas'][0], g['betas'][1],
                      g['eps'], self.state_step)
        ops.bump_param_generation(self.flat)


class AxialToLateralGANApolloModel(BaseModel):
    @staticmethod
    def modify_commandline_options(parser, is_train=True):
        parser.set_defaults(no_dropout=True)
        if is_train:
            parser.add_argument('--lambda_A', type=float, default=10.0, help='weight for cycle loss (A -> B -> A)')
            parser.add_argument('--gan_mode', type=str, default='vanilla', help='[vanilla| lsgan | wgangp]')
            parser.add_argument('--lambda_plane', type=int, nargs='+', default=[1, 1, 1])
            parser.add_argument('--randomize_projection_depth', action='store_true')
            parser.add_argument('--projection_depth', type=int, default=10)
            parser.add_argument('--min_projection_depth', type=int, default=2)
        parser.add_argument('--netG_B', type=str, default='deep_linear_gen')
        return parser

    _d_streams_on = os.environ.get('NC_D_STREAMS', '1') != '0'
    _d_on_main = os.environ.get('NC_D_MAIN', '1') != '0'
    _d_nstreams = int(os.environ.get('NC_D_NSTREAMS', '4'))  # side streams the discriminator jobs are dealt onto (job i -> stream i % n)

    def __init__(self, opt):
        BaseModel.__init__(self, opt)
        self._d_streams = []
        self._fwd_done = None
        self.loss_names = ['D_A_lateral', 'D_A_axial', 'G_A', 'G_A_lateral', 'G_A_axial', 'cycle',
                           'D_B_lateral', 'D_B_axial', 'G_B', 'G_B_lateral', 'G_B_axial']
        self.gan_mode = opt.gan_mode
        self.gen_dimension, self.dis_dimension = 3, 2  # apollo:66-67
        self.randomize_projection_depth = opt.randomize_projection_depth
        if not self.randomize_projection_depth:
            self.projection_depth_custom = opt.projection_depth
        else:
            self.max_projection_depth = opt.projection_depth
            self.min_projection_depth = opt.min_projection_depth
        self.visual_names = ['real', 'fake', 'rec'] * 2
        tot = float(opt.lambda_plane[0] + opt.lambda_plane[1] + opt.lambda_plane[2])
        self.lambda_plane_target, self.lambda_slice, self.lambda_proj = [f / tot for f in opt.lambda_plane]
        self.lateral_axis, self.axial_1_axis, self.axial_2_axis = 0, 1, 2
        self.model_names = ['G_A', 'G_B', 'D_A_lateral', 'D_A_axial', 'D_B_lateral', 'D_B_axial'] if self.isTrain \
            else ['G_A', 'G_B']
        G = networks.define_G
        self.netG_A = G(opt.input_nc, opt.output_nc, opt.ngf, opt.netG, opt.norm, not opt.no_dropout, opt.init_type,
                        opt.init_gain, self.gpu_ids, dimension=self.gen_dimension)
        self.netG_B = G(opt.output_nc, opt.input_nc, opt.ngf, opt.netG_B, opt.norm, not opt.no_dropout, opt.init_type,
                        opt.init_gain, self.gpu_ids, dimension=self.gen_dimension)
        if self.isTrain:
            def D(nc):
                return networks.define_D(nc, opt.ndf, opt.netD, opt.n_layers_D, opt.norm, opt.init_type,
                                         opt.init_gain, False, self.gpu_ids, dimension=self.dis_dimension)
            self.netD_A_axial = D(opt.output_nc)
            self.netD_A_lateral = D(opt.output_nc)
            self.netD_B_axial = D(opt.input_nc)
            self.netD_B_lateral = D(opt.input_nc)
            self.criterionGAN = networks.GANLoss(opt.gan_mode).to(self.device)
            self.criterionCycle = ops.l1_loss
            self._make_optimizers(opt)

    def _make_optimizers(self, opt):
        """apollo:131-138.  Call again after loading new parameter tensors (load_state_dict copies in place, so the
        flat views stay valid)."""
        # the generators' backward runs on one stream: their gradient exchange is bucketed and overlapped with it; the
        # discriminators' backward passes run on four streams, their (smaller) exchange is one call after the join
        self.optimizer_G = FlatAdam(itertools.chain(self.netG_A.parameters(), self.netG_B.parameters()),
                                    lr=opt.lr, betas=(opt.beta1, 0.999), overlap_all_reduce=True)
        self.optimizer_D = FlatAdam(
            itertools.chain(self.netD_A_axial.parameters(), self.netD_A_lateral.parameters(),
                            self.netD_B_axial.parameters(), self.netD_B_lateral.parameters()),
            lr=opt.lr, betas=(opt.beta1, 0.999))
        self.optimizers = [self.optimizer_G, self.optimizer_D]

    def set_input(self, input):
        AtoB = self.opt.direction == 'AtoB'
        self.real = input['A' if AtoB else 'B'].to(self.device)
        self.image_paths = input['A_paths' if AtoB else 'B_paths']
        self.cube_shape = self.real.shape
        self.num_slice = self.cube_shape[-3]
        if not self.randomize_projection_depth:
            self.projection_depth = self.projection_depth_custom
        else:
            self.projection_depth = np.random.randint(max(2, self.min_projection_depth),
                                                      self.max_projection_depth + 1)

    _early_DA = os.environ.get('NC_D_EARLY', '0') != '0'

    def forward(self):
        self.fake = self.netG_A(self.real)
        # the generator-loss passes of D_A only need `fake`: NC_D_EARLY=1 starts them under G_B's forward instead of behind it.  Measured twice, slower
        # both times (round 5, under G_B's persistent 5^3 kernel: +0.6 ms; round 6, under its 0.8 ms of 7^3 launches: 24.98 against 24.64 ms, same-box
        # alternation profiles/r06_ab_d_early.txt): small kernels under a persistent one cost it more than the idle time they fill.  Default off
        self._fake_done = None
        if self.real.is_cuda and self._d_streams_on and self._early_DA and self.isTrain:
            self._fake_done = torch.cuda.Event()
            self._fake_done.record()
        self.rec = self.netG_B(self.fake)

    # -- Volume.get_slice / get_projection (apollo:322-351); num_slice = shape[-1] for every axis (:325).
    #    The random draw happens here, in the reference's order; the discriminator call is deferred so that several
    #    planes going through the SAME network are evaluated as one batch (InstanceNorm is per instance and the LSGAN
    #    means are taken per plane, so every number is unchanged -- only the launch count drops from 18 to 8 passes).
    def _slice(self, input, slice_axis):
        return ops.volume_slice(input, slice_axis, np.random.randint(input.shape[-1]))

    def _proj(self, input, slice_axis):
        start = np.random.randint(0, input.shape[-1] - self.projection_depth)
        return ops.volume_mip(input, slice_axis, start, self.projection_depth)

    @staticmethod
    def _D(netD, planes):
        # every plane is [N, C, A, B] (N = batch): plane i owns rows i*N .. (i+1)*N of the batched prediction, and the
        # LSGAN mean of a plane runs over its whole batch, as the reference's per-plane netD call does (apollo:169-193)
        if getattr(netD, 'one_plane_per_call', False):  # spectral norm: every call moves (u, v), keep the reference's calls
            return [netD(pl) for pl in planes]
        pred = netD(planes[0] if len(planes) == 1 else torch.cat(planes, 0))
        n = planes[0].shape[0]
        return [pred[i * n:(i + 1) * n] for i in range(len(planes))]

    def _D_many(self, jobs, loss_fn, after=None, backward=False):
        """Evaluate independent discriminators concurrently: jobs = [(netD, planes_fn)], loss_fn(i, preds) -> tensor or
        tuple of tensors.  Job i cuts its planes (np.random draws: jobs are built one after the other, so the draw
        order is the reference's), runs its network and its loss on HIP stream i -- and, because autograd replays
        every op on the stream of its forward, the backward too -- so the four chains of small 2-D kernels overlap
        instead of queueing behind each other.
        after: an event instead of "everything queued on the calling stream so far" as the start condition;
        backward=True: each job also calls backward() on the sum of its losses, on its own stream (the jobs touch
        disjoint parameter sets, so this equals the reference's separate loss.backward() calls).  The calling stream
        waits for all job streams before this returns.  NC_D_STREAMS=0: in sequence on the calling stream."""
        def run(i, net, planes_fn):
            res = loss_fn(i, self._D(net, planes_fn()))
            if backward:
                tot = res if torch.is_tensor(res) else sum(res[1:], res[0])
                tot.backward()
            return res
        if not self._d_streams_on or not self.real.is_cuda:
            return [run(i, net, fn) for i, (net, fn) in enumerate(jobs)]
        main = torch.cuda.current_stream()
        # The runtime maps a process's streams onto FOUR hardware queues (neuroclear_amd/__init__.py): the calling stream's and three others.
        # Four side streams therefore share three queues, and the two chains that share one run one after the other -- while the calling
        # stream has nothing to do but wait for them.  So the LAST job runs on the calling stream itself (NC_D_MAIN=0: on a side stream).
        on_main = self._d_on_main and len(jobs) > 3 and not backward  # (the discriminators' own update overlaps the next step's forward: not there)
        nside = len(jobs) - 1 if on_main else len(jobs)
        ns = min(nside, self._d_nstreams)
        while len(self._d_streams) < ns:
            self._d_streams.append(torch.cuda.Stream(device=self.device))
        out = []
        for i, (net, fn) in enumerate(jobs[:nside]):
            st = self._d_streams[i % ns]
            ev = after[i] if isinstance(after, (list, tuple)) else after
            if ev is not None:
                st.wait_event(ev)
            else:
                st.wait_stream(main)
            with torch.cuda.stream(st):
                out.append(run(i, net, fn))
        if on_main:
            out.append(run(nside, *jobs[nside]))
        for i in range(ns):
            main.wait_stream(self._d_streams[i])
        return out

    def iter_f(self, input, function, slice_axis):
        return function(self._slice(input, slice_axis))

    def proj_f(self, input, function, slice_axis):
        return function(self._proj(input, slice_axis))

    def _d_loss(self, pred_real, pred_fake):
        return (self.criterionGAN(pred_real, True) + self.criterionGAN(pred_fake, False)) * 0.5

    def backward_D_slice(self, netD, real, fake, slice_axis_real, slice_axis_fake):
        pr, pf = self._D(netD, [self._slice(real, slice_axis_real), self._slice(fake.detach(), slice_axis_fake)])
        loss_D = self._d_loss(pr, pf)
        loss_D.backward()
        return loss_D

    def backward_D_projection(self, netD, real, fake, slice_axis_real, slice_axis_fake):
        pr, pf = self._D(netD, [self._slice(real, slice_axis_real), self._proj(fake.detach(), slice_axis_fake)])
        loss_D = self._d_loss(pr, pf)
        loss_D.backward()
        return loss_D

    def backward_D_A_lateral(self):
        self.loss_D_A_lateral = self.backward_D_projection(self.netD_A_lateral, self.real, self.fake,
                                                           self.lateral_axis, self.lateral_axis)

    def backward_D_A_axial(self):
        """apollo:225-239: two (real slice, fake MIP) pairs through netD_A_axial -- one batch of four planes."""
        fd = self.fake.detach()
        planes = [self._slice(self.real, self.lateral_axis), self._proj(fd, self.axial_1_axis),
                  self._slice(self.real, self.lateral_axis), self._proj(fd, self.axial_2_axis)]
        p = self._D(self.netD_A_axial, planes)
        self.loss_D_A_axial_1 = self._d_loss(p[0], p[1])
        self.loss_D_A_axial_2 = self._d_loss(p[2], p[3])
        (self.loss_D_A_axial_1 + self.loss_D_A_axial_2).backward()
        self.loss_D_A_axial = (self.loss_D_A_axial_1 + self.loss_D_A_axial_2) * 0.5

    def backward_D_B_lateral(self):
        self.loss_D_B_lateral = self.backward_D_slice(self.netD_B_lateral, self.real, self.rec, self.lateral_axis,
                                                      self.lateral_axis)

    def backward_D_B_axial(self):
        """apollo:241-253"""
        rd = self.rec.detach()
        planes = [self._slice(self.real, self.axial_1_axis), self._slice(rd, self.axial_1_axis),
                  self._slice(self.real, self.axial_2_axis), self._slice(rd, self.axial_2_axis)]
        p = self._D(self.netD_B_axial, planes)
        self.loss_D_B_axial_1 = self._d_loss(p[0], p[1])
        self.loss_D_B_axial_2 = self._d_loss(p[2], p[3])
        (self.loss_D_B_axial_1 + self.loss_D_B_axial_2).backward()
        self.loss_D_B_axial = (self.loss_D_B_axial_1 + self.loss_D_B_axial_2) * 0.5

    def backward_G(self):
        """apollo:255-283"""
        lambda_A = self.opt.lambda_A
        g = self.criterionGAN
        f, r = self.fake, self.rec
        la, a1, a2 = self.lateral_axis, self.axial_1_axis, self.axial_2_axis
        # draw order of the reference: A_lateral, A_axial x2, B_lateral, B_axial x2
        jobs = [(self.netD_A_lateral, lambda: [self._proj(f, la)]),
                (self.netD_A_axial, lambda: [self._proj(f, a1), self._proj(f, a2)]),
                (self.netD_B_lateral, lambda: [self._slice(r, la)]),
                (self.netD_B_axial, lambda: [self._slice(r, a1), self._slice(r, a2)])]

        def loss(i, p):
            if i % 2 == 0:
                return g(p[0], True) * self.lambda_plane_target
            return g(p[0], True) * self.lambda_slice + g(p[1], True) * self.lambda_slice
        fd = getattr(self, '_fake_done', None)
        (self.loss_G_A_lateral, self.loss_G_A_axial, self.loss_G_B_lateral, self.loss_G_B_axial) = \
            self._D_many(jobs, loss, after=[fd, fd, None, None] if fd is not None else None)
        self.loss_G_A = self.loss_G_A_lateral + self.loss_G_A_axial * 0.5
        self.loss_G_B = self.loss_G_B_lateral + self.loss_G_B_axial * 0.5
        self.loss_cycle = self.criterionCycle(self.rec, self.real) * lambda_A
        self.loss_G = self.loss_G_A + self.loss_G_B + self.loss_cycle
        self.loss_G.backward()

    def backward_D_all(self):
        """apollo:297-305: backward_D_A_lateral, backward_D_A_axial, backward_D_B_lateral, backward_D_B_axial, planes
        cut in the reference's draw order.  The discriminator step only needs fake / rec (detached) and the
        discriminators' own parameters, none of which the generators' backward pass or optimizer_G.step touch: the
        four jobs start as soon as forward() is done on the GPU (self._fwd_done) and run -- forward, losses and
        backward -- on the discriminator streams underneath the generators' backward pass."""
        fd, rd = self.fake.detach(), self.rec.detach()
        real = self.real
        la, a1, a2 = self.lateral_axis, self.axial_1_axis, self.axial_2_axis
        sl, pj = self._slice, self._proj
        jobs = [(self.netD_A_lateral, lambda: [sl(real, la), pj(fd, la)]),
                (self.netD_A_axial, lambda: [sl(real, la), pj(fd, a1), sl(real, la), pj(fd, a2)]),
                (self.netD_B_lateral, lambda: [sl(real, la), sl(rd, la)]),
                (self.netD_B_axial, lambda: [sl(real, a1), sl(rd, a1), sl(real, a2), sl(rd, a2)])]

        def loss(i, p):
            if i % 2 == 0:
                return (self._d_loss(p[0], p[1]),)
            return (self._d_loss(p[0], p[1]), self._d_loss(p[2], p[3]))
        (l_al,), (l_a1, l_a2), (l_bl,), (l_b1, l_b2) = self._D_many(jobs, loss, after=self._fwd_done, backward=True)
        self.loss_D_A_lateral = l_al
        self.loss_D_A_axial_1, self.loss_D_A_axial_2 = l_a1, l_a2
        self.loss_D_A_axial = (l_a1 + l_a2) * 0.5
        self.loss_D_B_lateral = l_bl
        self.loss_D_B_axial_1, self.loss_D_B_axial_2 = l_b1, l_b2
        self.loss_D_B_axial = (l_b1 + l_b2) * 0.5

    def optimize_parameters(self):
        """apollo:285-307"""
        Ds = [self.netD_A_lateral, self.netD_A_axial, self.netD_B_lateral, self.netD_B_axial]
        self.forward()
        self._fwd_done = None
        if self.real.is_cuda and self._d_streams_on:
            self._fwd_done = torch.cuda.Event()
            self._fwd_done.record()
        self.set_requires_grad(Ds, False)
        self.optimizer_G.zero_grad()
        self.backward_G()
        self.optimizer_G.all_reduce_mean()
        self.optimizer_G.step()
        self.set_requires_grad(Ds, True)
        # the discriminators' flat gradient buffer is zeroed right after optimizer_D.step (below), i.e. before this
        # step's forward(): zeroing it here, on the calling stream, would run after the early discriminator backward
        self.optimizer_D.zero_grad(fill=False)  # (re-attaches the .grad views only; the buffer starts as zeros)
        self.backward_D_all()
        self.optimizer_D.all_reduce_mean()
        self.optimizer_D.step()
        self.optimizer_D.grad.zero_()
