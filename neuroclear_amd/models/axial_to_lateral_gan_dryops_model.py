"""Dryops training step on MI355X (reference: models/axial_to_lateral_gan_dryops_model.py:7-300): the Apollo ablation
with no backward path -- G_A and the two D_A PatchGANs only.  Same option surface, loss names, np.random draw order
and optimizer grouping as the reference; kernels, flat Adam buffers and the per-phase gradient all-reduce are the
Apollo ones (axial_to_lateral_gan_apollo_model.py in this package)."""
import itertools

from . import networks
from .axial_to_lateral_gan_apollo_model import AxialToLateralGANApolloModel, FlatAdam
from .base_model import BaseModel


class AxialToLateralGANDryopsModel(AxialToLateralGANApolloModel):
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
        return parser

    def __init__(self, opt):
        BaseModel.__init__(self, opt)
        self.loss_names = ['D_A_lateral', 'D_A_axial', 'G_A', 'G_A_lateral', 'G_A_axial']  # dryops:63
        self.gan_mode = opt.gan_mode
        self.gen_dimension, self.dis_dimension = 3, 2
        self.randomize_projection_depth = opt.randomize_projection_depth
        if not self.randomize_projection_depth:
            self.projection_depth_custom = opt.projection_depth
        else:
            self.max_projection_depth = opt.projection_depth
            self.min_projection_depth = opt.min_projection_depth
        self.visual_names = ['real', 'fake']
        tot = float(opt.lambda_plane[0] + opt.lambda_plane[1] + opt.lambda_plane[2])
        self.lambda_plane_target, self.lambda_slice, self.lambda_proj = [f / tot for f in opt.lambda_plane]
        self.lateral_axis, self.axial_1_axis, self.axial_2_axis = 0, 1, 2
        self.model_names = ['G_A', 'D_A_lateral', 'D_A_axial'] if self.isTrain else ['G_A']
        self.netG_A = networks.define_G(opt.input_nc, opt.output_nc, opt.ngf, opt.netG, opt.norm, not opt.no_dropout,
                                        opt.init_type, opt.init_gain, self.gpu_ids, dimension=self.gen_dimension)
        if self.isTrain:
            def D():
                return networks.define_D(opt.output_nc, opt.ndf, opt.netD, opt.n_layers_D, opt.norm, opt.init_type,
                                         opt.init_gain, False, self.gpu_ids, dimension=self.dis_dimension)
            self.netD_A_axial = D()
            self.netD_A_lateral = D()
            self.criterionGAN = networks.GANLoss(opt.gan_mode).to(self.device)
            self._make_optimizers(opt)

    def _make_optimizers(self, opt):
        """dryops:101-106"""
        self.optimizer_G = FlatAdam(self.netG_A.parameters(), lr=opt.lr, betas=(opt.beta1, 0.999),
                                    overlap_all_reduce=True)
        self.optimizer_D = FlatAdam(itertools.chain(self.netD_A_axial.parameters(), self.netD_A_lateral.parameters()),
                                    lr=opt.lr, betas=(opt.beta1, 0.999))
        self.optimizers = [self.optimizer_G, self.optimizer_D]

    def forward(self):
        self.fake = self.netG_A(self.real)

    def backward_G(self):
        """dryops:219-232"""
        g = self.criterionGAN
        (p_lat,) = self._D(self.netD_A_lateral, [self._proj(self.fake, self.lateral_axis)])
        p_ax = self._D(self.netD_A_axial, [self._proj(self.fake, self.axial_1_axis),
                                           self._proj(self.fake, self.axial_2_axis)])
        self.loss_G_A_lateral = g(p_lat, True) * self.lambda_plane_target
        self.loss_G_A_axial = g(p_ax[0], True) * self.lambda_slice + g(p_ax[1], True) * self.lambda_slice
        self.loss_G_A = self.loss_G_A_lateral + self.loss_G_A_axial * 0.5
        self.loss_G = self.loss_G_A
        self.loss_G.backward()

    def optimize_parameters(self):
        """dryops:234-258"""
        Ds = [self.netD_A_lateral, self.netD_A_axial]
        self.forward()
        self.set_requires_grad(Ds, False)
        self.optimizer_G.zero_grad()
        self.backward_G()
        self.optimizer_G.all_reduce_mean()
        self.optimizer_G.step()
        self.set_requires_grad(Ds, True)
        self.optimizer_D.zero_grad()
        self.backward_D_A_lateral()
        self.backward_D_A_axial()
        self.optimizer_D.all_reduce_mean()
        self.optimizer_D.step()
