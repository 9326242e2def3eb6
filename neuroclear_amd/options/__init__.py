"""Command-line surface of the hot path (reference: options/base_options.py, train_options.py, test_options.py).

Only the flags the MI355X path consumes are declared ('Config / flags'); the two-stage parse that
lets the chosen model add its own flags (options/base_options.py:75-101) is kept."""
import argparse

import torch

from .. import models


def _base(parser):
    """options/base_options.py:22-73 (hot-path subset, same names / defaults)."""
    a = parser.add_argument
    a('--dataroot', required=True)
    a('--name', type=str, default='experiment_name')
    a('--gpu_ids', type=str, default='0')
    a('--checkpoints_dir', type=str, default='./checkpoints')
    a('--image_dimension', default=3, type=int)
    a('--model', type=str, default='cycle_gan')
    a('--input_nc', type=int, default=1)
    a('--output_nc', type=int, default=1)
    a('--ngf', type=int, default=64)
    a('--ndf', type=int, default=64)
    a('--netD', type=str, default='basic')
    a('--netG', type=str, default='unet_deconv')
    a('--n_layers_D', type=int, default=3)
    a('--norm', type=str, default='instance')
    a('--init_type', type=str, default='normal')
    a('--init_gain', type=float, default=0.02)
    a('--no_dropout', action='store_true')
    a('--dataset_mode', type=str, default='singlevolume')
    a('--direction', type=str, default='AtoB')
    a('--serial_batches', action='store_true')
    a('--batch_size', type=int, default=1)
    a('--crop_size', type=int, nargs='+', default=[0, 0, 0])
    a('--dice_size', type=int, nargs='+', default=[0, 0, 0])
    a('--preprocess', type=str, default='none')
    a('--epoch', type=str, default='latest')
    a('--load_iter', type=int, default=0)
    a('--verbose', action='store_true')
    a('--overlap', type=int, default=0)      # added by the dice dataset in the reference (diceImage_dataset.py:16-21)
    a('--border_cut', type=int, default=0)
    # not in the reference (fp32 only): arithmetic of the 3^3 / 5^3 convolutions, BASELINE.json configs[3].  bf16 / fp16 =
    # 16-bit MFMA operands with fp32 accumulation; InstanceNorm, losses, Adam and the master weights stay fp32.
    a('--precision', type=str, default='fp32', choices=['fp32', 'bf16', 'fp16'])
    return parser


def _train(parser):
    """options/train_options.py (hot-path subset)."""
    a = parser.add_argument
    a('--print_freq', type=int, default=500)
    a('--save_latest_freq', type=int, default=500)
    a('--save_by_iter', action='store_true')
    a('--continue_train', action='store_true')
    a('--epoch_count', type=int, default=1)
    a('--phase', type=str, default='train')
    a('--n_epochs', type=int, default=50000000)
    a('--n_epochs_decay', type=int, default=100)
    a('--beta1', type=float, default=0.1)
    a('--lr', type=float, default=0.0001)
    a('--lr_policy', type=str, default='linear')
    a('--lr_decay_iters', type=int, default=50)
    a('--max_iters', type=int, default=0, help='stop after this many iterations (0 = run until killed, as the '
                                                'reference does)')
    return parser


def _test(parser):
    """options/test_options.py (hot-path subset)."""
    a = parser.add_argument
    a('--results_dir', type=str, default='./results/')
    a('--phase', type=str, default='test')
    a('--eval', action='store_true')
    a('--data_type', type=str, default='uint16')
    a('--histogram_match', action='store_true')
    a('--normalize_intensity', action='store_true')
    a('--sat_level', type=float, nargs='+', default=[0.25, 99.75])
    a('--save_volume', action='store_true')
    a('--skip_real', action='store_true')
    parser.set_defaults(model='test', dataset_mode='diceImage')
    return parser


class _Options:
    isTrain = True

    def parse(self, argv=None):
        parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter, allow_abbrev=False)
        parser = _base(parser)
        parser = _train(parser) if self.isTrain else _test(parser)
        opt, _ = parser.parse_known_args(argv)
        # second pass: the model adds its flags (base_options.py:88-93)
        parser = models.get_option_setter(opt.model)(parser, self.isTrain)
        opt = parser.parse_args(argv)
        opt.isTrain = self.isTrain
        opt.gpu_ids = [int(g) for g in opt.gpu_ids.split(',') if int(g) >= 0]
        if opt.gpu_ids:
            torch.cuda.set_device(opt.gpu_ids[0])
        self.opt = opt
        return opt


class TrainOptions(_Options):
    isTrain = True


class TestOptions(_Options):
    isTrain = False
