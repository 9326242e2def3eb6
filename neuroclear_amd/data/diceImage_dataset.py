"""DiceImageDataSet on the device (reference: data/diceImage_dataset.py:9-123 + the inference subset of
data/base_dataset.py:87-143,291-301).

The reference reads the volume, zero-pads it to the dice grid, reflect-pads by border_cut and slices cube i on the
host; every cube then crosses PCIe.  Here the ORIGINAL uint16/uint8 volume is uploaded once and stays resident in HBM;
`__getitem__(i)` launches nc_dice_cut_cube, which produces the same (R+2b)^3 float32 cube -- zero dice pad, reflect
border, v/65535 computed in fp64 then rounded to fp32 like numpy does -- directly on the GPU."""
import os

import numpy as np
import torch

from .._lib import I, P, check, lib
from ..util import util


def _load_volume(path):
    if path.endswith('.npy'):
        return np.load(path)
    from ..util import tiff
    return tiff.imread(path)  # uncompressed grayscale TIFF stacks (what the reference's skimage.io.imsave writes)


class DiceImageDataSet:
    @staticmethod
    def modify_commandline_options(parser, is_train=False):
        parser.add_argument('--overlap', type=int, default=0)
        parser.add_argument('--border_cut', default=0, type=int)
        return parser

    def __init__(self, opt, volume=None):
        """opt: dataroot (directory holding one .npy/.tif volume) or `volume` given directly (numpy / torch, uint8 or
        uint16); dice_size, overlap, border_cut, gpu_ids as in the reference's options."""
        self.opt = opt
        self.roi_size = opt.dice_size[0]
        self.overlap = opt.overlap
        self.border_cut = opt.border_cut
        if self.border_cut < 1:
            raise ValueError('border_cut must be >= 1: the reference crops cube[b:-b], which is empty for b = 0 '
                             '(util/assemble_dice.py:143-145)')
        if volume is None:
            names = sorted(f for f in os.listdir(opt.dataroot) if not f.startswith('.') and
                           f.endswith(('.npy', '.tif', '.tiff')))
            if not names:
                raise FileNotFoundError('no volume (.npy/.tif) under %s' % opt.dataroot)
            volume = _load_volume(os.path.join(opt.dataroot, names[0]))
        if isinstance(volume, np.ndarray):
            if volume.dtype not in (np.uint8, np.uint16):
                raise TypeError('input volume must be uint8 or uint16 (data/base_dataset.py:134-143), got %s'
                                % volume.dtype)
            self.is_u16 = volume.dtype == np.uint16
            # torch has no uint16 arithmetic; the bytes are only ever read by the HIP kernel
            host = torch.from_numpy(volume.view(np.int16) if self.is_u16 else volume)
        else:
            self.is_u16 = volume.dtype in (torch.int16, torch.uint16)
            host = volume
        if host.dim() != 3:
            raise ValueError('expected a 3-D volume')
        self.device = torch.device('cuda', opt.gpu_ids[0]) if getattr(opt, 'gpu_ids', None) else torch.device('cuda')
        self.volume = host.contiguous().to(self.device)
        self.image_size_original = tuple(int(s) for s in host.shape)
        self.image_size = util.padded_shape(self.image_size_original, self.roi_size, self.overlap)
        self.steps = util.grid_steps(self.image_size, self.roi_size, self.overlap)

    def __len__(self):
        return self.steps[0] * self.steps[1] * self.steps[2]

    def __getitem__(self, index):
        if index < 0 or index >= len(self):
            raise IndexError(index)
        E = self.roi_size + 2 * self.border_cut
        cube = torch.empty((1, E, E, E), dtype=torch.float32, device=self.device)
        L0, L1, L2 = self.image_size_original
        check(lib().nc_dice_cut_cube(P(self.volume.data_ptr()), I(1 if self.is_u16 else 0), I(L0), I(L1), I(L2),
                                     I(self.roi_size), I(self.overlap), I(self.border_cut), I(int(index)),
                                     P(cube.data_ptr()), P(torch.cuda.current_stream().cuda_stream)),
              'nc_dice_cut_cube')
        return {'A': cube, 'A_paths': str(index)}

    def shape(self):
        return self.steps

    def size(self):
        return self.image_size

    def size_original(self):
        return self.image_size_original
