"""SingleVolumeDataset on the device (reference: data/singlevolume_dataset.py:20-55 with the crop / flip / normalise
part of data/base_dataset.py:87-143,187-240,279-301).

The volume is uploaded once as uint16/uint8 and every `__getitem__` cuts a fresh random crop ON THE GPU (the
reference transforms the whole volume on the host for every iteration -- 1.67 s of its 3.64 s iteration).
Random draws follow the reference's order: `random.randint` x3 for the crop origin (base_dataset.py:195-197), then
`random.shuffle` + `np.random.uniform` x3 for the flips (:279-289).  `random3Drotate` / `random90rotate` (per-slice
rotation + inscribed-rectangle crop of the whole volume, base_dataset.py:306-460) are folded into the crop: the
rotation angle is drawn first (`random.randint(0, 359)` / `np.random.choice` of the six right angles), the crop origin
is drawn against the shape the rotated volume would have, and nc_rotate_crop samples only the crop's voxels."""
import os
import random

import numpy as np
import torch

from .._lib import I, P, check, lib
from . import rotation
from .diceImage_dataset import _load_volume


class SingleVolumeDataset:
    def __init__(self, opt, volume=None):
        self.opt = opt
        self.rot3d = 'random3Drotate' in opt.preprocess
        self.rot90 = 'random90rotate' in opt.preprocess
        if self.rot3d and self.rot90:
            raise NotImplementedError('random3Drotate and random90rotate together (two successive whole-volume '
                                      'rotations) are not fused; use one of them')
        if (self.rot3d or self.rot90) and 'randomcrop' not in opt.preprocess:
            raise NotImplementedError('rotation augmentation is fused with randomcrop: add randomcrop to --preprocess')
        if volume is None:
            names = sorted(f for f in os.listdir(opt.dataroot) if f.endswith(('.npy', '.tif', '.tiff')))
            self.A_path = os.path.join(opt.dataroot, names[0])
            volume = _load_volume(self.A_path)
        else:
            self.A_path = 'memory'
        if volume.dtype not in (np.uint8, np.uint16):
            raise TypeError('input volume must be uint8 or uint16 (data/base_dataset.py:134-143)')
        self.den = 255.0 if volume.dtype == np.uint8 else 65535.0
        self.device = torch.device('cuda', opt.gpu_ids[0])
        host = torch.from_numpy(volume.astype(np.int32))  # torch has no uint16 arithmetic
        self.volume = host.to(self.device)
        # the rotation kernel reads the original integer type (as nc_dice_cut_cube does)
        self.is_u16 = volume.dtype == np.uint16
        self.raw = torch.from_numpy(np.ascontiguousarray(volume).view(np.uint8)).to(self.device) \
            if (self.rot3d or self.rot90) else None
        self.isTrain = opt.isTrain

    def __len__(self):
        return 10  # singlevolume_dataset.py:55

    def _rotated_crop(self):
        """__randomrotate_clean_3D_xy / __random90rotate (base_dataset.py:144-151, 455-460) + __randomcrop (:187-206) +
        __normalize (:134-143) in one kernel launch."""
        D, H, W = self.volume.shape
        angle = random.randint(0, 359) if self.rot3d else int(np.random.choice((-90, 90, -180, 180, -270, 270)))
        inv, (x1, y1, x2, y2) = rotation.rotate_clean_plan(H, W, angle)
        cz, cy, cx = self.opt.crop_size
        if min(cz, cy, cx) < 1:
            raise NotImplementedError('rotation augmentation needs a positive --crop_size on every axis')
        rh, rw = y2 - y1, x2 - x1  # shape of the rotated, cleaned slices the reference would crop from
        assert D - cz >= 0 and rh - cy >= 0 and rw - cx >= 0
        z = random.randint(0, D - cz)
        y = random.randint(0, rh - cy)
        x = random.randint(0, rw - cx)
        out = torch.empty((cz, cy, cx), dtype=torch.float32, device=self.device)
        m = inv.ctypes.data_as(P)
        check(lib().nc_rotate_crop(P(self.raw.data_ptr()), I(1 if self.is_u16 else 0), I(D), I(H), I(W), I(z), I(y1 + y),
                                   I(x1 + x), I(cz), I(cy), I(cx), m, P(out.data_ptr()),
                                   P(torch.cuda.current_stream().cuda_stream)), 'nc_rotate_crop')
        return out

    def __getitem__(self, index):
        v = self.volume
        if self.rot3d or self.rot90:
            a = self._rotated_crop()
        elif 'randomcrop' in self.opt.preprocess:
            cz, cy, cx = self.opt.crop_size
            assert v.shape[0] >= cz and v.shape[1] >= cy and v.shape[2] >= cx
            z = random.randint(0, v.shape[0] - cz)
            y = random.randint(0, v.shape[1] - cy)
            x = random.randint(0, v.shape[2] - cx)
            # crop size 0 on an axis = keep the full axis: the draw still happens (RNG order), the origin is 0
            # (data/base_dataset.py:199-214)
            z, y, x = (z if cz else 0), (y if cy else 0), (x if cx else 0)
            v = v[z:z + cz if cz else None, y:y + cy if cy else None, x:x + cx if cx else None]
        if not (self.rot3d or self.rot90):
            a = (v.to(torch.float64) / self.den).to(torch.float32)  # __normalize: float64 division, then .float()
        if 'randomflip' in self.opt.preprocess:
            axis_list = [0, 1, 2]
            random.shuffle(axis_list)
            for _ in range(3):
                if np.random.uniform(0, 1) < 0.5:
                    a = torch.flip(a, (axis_list.pop(),))
        if 'addColorChannel' in self.opt.preprocess:
            a = a.unsqueeze(0)
        if 'addBatchChannel' in self.opt.preprocess:
            a = a.unsqueeze(0)
        return {'A': a.contiguous(), 'A_paths': self.A_path}
