"""SingleVolumeDataset on the device (reference: data/singlevolume_dataset.py:20-55 with the crop / flip / normalise
part of data/base_dataset.py:87-143,187-240,279-301).

The volume is uploaded once as uint16/uint8 and every `__getitem__` cuts a fresh random crop ON THE GPU (the
reference transforms the whole volume on the host for every iteration -- 1.67 s of its 3.64 s iteration).
Random draws follow the reference's order: `random.randint` x3 for the crop origin (base_dataset.py:195-197), then
`random.shuffle` + `np.random.uniform` x3 for the flips (:279-289).  `random3Drotate` (cv2 per-slice rotation,
base_dataset.py:306-460) is a next-row item and raises."""
import os
import random

import numpy as np
import torch

from .diceImage_dataset import _load_volume


class SingleVolumeDataset:
    def __init__(self, opt, volume=None):
        self.opt = opt
        if 'random3Drotate' in opt.preprocess or 'random90rotate' in opt.preprocess:
            raise NotImplementedError('rotation augmentation is outside the MI355X hot path (SURVEY.md 8f rank 1); '
                                      'use --preprocess randomcrop_randomflip_addColorChannel_addBatchChannel')
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
        self.isTrain = opt.isTrain

    def __len__(self):
        return 10  # singlevolume_dataset.py:55

    def __getitem__(self, index):
        v = self.volume
        if 'randomcrop' in self.opt.preprocess:
            cz, cy, cx = self.opt.crop_size
            assert v.shape[0] >= cz and v.shape[1] >= cy and v.shape[2] >= cx
            z = random.randint(0, v.shape[0] - cz)
            y = random.randint(0, v.shape[1] - cy)
            x = random.randint(0, v.shape[2] - cx)
            v = v[z:z + cz if cz else None, y:y + cy if cy else None, x:x + cx if cx else None]
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
