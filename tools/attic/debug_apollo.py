import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from neuroclear_amd import ops
from neuroclear_amd.models import networks
from neuroclear_amd.util import seed as S
from oracle import nets as onets

dev = 'cuda'
sdn = S.weights_from_seed(S.patchgan_spec(2), 44)
net = networks.define_D(1, 64, 'basic', 3, 'instance', 'kaiming', 0.02, False, [0], dimension=2)
net.load_state_dict(S.state_dict_from_seed(S.patchgan_spec(2), 44, dev))
vol = torch.rand(1, 1, 36, 36, 36, generator=torch.Generator().manual_seed(1)) * 3 - 1
for axis in range(3):
    idx = 5 + axis
    sl = vol.select(axis + 2, idx)
    mine_sl = ops.volume_slice(vol.to(dev), axis, idx)
    print('axis', axis, 'slice equal', torch.equal(mine_sl.cpu(), sl.contiguous()), tuple(sl.shape), tuple(mine_sl.shape))
    yo = onets.patchgan(onets.to_torch(sdn), sl)
    ym = net(mine_sl)
    ym2 = net(sl.contiguous().to(dev))
    print('  D out oracle', yo.flatten()[:4].tolist(), 'mine', ym.flatten()[:4].tolist(), 'mine(torch slice)', ym2.flatten()[:4].tolist())
