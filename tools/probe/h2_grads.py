"""Per-parameter difference of the whole-network training calls between the two-term and the three-term form (debugging aid)."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from neuroclear_amd.models import networks
from neuroclear_amd.util import seed as S
from neuroclear_amd._lib import lib
size = int(sys.argv[1]) if len(sys.argv) > 1 else 32
net = networks.define_G(1, 1, 64, 'unet_deconv', 'instance', False, 'kaiming', 0.02, [0])
net.load_state_dict(S.state_dict_from_seed(S.unet_deconv_spec(), 5, 'cuda'))
x0 = torch.rand(1, 1, size, size, size, device='cuda')
r = torch.rand(1, 1, size, size, size, device='cuda')
res = {}
for terms in (3, 2):
    lib().nc_set_split_terms(terms)
    for p in net.parameters():
        p.grad = None
    x = x0.clone().requires_grad_(True)
    y = net(x)
    (y * r).mean().backward()
    res[terms] = (y.detach().clone(), x.grad.clone(), {k: p.grad.clone() for k, p in net.named_parameters()})
print('y', float((res[2][0] - res[3][0]).abs().max()), 'dx rel', float((res[2][1] - res[3][1]).norm() / res[3][1].norm()))
for k in res[3][2]:
    a, b = res[2][2][k], res[3][2][k]
    if a.dim() > 1:
        print('%-40s rel %.2e' % (k, float((a - b).norm() / b.norm())))
