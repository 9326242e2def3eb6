import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from neuroclear_amd import ops
from neuroclear_amd.models import networks
from neuroclear_amd.util import seed as S
from oracle import nets as onets

dev = 'cuda'
size = int(sys.argv[1]) if len(sys.argv) > 1 else 32
spec = S.unet_deconv_spec()
net = networks.define_G(1, 1, 64, 'unet_deconv', 'instance', False, 'kaiming', 0.02, [0])
net.load_state_dict(S.state_dict_from_seed(spec, 2, dev))
x_np = np.random.default_rng(102).random((1, 1, size, size, size), dtype=np.float32)
r_np = np.random.default_rng(202).random((1, 1, size, size, size), dtype=np.float32)
for force in (False, True):
    ops.set_force_direct(force)
    net.zero_grad()
    x = torch.from_numpy(x_np).to(dev).requires_grad_(True)
    y = net(x)
    (y * torch.from_numpy(r_np).to(dev)).mean().backward()
    ops.set_force_direct(False)
    # torch (MIOpen) reference in fp64 on the GPU with the same weights
    sd = {k: v.detach().double().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    xr = torch.from_numpy(x_np).to(dev).double().requires_grad_(True)
    yr = onets.unet_deconv(sd, xr)
    (yr * torch.from_numpy(r_np).to(dev).double()).mean().backward()
    print('force_direct', force, 'y maxerr', float((y.double() - yr).abs().max()),
          'dx rel2', float((x.grad.double() - xr.grad).norm() / xr.grad.norm()),
          'dx relmax', float((x.grad.double() - xr.grad).abs().max() / xr.grad.abs().max()))
    for (k, p) in net.named_parameters():
        g, gr = p.grad.double(), sd[k].grad
        print('   %-40s rel2 %.2e  |g| %.2e' % (k, float((g - gr).norm() / gr.norm().clamp_min(1e-30)), float(gr.norm())))
