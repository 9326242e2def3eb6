"""smoke()'s gradient comparison (16^3 unet_deconv, GPU against the CPU oracle) over several inputs and both values of nc_set_s3x_w64: isolated ReLU /
max-pool decision flips show as occasional 1e-3 spikes of the L2 error in BOTH modes; a systematic error would show in every seed of one mode."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from neuroclear_amd.models import networks
from neuroclear_amd.util import seed as S
from neuroclear_amd._lib import lib
from oracle import nets as onets
dev = 'cuda:0'
sd_np = S.weights_from_seed(S.unet_deconv_spec(), 3)
net = networks.define_G(1, 1, 64, 'unet_deconv', 'instance', False, 'kaiming', 0.02, [0])
net.load_state_dict(S.state_dict_from_seed(S.unet_deconv_spec(), 3, dev))
for seed in range(5, 17):
    x_np = np.random.default_rng(seed).random((1, 1, 16, 16, 16), dtype=np.float32)
    sd = onets.to_torch(sd_np, requires_grad=True)
    xo = torch.from_numpy(x_np).requires_grad_(True)
    onets.unet_deconv(sd, xo).mean().backward()
    sd64 = {k: v.double().detach().requires_grad_(True) for k, v in onets.to_torch(sd_np).items()}
    x64 = torch.from_numpy(x_np).double().requires_grad_(True)
    onets.unet_deconv(sd64, x64).mean().backward()
    out = []
    for on in (0, 1):
        lib().nc_set_s3x_w64(on)
        x = torch.from_numpy(x_np).to(dev).requires_grad_(True)
        net(x).mean().backward()
        g = x.grad.cpu()
        out.append('w64=%d vs fp32 oracle %.2e vs fp64 oracle %.2e' % (on, float((g - xo.grad).norm() / xo.grad.norm()), float((g.double() - x64.grad).norm() / x64.grad.norm())))
    print('seed %2d: %s | %s | fp32 oracle vs fp64 oracle %.2e' % (seed, out[0], out[1], float((xo.grad.double() - x64.grad).norm() / x64.grad.norm())), flush=True)
