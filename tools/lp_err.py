"""Development tool: deviation of the 16-bit conv path from the fp32 path on the seeded U-Net (output, per-layer weight
gradients), next to the deviation a 1e-3 input perturbation causes in pure fp32 (the noise floor of ReLU/max-pool flips)."""
import sys, torch
sys.path.insert(0, '.')
from neuroclear_amd import ops
from neuroclear_amd.models import networks
from neuroclear_amd.util import seed as S
net = networks.define_G(1, 1, 64, 'unet_deconv', 'instance', False, 'kaiming', 0.02, [0])
net.load_state_dict(S.state_dict_from_seed(S.unet_deconv_spec(), 11, device='cuda'))
size = 48
gen = torch.Generator(device='cuda').manual_seed(5)
x = torch.rand(1, 1, size, size, size, device='cuda', generator=gen)
r = torch.randn(1, 1, size, size, size, device='cuda', generator=gen)
res = {}
for prec in ('fp32', 'bf16', 'fp16', 'pert'):
    ops.set_conv_precision('fp32' if prec == 'pert' else prec)
    for p in net.parameters():
        p.grad = None
    xi = x * (1 + 1e-3 * torch.randn_like(x)) if prec == 'pert' else x
    y = net(xi)
    (y * r).mean().backward()
    res[prec] = (y.detach().clone(), {n: p.grad.clone() for n, p in net.named_parameters()})
ops.set_conv_precision('fp32')
for prec in ('bf16', 'fp16', 'pert'):
    d = (res[prec][0] - res['fp32'][0]).abs()
    print(prec, 'out max %.2e mean %.2e' % (d.max().item(), d.mean().item()))
    for n, g in res['fp32'][1].items():
        if g.dim() == 5:
            print('   %-40s |g| %.2e rel %.2e' % (n, g.norm().item(), (res[prec][1][n] - g).norm().item() / g.norm().item()))
