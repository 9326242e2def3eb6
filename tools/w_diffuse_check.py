"""Experiment: systematic output error of deep_linear_gen's 16-bit whole-network forward on a nearly constant input, relative
to the un-cancelled magnitude A (the same network with |weights|), for a few weight seeds.  NC_W_DIFFUSE=0: round-to-nearest."""
import sys
import torch
sys.path.insert(0, '.')
from neuroclear_amd import ops
from neuroclear_amd.models import networks

for seed in (1, 2, 3, 21, 22):
    torch.manual_seed(seed)
    net = networks.define_G(1, 1, 64, 'deep_linear_gen', 'instance', False, 'kaiming', 0.02, [0])
    x = 0.5 + 0.0116 * torch.randn(2, 1, 40, 40, 40, device='cuda')
    with torch.enable_grad():
        y32 = net(x).detach()
        ops.set_conv_precision('bf16')
        y16 = net(x).detach()
        ops.set_conv_precision('fp32')
        saved = [p.detach().clone() for p in net.parameters()]
        for p in net.parameters():
            p.data.abs_()
        A = float(net(x).detach().mean())
        for p, s in zip(net.parameters(), saved):
            p.data.copy_(s)
    d = (y16 - y32).double()
    print('seed %2d  mean y32 %+.4f  A %.3f  mean diff %+.5f (%.2e A)  rms diff %.5f (%.2e A)' % (
        seed, float(y32.mean()), A, float(d.mean()), abs(float(d.mean())) / A, float(d.pow(2).mean().sqrt()), float(d.pow(2).mean().sqrt()) / A))
