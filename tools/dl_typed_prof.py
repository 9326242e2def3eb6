import sys, torch
sys.path.insert(0, '.')
from neuroclear_amd._lib import lib, I
from neuroclear_amd.models import networks
L = lib()
torch.manual_seed(3)
net = networks.define_G(1, 1, 64, 'deep_linear_gen', 'instance', False, 'kaiming', 0.02, [0])
x = torch.rand(1, 1, 108, 108, 108, device='cuda'); r = torch.randn(1, 1, 108, 108, 108, device='cuda')
L.nc_set_dl_collapse(I(int(sys.argv[1]) if len(sys.argv) > 1 else 2))
for _ in range(6):
    for q in net.parameters(): q.grad = None
    xi = x.clone().requires_grad_(True)
    (net(xi) * r).sum().backward()
torch.cuda.synchronize()
