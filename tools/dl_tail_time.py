"""deep_linear_gen forward + backward at 108^3, 12 times (run under rocprofv3 --kernel-trace --stats to see the collapsed tail's kernels:
k_conv_to1_k3, k_dl_*, k_conv_c1k3, k_wgrad_c1<3>).  usage: python tools/dl_tail_time.py [size]"""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from neuroclear_amd.models import networks
from neuroclear_amd.util import seed as S
E = int(sys.argv[1]) if len(sys.argv) > 1 else 108
net = networks.define_G(1, 1, 64, 'deep_linear_gen', 'instance', False, 'kaiming', 0.02, [0])
net.load_state_dict(S.state_dict_from_seed(S.deep_linear_spec(), 3, 'cuda'))
x = torch.rand(1, 1, E, E, E, device='cuda').requires_grad_(True)
for it in range(12):
    if it == 2:
        torch.cuda.synchronize(); t0 = time.time()
    y = net(x)
    y.mean().backward()
torch.cuda.synchronize()
print('deep_linear fwd + bwd at %d^3: %.3f ms' % (E, (time.time() - t0) / 10 * 1e3))
