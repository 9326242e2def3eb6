"""PMC run: the 256 -> 512 PatchGAN layer at 216 planes, forward and data gradient on k_sconv (run under rocprofv3 --pmc)."""
import sys
import torch
sys.path.insert(0, '.')
from neuroclear_amd import ops

x = torch.randn(216, 256, 13, 13, device='cuda')
w = torch.randn(512, 256, 4, 4, device='cuda') * 0.02
y = ops.conv_fwd_raw(x, w, None, 1, 1)
dy = torch.randn_like(y)
for _ in range(3):
    ops.conv_fwd_raw(x, w, None, 1, 1)
    ops.conv_dgrad_raw(dy, w, x.shape, 1, 1)
    ops.conv_wgrad_raw(x, dy, w.shape, 1, 1, False)
torch.cuda.synchronize()
print('done')
