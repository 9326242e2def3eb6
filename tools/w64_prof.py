"""Kernel-only durations of the 3^3 two-term forward at the step's / the cube's largest shapes (run under rocprofv3 --kernel-trace --stats)."""
import sys
import torch
sys.path.insert(0, '.')
from neuroclear_amd import ops
from neuroclear_amd._lib import lib, I
lib().nc_set_split_terms(I(2))
g = torch.Generator(device='cuda').manual_seed(3)
for C, K, E in ((64, 64, 108), (128, 64, 108), (64, 64, 140), (128, 64, 140)):
    x = torch.randn(1, C, E, E, E, device='cuda', generator=g).clamp_min(0)
    w = torch.randn(K, C, 3, 3, 3, device='cuda', generator=g) * 0.02
    for _ in range(6):
        ops.conv_fwd_raw(x, w, None, 1, 1)
    torch.cuda.synchronize()
    del x, w
