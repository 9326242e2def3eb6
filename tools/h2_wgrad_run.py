"""The two-term and three-term weight gradient at 64 -> 64 / 108^3 (3^3 and 5^3), a few launches each: run under rocprofv3 (--stats or --pmc)
to read k_wgrad_s3x<KS, NT, dtype>'s duration and counters.  usage: rocprofv3 ... -- python3 tools/h2_wgrad_run.py"""
import sys
import torch
sys.path.insert(0, '.')
from neuroclear_amd import ops
from neuroclear_amd._lib import lib
g = torch.Generator(device='cuda').manual_seed(2)
x = torch.randn(1, 64, 108, 108, 108, device='cuda', generator=g).clamp_min(0)
dy = torch.randn(1, 64, 108, 108, 108, device='cuda', generator=g) * 1e-3
for terms in (3, 2):
    lib().nc_set_split_terms(terms)
    for ks in (3, 5):
        for _ in range(6):
            ops.conv_wgrad_raw(x, dy, (64, 64, ks, ks, ks), 1, ks // 2, False)
torch.cuda.synchronize()
