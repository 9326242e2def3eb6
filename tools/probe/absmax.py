"""Time the measuring pass of the two-term form (k_absmax) alone: a layer-level forward with and without it is not separable, so this
calls nc_conv_fwd on a 64 -> 64 layer under rocprofv3 --stats and reads k_absmax's average from the CSV.  usage: rocprofv3 ... -- python tools/probe/absmax.py"""
import sys
import torch
sys.path.insert(0, '.')
from neuroclear_amd import ops
x = torch.randn(1, 64, 108, 108, 108, device='cuda')
w = torch.randn(64, 64, 3, 3, 3, device='cuda') * 0.02
for _ in range(10):
    ops.conv_fwd_raw(x, w, None, 1, 1)
torch.cuda.synchronize()
