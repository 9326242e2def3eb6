"""64 -> 64 3^3 two-term forward at several plane shapes (one launch of whole tiles each), a few calls in a row: per-dispatch durations / counters
under rocprofv3 tell whether the time per (tile, k-step) depends on the row pitch.  usage: python tools/w64_shapes.py"""
import sys
import torch
sys.path.insert(0, '.')
from neuroclear_amd import ops
from neuroclear_amd._lib import lib, I
lib().nc_set_split_terms(I(2))
g = torch.Generator(device='cuda').manual_seed(3)
for D, H, W in ((108, 108, 108), (108, 108, 104), (108, 108, 112), (108, 108, 124), (108, 140, 140), (108, 126, 94), (108, 108, 110)):
    x = torch.randn(1, 64, D, H, W, device='cuda', generator=g).clamp_min(0)
    w = torch.randn(64, 64, 3, 3, 3, device='cuda', generator=g) * 0.02
    for _ in range(5):
        ops.conv_fwd_raw(x, w, None, 1, 1)
    torch.cuda.synchronize()
    del x, w
