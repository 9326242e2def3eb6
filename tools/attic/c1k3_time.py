import sys, torch
sys.path.insert(0, '.')
from neuroclear_amd import ops
import os
for S in (108, 140):
    x = torch.randn(1, 1, S, S, S, device='cuda'); w = torch.randn(64, 1, 3, 3, 3, device='cuda'); b = torch.randn(64, device='cuda')
    for _ in range(3): ops.conv_fwd_raw(x, w, b, 1, 1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.conv_fwd_raw(x, w, b, 1, 1)
    e1.record(); torch.cuda.synchronize()
    print('NC_C1K3=%s S=%d %.3f ms' % (os.environ.get('NC_C1K3', '1'), S, e0.elapsed_time(e1) / 20))
