"""PMC traffic run (under `rocprofv3 --pmc FETCH_SIZE` and, separately, `--pmc WRITE_SIZE`): calibration kernels with a
known byte count in the SAME access patterns the conv kernels use (4-byte-per-lane loads / stores, and the 16-byte
pattern for reference), then the dominant convolution launches of the 108^3 step."""
import sys

import torch

sys.path.insert(0, '.')
from neuroclear_amd import ops  # noqa: E402

dev = 'cuda'
x = torch.randn(1, 64, 108, 108, 108, device=dev)
for _ in range(2):
    ops.leaky_relu(x, 0.2)            # k_lrelu_fwd: dword loads + dword stores, 322.5 MB each way
    ops.instance_norm_act(x, 0.0)     # k_in_stats (float4 read 322.5 MB), k_in_act_fwd (float4 read + write)
w3 = torch.randn(64, 64, 3, 3, 3, device=dev) * 0.05
w5 = torch.randn(64, 64, 5, 5, 5, device=dev) * 0.05
x128 = torch.randn(1, 128, 108, 108, 108, device=dev)
w128 = torch.randn(64, 128, 3, 3, 3, device=dev) * 0.05
for _ in range(2):
    y = ops.conv_fwd_raw(x, w3, None, 1, 1)
    ops.conv_wgrad_raw(x, y, w3.shape, 1, 1, False)
    ops.conv_fwd_raw(x, w5, None, 1, 2)
    ops.conv_wgrad_raw(x, y, w5.shape, 1, 2, False)
    y2 = ops.conv_fwd_raw(x128, w128, None, 1, 1)
    ops.conv_wgrad_raw(x128, y2, w128.shape, 1, 1, False)
torch.cuda.synchronize()
print('done')
