"""PMC run of the 16-bit conv kernels (under `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE ...` and, in
separate passes, `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE`): the dominant launches of the 148^3 x 4 step (configs[3])."""
import sys

import torch

sys.path.insert(0, '.')
from neuroclear_amd import ops  # noqa: E402

dev = 'cuda'
ops.set_conv_precision('bf16')
N, S = 4, 148
x = torch.randn(N, 64, S, S, S, device=dev)
dy = torch.randn(N, 64, S, S, S, device=dev)
w3 = torch.randn(64, 64, 3, 3, 3, device=dev) * 0.05
w5 = torch.randn(64, 64, 5, 5, 5, device=dev) * 0.05
xh, dyh = ops.to_c8(x, 2), ops.to_c8(dy, 2)
for _ in range(2):
    ops.conv_fwd_raw(x, w3, None, 1, 1, xh=xh)
    ops.conv_wgrad_raw(None, dy, w3.shape, 1, 1, False, xh=xh, dyh=dyh, x_shape=x.shape)
    ops.conv_fwd_raw(x, w5, None, 1, 2, xh=xh)
    ops.conv_wgrad_raw(None, dy, w5.shape, 1, 2, False, xh=xh, dyh=dyh, x_shape=x.shape)
torch.cuda.synchronize()
print('done')
