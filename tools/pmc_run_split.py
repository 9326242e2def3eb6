"""PMC run of the split-operand kernels (under `rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES
SQ_WAIT_ANY SQ_BUSY_CU_CYCLES`): the dominant launches of the 108^3 step and of the 140^3 inference cube, 64 -> 64 channels."""
import sys

import torch

sys.path.insert(0, '.')
from neuroclear_amd import ops  # noqa: E402

dev = 'cuda'
ops.set_conv_split(True)
for S in (108, 140):
    x = torch.randn(1, 64, S, S, S, device=dev)
    dy = torch.randn(1, 64, S, S, S, device=dev)
    w3 = torch.randn(64, 64, 3, 3, 3, device=dev) * 0.05
    w5 = torch.randn(64, 64, 5, 5, 5, device=dev) * 0.05
    for _ in range(2):
        ops.conv_fwd_raw(x, w3, None, 1, 1)
        ops.conv_wgrad_raw(x, dy, w3.shape, 1, 1, False)
        if S == 108:
            ops.conv_fwd_raw(x, w5, None, 1, 2)
            ops.conv_wgrad_raw(x, dy, w5.shape, 1, 2, False)
    torch.cuda.synchronize()
print('done')
