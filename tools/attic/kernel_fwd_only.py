"""PMC target: forward MFMA conv launches only (3^3 64->64, 128->64 and 5^3 64->64 at 108^3), two launches each."""
import sys

import torch

sys.path.insert(0, '.')
from neuroclear_amd import ops  # noqa: E402

dev = 'cuda'
for C, K, k, p in ((64, 64, 3, 1), (128, 64, 3, 1), (64, 64, 5, 2)):
    x = torch.randn(1, C, 108, 108, 108, device=dev)
    w = torch.randn(K, C, k, k, k, device=dev) * 0.05
    for _ in range(2):
        ops.conv_fwd_raw(x, w, None, 1, p)
torch.cuda.synchronize()
print('done')
