"""One layer shape through nc_conv_fwd_c8, a few launches (for rocprofv3 passes).  usage: python tools/c8x_one.py N C K E ks mode [reps]"""
import ctypes
import sys

import torch

sys.path.insert(0, '.')
from neuroclear_amd import _lib, ops  # noqa: E402

L = _lib.lib()
N, C, K, E, ks, mode = (int(v) for v in sys.argv[1:7])
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 4
P = lambda t: ctypes.c_void_p(t.data_ptr())
V = E ** 3
xh = ops.to_c8(torch.randn(N, C, E, E, E, device='cuda'), 2)
w = torch.randn(K, C, ks, ks, ks, device='cuda') * 0.05
yh = torch.empty(N * K * V * 2, dtype=torch.uint8, device='cuda')
ws = torch.empty(L.nc_conv_lp_ws_bytes(N, C, E, E, E, K, ks, ks, ks, 1, ks // 2) + 256, dtype=torch.uint8, device='cuda')
L.nc_set_c8x_mode(mode)
for _ in range(reps):
    assert L.nc_conv_fwd_c8(P(xh), P(w), None, P(yh), K, 0, N, C, E, E, E, K, ks, ks, ks, 1, ks // 2, 2, P(ws), ctypes.c_size_t(ws.numel()), None) == 0
torch.cuda.synchronize()
