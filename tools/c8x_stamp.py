"""Per-interval cycle sums of one wave of k_conv_c8x (NC_C8X_STAMP build, tools/variant.sh conv_c8x stamp -DNC_C8X_STAMP):
   NC_HIP_LIB=neuroclear_amd/csrc/abl/libnc_hip_conv_c8x_stamp.so python tools/c8x_stamp.py N C K E ks"""
import ctypes
import sys

import torch

sys.path.insert(0, '.')
from neuroclear_amd import _lib, ops  # noqa: E402

L = _lib.lib()
N, C, K, E, ks = (int(v) for v in sys.argv[1:6])
P = lambda t: ctypes.c_void_p(t.data_ptr())
V = E ** 3
xh = ops.to_c8(torch.randn(N, C, E, E, E, device='cuda'), 2)
w = torch.randn(K, C, ks, ks, ks, device='cuda') * 0.05
yh = torch.empty(N * K * V * 2, dtype=torch.uint8, device='cuda')
nb = L.nc_conv_lp_ws_bytes(N, C, E, E, E, K, ks, ks, ks, 1, ks // 2)
ws = torch.zeros(nb + 4096, dtype=torch.uint8, device='cuda')
L.nc_set_c8x_mode(2)
for _ in range(3):
    assert L.nc_conv_fwd_c8(P(xh), P(w), None, P(yh), K, 0, N, C, E, E, E, K, ks, ks, ks, 1, ks // 2, 2, P(ws), ctypes.c_size_t(ws.numel()), None) == 0
torch.cuda.synchronize()
# the packed weights sit behind the (absent) input conversion at the start of the workspace; the stamps behind them
pk = (K // 64) * (ks ** 3 * (C // 8) // 4) * 4096
d = ws[pk:pk + 48].view(torch.int64).cpu().tolist()
n = max(d[5], 1)
names = ['loop overhead (stamp 3 -> next 0)', 'A request + vmcnt wait (0 -> 1)', 'arrival: barrier + DMA issue (1 -> 2)', 'B reads + 32 MFMA issue (2 -> 3)', 'epilogue + tile switch']
print('k-steps stamped:', d[5])
for nm, v in zip(names, d[:5]):
    print('  %-42s %10d cycles total, %7.1f per k-step' % (nm, v, v / n))
print('  sum per k-step: %.1f' % (sum(d[:5]) / n))
