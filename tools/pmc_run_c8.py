"""PMC / timing run of the 16-bit end-to-end kernels at the configs[3] shape (64 -> 64 channels, 4 x 148^3): the C8-in /
C8-out 3^3 and 5^3 forward (nc_conv_fwd_c8), their data gradient, the weight gradients and one InstanceNorm forward /
backward on C8.  Run under `rocprofv3 --kernel-trace --stats`, `--pmc FETCH_SIZE`, `--pmc WRITE_SIZE` (separate passes)."""
import ctypes
import sys

import torch

sys.path.insert(0, '.')
from neuroclear_amd import _lib, ops  # noqa: E402

L = _lib.lib()
dev = 'cuda'
ops.set_conv_precision('bf16')
N, S, C = 4, 148, 64
V = S ** 3
x = torch.randn(N, C, S, S, S, device=dev)
xh = ops.to_c8(x, 2)
del x
dyh = ops.to_c8(torch.randn(N, C, S, S, S, device=dev), 2)
w3 = torch.randn(64, 64, 3, 3, 3, device=dev) * 0.05
w5 = torch.randn(64, 64, 5, 5, 5, device=dev) * 0.05
yh = torch.empty(N * C * V * 2, dtype=torch.uint8, device=dev)
dw3, dw5 = torch.empty_like(w3), torch.empty_like(w5)
mean = torch.empty(N * C, device=dev)
rstd = torch.empty(N * C, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())


def ws_for(ks):
    return torch.empty(L.nc_conv_lp_ws_bytes(N, C, S, S, S, C, ks, ks, ks, 1, ks // 2) + 256, dtype=torch.uint8, device=dev)


ws3, ws5 = ws_for(3), ws_for(5)
wsn = torch.empty(L.nc_c8_instnorm_ws_bytes(N, C, ctypes.c_long(V)) + 256, dtype=torch.uint8, device=dev)
for _ in range(3):
    for w, ws, ks in ((w3, ws3, 3), (w5, ws5, 5)):
        assert L.nc_conv_fwd_c8(P(xh), P(w), None, P(yh), C, 0, N, C, S, S, S, C, ks, ks, ks, 1, ks // 2, 2, P(ws), ctypes.c_size_t(ws.numel()), None) == 0
        assert L.nc_conv_dgrad_c8(P(dyh), P(w), P(yh), N, C, S, S, S, C, ks, ks, ks, 1, ks // 2, 2, P(ws), ctypes.c_size_t(ws.numel()), None) == 0
    assert L.nc_conv_wgrad_lp(None, P(xh), None, P(dyh), P(dw3), None, N, C, S, S, S, C, 3, 3, 3, 1, 1, 2, P(ws3), ctypes.c_size_t(ws3.numel()), None) == 0
    assert L.nc_conv_wgrad_lp(None, P(xh), None, P(dyh), P(dw5), None, N, C, S, S, S, C, 5, 5, 5, 1, 2, 2, P(ws5), ctypes.c_size_t(ws5.numel()), None) == 0
    assert L.nc_c8_instnorm_stats(P(xh), N, C, ctypes.c_long(V), ctypes.c_float(1e-5), P(mean), P(rstd), 2, P(wsn), ctypes.c_size_t(wsn.numel()), None) == 0
    assert L.nc_c8_instnorm_act_fwd(P(xh), P(mean), P(rstd), ctypes.c_float(0.0), P(yh), C, 0, N, C, ctypes.c_long(V), 2, None) == 0
    assert L.nc_c8_instnorm_act_bwd(P(dyh), C, 0, P(xh), P(mean), P(rstd), ctypes.c_float(0.0), P(yh), None, N, C, ctypes.c_long(V), 2, P(wsn), ctypes.c_size_t(wsn.numel()), None) == 0
torch.cuda.synchronize()
print('done')
