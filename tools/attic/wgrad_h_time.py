"""Time and check the 16-bit weight gradient (nc_conv_wgrad_lp on C8 operands) at configs[3] layer shapes.
env NC_HX_WGRAD=0: conv_h.hip's k_wgrad_h (32x32x16) instead of the 16x16x32 kernel of conv_split.hip."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, '.')
from neuroclear_amd import _lib, ops  # noqa: E402

L = _lib.lib()
dev = 'cuda'
ops.set_conv_precision('bf16')
P = lambda t: ctypes.c_void_p(t.data_ptr())
tag = 'NC_HX_WGRAD=%s' % os.environ.get('NC_HX_WGRAD', '1')


def wgrad(xh, dyh, N, C, K, E, ks, dt, ws):
    dw = torch.empty(K, C, ks, ks, ks, device=dev)
    e = L.nc_conv_wgrad_lp(None, P(xh), None, P(dyh), P(dw), None, N, C, E[0], E[1], E[2], K, ks, ks, ks, 1, ks // 2, dt, P(ws), ctypes.c_size_t(ws.numel()), None)
    assert e == 0, _lib.last_error() if hasattr(_lib, 'last_error') else e
    return dw


def ws_for(N, C, K, E, ks):
    return torch.empty(L.nc_conv_lp_ws_bytes(N, C, E[0], E[1], E[2], K, ks, ks, ks, 1, ks // 2) + 256, dtype=torch.uint8, device=dev)


torch.manual_seed(0)
for dt, name in ((2, 'bf16'), (1, 'f16')):
    for ks, E in ((3, (9, 30, 37)), (5, (7, 22, 54)), (3, (5, 148, 148))):
        N, C, K = 2, 64, 64
        x = torch.randn(N, C, *E, device=dev)
        dy = torch.randn(N, K, *E, device=dev)
        cast = torch.bfloat16 if dt == 2 else torch.float16
        xr, dyr = x.to(cast).double(), dy.to(cast).double()
        ref = torch.nn.grad.conv3d_weight(xr, (K, C, ks, ks, ks), dyr, padding=ks // 2)
        dw = wgrad(ops.to_c8(x, dt), ops.to_c8(dy, dt), N, C, K, E, ks, dt, ws_for(N, C, K, E, ks))
        e = dw.double() - ref
        sc = ref.pow(2).mean().sqrt().item()
        print('[%s] %s ks %d %s err vs fp64 of rounded operands: max %.2e rms %.2e' % (tag, name, ks, E, e.abs().max().item() / sc, e.pow(2).mean().sqrt().item() / sc))
tot = 0.0
for nm, N, C, K, S, ks in [('64->64 148^3x4', 4, 64, 64, 148, 3), ('128->64 148^3x4', 4, 128, 64, 148, 3), ('128->128 74^3x4', 4, 128, 128, 74, 3),
                           ('256->256 37^3x4', 4, 256, 256, 37, 3), ('5^3 64->64 148^3x4', 4, 64, 64, 148, 5)]:
    E = (S, S, S)
    xh = ops.to_c8(torch.randn(N, C, *E, device=dev), 2)
    dyh = ops.to_c8(torch.randn(N, K, *E, device=dev), 2)
    ws = ws_for(N, C, K, E, ks)
    for _ in range(3):
        wgrad(xh, dyh, N, C, K, E, ks, 2, ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        wgrad(xh, dyh, N, C, K, E, ks, 2, ws)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10
    tot += t
    print('[%s] %-20s %.3f ms  %.0f TF' % (tag, nm, t, 2.0 * ks ** 3 * C * K * N * S ** 3 / 1e9 / t))
    del xh, dyh, ws
print('[%s] total %.3f ms' % (tag, tot))
