"""Time and check the 16-bit forward / data-gradient convolution (nc_conv_fwd_c8 / nc_conv_dgrad_c8: C8 in, C8 out) at configs[3] shapes."""
import ctypes
import sys

import torch

sys.path.insert(0, '.')
from neuroclear_amd import _lib, ops  # noqa: E402

L = _lib.lib()
dev = 'cuda'
ops.set_conv_precision('bf16')
P = lambda t: ctypes.c_void_p(t.data_ptr())
tag = sys.argv[1] if len(sys.argv) > 1 else ''


def ws_for(N, C, K, E, ks):
    return torch.empty(L.nc_conv_lp_ws_bytes(N, C, E[0], E[1], E[2], K, ks, ks, ks, 1, ks // 2) + 256, dtype=torch.uint8, device=dev)


def fwd(xh, w, b, N, C, K, E, ks, dt, ws):
    V = E[0] * E[1] * E[2]
    yh = torch.empty(N * K * V * 2, dtype=torch.uint8, device=dev)
    e = L.nc_conv_fwd_c8(P(xh), P(w), P(b) if b is not None else None, P(yh), K, 0, N, C, E[0], E[1], E[2], K, ks, ks, ks, 1, ks // 2, dt, P(ws), ctypes.c_size_t(ws.numel()), None)
    assert e == 0, e
    return yh


def dgrad(dyh, w, N, C, K, E, ks, dt, ws):
    V = E[0] * E[1] * E[2]
    dxh = torch.empty(N * C * V * 2, dtype=torch.uint8, device=dev)
    e = L.nc_conv_dgrad_c8(P(dyh), P(w), P(dxh), N, C, E[0], E[1], E[2], K, ks, ks, ks, 1, ks // 2, dt, P(ws), ctypes.c_size_t(ws.numel()), None)
    assert e == 0, e
    return dxh


def from_c8(h, N, C, E, dt):
    V = E[0] * E[1] * E[2]
    y = torch.empty(N, C, *E, device=dev)
    assert L.nc_from_c8(P(h), C, 0, P(y), N, C, ctypes.c_long(V), dt, None) == 0
    return y


torch.manual_seed(0)
for dt, cast in ((2, torch.bfloat16), (1, torch.float16)):
    for ks, E, C, K in ((3, (9, 30, 37), 64, 64), (5, (7, 22, 54), 64, 64), (3, (6, 37, 37), 128, 128)):
        N = 2
        x = torch.randn(N, C, *E, device=dev)
        w = torch.randn(K, C, ks, ks, ks, device=dev) * 0.05
        b = torch.randn(K, device=dev)
        ws = ws_for(N, C, K, E, ks)
        ref = torch.nn.functional.conv3d(x.to(cast).double(), w.to(cast).double(), b.double(), padding=ks // 2)
        y = from_c8(fwd(ops.to_c8(x, dt), w, b, N, C, K, E, ks, dt, ws), N, K, E, dt)
        sc = ref.pow(2).mean().sqrt().item()
        e1 = (y.double() - ref).abs().max().item() / sc
        dy = torch.randn(N, K, *E, device=dev)
        refd = torch.nn.grad.conv3d_input((N, C, *E), w.to(cast).double(), dy.to(cast).double(), padding=ks // 2)
        dx = from_c8(dgrad(ops.to_c8(dy, dt), w, N, C, K, E, ks, dt, ws), N, C, E, dt)
        e2 = (dx.double() - refd).abs().max().item() / refd.pow(2).mean().sqrt().item()
        print('[%s] dt %d ks %d %s %d->%d: fwd max err %.2e dgrad %.2e (16-bit output rounding: bf16 ~4e-3 x |y|max/rms, f16 ~5e-4 x)' % (tag, dt, ks, E, C, K, e1, e2))
tot = 0.0
for nm, N, C, K, S, ks in [('64->64 148^3x4', 4, 64, 64, 148, 3), ('128->64 148^3x4', 4, 128, 64, 148, 3), ('128->128 74^3x4', 4, 128, 128, 74, 3),
                           ('256->256 37^3x4', 4, 256, 256, 37, 3), ('5^3 64->64 148^3x4', 4, 64, 64, 148, 5)]:
    E = (S, S, S)
    xh = ops.to_c8(torch.randn(N, C, *E, device=dev), 2)
    w = torch.randn(K, C, ks, ks, ks, device=dev) * 0.05
    ws = ws_for(N, C, K, E, ks)
    for _ in range(3):
        fwd(xh, w, None, N, C, K, E, ks, 2, ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fwd(xh, w, None, N, C, K, E, ks, 2, ws)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10
    tot += t
    print('[%s] fwd %-20s %.3f ms  %.0f TF' % (tag, nm, t, 2.0 * ks ** 3 * C * K * N * S ** 3 / 1e9 / t))
    del xh, ws
print('[%s] total %.3f ms' % (tag, tot))
