"""Time the 16-bit forward / data-gradient kernels on the layer shapes of configs[3] (4 x 148^3 crops): k_conv_h (nc_set_c8x_mode(0))
against k_conv_c8x (mode 2), HIP events around `reps` back-to-back launches.  usage: python tools/c8x_time.py [reps [number of shapes [modes, e.g. 2 or 0,2]]]   (NC_HIP_LIB=... selects a variant build)"""
import ctypes
import sys

import torch

sys.path.insert(0, '.')
from neuroclear_amd import _lib, ops  # noqa: E402

L = _lib.lib()
dev = 'cuda'
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
nshapes = int(sys.argv[2]) if len(sys.argv) > 2 else 99
modes = [int(m) for m in sys.argv[3].split(',')] if len(sys.argv) > 3 else [0, 2]
P = lambda t: ctypes.c_void_p(t.data_ptr())
SHAPES = [(4, 64, 64, 148, 3), (4, 64, 64, 148, 5), (4, 128, 64, 148, 3), (4, 64, 128, 74, 3), (4, 128, 128, 74, 3), (4, 256, 128, 74, 3),
          (4, 128, 256, 37, 3), (4, 256, 256, 37, 3), (1, 64, 64, 108, 3)]
for N, C, K, E, ks in SHAPES[:nshapes]:
    V = E ** 3
    xh = ops.to_c8(torch.randn(N, C, E, E, E, device=dev), 2)
    w = torch.randn(K, C, ks, ks, ks, device=dev) * 0.05
    yh = torch.empty(N * K * V * 2, dtype=torch.uint8, device=dev)
    ws = torch.empty(L.nc_conv_lp_ws_bytes(N, C, E, E, E, K, ks, ks, ks, 1, ks // 2) + 256, dtype=torch.uint8, device=dev)
    flop = 2.0 * N * V * C * K * ks ** 3
    line = '%d x %3d^3 %3d->%3d k%d:' % (N, E, C, K, ks)
    for mode in modes:
        L.nc_set_c8x_mode(mode)
        def run():
            assert L.nc_conv_fwd_c8(P(xh), P(w), None, P(yh), K, 0, N, C, E, E, E, K, ks, ks, ks, 1, ks // 2, 2, P(ws), ctypes.c_size_t(ws.numel()), None) == 0
        run(); run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        line += '  mode %d %.3f ms %.0f TFLOP/s (uses_c8x %d)' % (mode, ms, flop / ms / 1e9, L.nc_conv_lp_uses_c8x(0, 0, N, C, E, E, E, K, ks))
    print(line, flush=True)
    del xh, w, yh, ws
