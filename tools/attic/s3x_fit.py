"""Per-tile fixed cost and per-k-step cost of the tap-stream kernel: 64 / 128 / 256 input channels -> 64 at 33 x 108 x 108
(1023 tiles of 384 positions = 4 rounds of 256 workgroups), pre-split input."""
import sys

import torch

sys.path.insert(0, '.')
from tools.split_conv import fwd_split, timeit, to_s3  # noqa: E402

dev = 'cuda'
res = []
for C in (64, 128, 256):
    x = torch.randn(1, C, 33, 108, 108, device=dev)
    w = torch.randn(64, C, 3, 3, 3, device=dev) * 0.05
    xs = to_s3(x)
    t = timeit(lambda: fwd_split(x, w, None, xs), iters=20, warm=5)
    res.append((C, t))
    print('C=%d: %.3f ms, per tile %.1f us' % (C, t, t * 1e3 / 4))
(c0, t0), (c1, t1), (c2, t2) = res
step = (t2 - t0) * 1e3 / 4 / (27 * (c2 - c0) / 32)
print('per k-step %.3f us; fixed per tile %.1f us (from 64/256), %.1f us (from 64/128)' % (
    step, t0 * 1e3 / 4 - step * 54, t0 * 1e3 / 4 - (t1 - t0) * 1e3 / 4 / 54 * 54))
