"""Time and check the split-operand weight gradient (pre-split operands) at the step's layer shapes.  env NC_S3X_WGRAD=0: the 32x32x16 kernel."""
import os
import sys

import torch

sys.path.insert(0, '.')
from tools.split_conv import timeit, to_s3, wgrad_split  # noqa: E402

dev = 'cuda'
torch.manual_seed(0)
tag = 'NC_S3X_WGRAD=%s' % os.environ.get('NC_S3X_WGRAD', '1')
for ks, n in ((3, (10, 30, 108)), (5, (7, 22, 54))):
    x = torch.randn(1, 64, *n, device=dev)
    dy = torch.randn(1, 64, *n, device=dev)
    ref = torch.nn.grad.conv3d_weight(x.double(), (64, 64, ks, ks, ks), dy.double(), padding=ks // 2)
    dw = wgrad_split(x, dy, ks)
    e = dw.double() - ref
    sc = ref.pow(2).mean().sqrt().item()
    print('[%s] ks %d err vs fp64: max %.2e rms %.2e' % (tag, ks, e.abs().max().item() / sc, e.pow(2).mean().sqrt().item() / sc))
tot = 0.0
for name, C, K, E, ks in [('64->64', 64, 64, 108, 3), ('128->64', 128, 64, 108, 3), ('128->128 /2', 128, 128, 54, 3), ('256->128 /2', 256, 128, 54, 3),
                          ('256->256 /4', 256, 256, 27, 3), ('5^3 64->64', 64, 64, 108, 5)]:
    x = torch.randn(1, C, E, E, E, device=dev)
    dy = torch.randn(1, K, E, E, E, device=dev)
    xs, dys = to_s3(x), to_s3(dy)
    t = timeit(lambda: wgrad_split(x, dy, ks, xs, dys), iters=10, warm=3)
    gf = 2.0 * ks ** 3 * C * K * E ** 3 / 1e9
    tot += t
    print('[%s] %-12s %.3f ms  %.0f TF' % (tag, name, t, gf / t))
print('[%s] total %.3f ms' % (tag, tot))
