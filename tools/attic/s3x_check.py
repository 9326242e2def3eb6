"""Quick parity check of the tap-stream kernel against torch fp64 on shapes that exercise each launch shape: main launch only,
tail only (third / half tiles), several tiles per workgroup, with and without bias, 3^3 and 5^3, forward and data gradient."""
import sys

import torch

sys.path.insert(0, '.')
from tools.split_conv import dgrad_split, fwd_split  # noqa: E402

dev = 'cuda'
torch.manual_seed(1)
bad = 0
for (N, C, K, n, ks, bias) in ((1, 64, 64, (20, 22, 27), 3, True), (1, 64, 64, (20, 22, 27), 3, False), (1, 64, 64, (12, 30, 108), 3, True),
                               (1, 64, 64, (44, 108, 108), 3, True), (2, 128, 64, (9, 54, 54), 3, True), (1, 64, 128, (27, 27, 27), 3, True),
                               (1, 64, 64, (11, 14, 19), 5, True), (1, 64, 64, (30, 108, 108), 5, True), (1, 256, 256, (27, 27, 27), 3, False),
                               (2, 64, 64, (7, 140, 140), 3, True)):
    x = torch.randn(N, C, *n, device=dev)
    w = torch.randn(K, C, ks, ks, ks, device=dev) * 0.03
    b = torch.randn(K, device=dev) if bias else None
    ref = torch.nn.functional.conv3d(x.double(), w.double(), b.double() if bias else None, padding=ks // 2)
    y = fwd_split(x, w, b)
    sc = ref.pow(2).mean().sqrt().item()
    e = (y.double() - ref).abs().max().item() / sc
    dy = torch.randn(N, K, *n, device=dev)
    refd = torch.nn.grad.conv3d_input(x.shape, w.double(), dy.double(), padding=ks // 2)
    dx = dgrad_split(dy, w)
    ed = (dx.double() - refd).abs().max().item() / refd.pow(2).mean().sqrt().item()
    ok = e < 1e-5 and ed < 1e-5
    bad += not ok
    print('%s N%d %d->%d %s ks%d bias=%s: fwd max err %.2e dgrad %.2e' % ('ok  ' if ok else 'FAIL', N, C, K, n, ks, bias, e, ed))
print('s3x_check: %d failure(s)' % bad)
sys.exit(1 if bad else 0)
