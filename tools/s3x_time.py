"""Time the split-operand forward kernel on pre-split inputs at the step's / the inference cube's layer shapes, and check it
against the fp64 reference on a slab.  Usage: python tools/s3x_time.py [108|140 ...]   (env: NC_S3X, NC_S3X_FLUSH, NC_S3X_TAIL)"""
import os
import sys

import torch

sys.path.insert(0, '.')
from tools.split_conv import fwd_split, timeit, to_s3  # noqa: E402

dev = 'cuda'


def main():
    torch.manual_seed(0)
    tag = ' '.join('%s=%s' % (k, os.environ[k]) for k in sorted(os.environ) if k.startswith('NC_S3'))
    # accuracy: 64 -> 64 3^3 and 5^3 at 24 x 30 x 108 against fp64
    for ks in (3, 5):
        x = torch.randn(1, 64, 12, 30, 108, device=dev)
        w = torch.randn(64, 64, ks, ks, ks, device=dev) * 0.03
        ref = torch.nn.functional.conv3d(x.double().cpu(), w.double().cpu(), None, padding=ks // 2)
        y = fwd_split(x, w, None)
        e = y.double().cpu() - ref
        sc = ref.pow(2).mean().sqrt().item()
        print('[%s] ks %d err vs fp64: max %.2e rms %.2e' % (tag, ks, e.abs().max().item() / sc, e.pow(2).mean().sqrt().item() / sc))
    sizes = [int(a) for a in sys.argv[1:]] or [108]
    tot = 0.0
    for S in sizes:
        for name, C, K, E, ks in [('64->64', 64, 64, S, 3), ('128->64', 128, 64, S, 3), ('128->128 /2', 128, 128, S // 2, 3),
                                  ('256->128 /2', 256, 128, S // 2, 3), ('256->256 /4', 256, 256, S // 4, 3), ('5^3 64->64', 64, 64, S, 5)]:
            x = torch.randn(1, C, E, E, E, device=dev)
            w = torch.randn(K, C, ks, ks, ks, device=dev) * 0.05
            xs = to_s3(x)
            t = timeit(lambda: fwd_split(x, w, None, xs), iters=10, warm=3)
            gf = 2.0 * ks ** 3 * C * K * E ** 3 / 1e9
            tot += t
            print('[%s] S=%d %-12s %.3f ms  %.0f TF' % (tag, S, name, t, gf / t))
    print('[%s] total %.3f ms' % (tag, tot))


if __name__ == '__main__':
    main()
