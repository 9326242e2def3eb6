"""Per-kernel timing on the GPU box (not a test): full-size layer shapes of the 108^3 Apollo step, HIP events."""
import json
import sys
import time

import torch

sys.path.insert(0, '.')
from neuroclear_amd import ops  # noqa: E402

dev = 'cuda'


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main(sizes):
    res = []
    S = sizes
    layers = [  # name, C, K, spatial edge, k, pad
        ('dc1.0 1->64', 1, 64, S, 3, 1), ('dc1.3 64->64', 64, 64, S, 3, 1), ('dc2.0 64->128', 64, 128, S // 2, 3, 1),
        ('dc2.3 128->128', 128, 128, S // 2, 3, 1), ('bot.0 128->256', 128, 256, S // 4, 3, 1),
        ('bot.3 256->256', 256, 256, S // 4, 3, 1), ('ex2.0 256->128', 256, 128, S // 2, 3, 1),
        ('ex1 128->64', 128, 64, S, 3, 1), ('gb 7^3 1->64', 1, 64, S, 7, 3), ('gb 5^3 64->64', 64, 64, S, 5, 2),
        ('gb 3^3 64->64', 64, 64, S, 3, 1), ('gb 1x1 64->32', 64, 32, S, 1, 0)]
    for name, C, K, E, k, p in layers:
        x = torch.randn(1, C, E, E, E, device=dev)
        w = torch.randn(K, C, k, k, k, device=dev) * 0.05
        y = ops.conv_fwd_raw(x, w, None, 1, p)
        dy = torch.randn_like(y)
        flop = 2.0 * C * K * k ** 3 * E ** 3
        t_f = timeit(lambda: ops.conv_fwd_raw(x, w, None, 1, p))
        t_d = timeit(lambda: ops.conv_dgrad_raw(dy, w, x.shape, 1, p)) if C > 1 or k == 7 else float('nan')
        t_w = timeit(lambda: ops.conv_wgrad_raw(x, dy, w.shape, 1, p, False))
        row = dict(layer=name, E=E, gflop=flop / 1e9, fwd_ms=t_f, dgrad_ms=t_d, wgrad_ms=t_w,
                   fwd_tf=flop / t_f / 1e9, dgrad_tf=flop / t_d / 1e9, wgrad_tf=flop / t_w / 1e9)
        print(json.dumps(row), flush=True)
        res.append(row)
        del x, w, y, dy
    x = torch.randn(1, 64, S, S, S, device=dev)
    t = timeit(lambda: ops.instance_norm_act(x, 0.0))
    print(json.dumps(dict(layer='IN+ReLU 64ch', ms=t, gbps=3 * x.numel() * 4 / t / 1e6)))
    return res


if __name__ == '__main__':
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 108)
