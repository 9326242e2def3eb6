"""Times fwd / wgrad of the dominant MFMA layer shapes of the 108^3 step; prints one JSON line {layer: ms}."""
import json
import sys

import torch

sys.path.insert(0, '.')
from neuroclear_amd import ops  # noqa: E402

dev = 'cuda'
S = int(sys.argv[1]) if len(sys.argv) > 1 else 108


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


res = {}
for name, C, K, E, k, p in [('3^3 64->64 @S', 64, 64, S, 3, 1), ('3^3 128->64 @S', 128, 64, S, 3, 1),
                            ('5^3 64->64 @S', 64, 64, S, 5, 2), ('3^3 128->128 @S/2', 128, 128, S // 2, 3, 1),
                            ('3^3 256->256 @S/4', 256, 256, S // 4, 3, 1)]:
    x = torch.randn(1, C, E, E, E, device=dev)
    w = torch.randn(K, C, k, k, k, device=dev) * 0.05
    y = ops.conv_fwd_raw(x, w, None, 1, p)
    res['fwd ' + name] = timeit(lambda: ops.conv_fwd_raw(x, w, None, 1, p))
    res['wgrad ' + name] = timeit(lambda: ops.conv_wgrad_raw(x, y, w.shape, 1, p, False))
print(json.dumps(res))
