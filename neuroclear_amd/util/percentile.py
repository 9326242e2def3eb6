"""np.percentile(volume, q) of a float32 device tensor without a sort: exact order statistics by radix select
(nc_radix_hist, three histogram passes per order statistic), then the linear interpolation of the reference's numpy
(1.21.2, conda_environment/neuroclear_env.yml:155): virtual index (n - 1) * q / 100 in float64, neighbours a <= b,
t = fractional part, p = a + (b - a) * t, or b - (b - a) * (1 - t) when t >= 0.5, with (b - a) taken in float32 and the
rest in float64 (numpy/lib/function_base.py `_lerp` of that version on 0-d float32 / float64 operands)."""
import ctypes

import numpy as np
import torch

from .._lib import I, L_, P, check, lib


def _key_to_float(key):
    u = np.uint32(key)
    u = np.uint32(u & np.uint32(0x7fffffff)) if (u & np.uint32(0x80000000)) else np.uint32(~u)
    return float(np.array([u], dtype=np.uint32).view(np.float32)[0])


def select_kth(x, k):
    """Exact k-th smallest element (0-based) of the float32 CUDA tensor x."""
    x = x.reshape(-1)
    n = x.numel()
    hist = torch.empty(4096, dtype=torch.int32, device=x.device)
    stream = P(torch.cuda.current_stream().cuda_stream)
    prefix, rank = 0, int(k)
    for p_, bits in ((0, 12), (1, 12), (2, 8)):
        check(lib().nc_radix_hist(P(x.data_ptr()), L_(n), I(p_), ctypes.c_uint(prefix), P(hist.data_ptr()), stream),
              'nc_radix_hist')
        h = hist.cpu().numpy().astype(np.int64)
        c = np.cumsum(h)
        b = int(np.searchsorted(c, rank, side='right'))
        rank -= int(c[b - 1]) if b else 0
        prefix = (prefix << bits) | b
    return _key_to_float(prefix)


def percentile(x, qs):
    """[np.percentile(x, q) for q in qs] with the arithmetic described in the module docstring; float64 results."""
    n = x.numel()
    out = []
    for q in qs:
        h = (q / 100.0) * (n - 1)
        lo = int(np.floor(h))
        hi = min(lo + 1, n - 1)
        t = h - lo
        a = np.float32(select_kth(x, lo))
        b = np.float32(select_kth(x, hi)) if hi != lo else a
        d = np.float64(np.float32(b - a))
        out.append(float(np.float64(b) - d * (1.0 - t)) if t >= 0.5 else float(np.float64(a) + d * t))
    return out
