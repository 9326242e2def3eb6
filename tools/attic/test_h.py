"""Quick GPU check + timing of the 16-bit conv kernels (development tool; the parity tests are in tests/test_gpu_lp.py)."""
import sys
import time
import torch
import torch.nn.functional as F
sys.path.insert(0, '.')
from neuroclear_amd import ops


def rnd(t, prec):
    return t.to(torch.bfloat16 if prec == 'bf16' else torch.float16).float()


def one(N, C, K, S, ks, prec, what, time_it=False):
    D, H, W = S if isinstance(S, tuple) else (S, S, S)
    g = torch.Generator(device='cuda').manual_seed(1)
    x = torch.randn(N, C, D, H, W, device='cuda', generator=g)
    w = torch.randn(K, C, ks, ks, ks, device='cuda', generator=g) * (1.0 / (C * ks ** 3) ** 0.5)
    b = torch.randn(K, device='cuda', generator=g)
    dy = torch.randn(N, K, D, H, W, device='cuda', generator=g)
    ops.set_conv_precision(prec)
    if what == 'fwd':
        y = ops.conv_fwd_raw(x, w, b, 1, ks // 2)
        ref = F.conv3d(rnd(x, prec), rnd(w, prec), b, padding=ks // 2)
        f = lambda: ops.conv_fwd_raw(x, w, b, 1, ks // 2)
    elif what == 'dgrad':
        y = ops.conv_dgrad_raw(dy, w, x.shape, 1, ks // 2)
        ref = F.conv_transpose3d(rnd(dy, prec), rnd(w, prec), padding=ks // 2)
        f = lambda: ops.conv_dgrad_raw(dy, w, x.shape, 1, ks // 2)
    else:
        y, _ = ops.conv_wgrad_raw(x, dy, w.shape, 1, ks // 2, False)
        xr, dyr = rnd(x, prec).requires_grad_(False), rnd(dy, prec)
        wz = torch.zeros_like(w, requires_grad=True)
        F.conv3d(xr, wz, None, padding=ks // 2).backward(dyr)
        ref = wz.grad
        f = lambda: ops.conv_wgrad_raw(x, dy, w.shape, 1, ks // 2, False)
    ops.set_conv_precision('fp32')
    err = (y - ref).abs().max().item() / ref.abs().max().item()
    msg = '%s %s N%d C%d K%d %s k%d: rel err %.2e' % (prec, what, N, C, K, S, ks, err)
    if time_it:
        ops.set_conv_precision(prec)
        for _ in range(2):
            f()
        torch.cuda.synchronize()
        t0 = time.time()
        n = 5
        for _ in range(n):
            f()
        torch.cuda.synchronize()
        dt = (time.time() - t0) / n
        ops.set_conv_precision('fp32')
        fl = 2.0 * N * C * K * ks ** 3 * D * H * W
        msg += '  %.3f ms  %.0f TFLOP/s (incl. convert+pack)' % (dt * 1e3, fl / dt / 1e12)
    print(msg, flush=True)
    return err


if __name__ == '__main__':
    whats = sys.argv[1].split(',') if len(sys.argv) > 1 else ['fwd', 'dgrad']
    bad = 0
    for what in whats:
        for prec in ('bf16', 'fp16'):
            for (N, C, K, S, ks) in [(1, 32, 64, 8, 3), (2, 32, 64, (5, 9, 13), 3), (1, 64, 128, 20, 3),
                                     (1, 128, 64, (7, 30, 37), 3), (1, 64, 64, 36, 3), (1, 64, 64, 20, 5),
                                     (2, 32, 64, (6, 11, 23), 5), (1, 64, 64, (9, 40, 52), 5)]:
                if what == 'dgrad':
                    C, K = K, C
                bad += one(N, C, K, S, ks, prec, what) > 2e-4
    for what in whats:
        one(1, 64, 64, 108, 3, 'bf16', what, True)
        one(1, 64, 64, 148, 3, 'bf16', what, True)
        one(1, 128, 128, 74, 3, 'bf16', what, True)
        one(1, 256, 256, 37, 3, 'bf16', what, True)
        one(1, 64, 64, 108, 5, 'bf16', what, True)
        one(1, 64, 64, 148, 5, 'bf16', what, True)
    print('BAD' if bad else 'OK', bad)
