"""Check + time the split-operand PatchGAN layer kernel (csrc/conv_p2d.hip) against fp64 and against the image-staged fp32 kernel
(ops.set_conv_split(False) sends the layer back to k_sconv).  usage: python tools/p2d_check.py"""
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, '.')
from neuroclear_amd import ops
from neuroclear_amd._lib import lib

L = lib()


def timeit(f, n=10):
    f(); f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for B, C, K, H, W, st in ((216, 256, 512, 13, 13, 1), (108, 256, 512, 13, 13, 1), (70, 128, 64, 17, 12, 1), (216, 64, 128, 54, 54, 2), (216, 128, 256, 27, 27, 2),
                          (108, 64, 128, 54, 54, 2), (108, 128, 256, 27, 27, 2), (50, 64, 64, 21, 30, 2), (40, 128, 128, 19, 17, 2)):
    g = torch.Generator(device='cuda').manual_seed(1)
    x = torch.randn(B, C, H, W, device='cuda', generator=g)
    w = torch.randn(K, C, 4, 4, device='cuda', generator=g) * (2.0 / (C * 16)) ** 0.5
    b = torch.randn(K, device='cuda', generator=g)
    ops.set_conv_split(True)
    y = ops.conv_fwd_raw(x, w, b, st, 1)
    dy = torch.randn(y.shape, device='cuda', generator=g)
    dx = ops.conv_dgrad_raw(dy, w, x.shape, st, 1)
    y2 = ops.conv_fwd_raw(x, w, b, st, 1)
    dx2 = ops.conv_dgrad_raw(dy, w, x.shape, st, 1)
    ops.set_conv_split(False)
    y0 = ops.conv_fwd_raw(x, w, b, st, 1)
    dx0 = ops.conv_dgrad_raw(dy, w, x.shape, st, 1)
    ops.set_conv_split(True)
    yr = F.conv2d(x.double(), w.double(), b.double(), stride=st, padding=1)
    dxr = torch.nn.grad.conv2d_input(x.shape, w.double(), dy.double(), stride=st, padding=1)
    fl = 2.0 * C * K * 16 * y.numel() / K

    def err(a, r):
        return float((a.double() - r).abs().max() / r.abs().max()), float((a.double() - r).pow(2).mean().sqrt() / r.pow(2).mean().sqrt())
    print('B %3d %3d->%3d %2dx%2d s%d | fwd err split %.2e/%.2e  fp32mfma %.2e/%.2e | dgrad split %.2e/%.2e fp32mfma %.2e/%.2e | rerun equal %s %s' % (
        (B, C, K, H, W, st) + err(y, yr) + err(y0, yr) + err(dx, dxr) + err(dx0, dxr) + (torch.equal(y, y2), torch.equal(dx, dx2))), flush=True)
    ts = timeit(lambda: ops.conv_fwd_raw(x, w, b, st, 1)), timeit(lambda: ops.conv_dgrad_raw(dy, w, x.shape, st, 1))
    ops.set_conv_split(False)
    t0 = timeit(lambda: ops.conv_fwd_raw(x, w, b, st, 1)), timeit(lambda: ops.conv_dgrad_raw(dy, w, x.shape, st, 1))
    ops.set_conv_split(True)
    print('      fwd %.3f ms %.0f TF (k_sconv %.3f ms %.0f TF) | dgrad %.3f ms %.0f TF (k_sconv %.3f ms %.0f TF)' % (
        ts[0], fl / ts[0] / 1e9, t0[0], fl / t0[0] / 1e9, ts[1], fl / ts[1] / 1e9, t0[1], fl / t0[1] / 1e9), flush=True)
    # weight gradient (csrc/wgrad_p2d.hip where it applies; k_swgrad otherwise and with the split kernels off)
    act = L.nc_conv2d_split_active(2, B, C, H, W, K, 4, st, 1)
    dw, _ = ops.conv_wgrad_raw(x, dy, w.shape, st, 1, False)
    dw2, _ = ops.conv_wgrad_raw(x, dy, w.shape, st, 1, False)
    ops.set_conv_split(False)
    dw0, _ = ops.conv_wgrad_raw(x, dy, w.shape, st, 1, False)
    tw0 = timeit(lambda: ops.conv_wgrad_raw(x, dy, w.shape, st, 1, False))
    ops.set_conv_split(True)
    tw = timeit(lambda: ops.conv_wgrad_raw(x, dy, w.shape, st, 1, False))
    dwr = torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double(), stride=st, padding=1)
    print('      wgrad split-active %d err %.2e/%.2e  k_swgrad %.2e/%.2e | rerun equal %s | %.3f ms %.0f TF (k_swgrad %.3f ms %.0f TF)' % (
        (act,) + err(dw, dwr) + err(dw0, dwr) + (torch.equal(dw, dw2), tw, fl / tw / 1e9, tw0, fl / tw0 / 1e9)), flush=True)
