"""fp32 3^3 convolution: the split-operand kernel (csrc/conv_split.hip) next to the fp32 MFMA kernel -- error of both against
fp64 at a small size, and time at the layer shapes of the 108^3 step / the 140^3 inference cube."""
import sys

import torch

sys.path.insert(0, '.')
from neuroclear_amd import ops  # noqa: E402
from neuroclear_amd._lib import I, L_, Z, check, lib  # noqa: E402

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


def split_ws(N, C, D, H, W, K, ks):
    nb = lib().nc_conv_split_ws_bytes(I(N), I(C), I(D), I(H), I(W), I(K), I(ks))
    return ops.workspace(nb, dev, 'ws_split')


def to_s3(x):
    N, C = x.shape[:2]
    S = x.numel() // (N * C)
    out = torch.empty(N * C * S * 6, dtype=torch.uint8, device=x.device)
    check(lib().nc_to_s3(ops._ptr(x), ops._ptr(out), I(N), I(C), L_(S), ops._stream()), 'nc_to_s3')
    return out


def fwd_split(x, w, b, xs=None):
    N, C, D, H, W = x.shape
    K = w.shape[0]
    y = torch.empty(N, K, D, H, W, device=x.device)
    ks = w.shape[2]
    ws = split_ws(N, C, D, H, W, K, ks)
    check(lib().nc_conv_fwd_split(ops._ptr(x), ops._ptr(xs), ops._ptr(w), ops._ptr(b), ops._ptr(y), I(N), I(C), I(D), I(H), I(W), I(K),
                                  I(ks), ops._ptr(ws), Z(ws.numel()), ops._stream()), 'nc_conv_fwd_split')
    return y


def dgrad_split(dy, w, dys=None):
    N, K, D, H, W = dy.shape
    C = w.shape[1]
    dx = torch.empty(N, C, D, H, W, device=dy.device)
    ks = w.shape[2]
    ws = split_ws(N, C, D, H, W, K, ks)
    check(lib().nc_conv_dgrad_split(ops._ptr(dy), ops._ptr(dys), ops._ptr(w), ops._ptr(dx), I(N), I(C), I(D), I(H), I(W), I(K),
                                    I(ks), ops._ptr(ws), Z(ws.numel()), ops._stream()), 'nc_conv_dgrad_split')
    return dx


def wgrad_split(x, dy, ks, xs=None, dys=None):
    N, C, D, H, W = x.shape
    K = dy.shape[1]
    dw = torch.empty(K, C, ks, ks, ks, device=x.device)
    ws = split_ws(N, C, D, H, W, K, ks)
    check(lib().nc_conv_wgrad_split(ops._ptr(x), ops._ptr(xs), ops._ptr(dy), ops._ptr(dys), ops._ptr(dw), I(N), I(C), I(D), I(H), I(W),
                                    I(K), I(ks), ops._ptr(ws), Z(ws.numel()), ops._stream()), 'nc_conv_wgrad_split')
    return dw


def main():
    torch.manual_seed(0)
    for (N, C, K, n, ks) in ((1, 64, 64, (20, 22, 27), 3), (2, 16, 128, (9, 17, 30), 3), (1, 128, 64, (12, 12, 12), 3),
                             (1, 64, 64, (11, 14, 19), 5), (2, 8, 64, (3, 30, 40), 5)):
        pd = ks // 2
        x = torch.randn(N, C, *n, device=dev)
        w = torch.randn(K, C, ks, ks, ks, device=dev) * 0.02
        b = torch.randn(K, device=dev)
        ref = torch.nn.functional.conv3d(x.double().cpu(), w.double().cpu(), b.double().cpu(), padding=pd)
        sc = ref.pow(2).mean().sqrt().item()
        y32 = ops.conv_fwd_raw(x, w, b, 1, pd)
        ys = fwd_split(x, w, b)
        for name, y in (('fp32', y32), ('split', ys)):
            e = y.double().cpu() - ref
            print('fwd   %s %-5s max %.2e rms %.2e' % ((N, C, K, n, ks), name, e.abs().max().item() / sc, e.pow(2).mean().sqrt().item() / sc))
        if C % 32 == 0:
            dy = torch.randn(N, K, *n, device=dev)
            refw = torch.nn.grad.conv3d_weight(x.double().cpu(), w.shape, dy.double().cpu(), padding=pd)
            sc = refw.pow(2).mean().sqrt().item()
            ops.set_conv_split(False)
            w32 = ops.conv_wgrad_raw(x, dy, w.shape, 1, pd, False)[0]
            ops.set_conv_split(True)
            wsp = wgrad_split(x, dy, ks)
            for name, y in (('fp32', w32), ('split', wsp)):
                e = y.double().cpu() - refw
                print('wgrad %s %-5s max %.2e rms %.2e' % ((N, C, K, n, ks), name, e.abs().max().item() / sc, e.pow(2).mean().sqrt().item() / sc))
        if C % 64:
            continue
        dy = torch.randn(N, K, *n, device=dev)
        refd = torch.nn.grad.conv3d_input(x.shape, w.double().cpu(), dy.double().cpu(), padding=pd)
        sc = refd.pow(2).mean().sqrt().item()
        d32 = ops.conv_dgrad_raw(dy, w, x.shape, 1, pd)
        dsp = dgrad_split(dy, w)
        for name, y in (('fp32', d32), ('split', dsp)):
            e = y.double().cpu() - refd
            print('dgrad %s %-5s max %.2e rms %.2e' % ((N, C, K, n, ks), name, e.abs().max().item() / sc, e.pow(2).mean().sqrt().item() / sc))
    sizes = [int(a) for a in sys.argv[1:]] or [108]
    for S in sizes:
        for name, C, K, E, ks in [('64->64 @S', 64, 64, S, 3), ('128->64 @S', 128, 64, S, 3), ('64->128 @S/2', 64, 128, S // 2, 3),
                                  ('128->128 @S/2', 128, 128, S // 2, 3), ('256->128 @S/2', 256, 128, S // 2, 3),
                                  ('128->256 @S/4', 128, 256, S // 4, 3), ('256->256 @S/4', 256, 256, S // 4, 3),
                                  ('5^3 64->64 @S', 64, 64, S, 5)]:
            if not lib().nc_conv_split_supported(I(0), I(1), I(C), I(E), I(E), I(E), I(K), I(ks), I(ks), I(ks), I(1), I(ks // 2)):
                print('S=%d %s: not covered' % (S, name))
                continue
            x = torch.randn(1, C, E, E, E, device=dev)
            w = torch.randn(K, C, ks, ks, ks, device=dev) * 0.05
            xs = to_s3(x)
            t32 = timeit(lambda: ops.conv_fwd_raw(x, w, None, 1, ks // 2))
            tsp = timeit(lambda: fwd_split(x, w, None))
            tpre = timeit(lambda: fwd_split(x, w, None, xs))
            tcv = timeit(lambda: to_s3(x))
            if lib().nc_conv_split_supported(I(2), I(1), I(C), I(E), I(E), I(E), I(K), I(ks), I(ks), I(ks), I(1), I(ks // 2)):
                dy = torch.randn(1, K, E, E, E, device=dev)
                ops.set_conv_split(False)
                tw32 = timeit(lambda: ops.conv_wgrad_raw(x, dy, w.shape, 1, ks // 2, False))
                ops.set_conv_split(True)
                tws = timeit(lambda: wgrad_split(x, dy, ks))
                dys = to_s3(dy)
                twp = timeit(lambda: wgrad_split(x, dy, ks, xs, dys))
                gfw = 2.0 * ks ** 3 * C * K * E ** 3 / 1e9
                print('S=%d %-16s wgrad fp32 %.3f ms (%.0f TF)   split %.3f ms (%.0f TF; pre-split operands %.3f = %.0f TF)' % (
                    S, name, tw32, gfw / tw32, tws, gfw / tws, twp, gfw / twp))
                del dy, dys
            gf = 2.0 * ks ** 3 * C * K * E ** 3 / 1e9
            print('S=%d %-16s fp32 %.3f ms (%.0f TF)   split %.3f ms (%.0f TF; pre-split input %.3f = %.0f TF; to_s3 %.3f)' % (
                S, name, t32, gf / t32, tsp, gf / tsp, tpre, gf / tpre, tcv))


if __name__ == '__main__':
    main()
