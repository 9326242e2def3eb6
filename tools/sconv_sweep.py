"""Timing experiment: every tile configuration of k_sconv (NC_SCONV_CFG=i, one process per i) on the PatchGAN layers at
Athena's batches; prints TFLOP/s of fwd / dgrad per layer (0 = configuration not applicable -> gather GEMM ran instead)."""
import os
import subprocess
import sys

if len(sys.argv) > 1 and sys.argv[1] == 'one':
    import torch
    sys.path.insert(0, '.')
    from neuroclear_amd import ops
    from neuroclear_amd._lib import lib

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
    out = []
    for N in (108, 216):
        for C, K, H, s in [(64, 128, 54, 2), (128, 256, 27, 2), (256, 512, 13, 1)]:
            x = torch.randn(N, C, H, H, device='cuda')
            w = torch.randn(K, C, 4, 4, device='cuda') * 0.02
            y = ops.conv_fwd_raw(x, w, None, s, 1)
            dy = torch.randn_like(y)
            fl = 2.0 * C * K * 16 * y.numel() / K
            tf = timeit(lambda: ops.conv_fwd_raw(x, w, None, s, 1))
            td = timeit(lambda: ops.conv_dgrad_raw(dy, w, x.shape, s, 1))
            out.append('%5.1f/%5.1f' % (fl / tf / 1e9, fl / td / 1e9))
    print('cfg %2s  ' % os.environ.get('NC_SCONV_CFG', '-') + '  '.join(out), flush=True)
else:
    print('        ' + '  '.join('B%d L%d f/d  ' % (n, l) for n in (108, 216) for l in (2, 3, 4)))
    for i in [-1] + list(range(14)):
        env = dict(os.environ)
        if i >= 0:
            env['NC_SCONV_CFG'] = str(i)
        subprocess.run([sys.executable, __file__, 'one'], env=env, stderr=subprocess.DEVNULL)
