"""Timing experiment (not a test): one 16-bit conv layer, kernel time with parts of the kernel switched off
(env NC_H_ABLATE, read once by the library): 0 full, 1 no stores, 2 no MFMA loop, 4 no LDS-DMA.
usage: python tools/ablate_h.py fwd|wgrad N C K S ks"""
import os
import subprocess
import sys

if len(sys.argv) > 1 and sys.argv[1] == '--child':
    import torch
    sys.path.insert(0, '.')
    from neuroclear_amd import ops
    what, N, C, K, S, ks = sys.argv[2], *map(int, sys.argv[3:8])
    x = torch.randn(N, C, S, S, S, device='cuda')
    w = torch.randn(K, C, ks, ks, ks, device='cuda') * 0.01
    dy = torch.randn(N, K, S, S, S, device='cuda')
    ops.set_conv_precision('bf16')
    ts = []
    for _ in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if what == 'fwd':
            ops.conv_fwd_raw(x, w, None, 1, ks // 2)
        else:
            ops.conv_wgrad_raw(x, dy, w.shape, 1, ks // 2, False)
        e1.record()
        ts.append((e0, e1))
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ts)[1]
    print('RESULT %.3f' % ms)
    sys.exit(0)

for ab in [int(a) for a in os.environ.get('ABL', '0,1,2,4,3,6,7').split(',')]:
    env = dict(os.environ, NC_H_ABLATE=str(ab))
    out = subprocess.run([sys.executable, __file__, '--child'] + sys.argv[1:], env=env, capture_output=True, text=True)
    r = [l for l in out.stdout.splitlines() if l.startswith('RESULT')]
    print('ablate %d: %s ms (whole call incl. to_c8 + pack)' % (ab, r[-1].split()[1] if r else 'FAILED ' + out.stderr[-300:]), flush=True)
