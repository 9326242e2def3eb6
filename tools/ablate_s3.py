"""Timing experiment: the split-operand forward kernel with parts switched off (library variants built with -DNC_S3_ABLATE=n into
neuroclear_amd/csrc/abl/, loaded through NC_HIP_LIB): 1 no LDS fragment reads, 4 no DMA, 8 no accumulator restart."""
import os
import subprocess
import sys

CODE = r'''
import sys, torch
sys.path.insert(0, '.')
from tools.split_conv import fwd_split, to_s3, timeit
for (C, K, E, ks) in ((64, 64, 108, 3), (64, 64, 140, 3), (64, 64, 108, 5)):
    x = torch.randn(1, C, E, E, E, device='cuda')
    w = torch.randn(K, C, ks, ks, ks, device='cuda') * 0.05
    xs = to_s3(x)
    print('%d->%d %d^3 k%d: %.3f ms' % (C, K, E, ks, timeit(lambda: fwd_split(x, w, None, xs))))
'''
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for v in [''] + ['s%s' % n for n in sys.argv[1:]]:
    env = dict(os.environ)
    if v:
        env['NC_HIP_LIB'] = os.path.join(root, 'neuroclear_amd', 'csrc', 'abl', 'libnc_hip_%s.so' % v)
    out = subprocess.run([sys.executable, '-c', CODE], cwd=root, env=env, capture_output=True, text=True)
    print('variant', v or 'product', '|', ' | '.join(out.stdout.strip().splitlines()) or out.stderr[-300:], flush=True)
