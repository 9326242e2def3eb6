"""Timing experiments (not a test): runs tools/kernel_one.py once per library variant in neuroclear_amd/csrc/abl/
(tools/ablate_build.sh) and prints the per-layer times side by side."""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = [None] + sorted(glob.glob(os.path.join(ROOT, 'neuroclear_amd', 'csrc', 'abl', 'libnc_hip_*.so')))
rows = {}
for lib in libs:
    env = dict(os.environ)
    tag = 'base'
    if lib:
        env['NC_HIP_LIB'] = lib
        tag = os.path.basename(lib)[len('libnc_hip_'):-3]
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'kernel_one.py')] + sys.argv[1:], env=env,
                         capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith('{')]
    if not line:
        print(tag, 'FAILED', out.stderr[-400:])
        continue
    rows[tag] = json.loads(line[-1])
names = list(next(iter(rows.values())).keys())
print('%-28s' % 'layer' + ''.join('%10s' % t for t in rows))
for n in names:
    print('%-28s' % n + ''.join('%10.3f' % rows[t][n] for t in rows))
