"""Where a tile's time goes: s_memtime stamps of workgroup 0 / wave 0 of the tap-stream kernel (a -DNC_S3X_STAMP build of the library,
tools/s3x_variant.sh stamp -DNC_S3X_STAMP; run with NC_HIP_LIB=neuroclear_amd/csrc/abl/libnc_hip_s3x_stamp.so).  The stamps sit in the
workspace behind the packed weights.  s_memtime ticks at 100 MHz."""
import sys

import torch

sys.path.insert(0, '.')
from neuroclear_amd import ops  # noqa: E402
from tools.split_conv import fwd_split, split_ws, to_s3  # noqa: E402

dev = 'cuda'
C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
DZ = int(sys.argv[2]) if len(sys.argv) > 2 else 33
x = torch.randn(1, C, DZ, 108, 108, device=dev)
w = torch.randn(64, C, 3, 3, 3, device=dev) * 0.05
xs = to_s3(x)
for _ in range(3):
    fwd_split(x, w, None, xs)
ws = split_ws(1, C, DZ, 108, 108, 64, 3)
ws.zero_()
fwd_split(x, w, None, xs)
torch.cuda.synchronize()
packed = 64 * C * 27 * 6
st = ws[packed:packed + 32000].view(torch.int64).cpu().numpy()
st = st[st != 0]
wg = ws[packed + 32000:packed + 32000 + 256 * 16].view(torch.int64).cpu().numpy().reshape(256, 2)
t0 = wg[:, 0].min()
import numpy as np  # noqa: E402
dur = (wg[:, 1] - wg[:, 0]) / 100.0
print('last launch: workgroup start spread %.1f us; duration min %.1f mean %.1f max %.1f us; end spread %.1f us; kernel span %.1f us' % (
    (wg[:, 0].max() - t0) / 100.0, dur.min(), dur.mean(), dur.max(), (wg[:, 1].max() - wg[:, 1].min()) / 100.0, (wg[:, 1].max() - t0) / 100.0))
for x8 in range(8):
    print('   xcd %d: mean duration %.1f us' % (x8, dur[x8::8].mean()))
NS = 27 * C // 32
per_tile = 1 + 5 + (NS - 4) // 2  # tile top, around the 4 peeled steps, every pair of later steps
print("stamps", len(st), 'per tile', per_tile)
for t in range(min(4, len(st) // per_tile)):
    s = st[t * per_tile:(t + 1) * per_tile + 1]
    d = (s[1:] - s[:-1])  # shader cycles
    print('tile %d: top->step0 %d cyc; steps 0..3: %s cyc; step pairs: mean %.0f min %d max %d cyc; last pair -> next tile top %s' % (
        t, d[0], list(d[1:5]), d[5:per_tile - 1].mean(), d[5:per_tile - 1].min(), d[5:per_tile - 1].max(),
        d[per_tile - 1] if len(d) >= per_tile else '-'))
    print('   pairs:', list(d[5:per_tile - 1]))
