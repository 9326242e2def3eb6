"""One step of ONE side queue, kernel by kernel (start relative to the step's first kernel on that queue, duration, gap to the previous kernel):
the discriminator chains whose latency the generators' backward waits for.  usage: python tools/trace_chain.py <dir> [queue rank by kernel count]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r'(k_[A-Za-z0-9_]+(<[^>]*>)?)', name)
    return m.group(1) if m else name[:36]


d = sys.argv[1]
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = []
for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
    rows += list(csv.DictReader(open(f)))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'], short(r['Kernel_Name'])) for r in rows)
byq = defaultdict(list)
for s, e, q, n in ev:
    byq[q].append((s, e, n))
qs = sorted(byq, key=lambda q: -len(byq[q]))
main_q = max(byq, key=lambda q: sum(e - s for s, e, _ in byq[q]))
side = [q for q in qs if q != main_q][rank - 1]
l = byq[side]
# steps: split at gaps > 3 ms on this queue; take the last complete one
cuts = [0] + [i for i in range(1, len(l)) if l[i][0] - l[i - 1][1] > 3e6] + [len(l)]
segs = [l[a:b] for a, b in zip(cuts, cuts[1:]) if b - a > 20]
seg = segs[-2] if len(segs) > 1 else segs[-1]
print('queue %s: %d kernels in the segment, span %.0f us, busy %.0f us' % (side, len(seg), (seg[-1][1] - seg[0][0]) / 1e3, sum(e - s for s, e, _ in seg) / 1e3))
t0, prev = seg[0][0], seg[0][0]
for s, e, n in seg:
    print('  +%8.1f us  gap %6.1f  %-44s %7.1f us' % ((s - t0) / 1e3, (s - prev) / 1e3, n, (e - s) / 1e3))
    prev = e
