"""What does the main queue wait for?  From a rocprofv3 --kernel-trace run of `bench.py --workload train`: the steady part's main-queue gaps
longer than 200 us (one or two per step: the generators' backward waiting for the discriminator chains of the generator loss), and for the
LAST such gap per step the side-queue kernels that run inside it, queue by queue, in order.  usage: python tools/trace_wait.py <dir>
(Under the tracer the host needs ~2 x as long per launch and the step stretches from 24 to ~43 ms: gap LENGTHS of a traced run say little; the order
and the durations of the side-queue kernels inside a gap are what this is for.)"""
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
rows = []
for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
    rows += list(csv.DictReader(open(f)))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'], short(r['Kernel_Name'])) for r in rows)
byq = defaultdict(list)
for s, e, q, n in ev:
    byq[q].append((s, e, n))
main_q = max(byq, key=lambda q: sum(e - s for s, e, _ in byq[q]))
l = byq[main_q]
cut = max(range(1, len(l)), key=lambda i: l[i][0] - l[i - 1][1])
ls = l[cut:]
gaps = [(ls[i][1], ls[i + 1][0], ls[i][2], ls[i + 1][2]) for i in range(len(ls) - 1) if ls[i + 1][0] - ls[i][1] > 200e3]
print('%d main-queue gaps > 200 us in the steady part; by (kernel before -> after):' % len(gaps))
agg = defaultdict(lambda: [0, 0])
for a, b, n0, n1 in gaps:
    agg[(n0, n1)][0] += 1
    agg[(n0, n1)][1] += b - a
for k, v in sorted(agg.items(), key=lambda x: -x[1][1]):
    print('  %-34s -> %-34s n=%d  %.0f us each' % (k[0], k[1], v[0], v[1] / v[0] / 1e3))
if gaps:
    key = max(agg, key=lambda k: agg[k][1])
    a, b = [g for g in gaps if (g[2], g[3]) == key][-1][:2]
    print('inside the last "%s -> %s" gap (%.0f us):' % (key[0], key[1], (b - a) / 1e3))
    for q, lq in byq.items():
        if q == main_q:
            continue
        ins = [(s, e, n) for s, e, n in lq if e > a and s < b]
        if not ins:
            continue
        busy = sum(min(e, b) - max(s, a) for s, e, n in ins)
        print(' queue %s: %d kernels, busy %.0f us of %.0f' % (q, len(ins), busy / 1e3, (b - a) / 1e3))
        for s, e, n in ins:
            print('    +%7.1f us  %-40s %6.1f us' % ((s - a) / 1e3, n, (e - s) / 1e3))
