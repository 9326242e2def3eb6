"""Timeline summary of a rocprofv3 --kernel-trace run of `bench.py --workload train`: per queue busy time, idle gaps of the busiest
queue, and the top kernels by time.  usage: python tools/trace_gaps.py <dir> [steps]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r'(k_[A-Za-z0-9_]+(<[^>]*>)?)', name)
    return m.group(1) if m else name[:40]


def main(d, steps=3):
    rows = []
    for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
        rows += list(csv.DictReader(open(f)))
    ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'], short(r['Kernel_Name'])) for r in rows))
    # keep the last `steps` thirds of the run?  simpler: everything after the first k_adam launch burst is steady state; report per step
    t0, t1 = ev[0][0], ev[-1][1]
    print('kernels %d span %.1f ms' % (len(ev), (t1 - t0) / 1e6))
    byq = defaultdict(list)
    for s, e, q, n in ev:
        byq[q].append((s, e, n))
    for q, l in sorted(byq.items(), key=lambda x: -sum(e - s for s, e, _ in x[1])):
        busy = sum(e - s for s, e, _ in l)
        print('queue %s: %d kernels, busy %.1f ms' % (q, len(l), busy / 1e6))
    main_q = max(byq, key=lambda q: sum(e - s for s, e, _ in byq[q]))
    l = byq[main_q]
    gaps = defaultdict(lambda: [0, 0])
    for (s0, e0, n0), (s1, e1, n1) in zip(l, l[1:]):
        g = s1 - e0
        if g > 0:
            k = '%s -> %s' % (n0, n1)
            gaps[k][0] += 1
            gaps[k][1] += g
    tot = sum(v[1] for v in gaps.values())
    print('main queue idle between kernels: %.2f ms total' % (tot / 1e6))
    for k, v in sorted(gaps.items(), key=lambda x: -x[1][1])[:14]:
        print('   %-70s n=%d %.3f ms (%.1f us each)' % (k, v[0], v[1] / 1e6, v[1] / v[0] / 1e3))
    # steady state: the last `steps` steps = the part of the main queue after its longest idle gap (set-up / warm-up compiles end there)
    cut = max(range(1, len(l)), key=lambda i: l[i][0] - l[i - 1][1]) if len(l) > 1 else 0
    ls = l[cut:]
    span = ls[-1][1] - ls[0][0]
    busy = sum(e - s for s, e, _ in ls)
    print('main queue after its longest gap: %d kernels, span %.2f ms, busy %.2f ms, idle %.2f ms' % (len(ls), span / 1e6, busy / 1e6, (span - busy) / 1e6))
    for name, sel in (('main queue', lambda q: q == main_q), ('other queues', lambda q: q != main_q)):
        tk = defaultdict(lambda: [0, 0])
        for s, e, q, n in ev:
            if sel(q) and s >= ls[0][0]:
                tk[n][0] += 1
                tk[n][1] += e - s
        print('--- %s, steady part: %.2f ms of kernels' % (name, sum(v[1] for v in tk.values()) / 1e6))
        for n, v in sorted(tk.items(), key=lambda x: -x[1][1])[:22]:
            print('%-40s n=%5d %.2f ms total  %.1f us avg' % (n, v[0], v[1] / 1e6, v[1] / v[0] / 1e3))


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3)
