"""Raw per-kernel averages of every counter in a rocprofv3 --kernel-trace --pmc run (sums over the counter's dimensions, mean over the
dispatches of a kernel), plus the mean duration.  usage: python tools/pmc_raw.py <dir> [kernel name filter]"""
import collections
import csv
import glob
import os
import re
import sys


def short(name):
    m = re.search(r'(k_[A-Za-z0-9_]+(<[^>]*>)?)', name)
    return m.group(1) if m else name[:40]


d, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else None)
disp = {}
for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        k = (f, int(r['Dispatch_Id']))
        e = disp.setdefault(k, dict(kernel=short(r['Kernel_Name']), dur=(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, c=collections.Counter()))
        e['c'][r['Counter_Name']] += float(r['Counter_Value'])
agg = {}
for e in disp.values():
    if flt and flt not in e['kernel']:
        continue
    a = agg.setdefault(e['kernel'], dict(n=0, dur=0.0, c=collections.Counter()))
    a['n'] += 1
    a['dur'] += e['dur']
    a['c'].update(e['c'])
for k, a in sorted(agg.items()):
    print('%s  n=%d  dur_us=%.1f' % (k, a['n'], a['dur'] / a['n']))
    for c, v in sorted(a['c'].items()):
        print('    %-40s %.4g' % (c, v / a['n']))
