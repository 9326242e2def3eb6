"""Per-KERNEL table from a rocprofv3 --kernel-trace --pmc <SQ counters> run (one row per kernel name: launches, mean duration, clock, MFMA pipe
busy, wait shares), for the matrix kernels of a whole bench run.  usage: python tools/pmc_group.py <dir> <out.csv> [name filter ...]
busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs); clock = GRBM_GUI_ACTIVE / 8 / duration (MI355X_MICROARCH.md)."""
import csv
import glob
import os
import re
import sys


def short(name):
    m = re.search(r'(k_[A-Za-z0-9_]+(<[^>]*>)?)', name)
    return m.group(1) if m else name[:40]


d, out, flts = sys.argv[1], sys.argv[2], sys.argv[3:]
disp = {}
for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = (f, int(r['Dispatch_Id']))
            e = disp.setdefault(k, dict(kernel=short(r['Kernel_Name']), dur=(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, c={}))
            e['c'][r['Counter_Name']] = e['c'].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
agg = {}
for e in disp.values():
    if flts and not any(f in e['kernel'] for f in flts):
        continue
    if e['dur'] < 20.0:  # launches too short for the counters to mean anything
        continue
    gui = e['c'].get('GRBM_GUI_ACTIVE', 0.0) / 8
    wc = e['c'].get('SQ_WAVE_CYCLES', 0.0) or 1.0
    if not gui:
        continue
    a = agg.setdefault(e['kernel'], dict(n=0, dur=0.0, clk=0.0, busy=0.0, wa=0.0, wi=0.0, ai=0.0))
    a['n'] += 1
    a['dur'] += e['dur']
    a['clk'] += gui / (e['dur'] * 1e3)
    a['busy'] += e['c'].get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (gui * 1024)
    a['wa'] += e['c'].get('SQ_WAIT_ANY', 0.0) / wc
    a['wi'] += e['c'].get('SQ_WAIT_INST_ANY', 0.0) / wc
    a['ai'] += e['c'].get('SQ_ACTIVE_INST_ANY', 0.0) / wc
lines = ['kernel,launches,mean_duration_us,clock_GHz,mfma_busy,wait_any,wait_inst,active_inst']
for k, a in sorted(agg.items(), key=lambda kv: -kv[1]['dur']):
    n = a['n']
    lines.append('"%s",%d,%.1f,%.2f,%.3f,%.3f,%.3f,%.3f' % (k, n, a['dur'] / n, a['clk'] / n, a['busy'] / n, a['wa'] / n, a['wi'] / n, a['ai'] / n))
open(out, 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
