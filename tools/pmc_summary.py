"""Per-dispatch table from a rocprofv3 --kernel-trace --pmc <SQ counters> run: kernel name (short), duration, clock, MFMA pipe busy,
wait shares.  usage: python tools/pmc_summary.py <dir with *_counter_collection.csv> [out.csv] [name filter]
busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs); clock = GRBM_GUI_ACTIVE / 8 / duration (MI355X_MICROARCH.md)."""
import csv
import glob
import os
import re
import sys


def short(name):
    m = re.search(r'(k_[A-Za-z0-9_]+(<[^>]*>)?)', name)
    return m.group(1) if m else name[:40]


def main(d, out=None, flt=None):
    rows = {}
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = int(r['Dispatch_Id'])
                e = rows.setdefault(k, dict(kernel=short(r['Kernel_Name']), dur=(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
                e[r['Counter_Name']] = e.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    cols = ['dispatch', 'kernel', 'duration_us', 'clock_GHz', 'mfma_busy', 'wait_any', 'wait_inst', 'active_inst']
    lines = [cols]
    for k in sorted(rows):
        e = rows[k]
        if flt and flt not in e['kernel']:
            continue
        gui = e.get('GRBM_GUI_ACTIVE', 0.0) / 8
        wc = e.get('SQ_WAVE_CYCLES', 0.0) or 1.0
        lines.append([k, e['kernel'], '%.1f' % e['dur'], '%.2f' % (gui / (e['dur'] * 1e3)) if e['dur'] else '',
                      '%.3f' % (e.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (gui * 1024)) if gui else '',
                      '%.3f' % (e.get('SQ_WAIT_ANY', 0.0) / wc), '%.3f' % (e.get('SQ_WAIT_INST_ANY', 0.0) / wc),
                      '%.3f' % (e.get('SQ_ACTIVE_INST_ANY', 0.0) / wc)])
    q = lambda c: '"%s"' % c if ',' in str(c) else str(c)
    txt = '\n'.join(','.join(q(c) for c in l) for l in lines)
    if out:
        open(out, 'w').write(txt + '\n')
    print(txt)


if __name__ == '__main__':
    main(*sys.argv[1:4])
