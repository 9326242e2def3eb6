"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same bench command) into
profiles/<round>_pmc_traffic.json: HBM bytes per launch for each MFMA kernel class.

usage: python tools/pmc_aggregate.py <fetch_dir> <write_dir> <out.json>

Corrections (MI355X_MICROARCH.md, HBM / rocprofv3 section; re-calibrated in profiles/r01_pmc_calib_*.csv): both counters
are in KiB; FETCH_SIZE under-reports by 2x on gfx950, WRITE_SIZE is exact."""
import csv
import glob
import json
import os
import sys


def kernel_class(name):
    if 'k_conv_mfma<3' in name:
        return 'conv_mfma_k3'
    if 'k_conv_mfma<5' in name:
        return 'conv_mfma_k5'
    if 'k_wgrad_mfma<3' in name or 'k_wgrad_dma<3' in name or 'k_wgrad_rows<3' in name:
        return 'wgrad_mfma_k3'
    if 'k_wgrad_mfma<5' in name or 'k_wgrad_dma<5' in name:
        return 'wgrad_mfma_k5'
    if 'k_conv_s3x<7' in name:  # round 6: Conv3d(1, 64, 7) in pseudo-channel form (last template argument: 1 forward, 2 data gradient)
        return 'conv_c1k7_fwd' if name.split('>')[0].rstrip().endswith('1') else 'conv_c1k7_dgrad'
    if 'k_conv_s3x<5, 4, 2, false, true' in name or 'k_conv_s3x<5, 2, 2, false, true' in name:
        return 'conv_split_k5_k32'   # the 32-output-channel tile (deep_linear_gen's forward without act1)
    if 'k_wgrad_c1<7' in name:
        return 'wgrad_c1_k7'
    if 'k_fold_c1k7' in name:
        return 'fold_c1k7'
    for kern, cls in (('k_conv_s3w<', 'conv_split_k3'), ('k_conv_s3x<3, 8', 'conv_split_k3'), ('k_conv_s3x<3, 7', 'conv_split_k3'), ('k_conv_s3x<5, 7', 'conv_split_k5'), ('k_conv_s3x<3, 6', 'conv_split_k3_small'), ('k_conv_s3x<3, 4', 'conv_split_k3_small'),
                      ('k_conv_s3x<3, 2', 'conv_split_k3_tail'), ('k_conv_s3x<5, 8', 'conv_split_k5'), ('k_conv_s3x<5', 'conv_split_k5_tail'),
                      ('k_conv_s3<3', 'conv_split_k3'), ('k_conv_s3<5', 'conv_split_k5'), ('k_wgrad_s3x<3', 'wgrad_split_k3'),
                      ('k_wgrad_s3x<5', 'wgrad_split_k5'), ('k_wgrad_s3<3', 'wgrad_split_k3'), ('k_wgrad_s3<5', 'wgrad_split_k5'),
                      ('k_split3', 'split3'), ('k_act_split3', 'act_split3'), ('k_in_bwd_apply_s3', 'in_bwd_apply_s3'),
                      ('k_split2h', 'split2h'), ('k_act_split2h', 'act_split2h'), ('k_in_bwd_apply_h2', 'in_bwd_apply_h2'), ('k_absmax(', 'absmax')):
        if kern in name:
            return cls
    return None


def collect(d, counter):
    tot = {}
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r['Counter_Name'] != counter:
                    continue
                c = kernel_class(r['Kernel_Name'])
                if c is None:
                    continue
                t = tot.setdefault(c, [0, 0.0])
                t[0] += 1
                t[1] += float(r['Counter_Value'])
    return tot


def main(fetch_dir, write_dir, out):
    fe = collect(fetch_dir, 'FETCH_SIZE')
    wr = collect(write_dir, 'WRITE_SIZE')
    classes = {}
    for c in sorted(fe):
        n, kib = fe[c]
        nw, kibw = wr.get(c, [0, 0.0])
        fb = kib * 1024 * 2 / n
        wb = kibw * 1024 / nw if nw else 0.0
        classes[c] = dict(launches=n, hbm_bytes_per_launch=fb + wb, fetch_bytes_per_launch=fb, write_bytes_per_launch=wb)
    doc = dict(source='rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `python bench.py --steps 2 '
                      '--warmup 1 --no-cpu-baseline --no-prof`; FETCH_SIZE x2 (gfx950 correction, calibrated with '
                      'tools/pmc_run.py), WRITE_SIZE x1; KiB -> bytes; aggregated by tools/pmc_aggregate.py',
               classes=classes)
    with open(out, 'w') as f:
        json.dump(doc, f, indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == '__main__':
    main(*sys.argv[1:4])
