"""HBM bytes per 140^3 cube of the diced inference from two --pmc passes (FETCH_SIZE, WRITE_SIZE; tools/prof_r03.sh): all kernels of the
run summed and divided by the number of cube forwards the run made, COUNTED from the same files (one k_sigmoid_fwd -- or, with the fused
inference tail, k_in_act_tail -- dispatch per nc_unet_deconv_fwd call; warm-up cubes included in both numerator and denominator).
python tools/pmc_cube.py <fetch_dir> <write_dir> -> JSON on stdout."""
import csv
import glob
import json
import os
import sys


def total(d, counter):
    t, cubes = 0.0, 0
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r['Counter_Name'] == counter:
                    t += float(r['Counter_Value'])
                    if 'k_sigmoid_fwd' in r['Kernel_Name'] or 'k_in_act_tail' in r['Kernel_Name']:  # (one of the two per cube forward)
                        cubes += 1
    return t, cubes


fe, n1 = total(sys.argv[1], 'FETCH_SIZE')
wr, n2 = total(sys.argv[2], 'WRITE_SIZE')
fe *= 1024 * 2  # KiB; gfx950: FETCH_SIZE under-reports 2 x (MI355X_MICROARCH.md)
wr *= 1024
assert n1 == n2 and n1 > 0, (n1, n2)
print(json.dumps(dict(cubes=n1, fetch_bytes_per_cube=fe / n1, write_bytes_per_cube=wr / n1, hbm_bytes_per_cube=(fe + wr) / n1)))
