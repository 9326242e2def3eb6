# The 16-bit kernels at 64 -> 64 channels, 4 x 148^3 (tools/pmc_run_c8.py): kernel stats, SQ counters, HBM bytes per launch -- three passes.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c8; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o t -- python3 tools/pmc_run_c8.py > $O/st.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq -o t -- python3 tools/pmc_run_c8.py > $O/sq.log 2>&1
python3 tools/pmc_summary.py $O/sq $O/c8_sq_counters.csv k_ > /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/tf -o t -- python3 tools/pmc_run_c8.py > $O/tf.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/tw -o t -- python3 tools/pmc_run_c8.py > $O/tw.log 2>&1
python3 - <<PY
import csv, collections, json
def per_kernel(d, counter):
    agg = collections.defaultdict(lambda: [0.0, 0])
    disp = collections.defaultdict(float); names = {}
    for r in csv.DictReader(open(d + '/t_counter_collection.csv')):
        if r['Counter_Name'] != counter: continue
        disp[r['Dispatch_Id']] += float(r['Counter_Value']); names[r['Dispatch_Id']] = r['Kernel_Name']
    for k, v in disp.items():
        n = names[k]; n = n[n.find('k_'):].split('(')[0]
        agg[n][0] += v; agg[n][1] += 1
    return {n: a[0] / a[1] for n, a in agg.items()}
f = per_kernel('$O/tf', 'FETCH_SIZE'); w = per_kernel('$O/tw', 'WRITE_SIZE')
out = {n: {'fetch_bytes_per_launch': f[n] * 2 * 1024, 'write_bytes_per_launch': w.get(n, 0) * 1024} for n in f}
json.dump(out, open('$O/c8_traffic.json', 'w'), indent=1)
for n, v in sorted(out.items()):
    if 'conv_h' in n or 'wgrad' in n or 'c8' in n: print('%-48s %.2f GB read %.2f GB written' % (n[:48], v['fetch_bytes_per_launch'] / 1e9, v['write_bytes_per_launch'] / 1e9))
PY
grep -E "conv_h|wgrad_s3x" $O/c8_sq_counters.csv | tail -8
python3 - <<PY
import csv
for r in list(csv.DictReader(open('$O/st/t_kernel_stats.csv')))[:12]:
    print(r['Name'][:80], r['Calls'], round(float(r['AverageNs']) / 1e6, 3))
PY
rm -rf $O/tf $O/tw $O/sq/t_counter_collection.csv; find $O -name "*kernel_trace.csv" -delete
