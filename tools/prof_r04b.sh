# Round-4 evidence, second part (after the tile-order change): the 16-bit kernels' stats / SQ counters / HBM traffic at 64 -> 64, 4 x 148^3,
# kernel stats of the configs[3] step, its same-box A/B (NC_C8X), and the default bench line.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b
rm -rf $O; mkdir -p $O
T="timeout 600"
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -o t -- python3 bench.py --workload train --crop 148 --batch 4 --precision bf16 --steps 4 --warmup 2 --no-cpu-baseline > $O/c3.log 2>&1
for m in 2 0; do
  export NC_C8X=$m
  $T rocprofv3 --kernel-trace --stats --output-format csv -d $O/c8st$m -o t -- python3 tools/pmc_run_c8.py > $O/c8st$m.log 2>&1
  $T rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/c8sq$m -o t -- python3 tools/pmc_run_c8.py > $O/c8sq$m.log 2>&1
  python3 tools/pmc_summary.py $O/c8sq$m $O/c8_sq_counters_mode$m.csv k_ > /dev/null
  $T rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/c8tf$m -o t -- python3 tools/pmc_run_c8.py > $O/c8tf$m.log 2>&1
  $T rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/c8tw$m -o t -- python3 tools/pmc_run_c8.py > $O/c8tw$m.log 2>&1
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
f = per_kernel('$O/c8tf$m', 'FETCH_SIZE'); w = per_kernel('$O/c8tw$m', 'WRITE_SIZE')
out = {n: {'fetch_bytes_per_launch': f[n] * 2 * 1024, 'write_bytes_per_launch': w.get(n, 0) * 1024} for n in f}
json.dump(out, open('$O/c8_traffic_mode$m.json', 'w'), indent=1)
PY
done
unset NC_C8X
rm -rf $O/c8tf* $O/c8tw*; find $O -name "*counter_collection.csv" -delete
find $O -name "*kernel_trace.csv" -delete; find $O -name "t_agent_info.csv" -delete
for m in 1 0 1 0 2; do
  echo "configs[3] 4x148^3 bf16 NC_C8X=$m $(NC_C8X=$m timeout 600 python3 bench.py --crop 148 --batch 4 --precision bf16 --workload train --no-cpu-baseline --steps 8 --warmup 3 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.readline()); print('ms_per_step %.2f' % j['ms_per_step'], {k: v['tflops'] for k, v in j['roofline']['classes'].items() if '_lp_k' in k})")" >> $O/ab_r04.txt
done
timeout 900 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
cat $O/ab_r04.txt; cut -c1-300 $O/bench_default.json
