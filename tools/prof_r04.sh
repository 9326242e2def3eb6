# Round-4 evidence in one GPU call (every profiler run under its own timeout): kernel stats of the default train / inference / Athena /
# configs[3] runs, SQ counters + HBM traffic of the 16-bit kernels at 64 -> 64, 4 x 148^3 (k_conv_c8x against k_conv_h), the same-box A/B of
# the new kernel in the configs[3] step, the default bench line and the 2-rank dry run.  Outputs under gpurun_out/r04; summaries are copied
# into profiles/ by hand.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
rm -rf $O; mkdir -p $O
T="timeout 600"
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/train108 -o t -- python3 bench.py --workload train --steps 10 --warmup 3 --no-cpu-baseline > $O/train108.log 2>&1
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/infer -o t -- python3 bench.py --workload infer --steps 1 --warmup 1 --no-cpu-baseline > $O/infer.log 2>&1
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/athena -o t -- python3 bench.py --workload train --model athena --data structured --steps 6 --warmup 3 --no-cpu-baseline > $O/athena.log 2>&1
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -o t -- python3 bench.py --workload train --crop 148 --batch 4 --precision bf16 --steps 4 --warmup 2 --no-cpu-baseline > $O/c3.log 2>&1
# the 16-bit kernels at 64 -> 64, 4 x 148^3: stats, SQ counters, traffic (separate passes); mode 1 (default: k_conv_c8x) and NC_C8X=0 (k_conv_h)
for m in 1 0; do
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
# same-box A/B of the configs[3] step and of the other workloads' sensitivity to it
for m in 1 0 1 0; do
  echo "configs[3] 4x148^3 bf16 NC_C8X=$m $(NC_C8X=$m timeout 600 python3 bench.py --crop 148 --batch 4 --precision bf16 --workload train --no-cpu-baseline --steps 8 --warmup 3 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.readline()); print('ms_per_step %.2f' % j['ms_per_step'], {k: v['tflops'] for k, v in j['roofline']['classes'].items() if '_lp_k' in k})")" >> $O/ab_r04.txt
done
timeout 900 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
NC_DIST_BACKEND=gloo timeout 900 python3 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_2rank_dry.json 2> $O/bench_2rank_dry.err
for f in train108 infer athena c3; do tail -1 $O/$f.log | cut -c1-300; done
cat $O/ab_r04.txt; cut -c1-600 $O/bench_default.json; cut -c1-400 $O/bench_2rank_dry.json
grep -E "conv_c8x|conv_h<2, 3|conv_h<2, 5|wgrad_s3x" $O/c8_sq_counters_mode1.csv | tail -8; grep -E "conv_h<2, 3|conv_h<2, 5" $O/c8_sq_counters_mode0.csv | tail -4
