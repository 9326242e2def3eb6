cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c3; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -o t -- python3 bench.py --workload train --crop 148 --batch 4 --precision bf16 --steps 4 --warmup 2 --no-cpu-baseline > $O/c3.log 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "t_agent_info.csv" -delete
python3 - <<PY
import csv
rows = list(csv.DictReader(open('$O/c3/t_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms per step (6 steps):', tot / 6e6)
for r in rows[:32]:
    print('%-90s %6s %9.3f ms/step %7.3f avg_ms %5.1f%%' % (r['Name'][:90], r['Calls'], float(r['TotalDurationNs']) / 6e6, float(r['AverageNs']) / 1e6, float(r['Percentage'])))
PY
