# Round-5 evidence, second half (final tree): kernel stats of the Athena step (configs[4]) and the configs[3] step, each under its own timeout;
# summaries are copied into profiles/ by hand (profiles/README.md).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05b
rm -rf $O; mkdir -p $O
T="timeout 700"
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/athena -o t -- python3 bench.py --workload train --model athena --data structured --steps 6 --warmup 3 --no-cpu-baseline > $O/athena.log 2>&1
python3 bench.py --workload train --model athena --data structured --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_athena.json
python3 bench.py --workload train --crop 148 --batch 4 --precision bf16 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_config3.json
rm -f $O/athena/*kernel_trace.csv
python3 -c "
import json
for f in ('bench_athena','bench_config3'):
    j=json.load(open('$O/%s.json'%f)); print(f, round(j['ms_per_step'],2), j['value'])
"
