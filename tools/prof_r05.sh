# Round-5 evidence in one GPU call (every profiler run under its own timeout): the default bench line, kernel stats of the train step and of a
# ONE-stream 480^3 inference, and the guard-off / guard-on A/B of the train step.  Outputs under gpurun_out/r05; summaries are copied into
# profiles/ by hand (profiles/README.md).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05
rm -rf $O; mkdir -p $O
T="timeout 600"
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/train108 -o t -- python3 bench.py --workload train --steps 10 --warmup 3 --no-cpu-baseline > $O/train108.log 2>&1
NC_INFER_STREAMS=1 $T rocprofv3 --kernel-trace --stats --output-format csv -d $O/infer1 -o t -- python3 bench.py --workload infer --volume 480 --steps 1 --warmup 1 --no-cpu-baseline > $O/infer1.log 2>&1
for i in 1 2; do
  for g in 0 1; do
    NC_H2_GUARD=$g python3 bench.py --workload train --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('NC_H2_GUARD=$g ms_per_step %.3f' % j['ms_per_step'], j.get('two_term_range_guard'))" >> $O/ab_guard.txt
  done
done
python3 tools/trace_gaps.py $O/train108 > $O/train108_gaps.txt 2>&1
cat $O/ab_guard.txt
