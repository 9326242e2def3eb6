# Round-6 evidence in one GPU call (every profiler run under its own timeout): kernel stats of the train step and of a ONE-stream, ONE-cube-per-call
# 480^3 inference, main-queue gaps.  Outputs under gpurun_out/r06; summaries are copied into profiles/ by hand (profiles/README.md).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06
mkdir -p $O; rm -rf $O/train108 $O/infer1
T="timeout 600"
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/train108 -o t -- python3 bench.py --workload train --steps 10 --warmup 3 --no-cpu-baseline > $O/train108.log 2>&1
python3 tools/trace_gaps.py $O/train108 > $O/train108_gaps.txt 2>&1
NC_INFER_STREAMS=1 $T rocprofv3 --kernel-trace --stats --output-format csv -d $O/infer1 -o t -- python3 bench.py --workload infer --volume 480 --steps 1 --warmup 1 --no-cpu-baseline > $O/infer1.log 2>&1
find $O/train108 -name "*kernel_stats.csv" -exec cp {} $O/train108_kernel_stats.csv \;
find $O/infer1 -name "*kernel_stats.csv" -exec cp {} $O/infer480_one_stream_kernel_stats.csv \;
rm -rf $O/train108 $O/infer1
head -40 $O/train108_gaps.txt
