# Athena step (configs[4]): kernel stats + main-queue gaps, and the smoke line with the W64 switch on / off
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O; rm -rf $O/athena
NC_S3X_W64=0 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
NC_S3X_W64=1 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for v in 0 1; do NC_S3X_W64=$v python3 bench.py --workload train --model athena --data structured --steps 15 --warmup 4 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('athena W64=$v ms', j['ms_per_step'])"; done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/athena -o t -- python3 bench.py --workload train --model athena --data structured --steps 8 --warmup 3 --no-cpu-baseline > $O/athena.log 2>&1
python3 tools/trace_gaps.py $O/athena > $O/athena_gaps.txt 2>&1
find $O/athena -name "*kernel_stats.csv" -exec cp {} $O/athena_kernel_stats.csv \;
rm -rf $O/athena
head -30 $O/athena_gaps.txt
