cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train108 -o t -- python3 bench.py --workload train --steps 8 --warmup 2 --no-cpu-baseline > $O/train108.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -o t -- python3 bench.py --workload train --crop 148 --batch 4 --precision bf16 --steps 4 --warmup 2 --no-cpu-baseline > $O/c3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/infer -o t -- python3 bench.py --workload infer --steps 1 --warmup 1 --no-cpu-baseline > $O/infer.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c8k -o t -- python3 tools/pmc_run_c8.py > $O/c8k.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/c8_fetch -o t -- python3 tools/pmc_run_c8.py > $O/c8_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/c8_write -o t -- python3 tools/pmc_run_c8.py > $O/c8_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/t108_fetch -o t -- python3 bench.py --workload train --steps 2 --warmup 1 --no-cpu-baseline --no-prof > $O/t108_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/t108_write -o t -- python3 bench.py --workload train --steps 2 --warmup 1 --no-cpu-baseline --no-prof > $O/t108_write.log 2>&1
rm -f $O/infer/*kernel_trace.csv $O/*/t_agent_info.csv
find $O -name "*.csv" | xargs ls -la | awk '{print $5, $9}'
tail -2 $O/train108.log | cut -c1-300; tail -1 $O/infer.log | cut -c1-600; tail -1 $O/c3.log | cut -c1-200
