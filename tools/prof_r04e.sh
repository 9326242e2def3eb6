#!/bin/bash
# Final round-4 tree: kernel stats of the three bench workloads + the default bench line.  ONE call, every profiler run under its own timeout.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04f; mkdir -p $O
T="timeout 600"
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -o t -- python3 bench.py --workload train --steps 10 --warmup 3 --no-cpu-baseline > $O/train.log 2>&1
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/infer -o t -- python3 bench.py --workload infer --steps 1 --warmup 1 --no-cpu-baseline > $O/infer.log 2>&1
$T rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq -o t -- python3 tools/h2_check.py quick > $O/sq.log 2>&1
python3 tools/pmc_summary.py $O/sq $O/h2_sq_counters.csv k_conv_s3x > /dev/null
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "t_agent_info.csv" -delete
timeout 300 python3 tools/h2_check.py > $O/h2_layers.txt 2>&1
timeout 900 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
grep "^{" $O/train.log | cut -c1-200; grep "^{" $O/infer.log | cut -c1-200; cut -c1-300 $O/bench_default.json; tail -4 $O/h2_sq_counters.csv
