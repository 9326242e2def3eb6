#!/bin/bash
# Round 6, final tree: (1) SQ counters (clock, MFMA pipe busy, wait shares) per matrix-kernel instantiation in the real mixes -- the train step and a
# one-stream 300^3 inference (ST forms) -- and per micro-benchmark mode; (2) HBM bytes per launch of the kernel classes and per 140^3 cube
# (separate --pmc passes, each under its own timeout).  Outputs under gpurun_out/r06pmc; summaries are copied into profiles/ by hand.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06pmc
rm -rf $O; mkdir -p $O
T="timeout 900"
SQ="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
$T rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/sqt -o t -- python3 bench.py --workload train --steps 2 --warmup 1 --no-cpu-baseline --no-prof > $O/sqt.log 2>&1
python3 tools/pmc_group.py $O/sqt $O/train_sq_counters.csv k_conv_s3w k_conv_s3x k_wgrad_s3x k_conv_mfma k_wgrad_c1 k_dgrad_to1 k_conv_gemm k_convT > /dev/null
NC_INFER_STREAMS=1 $T rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/sqi -o t -- python3 bench.py --workload infer --volume 300 --steps 1 --warmup 1 --no-cpu-baseline --no-prof > $O/sqi.log 2>&1
python3 tools/pmc_group.py $O/sqi $O/infer_sq_counters.csv k_conv_s3w k_conv_s3x k_convT k_conv_c1k3 > /dev/null
rm -rf $O/sqt $O/sqi
$T rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/tf -o t -- python3 bench.py --workload train --steps 5 --warmup 2 --no-cpu-baseline --no-prof > $O/tf.log 2>&1
$T rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/tw -o t -- python3 bench.py --workload train --steps 5 --warmup 2 --no-cpu-baseline --no-prof > $O/tw.log 2>&1
python3 tools/pmc_aggregate.py $O/tf $O/tw $O/train.json > /dev/null
$T rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/if -o t -- python3 bench.py --workload infer --volume 480 --steps 1 --warmup 1 --no-cpu-baseline --no-prof > $O/if.log 2>&1
$T rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/iw -o t -- python3 bench.py --workload infer --volume 480 --steps 1 --warmup 1 --no-cpu-baseline --no-prof > $O/iw.log 2>&1
python3 tools/pmc_cube.py $O/if $O/iw > $O/cube.json
rm -rf $O/tf $O/tw $O/if $O/iw
cat $O/train_sq_counters.csv; cat $O/infer_sq_counters.csv; head -c 3000 $O/train.json; cat $O/cube.json
