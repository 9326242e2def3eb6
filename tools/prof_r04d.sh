#!/bin/bash
# Round-4 profiles after the two-term form became the default: ONE call on one MI355X.  Every profiler run under its own timeout.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04d; mkdir -p $O
T="timeout 600"
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -o t -- python3 bench.py --workload train --steps 10 --warmup 3 --no-cpu-baseline > $O/train.log 2>&1
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/infer -o t -- python3 bench.py --workload infer --steps 1 --warmup 1 --no-cpu-baseline > $O/infer.log 2>&1
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/athena -o t -- python3 bench.py --workload train --model athena --data structured --steps 6 --warmup 3 --no-cpu-baseline > $O/athena.log 2>&1
$T rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq -o t -- python3 tools/h2_check.py quick > $O/sq.log 2>&1
python3 tools/pmc_summary.py $O/sq $O/h2_sq_counters.csv k_conv_s3x > /dev/null
$T rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o t -- python3 tools/h2_check.py quick > $O/fetch.log 2>&1
$T rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o t -- python3 tools/h2_check.py quick > $O/write.log 2>&1
python3 tools/pmc_raw.py $O/fetch k_conv_s3x > $O/h2_fetch.txt 2>&1
python3 tools/pmc_raw.py $O/write k_conv_s3x > $O/h2_write.txt 2>&1
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "t_agent_info.csv" -delete
timeout 300 python3 tools/h2_check.py > $O/h2_layers.txt 2>&1
timeout 300 python3 tools/p2d_check.py > $O/p2d_layers.txt 2>&1
for m in 2 3 2 3; do echo "athena 108^3 structured NC_SPLIT_TERMS=$m $(NC_SPLIT_TERMS=$m timeout 600 python3 bench.py --workload train --model athena --data structured --no-cpu-baseline --steps 8 --warmup 3 2>/dev/null | python3 -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('ms_per_step %.2f' % j['ms_per_step'], 'first-step G_A %.5f' % j['config']['first_step_losses']['G_A'])")" >> $O/ab_terms.txt; done
for m in 2 3 2 3; do echo "apollo 4x148^3 bf16 (configs[3], 16-bit path: unaffected) NC_SPLIT_TERMS=$m $(NC_SPLIT_TERMS=$m timeout 600 python3 bench.py --workload train --crop 148 --batch 4 --precision bf16 --no-cpu-baseline --steps 6 --warmup 3 2>/dev/null | python3 -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('ms_per_step %.2f' % j['ms_per_step'])")" >> $O/ab_terms.txt; done
cat $O/ab_terms.txt; head -12 $O/train/t_kernel_stats.csv | cut -c1-160; tail -5 $O/h2_sq_counters.csv; cat $O/h2_fetch.txt | tail -3; cat $O/h2_write.txt | tail -3
