# Round-3 evidence in one GPU call: kernel stats of the default train / inference / Athena runs, SQ counters of the dominant kernels
# (with kernel names), HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes).  Outputs under gpurun_out/r03; the summaries are copied
# into profiles/ by hand (tools/pmc_summary.py, tools/pmc_aggregate.py, tools/pmc_cube.py).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train108 -o t -- python3 bench.py --workload train --steps 10 --warmup 3 --no-cpu-baseline > $O/train108.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/infer -o t -- python3 bench.py --workload infer --steps 1 --warmup 1 --no-cpu-baseline > $O/infer.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/athena -o t -- python3 bench.py --workload train --model athena --data structured --steps 6 --warmup 3 --no-cpu-baseline > $O/athena.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -o t -- python3 bench.py --workload train --crop 148 --batch 4 --precision bf16 --steps 4 --warmup 2 --no-cpu-baseline > $O/c3.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq -o t -- python3 tools/pmc_run_split.py > $O/sq.log 2>&1
python3 tools/pmc_summary.py $O/sq $O/pmc_split_sq_counters.csv k_ > /dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/tf -o t -- python3 bench.py --workload train --steps 2 --warmup 1 --no-cpu-baseline --no-prof > $O/tf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/tw -o t -- python3 bench.py --workload train --steps 2 --warmup 1 --no-cpu-baseline --no-prof > $O/tw.log 2>&1
python3 tools/pmc_aggregate.py $O/tf $O/tw $O/pmc_train.json > /dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/if -o t -- python3 bench.py --workload infer --volume 300 --steps 1 --warmup 1 --no-cpu-baseline --no-prof > $O/if.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/iw -o t -- python3 bench.py --workload infer --volume 300 --steps 1 --warmup 1 --no-cpu-baseline --no-prof > $O/iw.log 2>&1
python3 tools/pmc_cube.py $O/if $O/iw > $O/pmc_cube.json
rm -rf $O/tf $O/tw $O/if $O/iw $O/sq/t_counter_collection.csv
find $O -name "*kernel_trace.csv" -delete; find $O -name "t_agent_info.csv" -delete
ls -la $O $O/*/ | head -40
for f in train108 infer athena c3; do tail -1 $O/$f.log | cut -c1-400; done
cat $O/pmc_cube.json; head -c 1500 $O/pmc_train.json
