# Round-4 evidence, third part (the split-operand PatchGAN kernels): kernel stats of the Athena step, its same-box A/B (NC_P2D), the per-layer
# timings, SQ counters of the new kernels, and the full default bench (the Apollo headline is not affected: its discriminators see 1-4 planes).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c
rm -rf $O; mkdir -p $O
T="timeout 600"
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/athena -o t -- python3 bench.py --workload train --model athena --data structured --steps 6 --warmup 3 --no-cpu-baseline > $O/athena.log 2>&1
$T rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq -o t -- python3 tools/p2d_check.py > $O/sq.log 2>&1
python3 tools/pmc_summary.py $O/sq $O/p2d_sq_counters.csv k_conv_p2d > /dev/null
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "t_agent_info.csv" -delete
for m in 3 0 3 0 1; do
  echo "athena 108^3 structured NC_P2D=$m $(NC_P2D=$m timeout 600 python3 bench.py --workload train --model athena --data structured --no-cpu-baseline --steps 8 --warmup 3 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.readline()); print('ms_per_step %.2f' % j['ms_per_step'], 'first-step G_A %.5f D_B_xz %.5f' % (j['config']['first_step_losses']['G_A'], j['config']['first_step_losses']['D_B_xz']))")" >> $O/ab_athena.txt
done
timeout 400 python3 tools/p2d_check.py 2>&1 | grep -v amdgpu.ids > $O/p2d_layers.txt
timeout 900 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
cat $O/ab_athena.txt; cat $O/p2d_layers.txt | grep -A1 "^B 216\|^B 108"; cut -c1-200 $O/bench_default.json; tail -12 $O/p2d_sq_counters.csv
