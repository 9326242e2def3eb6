# Round 6: the operand-delivery micro-benchmark (tools/mfma_feed.hip, modes 0-13) plain, then under the SQ counters (clock, MFMA pipe busy,
# wait shares per mode) and the LDS counters.  Outputs under gpurun_out/r06; summaries are copied into profiles/ by hand.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06
mkdir -p $O
T="timeout 300"
$T ./tools/mfma_feed.bin > $O/mfma_feed.txt 2>&1
$T rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/feed_sq -o t -- ./tools/mfma_feed.bin > $O/feed_sq.log 2>&1
python3 tools/pmc_summary.py $O/feed_sq $O/mfma_feed_sq_counters.csv > /dev/null
$T rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $O/feed_lds -o t -- ./tools/mfma_feed.bin > $O/feed_lds.log 2>&1
python3 tools/pmc_raw.py $O/feed_lds > $O/mfma_feed_lds_counters.txt 2>&1
rm -rf $O/feed_sq $O/feed_lds
cat $O/mfma_feed.txt; cat $O/mfma_feed_sq_counters.csv
