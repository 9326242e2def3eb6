cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_c8x; rm -rf $O; mkdir -p $O
A="4 64 64 148 3 2 4"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS --output-format csv -d $O/p1 -o t -- python3 tools/c8x_one.py $A > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc TA_TA_BUSY TA_BUFFER_TOTAL_CYCLES TA_BUFFER_WAVEFRONTS TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES --output-format csv -d $O/p2 -o t -- python3 tools/c8x_one.py $A > $O/p2.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM GRBM_GUI_ACTIVE --output-format csv -d $O/p3 -o t -- python3 tools/c8x_one.py $A > $O/p3.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $O/p4 -o t -- python3 tools/c8x_one.py $A > $O/p4.log 2>&1
for p in p1 p2 p3 p4; do python3 tools/pmc_raw.py $O/$p k_conv_c8x; done > $O/summary.txt 2>&1
cat $O/summary.txt; tail -3 $O/p2.log
find $O -name "*.csv" -delete
