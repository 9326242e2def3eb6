cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_c8x; rm -rf $O; mkdir -p $O
A="4 64 64 148 3 2 4"
timeout 150 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/p1 -o t -- python3 tools/c8x_one.py $A > $O/p1.log 2>&1
python3 tools/pmc_summary.py $O/p1 $O/sq.csv k_conv_c8x | tail -3
timeout 150 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/p2 -o t -- python3 tools/c8x_one.py $A > $O/p2.log 2>&1
python3 tools/pmc_raw.py $O/p2 k_conv_c8x
timeout 150 rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/p3 -o t -- python3 tools/c8x_one.py $A > $O/p3.log 2>&1
python3 tools/pmc_raw.py $O/p3 k_conv_c8x
A="4 64 64 148 3 0 4"
timeout 150 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/p4 -o t -- python3 tools/c8x_one.py $A > $O/p4.log 2>&1
python3 tools/pmc_summary.py $O/p4 $O/sq_h.csv k_conv_h | tail -3
find $O -name "*.csv" -path "*/p?/*" -delete
