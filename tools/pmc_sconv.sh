# PMC passes over tools/pmc_sconv.py (one counter group per pass; no trace domains besides --kernel-trace)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02/pmc_sconv
mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/a -o t -- python3 tools/pmc_sconv.py > $O/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $O/b -o t -- python3 tools/pmc_sconv.py > $O/b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM --output-format csv -d $O/c -o t -- python3 tools/pmc_sconv.py > $O/c.log 2>&1
ls $O/*
