#!/bin/bash
# kernel-only times of the tap-stream kernel (two-term and three-term) for the ablation builds in csrc/abl/: rocprofv3 --stats of tools/h2_check.py quick
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/abl; mkdir -p $O
for v in base nob noa nobar nodma; do
  if [ $v = base ]; then unset NC_HIP_LIB; else export NC_HIP_LIB=$GRAFT_REPO_ROOT/neuroclear_amd/csrc/abl/libnc_hip_s3x_$v.so; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$v -o t -- python3 tools/h2_check.py quick > $O/$v.log 2>&1
  echo "== $v"; grep "k_conv_s3x<3, 8" $O/$v/t_kernel_stats.csv | cut -d, -f1-4 | sed 's/void nc::(anonymous namespace):://'
done
