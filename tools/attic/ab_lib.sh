#!/bin/bash
# same-box A/B of two library builds on kernel-only times (rocprofv3 --stats of tools/h2_check.py) and on the train step: default lib vs $1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ALT=$GRAFT_REPO_ROOT/neuroclear_amd/csrc/abl/$1
for v in new alt new alt; do
  if [ $v = new ]; then unset NC_HIP_LIB; else export NC_HIP_LIB=$ALT; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_$v -o t -- python3 tools/h2_check.py > gpurun_out/ab_$v.log 2>&1
  echo "== $v"; grep "k_conv_s3x<[35], [87], 2" gpurun_out/ab_$v/t_kernel_stats.csv | cut -d, -f1-4 | cut -c30-150
done
for v in new alt new alt; do
  if [ $v = new ]; then unset NC_HIP_LIB; else export NC_HIP_LIB=$ALT; fi
  echo "train $v $(timeout 600 python3 bench.py --workload train --no-cpu-baseline --steps 8 --warmup 3 2>/dev/null | python3 -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('ms_per_step %.2f' % j['ms_per_step'])")"
done
