#!/bin/bash
# kernel-only times (rocprofv3 --stats of tools/h2_check.py) of several library builds, alternating with the default one
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in base "$@" base "$@"; do
  if [ $v = base ]; then unset NC_HIP_LIB; else export NC_HIP_LIB=$GRAFT_REPO_ROOT/neuroclear_amd/csrc/abl/libnc_hip_s3x_$v.so; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abv_$v -o t -- python3 tools/h2_check.py > gpurun_out/abv_$v.log 2>&1
  echo "== $v"; python3 - <<P
import csv
for r in csv.DictReader(open('gpurun_out/abv_$v/t_kernel_stats.csv')):
    if 'k_conv_s3x<' in r['Name'] and ', 2>' in r['Name'] and ('8, 2' in r['Name'] or '7, 2' in r['Name']): print(r['Name'][38:58], r['Calls'], 'avg %.1f us min %.1f' % (float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
P
done
