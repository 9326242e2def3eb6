#!/bin/bash
# tile-order change of k_conv_s3x: tests, kernel times and HBM bytes (tools/h2_check.py quick under --pmc FETCH_SIZE / WRITE_SIZE), step time
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/order; mkdir -p $O
timeout 600 python3 -m pytest tests/test_gpu_h2.py tests/test_gpu_split.py -q 2>&1 | tail -2
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o t -- python3 tools/h2_check.py quick > $O/fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o t -- python3 tools/h2_check.py quick > $O/write.log 2>&1
python3 tools/pmc_raw.py $O/fetch k_conv_s3x 2>&1 | grep -A1 "3, 8, 2\|3, 8, 3"
python3 tools/pmc_raw.py $O/write k_conv_s3x 2>&1 | grep -A1 "3, 8, 2\|3, 8, 3"
rm -rf $O/fetch $O/write
for i in 1 2 3; do timeout 600 python3 bench.py --workload train --no-cpu-baseline --steps 8 --warmup 3 2>/dev/null | python3 -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('ms_per_step %.2f' % j['ms_per_step'])"; done
timeout 600 python3 bench.py --workload infer --no-cpu-baseline --steps 1 --warmup 1 2>/dev/null | grep "^{" | python3 -c "import sys,json; j=json.loads(sys.stdin.readline()); print('infer', j['ms_per_step'])"
