cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_c8x.py tests/test_gpu_c8.py tests/test_gpu_lp.py -x -q -m gpu 2>&1 | tail -3
timeout 200 python tools/c8x_time.py 5 3 0 2>&1 | grep "x "
O=gpurun_out/grp; rm -rf $O; mkdir -p $O
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/tf -o t -- python3 tools/c8x_one.py 4 64 64 148 5 0 3 > $O/tf.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/tw -o t -- python3 tools/c8x_one.py 4 64 64 148 5 0 3 > $O/tw.log 2>&1
python3 tools/pmc_raw.py $O/tf k_conv_h; python3 tools/pmc_raw.py $O/tw k_conv_h
rm -rf $O
