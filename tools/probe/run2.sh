timeout 1200 python -m pytest tests/test_gpu_c8x.py -x -q -m gpu 2>&1 | tail -30 > gpurun_out/c8x_test.log
NC_C8X=2 timeout 1200 python -m pytest tests/test_gpu_c8.py tests/test_gpu_lp.py -x -q -m gpu 2>&1 | tail -15 >> gpurun_out/c8x_test.log
cat gpurun_out/c8x_test.log
