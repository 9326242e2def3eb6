timeout 900 python -m pytest tests/test_gpu_dist.py -x -q -m gpu 2>&1 | tail -30 > gpurun_out/dist_test.log
tail -30 gpurun_out/dist_test.log
