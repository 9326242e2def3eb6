timeout 900 python -m pytest tests/test_gpu_c8x.py -x -q -m gpu 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_grad_fp64.py -x -q -m gpu -s 2>&1 | grep -v "^$" | tail -80
