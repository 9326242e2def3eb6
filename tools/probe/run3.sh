timeout 900 python -m pytest tests/test_gpu_c8x.py -x -q -m gpu 2>&1 | tail -5
timeout 300 python tools/c8x_time.py 5 2>&1 | grep "x "
NC_C8X_GRID=256 timeout 300 python tools/c8x_time.py 5 1 2 2>&1 | grep "x "
NC_HIP_LIB=$PWD/neuroclear_amd/csrc/abl/libnc_hip_conv_c8x_stamp.so python tools/c8x_stamp.py 4 64 64 148 3 2>&1 | grep -v amdgpu.ids
