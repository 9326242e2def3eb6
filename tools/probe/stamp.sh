export NC_HIP_LIB=$PWD/neuroclear_amd/csrc/abl/libnc_hip_conv_c8x_stamp.so
python tools/c8x_stamp.py 4 64 64 148 3
NC_C8X_GRID=256 python tools/c8x_stamp.py 4 64 64 148 3
python tools/c8x_stamp.py 4 64 64 148 5
