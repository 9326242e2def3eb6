for a in base a1 a64 a65; do
  if [ $a = base ]; then unset NC_HIP_LIB; else export NC_HIP_LIB=$PWD/neuroclear_amd/csrc/abl/libnc_hip_conv_c8x_$a.so; fi
  echo "== $a"; timeout 120 python tools/c8x_time.py 5 2 2 2>&1 | grep "x "
done
