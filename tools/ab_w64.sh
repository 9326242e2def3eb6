# same-box timing of k_conv_s3w variants (tools/s3x_variant.sh): default (working tree) against variant libraries under neuroclear_amd/csrc/abl
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06w; mkdir -p $O
A=$GRAFT_REPO_ROOT/neuroclear_amd/csrc/abl
timeout 900 python -m pytest tests/test_gpu_h2.py -q -m gpu -x -k "w64 or epilogue or layer_against" > $O/tests3.log 2>&1; tail -3 $O/tests3.log
for i in 1 2 3; do
for v in default ${VARIANTS:-base}; do
  if [ $v = default ]; then unset NC_HIP_LIB; else export NC_HIP_LIB=$A/libnc_hip_s3x_$v.so; fi
  timeout 300 python tools/h2_time.py 2 2>&1 | tail -1 | sed "s/^/$v /"
done; done | tee $O/h2_time_variants.log
