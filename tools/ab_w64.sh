cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06w; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_h2.py -q -m gpu -x -k "w64 or layer_against or epilogue" > $O/tests2.log 2>&1; tail -5 $O/tests2.log
for i in 1 2; do
NC_S3X_W64=0 timeout 300 python tools/h2_time.py 2 2>&1 | tail -1
NC_S3X_W64=1 timeout 300 python tools/h2_time.py 2 2>&1 | tail -1
done | tee $O/h2_time.log
