#!/bin/bash
# Timing experiments only: builds libnc_hip variants of the tap-stream kernel with extra -D flags into neuroclear_amd/csrc/abl/
# (git-ignored, ships to the GPU box).  usage: tools/s3x_variant.sh <tag> <flags...>   e.g.  tools/s3x_variant.sh stamp -DNC_S3X_STAMP
set -e
cd "$(dirname "$0")/../neuroclear_amd/csrc"
make -j8 >/dev/null
mkdir -p abl
tag=$1; shift
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-int-to-pointer-cast -Wno-inline-asm"
/opt/rocm/bin/hipcc $FL "$@" -c conv_s3x.hip -o abl/conv_s3x_$tag.o
objs=$(ls *.o | grep -v "^conv_s3x.o$")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o abl/libnc_hip_s3x_$tag.so $objs abl/conv_s3x_$tag.o
ls -la abl/libnc_hip_s3x_$tag.so
