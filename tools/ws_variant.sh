#!/bin/bash
# Timing experiments only: builds libnc_hip variants of conv_split.hip (the weight-gradient kernels) with extra -D flags into
# neuroclear_amd/csrc/abl/ (git-ignored, ships to the GPU box).  usage: tools/ws_variant.sh <tag> <flags...>
set -e
cd "$(dirname "$0")/../neuroclear_amd/csrc"
make -j8 >/dev/null
mkdir -p abl
tag=$1; shift
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-int-to-pointer-cast -Wno-inline-asm"
/opt/rocm/bin/hipcc $FL "$@" -c conv_split.hip -o abl/conv_split_$tag.o
objs=$(ls *.o | grep -v "^conv_split.o$")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o abl/libnc_hip_ws_$tag.so $objs abl/conv_split_$tag.o
ls -la abl/libnc_hip_ws_$tag.so
