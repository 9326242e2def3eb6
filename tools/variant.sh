#!/bin/bash
# Timing experiments only: builds a libnc_hip variant with ONE source recompiled under extra -D flags into neuroclear_amd/csrc/abl/
# (git-ignored, ships to the GPU box; select it with NC_HIP_LIB).  usage: tools/variant.sh <source without .hip> <tag> <flags...>
#   e.g.  tools/variant.sh conv_c8x nobar -DNC_C8X_ABL=1   ->  neuroclear_amd/csrc/abl/libnc_hip_conv_c8x_nobar.so
set -e
cd "$(dirname "$0")/../neuroclear_amd/csrc"
make -j8 >/dev/null
mkdir -p abl
src=$1; tag=$2; shift; shift
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-int-to-pointer-cast -Wno-inline-asm"
/opt/rocm/bin/hipcc $FL "$@" -c $src.hip -o abl/${src}_$tag.o
objs=$(ls *.o | grep -v "^$src.o$")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o abl/libnc_hip_${src}_$tag.so $objs abl/${src}_$tag.o
ls -la abl/libnc_hip_${src}_$tag.so
