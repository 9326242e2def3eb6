#!/bin/bash
# Timing experiments only: builds libnc_hip variants with -DNC_ABLATE=<n> (fwd kernel) / -DNC_WG_ABLATE=<n> (wgrad
# kernel) into neuroclear_amd/csrc/abl/ (git-ignored, ships to the GPU box).  usage: tools/ablate_build.sh f1 f2 w1 ...
set -e
cd "$(dirname "$0")/../neuroclear_amd/csrc"
make -j8 >/dev/null
mkdir -p abl
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function"
for v in "$@"; do
  kind=${v:0:1}; n=${v:1}
  if [ "$kind" = f ]; then src=conv_mfma_fwd; def=-DNC_ABLATE=$n; else src=conv_mfma_wgrad; def=-DNC_WG_ABLATE=$n; fi
  ( /opt/rocm/bin/hipcc $FL $def -c $src.hip -o abl/${src}_$v.o
    objs=$(ls *.o | grep -v "^$src.o$")
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o abl/libnc_hip_$v.so $objs abl/${src}_$v.o ) &
done
wait
ls -la abl/*.so
