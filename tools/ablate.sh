#!/bin/bash
# dev tool: build side libraries with compile-time ablation flags (-DCVPCE_DBG=<flags>) of ONE kernel source
# usage: ablate.sh <source-stem, e.g. conv_igemm | vgg_stem2 | conv3x3_halo2> flags...
#   -> cvpce_amd/libcvpce_hip_<stem>_dbg<flags>.so ; select it with CVPCE_LIB=<path>
set -e
cd "$(dirname "$0")/../cvpce_amd/csrc"
mkdir -p build
make -s
which=$1; shift
others=$(ls build/*.o | grep -v "_dbg" | grep -v "build/$which.o")
for d in "$@"; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -DCVPCE_DBG=$d -c $which.hip -o build/${which}_dbg$d.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $others build/${which}_dbg$d.o -o ../libcvpce_hip_${which}_dbg$d.so
done
