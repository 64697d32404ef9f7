#!/bin/bash
# dev tool: build side libraries with compile-time ablation flags of the conv kernel (see CVPCE_DBG in conv_igemm.hip)
set -e
cd "$(dirname "$0")/../cvpce_amd/csrc"
mkdir -p build
for d in "$@"; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -DCVPCE_DBG=$d -c conv_igemm.hip -o build/conv_igemm_dbg$d.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/conv_igemm_dbg$d.o build/vgg_stem.o build/elementwise.o build/preproc.o build/detect.o build/match.o -o ../libcvpce_hip_dbg$d.so
done
