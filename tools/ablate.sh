#!/bin/bash
# dev tool: build side libraries with compile-time ablation flags of the conv kernel (see CVPCE_DBG in conv_igemm.hip)
set -e
cd "$(dirname "$0")/../cvpce_amd/csrc"
mkdir -p build
# usage: ablate.sh <conv|stem> flags...
which=$1; shift
for d in "$@"; do
  if [ "$which" = stem ]; then
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -DCVPCE_DBG=$d -c vgg_stem.hip -o build/vgg_stem_dbg$d.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/conv_igemm.o build/vgg_stem_dbg$d.o build/elementwise.o build/preproc.o build/detect.o build/match.o -o ../libcvpce_hip_stem$d.so
  else
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -DCVPCE_DBG=$d -c conv_igemm.hip -o build/conv_igemm_dbg$d.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/conv_igemm_dbg$d.o build/vgg_stem.o build/elementwise.o build/preproc.o build/detect.o build/match.o -o ../libcvpce_hip_dbg$d.so
  fi
done
