#!/bin/bash
# dev tool: build a variant of the C-ABI library with one source recompiled under extra -D flags
# usage: build_variant.sh <source.hip> <out name> <-Dflags...>   -> tools/dev/ab/lib_<name>.so (git-ignored, travels with gpurun)
set -e
cd "$(dirname "$0")/../../cvpce_amd/csrc"
src=$1; name=$2; shift 2
make -s
mkdir -p ../../tools/dev/ab
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-result "$@" -c $src -o ../../tools/dev/ab/${src%.hip}_$name.o
objs=$(ls build/*.o | grep -v "build/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs ../../tools/dev/ab/${src%.hip}_$name.o -o ../../tools/dev/ab/lib_$name.so
echo built tools/dev/ab/lib_$name.so
