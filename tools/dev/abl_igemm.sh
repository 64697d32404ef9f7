#!/bin/bash
# dev tool: timing-only ablations of the register-staged conv kernel on the detector's small-map layers (unsplit launches)
# build first (in the container): tools/ablate.sh conv_igemm 1024 2048 4096 8192 3072 12288
set -e
for d in "" 1024 2048 4096 8192 3072 12288; do
  if [ -z "$d" ]; then echo "== shipped"; unset CVPCE_LIB; else echo "== CVPCE_DBG=$d"; export CVPCE_LIB=$PWD/cvpce_amd/libcvpce_hip_conv_igemm_dbg$d.so; fi
  CVPCE_CONV_SPLITK=0 python tools/dev/bench_splitk.py 4 2>&1 | sed -n '1p;4p;6p' | cut -c1-70
done
