#!/bin/bash
# dev tool: the round-2 ablation table of conv3x3_halo2 (profiles/r02_ablation_halo2.md) on the current build
for d in 0 4 8 12 32 44; do
  lib=$PWD/cvpce_amd/libcvpce_hip_conv3x3_halo2_dbg$d.so; [ $d == 0 ] && lib=$PWD/cvpce_amd/libcvpce_hip.so
  echo "dbg $d: $(CVPCE_LIB=$lib timeout -k 10 120 python tools/bench_conv.py --layers vgg3_1,vgg3_2,vgg4_2,vgg5_1 --relu-input 2>&1 | grep -E 'vgg' | awk '{printf "%s %s %s | ", $1, $2, $3}')"
done
