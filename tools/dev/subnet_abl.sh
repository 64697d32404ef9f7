#!/bin/bash
# dev: timing-only ablations of the fused subnet kernel
for v in base a1 a2 a4 a8 a3 a7; do
  if [ $v = base ]; then lib=""; else lib="CVPCE_LIB=$PWD/tools/dev/ab/lib_gs_$v.so"; fi
  echo "== $v"; env $lib python tools/dev/subnet_time.py 2>&1 | grep "N=8"
done
