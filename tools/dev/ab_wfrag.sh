#!/bin/bash
# dev A/B: halo kernels with row-major weights (library built before the layout change + CVPCE_WEIGHT_ROWMAJOR=1) vs fragment-major
for i in 1 2 3; do
  echo "rowmajor: $(CVPCE_WEIGHT_ROWMAJOR=1 CVPCE_LIB=$PWD/tools/dev/ab/lib_oldw.so timeout -k 10 200 python tools/dev/embed_layers.py 2>&1 | grep -E 'skip=True|skip=False' | tr '\n' ' ')"
  echo "fragment: $(timeout -k 10 200 python tools/dev/embed_layers.py 2>&1 | grep -E 'skip=True|skip=False' | tr '\n' ' ')"
done
timeout -k 10 200 python tools/dev/embed_layers.py 2>&1 | head -12
for i in 1 2; do
  echo -n "det rowmajor "; CVPCE_WEIGHT_ROWMAJOR=1 CVPCE_LIB=$PWD/tools/dev/ab/lib_oldw.so python tools/dev/run_detector.py 8 200 40
  echo -n "det fragment "; python tools/dev/run_detector.py 8 200 40
done
