#!/bin/bash
# dev tool: the detector with the pointwise kernel's operand ring 2 / 3 / 4 deep, and with every eligible 1x1 conv through it
for rep in 1 2; do
for v in nb2 nb3 this; do
  lib=$PWD/tools/dev/ab/lib_$v.so; [ $v == this ] && lib=$PWD/cvpce_amd/libcvpce_hip.so
  echo -n "$v          "; CVPCE_LIB=$lib python tools/dev/run_detector.py 8 200 40
  echo -n "$v any-shape "; CVPCE_CONV1X1_ANY=1 CVPCE_LIB=$lib python tools/dev/run_detector.py 8 200 40
done
done
python tools/dev/bench_lateral.py 8
