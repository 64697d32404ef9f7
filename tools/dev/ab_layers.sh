#!/bin/bash
# dev tool: same-call A/B of two library builds on the REAL activations of chosen VGG layers, then the kernel tests
# usage: ab_layers.sh <other lib> <layers, comma separated> [pytest -k expression]
set -e
python tools/dev/real_layer_bench.py capture 2>&1 | tail -1
for r in 1 2 3; do
  echo "== other"; CVPCE_LIB=$1 python tools/dev/real_layer_bench.py time $2 2>&1 | grep -v "^W\|^E"
  echo "== this";  python tools/dev/real_layer_bench.py time $2 2>&1 | grep -v "^W\|^E"
done
if [ -n "$3" ]; then python -m pytest tests/test_gpu_kernels.py -q -x -k "$3" 2>&1 | tail -2; fi
