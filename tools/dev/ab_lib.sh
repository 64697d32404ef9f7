#!/bin/bash
# dev tool: same-call A/B of two builds of the C-ABI library on the detector-only and pipeline benches
# usage: ab_lib.sh <path of the other library> [pipeline]
set -e
other=$1
run() { python bench.py "${@}" --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-peaks 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"; }
for i in 1 2 3; do
  echo -n "det other "; CVPCE_LIB=$other run --workload detector
  echo -n "det this  "; run --workload detector
done
if [ "$2" == "pipeline" ]; then
for i in 1 2; do
  echo -n "pipe other "; CVPCE_LIB=$other run --no-parity --no-h2d
  echo -n "pipe this  "; run --no-parity --no-h2d
done
fi
