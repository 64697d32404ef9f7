#!/bin/bash
# dev tool: whole-pipeline bench with the crop kernel writing whole crops (CVPCE_CROP_CONTENT=0) against content only (default), alternating
set -e
run() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-peaks --no-workloads --no-parity --no-h2d --no-precision-leg --no-clocks "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d.get('value_lists_off'), d.get('value_planted_boxes'))"; }
for i in 1 2 3; do
  echo -n "whole crops:  "; CVPCE_CROP_CONTENT=0 run
  echo -n "content only: "; run
done
