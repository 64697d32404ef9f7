#!/bin/bash
# dev tool: pipeline bench at several embed chunk sizes (crops per pass of the VGG schedule), alternating, one gpurun call
set -e
run() { python bench.py --no-parity --no-h2d --steps 12 --warmup 2 --no-cpu-baseline --no-roofline --no-peaks 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"; }
for i in 1 2; do
  for b in "$@"; do echo -n "embed batch $b: "; CVPCE_EMBED_BATCH=$b run; done
done
