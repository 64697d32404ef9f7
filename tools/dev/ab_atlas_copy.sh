#!/bin/bash
# dev tool: detector-only bench with the one-launch atlas pack / unpack against the slice copies (CVPCE_ATLAS_COPY=0)
set -e
run() { python bench.py --workload detector --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-peaks 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"; }
for i in 1 2 3; do
  echo -n "slices   "; CVPCE_ATLAS_COPY=0 run
  echo -n "one-shot "; run
done
