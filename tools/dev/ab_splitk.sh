#!/bin/bash
# dev tool: detector-only bench (configs[1]: 4 images, dpi 1000; then the headline's 8 images, dpi 200) with split-K convs off / on, alternating
set -e
run() { python bench.py --workload detector --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-peaks "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"; }
for i in 1 2 3; do
  echo -n "configs[1] splitk=0: "; CVPCE_CONV_SPLITK=0 run
  echo -n "configs[1] splitk=1: "; run
done
for i in 1 2; do
  echo -n "8 images splitk=0: "; CVPCE_CONV_SPLITK=0 run --images-per-gpu 8 --detections-per-img 200
  echo -n "8 images splitk=1: "; run --images-per-gpu 8 --detections-per-img 200
done
