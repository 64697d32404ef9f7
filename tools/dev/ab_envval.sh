#!/bin/bash
# dev tool: detector-only bench with VAR=VALUE against the default, alternating.  usage: ab_envval.sh VAR VALUE
set -e
run() { python bench.py --workload detector --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-peaks 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"; }
for i in 1 2 3; do
  echo -n "$1=$2: "; export $1=$2; run; unset $1
  echo -n "default: "; run
done
