#!/bin/bash
# dev tool: pipeline bench with an environment switch off / on, alternating, with result digests.  usage: ab_env.sh VAR [rounds]
set -e
d() { python bench.py --verify --steps 12 --warmup 2 --no-parity --no-h2d --no-cpu-baseline --no-roofline --no-peaks 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['verify']['digest'][:16])"; }
for i in $(seq 1 ${2:-3}); do
  echo -n "$1=0: "; env $1=0 python -c "pass"; export $1=0; d; unset $1
  echo -n "default: "; d
done
