#!/bin/bash
# dev tool: tail pass folded (768 + 832) vs separate (768 + 768 + 64), alternating, with result digests
set -e
d() { python bench.py --verify --steps 12 --warmup 2 --no-parity --no-h2d --no-cpu-baseline --no-roofline --no-peaks 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['verify']['digest'][:16])"; }
for i in 1 2 3; do
  echo -n "separate tail: "; CVPCE_EMBED_MAX=768 d
  echo -n "folded tail:   "; d
done
