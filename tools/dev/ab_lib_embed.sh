#!/bin/bash
# dev A/B: per-layer embedder times and the detector with another build of the library ($1) vs this one, alternating
for i in 1 2 3; do
  echo "other: $(CVPCE_LIB=$1 timeout -k 10 200 python tools/dev/embed_layers.py 2>&1 | grep -E 'skip=True|@' | awk '{printf "%s %s | ", $1, $2}')"
  echo "this:  $(timeout -k 10 200 python tools/dev/embed_layers.py 2>&1 | grep -E 'skip=True|@' | awk '{printf "%s %s | ", $1, $2}')"
done
for i in 1 2; do
  echo -n "det other "; CVPCE_LIB=$1 python tools/dev/run_detector.py 8 200 40
  echo -n "det this  "; python tools/dev/run_detector.py 8 200 40
done
