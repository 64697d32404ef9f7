#!/bin/bash
# dev: timing-only ablations of match_big_kernel's K loop (results of the ablated builds are wrong by construction), same call
for i in 1 2; do
  for v in "" abl1 abl2 abl3; do
    lib=""; [ -n "$v" ] && lib=tools/dev/ab/lib_$v.so
    echo "${v:-shipped}: $(CVPCE_LIB=$lib timeout -k 10 100 python tools/dev/bench_match.py 1600,10000,1024 800,10000,1024 2>&1 | grep -o ' x [0-9]* x [0-9]*: auto [0-9.]*' | tr '\n' ' ')"
  done
done
