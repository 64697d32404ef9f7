#!/bin/bash
# dev tool: crop kernel durations inside the pipeline bench (kernel trace), whole crops vs content only
set -e
R=$PWD; cd /tmp; export TMPDIR=/tmp
for v in 0 1; do
  rm -rf $R/gpurun_out/prof_crop$v
  CVPCE_CROP_CONTENT=$v rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_crop$v -- python3 $R/bench.py --steps 4 --warmup 2 --windows 1 --no-cpu-baseline --no-roofline --no-peaks --no-workloads --no-parity --no-h2d --no-precision-leg --no-clocks --no-coheadlines > $R/gpurun_out/prof_crop$v.log 2>&1
  echo "CVPCE_CROP_CONTENT=$v"; grep -h "crop_resize2\|vgg_stem2" $R/gpurun_out/prof_crop$v/*/*kernel_stats.csv | awk -F'","' '{printf "  %-60s calls %5s avg %8.1f us\n", substr($1,2,58), $2, $4/1000}'
  find $R/gpurun_out/prof_crop$v -name '*_kernel_trace.csv' -delete
done
