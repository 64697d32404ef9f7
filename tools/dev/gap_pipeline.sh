#!/bin/bash
# dev tool: idle gaps inside one whole-pipeline step (kernel trace of a short bench run)
set -e
R=$PWD; cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_gap
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_gap -- python3 $R/bench.py --steps 4 --warmup 2 --windows 1 --no-cpu-baseline --no-roofline --no-peaks --no-workloads --no-parity --no-h2d --no-precision-leg --no-clocks --no-coheadlines > $R/gpurun_out/prof_gap.log 2>&1
cd $R; python3 tools/dev/gap_report.py gpurun_out/prof_gap
find gpurun_out/prof_gap -name '*_kernel_trace.csv' -size +30M -delete
