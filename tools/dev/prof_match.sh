#!/bin/bash
# dev tool (through gpurun, from the repo root): rocprofv3 evidence for the distance-GEMM kernels of BASELINE configs[3] ->
#   gpurun_out/match_<P>_<D>/{stats,fetch,write,sq,tcc}; counter passes separate from each other and never combined with system traces
set -e
R=$PWD; O=$R/gpurun_out
cd /tmp; export TMPDIR=/tmp
for case in "200 10000 512" "1600 10000 1024"; do
  set -- $case
  d=$O/match_$1_$3; rm -rf $d
  rocprofv3 --kernel-trace --stats --output-format csv -d $d/stats -- python3 $R/tools/dev/run_match.py $1 $2 $3 30 > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $d/fetch -- python3 $R/tools/dev/run_match.py $1 $2 $3 30 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $d/write -- python3 $R/tools/dev/run_match.py $1 $2 $3 30 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $d/sq -- python3 $R/tools/dev/run_match.py $1 $2 $3 30 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $d/tcc -- python3 $R/tools/dev/run_match.py $1 $2 $3 30 > /dev/null 2>&1
  echo "case $case done"
done
find $O/match_* -name '*_kernel_trace.csv' -size +20M -delete
du -sh $O/match_*
