#!/bin/bash
# Collect the rocprofv3 evidence of one build on the GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh <tag>   ->  gpurun_out/prof_<tag> (kernel trace + stats of bench.py --steps 5 --warmup 2)
#                                         gpurun_out/pmc_<tag>/{fetch,write,sq,tcc} (separate counter passes, bench.py --steps 1 --warmup 1)
# then, back in the container:  python tools/summarise_profiles.py <tag> prof_<tag> pmc_<tag>
# Counter passes are never combined with system traces (only --kernel-trace); the program sits directly after `--`.
set -e
tag=$1
R=$PWD; O=$R/gpurun_out; B=$R/bench.py
cd /tmp; export TMPDIR=/tmp
if [ -z "$SKIP_STATS" ]; then
# (the stats pass runs the headline steps and the roofline leg ONLY -- no lists-off / planted-box / fitted-scene / precision legs, which launch
#  the same template instances on other work: every LIST / STRIP launch in the CSV is then a launch of the headline step, and
#  tools/roofline_check.py can reproduce `roofline.achieved` from the CSV and the details file alone)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -- python3 $B --steps 5 --warmup 2 --no-coheadlines --no-workloads --no-parity --no-cpu-baseline --no-precision-leg --no-h2d --no-peaks --no-clocks --details $O/prof_${tag}_details.json > $O/prof_$tag.log 2>&1
echo "stats pass done"
fi
# (the counter passes end with ONE headline step: every side leg off, so that "the last pipeline pass" of the trace is that step)
Q="--steps 1 --warmup 1 --windows 1 --no-roofline --no-cpu-baseline --no-parity --no-h2d --no-peaks --no-workloads --no-precision-leg --no-coheadlines"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_$tag/fetch -- python3 $B $Q > /dev/null 2>&1
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_$tag/write -- python3 $B $Q > /dev/null 2>&1
echo "write pass done"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_$tag/sq -- python3 $B $Q > /dev/null 2>&1
echo "sq pass done"
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_$tag/tcc -- python3 $B $Q > /dev/null 2>&1
echo "tcc pass done"
[ -z "$SKIP_STATS" ] && grep -o '{"metric.*' $O/prof_$tag.log | tail -1 > $O/prof_$tag.json
# the traces are large: keep only what the summary needs (gpurun merges at most 64 MiB back)
find $O/prof_$tag -name '*_kernel_trace.csv' -size +40M -delete
du -sh $O/prof_$tag $O/pmc_$tag
