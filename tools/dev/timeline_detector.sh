#!/bin/bash
# dev tool: launch-by-launch timeline of one graph-replayed detector pass; args: images dpi [precision]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/prof_tl
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_tl -- python3 tools/dev/run_detector.py $1 $2 6 ${3:-bf16} > gpurun_out/tl.log 2>&1 || exit 1
python3 tools/trace_timeline.py gpurun_out/prof_tl > gpurun_out/timeline_$1_$2.txt
tail -45 gpurun_out/timeline_$1_$2.txt
