#!/bin/bash
# dev tool: per-kernel times of the detector's post-processing (tools/dev/bench_post.py under rocprofv3); args: images dpi [tag]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
tag=${3:-post}
rm -rf gpurun_out/prof_$tag
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 tools/dev/bench_post.py $1 $2 20 > gpurun_out/$tag.log 2>&1 || exit 1
grep -E "postprocess|digest" gpurun_out/$tag.log
python3 - "$tag" <<'PY'
import csv, glob, sys
f = glob.glob(f"gpurun_out/prof_{sys.argv[1]}/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r["Name"] for k in ("decode", "nms")):
        print(f'{r["Name"][:40]:40s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"]) / 1e3:7.1f} us  min {float(r["MinNs"]) / 1e3:7.1f}  max {float(r["MaxNs"]) / 1e3:7.1f}')
PY
