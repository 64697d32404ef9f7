#!/bin/bash
# dev tool: kernel durations (rocprofv3 --stats) of tools/dev/bench_splitk.py -- the wall-clock figures of that tool are bound by the
# host's ~20 us per op call for the short launches
set -e
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_splitk -- python3 $R/tools/dev/bench_splitk.py 4 > $R/gpurun_out/prof_splitk.log 2>&1
f=$(ls $R/gpurun_out/prof_splitk/*/*kernel_trace.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
# launches in order; group consecutive runs of the same (kernel, grid)
seq = []
for r in rows:
    name = r['Kernel_Name'].split('(')[0][-60:]
    key = (name, r.get('Grid_Size') or r.get('Grid_Size_X'))
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000
    if seq and seq[-1][0] == key: seq[-1][1].append(d)
    else: seq.append((key, [d]))
for key, ds in seq:
    if len(ds) >= 20 and ('igemm' in key[0] or 'splitk' in key[0]):
        ds = sorted(ds); print(f'{key[0]:62s} grid {key[1]:>8s}  n {len(ds):3d}  median {ds[len(ds)//2]:7.1f} us  min {ds[0]:7.1f}')
PY
