#!/bin/bash
# dev tool: TA / TCP counters of the pointwise conv kernel on the detector's 1x1 shapes (tools/dev/bench_1x1.py)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/pmc_1x1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TA_BUSY_avr TA_BUFFER_TOTAL_CYCLES_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum --output-format csv -d gpurun_out/pmc_1x1/a -- python3 tools/dev/bench_1x1.py > /dev/null 2>&1 || echo pass a failed
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum --output-format csv -d gpurun_out/pmc_1x1/b -- python3 tools/dev/bench_1x1.py > /dev/null 2>&1 || echo pass b failed
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc_1x1/c -- python3 tools/dev/bench_1x1.py > /dev/null 2>&1 || echo pass c failed
python3 - <<'PY'
import csv, glob, collections
for p in 'abc':
    fs = glob.glob(f'gpurun_out/pmc_1x1/{p}/**/*counter_collection.csv', recursive=True)
    if not fs:
        print(p, 'no file'); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        if 'conv1x1' in r['Kernel_Name']:
            agg[(r['Kernel_Name'][:44], r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in agg.items():
        print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
