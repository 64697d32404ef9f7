"""dev tool: reproduce bench.py's `roofline.achieved` from the rocprofv3 evidence alone.
  python tools/roofline_check.py <tag> <prof dir under gpurun_out> <details json of the SAME command>
-> profiles/<tag>_roofline_check.md: the dominant kernel's LIST and STRIP template instances as two rows (launches, total ms, average us from
the profiler's kernel-stats CSV; executed TFLOP from the details file's ConvProfile summary), the TFLOP/s that follows, and the figure the
bench line printed (HIP events around the same launches, un-profiled passes of the same command)."""
import csv, glob, json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(root, 'tools'))
tag, prof, details = sys.argv[1], sys.argv[2], sys.argv[3]
stats = glob.glob(f'{root}/gpurun_out/{prof}/*/*_kernel_stats.csv')[0]
d = json.load(open(details))
r = d['roofline']
rows = list(csv.DictReader(open(stats)))


def targs(name):
    name = name[5:] if name.startswith('void ') else name
    return [a.strip() for a in name[name.index('<') + 1:name.rindex('>')].split(',')] if '<' in name else []


def pick(pred):
    sel = [x for x in rows if x['Name'].startswith('void conv3x3_halo2_kernel') and pred(targs(x['Name']))]
    calls = sum(int(x['Calls']) for x in sel)
    tot = sum(float(x['TotalDurationNs']) for x in sel)
    return sel, calls, tot


passes = d['steps'] * d['windows']['n'] + d['warmup'] + 2 * d['steps'] + 2     # warm-up + timed windows + the roofline leg's two passes (+ 2 clock-leg-free extras tolerated)
out = [f'# {tag} — `roofline.achieved` reproduced from the profiler\'s own numbers\n',
       f'Command: the stats pass of `tools/collect_profiles.sh {tag}` (`bench.py --steps {d["steps"]} --warmup {d["warmup"]}`, headline steps + roofline leg only).',
       'Kernel durations: `' + os.path.basename(stats) + '` of that run (copied as `' + tag + '_bench_kernel_stats.csv`); executed FLOPs: the bench\'s own',
       'ConvProfile records (`roofline.executed_tflop`, `roofline.strip_launches.executed_tflop` in the details file of the same run): work-list launches compute',
       'only the listed tiles, so the FLOPs a launch executes are counted by the list builder (`csrc/skiplist.hip`), not derivable from the trace.\n',
       '| launches of `conv3x3_halo2_kernel<ElemBF16, …>` | template instances | calls in the CSV | total ms | avg µs / launch | executed TFLOP (bench, its ' + str(r['launches']) + ' profiled launches) | TFLOP/s = TFLOP per launch ÷ avg µs | of 2 500 |',
       '|---|---|---|---|---|---|---|---|']
for label, pred, key in (('LIST (work-list launches, bf16, the embedder\'s conv3_1 … conv5_3)', lambda a: len(a) >= 5 and a[0] == 'ElemBF16' and a[3] == 'true' and a[4] == 'false', None),
                         ('STRIP (three 4-row strips per tile)', lambda a: len(a) >= 5 and a[0] == 'ElemBF16' and a[3] == 'true' and a[4] == 'true', 'strip_launches')):
    sel, calls, tot = pick(pred)
    src = r if key is None else r.get(key, {})
    if not calls or not src:
        continue
    tf_per_launch = src['executed_tflop'] / src['launches']
    avg_us = tot / calls / 1e3
    rate = tf_per_launch / (avg_us * 1e-6)
    out.append(f'| {label} | {len(sel)} | {calls} | {tot / 1e6:.2f} | {avg_us:.1f} | {src["executed_tflop"]:.3f} | {rate:.0f} | {rate / 2500:.3f} |')
out += ['', f'The bench line of this run printed `roofline.achieved` = {r["achieved"]} TFLOP/s (`frac` {r["frac"]}) at `avg_launch_us` = {r["avg_launch_us"]} '
        f'(HIP events on the launch stream, {r["launches"]} LIST launches); strips: {r.get("strip_launches", {}).get("achieved")} TFLOP/s at '
        f'{r.get("strip_launches", {}).get("avg_launch_us")} µs.  The profiler\'s averages include the warm-up passes and run ≈ 2–3 % slower under tracing '
        '(MI355X_MICROARCH.md, DVFS item 2).', '']
open(f'{root}/profiles/{tag}_roofline_check.md', 'w').write('\n'.join(out))
print('\n'.join(out))
