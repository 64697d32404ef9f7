"""dev tool: summarise a rocprofv3 --kernel-trace CSV: per-kernel totals over the LAST `--steps` repetitions of a marker kernel,
GPU busy vs idle time inside that window (overlapping kernels on several streams are merged)."""
import argparse, csv, glob, collections, sys
ap = argparse.ArgumentParser()
ap.add_argument('dir')
ap.add_argument('--marker', default='gln_transform_batch_kernel', help='kernel that starts a step (first launch of a detector pass)')
ap.add_argument('--marker-per-step', type=int, default=1)
ap.add_argument('--steps', type=int, default=3)
ap.add_argument('--end-marker', default='nms_scan_kernel', help='kernel that ends the window of a step')
a = ap.parse_args()
f = glob.glob(a.dir + '/**/*_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if a.marker in r['Kernel_Name']]
ends = [i for i, r in enumerate(rows) if a.end_marker in r['Kernel_Name']]
starts = marks[::a.marker_per_step][-a.steps:]
tot = collections.defaultdict(lambda: [0, 0.0])
span = busy = 0.0
for s in starts:
    e = min(i for i in ends if i > s)
    win = rows[s:e + 1]
    t0, t1 = int(win[0]['Start_Timestamp']), max(int(r['End_Timestamp']) for r in win)
    span += (t1 - t0) / 1e3
    iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in win)
    cur_s, cur_e = iv[0]
    for s_, e_ in iv[1:]:
        if s_ > cur_e:
            busy += (cur_e - cur_s) / 1e3; cur_s, cur_e = s_, e_
        else:
            cur_e = max(cur_e, e_)
    busy += (cur_e - cur_s) / 1e3
    for r in win:
        d = tot[r['Kernel_Name'][:90]]; d[0] += 1; d[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
n = len(starts)
print(f'{n} steps: window {span / n:.1f} us per step, GPU busy (any kernel) {busy / n:.1f} us, idle {(span - busy) / n:.1f} us')
for k, (c, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f'{us / n:9.1f} us  x{c / n:5.1f}  {k}')
