"""dev tool: split a rocprofv3 --kernel-trace of bench.py into its pipeline passes (each starts with a burst of gln_transform
launches) and print, per pass, the window length and the dominant kernel's launch count / average duration.  The passes of
bench.py's roofline leg (ops.ConvProfile: HIP events around every launch, detector graph off, one stream) are the ones whose
average `roofline.avg_launch_us` is measured on; in the timed steps the detector's two head towers run their launches
concurrently on two streams, which stretches those launches' individual durations (not the step)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True)[0]
kernel = sys.argv[2] if len(sys.argv) > 2 else 'void conv3x3_halo2_kernel'


def targs(name):
    return [a.strip() for a in name[name.index('<') + 1:name.rindex('>')].split(',')] if '<' in name else []


def is_list(r, strip):
    """an embedder work-list launch of conv3x3_halo2_kernel<E, POOL, GMAX, LIST, STRIP, NW>: LIST (strip=False) or STRIP (strip=True) instance"""
    n = r['Kernel_Name']
    if 'conv3x3_halo2_kernel' not in n:
        return False
    a = targs(n[:n.rindex('(')] if n.endswith(')') else n)
    return len(a) >= 5 and a[0] == 'ElemBF16' and a[3] == 'true' and (a[4] == 'true') == strip
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if 'gln_transform_batch_kernel' in r['Kernel_Name']]
bursts = []
for m in marks:
    if not bursts or m - bursts[-1][-1] > 40:
        bursts.append([m])
    else:
        bursts[-1].append(m)
print('| pass | images | window ms | launches of the kernel | avg µs | sum ms | LIST launches | LIST avg µs | LIST sum ms | STRIP launches | STRIP avg µs | STRIP sum ms |\n|---|---|---|---|---|---|---|---|---|---|---|---|')
for bi, b in enumerate(bursts):
    s = b[0]
    e = bursts[bi + 1][0] if bi + 1 < len(bursts) else len(rows)
    if bi + 1 == len(bursts):       # last burst: stop at the end of its own pipeline pass (what follows are other legs)
        ends = [i for i in range(s, len(rows)) if 'match_merge_kernel' in rows[i]['Kernel_Name'] or rows[i]['Kernel_Name'].startswith('void match_kernel')]
        e = (ends[0] + 1) if ends else e
    h = [r for r in rows[s:e] if kernel in r['Kernel_Name']]
    d = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in h) / 1e3
    span = (max(int(r['End_Timestamp']) for r in rows[s:e]) - int(rows[s]['Start_Timestamp'])) / 1e6
    cols = ''
    for strip in (False, True):
        hh = [r for r in rows[s:e] if is_list(r, strip)]
        dd = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in hh) / 1e3
        cols += f' {len(hh)} | {dd / max(1, len(hh)):.1f} | {dd / 1e3:.2f} |'
    print(f'| {bi} | {len(b)} | {span:.1f} | {len(h)} | {d / max(1, len(h)):.1f} | {d / 1e3:.1f} |' + cols)
