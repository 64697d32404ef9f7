"""dev tool: idle time inside one pipeline step of a rocprofv3 --kernel-trace (csv): wall, union of kernel intervals, idle gaps.
usage: gap_report.py <trace dir>   (steps are delimited by gln_transform_batch_kernel launches; the last full step is reported)"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if 'gln_transform_batch_kernel' in r['Kernel_Name']]
s, e = marks[-2], marks[-1]
iv = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows[s:e]]
t0, t1 = iv[0][0], max(b for _, b, _ in iv)
busy, gaps, latest, prev = 0, [], iv[0][0], None
for a, b, n in iv:
    if a > latest:
        gaps.append(((a - latest) / 1e3, prev[:40] if prev else '', n[:40]))
        busy += 0
    if b > latest:
        busy += b - max(a, latest)
        latest, prev = b, n
print(f'step wall {(t1 - t0) / 1e6:.3f} ms (to the next step\'s first launch {(int(rows[e]["Start_Timestamp"]) - t0) / 1e6:.3f}), '
      f'{len(iv)} launches, some kernel running {busy / 1e6:.3f} ms, idle {sum(g for g, _, _ in gaps) / 1e3:.3f} ms in {len(gaps)} gaps')
for g, p, n in sorted(gaps, reverse=True)[:12]:
    print(f'  {g:7.1f} us  after {p:40s} before {n}')
small = [g for g, _, _ in gaps]
print('  gap histogram (us): <2:', sum(1 for g in small if g < 2), ' 2-5:', sum(1 for g in small if 2 <= g < 5), ' 5-10:', sum(1 for g in small if 5 <= g < 10),
      ' 10-20:', sum(1 for g in small if 10 <= g < 20), ' >=20:', sum(1 for g in small if g >= 20))
