"""dev tool: one step of a rocprofv3 --kernel-trace as a timeline (start offset, duration, gap since the latest end so far, kernel)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True)[0]
per_step = 1   # one batched transform launch per detector pass (argv[2] kept for old command lines)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if 'gln_transform_batch_kernel' in r['Kernel_Name']]
s, e = marks[-2 * per_step], marks[-per_step]
t0 = int(rows[s]['Start_Timestamp']); latest = t0
for r in rows[s:e]:
    a, b = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (a - latest) / 1e3
    print(f'{(a - t0) / 1e3:8.1f} us  {(b - a) / 1e3:7.1f} us  gap {gap:6.1f}  {r["Kernel_Name"][:70]}')
    latest = max(latest, b)
