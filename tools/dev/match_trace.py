import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(list(rows[0].keys()))
agg = collections.defaultdict(list)
for r in rows:
    k = r["Kernel_Name"]
    if "match" in k or "row_norm" in k:
        g = r.get("Grid_Size_X") or r.get("Grid_Size") or "?"
        agg[(k[:40], g)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in agg.items():
    v = sorted(v); print(k, len(v), "median us", v[len(v)//2], "min", v[0])
