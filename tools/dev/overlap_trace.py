"""dev tool: from a rocprofv3 --kernel-trace of tools/dev/overlap_probe.py: how much of the detector kernels' time overlaps embedder kernels, per queue."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
emb_names = ('conv3x3_halo2', 'conv3x3_halo3', 'vgg_stem2')
queues = collections.Counter()
emb = [(int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows if any(n in r['Kernel_Name'] for n in emb_names) and 'Halo2Args' in r['Kernel_Name'] or 'vgg_stem2' in r['Kernel_Name']]
print('columns:', list(rows[0].keys()))
for r in rows[-400:]:
    queues[(r.get('Queue_Id'), r['Kernel_Name'][:40])] += 1
for k, v in sorted(queues.items(), key=lambda kv: -kv[1])[:25]:
    print(v, k)
# overlap of decode_topk (detector only) with the embedder's stem kernel intervals
stem = [(int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows if 'vgg_stem2' in r['Kernel_Name']]
dec = [(int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows if 'decode_topk' in r['Kernel_Name']]
def inside(t, ivs):
    return any(a <= t <= b for a, b in ivs)
big = [(int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows if 'Halo2Args' in r['Kernel_Name'] and int(r['End_Timestamp']) - int(r['Start_Timestamp']) > 400000]
print('decode_topk launches:', len(dec), 'of which started while a >0.4 ms halo2 launch was running:', sum(inside(a, big) for a, b in dec))
