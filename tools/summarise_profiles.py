"""dev tool: turn the rocprofv3 output of one round (gpurun_out/<prof>, gpurun_out/<pmc>/{fetch,write,sq,tcc}) into the
summaries committed under profiles/ (kernel stats CSV, HBM traffic md + hbm_traffic.json, per-kernel counter table)."""
import csv, glob, collections, json, shutil, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, prof, pmc = sys.argv[1], sys.argv[2], sys.argv[3]          # e.g. r01g prof_r1g pmc7


def short_name(k):
    """a rocprof kernel name without `void ` and the argument list, the template arguments kept WHOLE: round 5 cut names at 46 characters
    and merged the LIST, STRIP and plain instances of conv3x3_halo2_kernel (three different kernels) into one row."""
    k = k[5:] if k.startswith('void ') else k
    k = k.replace('(anonymous namespace)::', '')
    depth = 0
    for i, ch in enumerate(k):
        depth += ch == '<'
        depth -= ch == '>'
        if ch == '(' and depth == 0:
            return k[:i]
    return k


def template_args(k):
    k = short_name(k)
    return [a.strip() for a in k[k.index('<') + 1:k.rindex('>')].split(',')] if '<' in k else []


def prof_name(k):    # rocprof kernel name -> the name ops.ConvProfile / bench.py use (roofline.traffic, roofline.hbm_stages)
    # the detector (fp16 storage by default) runs ElemF16 instances of the 3x3 halo kernels: filed apart, like bench.py's profile does
    for p in ('conv3x3_halo2_kernel', 'conv3x3_halo3_kernel'):
        if (k.startswith(p) or k.startswith('void ' + p)) and 'ElemF16' in k:
            return p + '[detector]'
    # conv3x3_halo2_kernel<E, POOL, GMAX, LIST, STRIP, NW>: the STRIP instances are launches of their own (bench.py files them apart too)
    if k.startswith('void conv3x3_halo2_kernel') or k.startswith('conv3x3_halo2_kernel'):
        ta = template_args(k)
        if len(ta) >= 5 and ta[4] == 'true':
            return 'conv3x3_halo2_kernel[strips]'
    for p, n in (('conv3x3_halo2_kernel', 'conv3x3_halo2_kernel'), ('conv3x3_halo3_kernel', 'conv3x3_halo3_kernel'), ('vgg_stem2_kernel', 'vgg_stem2_kernel'),
                 ('gln_transform_batch_kernel', 'gln_transform_batch_kernel'), ('crop_resize2_kernel', 'crop_resize_kernel'), ('crop_resize_kernel', 'crop_resize_kernel'),
                 ('conv1x1_stream_wreg_kernel', 'conv1x1_kernel'), ('conv1x1_stream_kernel', 'conv1x1_kernel'), ('conv1x1_kernel', 'conv1x1_kernel'),
                 ('thin3x3_kernel', 'thin3x3_kernel'), ('gauss_tail_kernel', 'gauss_tail_kernel'), ('gauss_subnet_kernel', 'gauss_subnet_kernel')):
        if k.startswith(p) or k.startswith('void ' + p):
            return n
    return None


f = glob.glob(f'{root}/gpurun_out/{prof}/*/*_kernel_stats.csv')[0]
shutil.copy(f, f'{root}/profiles/{tag}_bench_kernel_stats.csv')
out = {}
for kind in ('fetch', 'write'):
    f = glob.glob(f'{root}/gpurun_out/{pmc}/{kind}/*/*_counter_collection.csv')[0]
    agg = collections.defaultdict(list)
    rows_ = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Dispatch_Id']))
    # only the launches of the LAST pipeline pass (= one timed step of bench.py: it starts with the 8 transform launches of the
    # step's images); the gallery build and the warm-up pass launch the same kernels on other batch sizes
    # (round 3: the input transform of a step's 8 images is ONE launch of gln_transform_batch_kernel)
    marks = [int(r['Dispatch_Id']) for r in rows_ if 'gln_transform_batch_kernel' in r['Kernel_Name']]
    first = marks[-1] if marks else 0
    for r in rows_:
        if int(r['Dispatch_Id']) >= first:
            agg[r['Kernel_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        out.setdefault(k, {})[kind] = (len(v), sum(v))
rows = []
for k, d in out.items():
    if 'fetch' in d and 'write' in d:
        rows.append((k, d['fetch'][0], d['fetch'][1] * 1024 * 2 / d['fetch'][0], d['write'][1] * 1024 / d['write'][0]))
rows.sort(key=lambda r: -(r[2] + r[3]) * r[1])
byname = collections.defaultdict(lambda: [0, 0.0, 0.0])
with open(f'{root}/profiles/{tag}_pmc_hbm_traffic.md', 'w') as fo:
    fo.write(f'# HBM traffic per launch over ONE step of bench.py (the last pipeline pass of `--steps 1 --warmup 1`: 8 images, 1600 crops), MI355X, {tag} build\n\n')
    fo.write('Two separate `rocprofv3 --pmc` passes (FETCH_SIZE, WRITE_SIZE; KB units), averaged per launch.\n')
    fo.write('FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests of wide reads at 64 B); WRITE_SIZE as is.\n\n')
    fo.write('| kernel | launches | read MB/launch | write MB/launch | total MB/launch |\n|---|---|---|---|---|\n')
    for k, n, fb, wb in rows[:14]:
        fo.write(f'| `{short_name(k)}` | {n} | {fb / 1e6:.1f} | {wb / 1e6:.1f} | {(fb + wb) / 1e6:.1f} |\n')
    for k, n, fb, wb in rows:
        pn = prof_name(k)
        if pn:
            b = byname[pn]; b[0] += n; b[1] += fb * n; b[2] += wb * n
    fo.write('\n`hbm_traffic.json` holds the same figures keyed by the names bench.py uses for its conv profile\n'
             '(template instantiations that differ only in the fused pooling are merged, launch-weighted).\n')
js = {pn: {'launches': n, 'read_bytes_per_launch': fb / n, 'write_bytes_per_launch': wb / n} for pn, (n, fb, wb) in byname.items()}
# stamped with the library the counters were collected on (the .so that travelled to the GPU box = the one in the tree now, as long as
# nothing was rebuilt in between): bench.py reports `roofline.traffic` only when the library it loads is this one
import hashlib
js['_library_sha256'] = hashlib.sha256(open(f'{root}/cvpce_amd/libcvpce_hip.so', 'rb').read()).hexdigest()
js['_collected_as'] = tag
json.dump(js, open(f'{root}/profiles/hbm_traffic.json', 'w'), indent=1)
res = collections.defaultdict(dict)
for kind in ('sq', 'tcc'):
    f = glob.glob(f'{root}/gpurun_out/{pmc}/{kind}/*/*_counter_collection.csv')[0]
    kt = glob.glob(f'{root}/gpurun_out/{pmc}/{kind}/*/*_kernel_trace.csv')[0]
    dur = {r['Dispatch_Id']: int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in csv.DictReader(open(kt))}
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = short_name(r['Kernel_Name'])
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
        if kind == 'sq' and r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
            agg[k]['_dur'].append(dur[r['Dispatch_Id']])
    for k, v in agg.items():
        for c, vals in v.items():
            res[k][c] = sum(vals) / len(vals)
lines = []
for k, v in res.items():
    if v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) > 2e6:
        cyc = v['GRBM_GUI_ACTIVE'] / 8
        busy = v['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc)
        lines.append((v['_dur'], '| `%s` | %.0f | %.2f | %.2f | %.2f | %.2f | %.2f | %.2f | %.2f |' % (
            k, v['_dur'] / 1e3, busy, cyc / v['_dur'], busy * cyc / v['_dur'] / 2.4, v['SQ_WAIT_INST_ANY'] / v['SQ_WAVE_CYCLES'],
            v['SQ_LDS_BANK_CONFLICT'] / max(1, v['SQ_LDS_IDX_ACTIVE']), v['SQ_LDS_IDX_ACTIVE'] / (256 * cyc),
            v['TCC_HIT_sum'] / (v['TCC_HIT_sum'] + v['TCC_MISS_sum']))))
lines.sort(reverse=True)
hdr = ('| kernel | avg µs | MFMA busy | eff. GHz | busy × GHz / 2.4 | WAIT_INST_ANY / WAVE_CYCLES | LDS conflict / LDS active | LDS active / cycle | L2 hit |\n'
       '|---|---|---|---|---|---|---|---|---|\n')
open(f'{root}/profiles/{tag}_pmc_kernels.md', 'w').write(
    f'# {tag} — SQ / TCC counters per kernel, real data (bench.py --steps 1 --warmup 1), MI355X\n\n'
    'Two separate `rocprofv3 --pmc` passes (SQ_*+GRBM, TCC_*), per-launch averages.  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / '
    '(4 SIMD x 256 CU x GRBM_GUI_ACTIVE per XCD); eff. GHz = GRBM_GUI_ACTIVE per XCD / kernel duration; their product over the\n'
    '2.4 GHz the nominal 2.5 PFLOP/s assumes is the fraction of the nominal MFMA peak the kernel can reach at most.\n\n'
    + hdr + '\n'.join(l for _, l in lines) + '\n')
print(open(f'{root}/profiles/{tag}_pmc_kernels.md').read())
print(js)
