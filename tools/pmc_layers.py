"""dev tool: per-kernel SQ counter ratios from rocprofv3 --pmc passes over tools/bench_conv.py (one counter_collection.csv per pass).
usage: pmc_layers.py <dir with pass sub-directories>"""
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
order = []
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'halo' not in k and 'stem' not in k:
            continue
        key = (k[:44], int(r['Grid_Size']))
        if key not in order:
            order.append(key)
        agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
print('ratios to SQ_WAVE_CYCLES unless noted; mean over launches')
for key in order:
    c = {n: sum(v) / len(v) for n, v in agg[key].items()}
    wc = c.get('SQ_WAVE_CYCLES', 0) or 1
    g = c.get('GRBM_GUI_ACTIVE', 0) or 1
    out = [f'{key[0]:44s} grid {key[1]:6d}']
    for n in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VMEM', 'SQ_ACTIVE_INST_SCA',
              'SQ_ACTIVE_INST_MISC', 'SQ_WAIT_INST_LDS', 'SQ_INST_CYCLES_VMEM_WR', 'SQ_INST_CYCLES_VMEM_RD', 'SQ_VMEM_WR_TA_DATA_FIFO_FULL'):
        if n in c:
            out.append(f'{n[3:]}={c[n] / wc:.3f}')
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in c:
        out.append(f'MFMA_BUSY/(4*256*GUI/8)={c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * 256 * g / 8):.3f}')
    if 'SQ_VALU_MFMA_COEXEC_CYCLES' in c and 'SQ_VALU_MFMA_BUSY_CYCLES' in c:
        out.append(f'COEXEC/MFMA_BUSY={c["SQ_VALU_MFMA_COEXEC_CYCLES"] / c["SQ_VALU_MFMA_BUSY_CYCLES"]:.3f}')
    if 'SQ_BUSY_CYCLES' in c:
        out.append(f'WAVE_CYCLES/BUSY_CYCLES={wc / c["SQ_BUSY_CYCLES"]:.2f}')
    print('  '.join(out))
