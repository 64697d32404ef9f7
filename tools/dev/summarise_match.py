"""dev tool: gpurun_out/match_<P>_<D>/{stats,fetch,write,sq,tcc} (tools/dev/prof_match.sh) -> profiles/r05_match_pmc.md"""
import csv, glob, collections, os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = ['# r05 — the distance-GEMM kernels of BASELINE configs[3] under rocprofv3 (MI355X, `tools/dev/prof_match.sh`, 30 eager launches per case)\n',
       'Kernel time from `--kernel-trace --stats`; HBM bytes from two separate `--pmc` passes (FETCH_SIZE doubled per MI355X_MICROARCH.md: gfx950 tallies',
       'the 128-byte requests of wide reads at 64 B; WRITE_SIZE as is; KB units); SQ / TCC counters from two more passes.  Algorithmic bytes = both',
       'operands once + norms + the (P, 1) int64 result.  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMD x 256 CU x GRBM_GUI_ACTIVE per XCD).\n']
for case in sorted(glob.glob(f'{root}/gpurun_out/match_*')):
    P, D = (int(v) for v in os.path.basename(case).split('_')[1:])
    G = 10000
    alg = (G * D + P * D) * 2 + (G + P) * 4 + P * 8
    flop = 2.0 * P * G * D
    stats = {r['Name']: r for r in csv.DictReader(open(glob.glob(f'{case}/stats/*/*_kernel_stats.csv')[0]))}
    cnt = collections.defaultdict(lambda: collections.defaultdict(list))
    for kind in ('fetch', 'write', 'sq', 'tcc'):
        for r in csv.DictReader(open(glob.glob(f'{case}/{kind}/*/*_counter_collection.csv')[0])):
            cnt[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
    out.append(f'## {P} queries x {G} rows x {D}  ({flop / 1e9:.2f} GFLOP, algorithmic {alg / 1e6:.2f} MB per launch)\n')
    out.append('| kernel | launches | avg µs | TFLOP/s | read MB | write MB | HBM / algorithmic | MFMA busy | eff. GHz | LDS conflict / active | L2 hit |')
    out.append('|---|---|---|---|---|---|---|---|---|---|---|')
    total = 0.0
    for name, s in stats.items():
        if 'match' not in name:
            continue
        c = {k: sum(v) / len(v) for k, v in cnt.get(name, {}).items()}
        us = float(s['AverageNs']) / 1e3
        total += us
        rd, wr = c.get('FETCH_SIZE', 0) * 1024 * 2, c.get('WRITE_SIZE', 0) * 1024
        cyc = c.get('GRBM_GUI_ACTIVE', 0) / 8
        busy = c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc) if cyc else 0
        hit = c.get('TCC_HIT_sum', 0) / max(1.0, c.get('TCC_HIT_sum', 0) + c.get('TCC_MISS_sum', 0))
        big = 'big' in name
        out.append(f"| `{name[:60]}` | {s['Calls']} | {us:.1f} | {flop / us / 1e6 if big or 'match_kernel' in name else 0:.0f} | {rd / 1e6:.2f} | {wr / 1e6:.2f} | "
                   f"{(rd + wr) / alg if big or 'match_kernel' in name else 0:.2f} | {busy:.2f} | {cyc / (us * 1e3) if us else 0:.2f} | "
                   f"{c.get('SQ_LDS_BANK_CONFLICT', 0) / max(1.0, c.get('SQ_LDS_IDX_ACTIVE', 0)):.2f} | {hit:.2f} |")
    out.append(f'\nsum of the launch\'s kernels {total:.1f} µs (eager launches; graph-replayed per-launch times: `bench.py` `workloads.match_stress_configs3`)\n')
open(f'{root}/profiles/r05_match_pmc.md', 'w').write('\n'.join(out) + '\n')
print('\n'.join(out))
