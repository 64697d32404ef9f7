"""dev tool: the bf16 distance GEMM + top-1 per launch (graph-replayed) for the two cores of cvpce_match_topk, same process:
the 128-row register-staged kernel (core 1) against the 256-row LDS-DMA kernel (core 2, every query-tile width), with the indices and
distances of every variant compared bit for bit (all bf16 kernels form identical distances).
usage: bench_match.py [P,G,D ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cvpce_amd import ops
from cvpce_amd._lib import lib

dev = torch.device('cuda:0')
cases = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]] or [(200, 10000, 512), (200, 10000, 1024), (400, 10000, 1024), (800, 10000, 1024),
                                                                        (1600, 10000, 512), (1600, 10000, 1024), (1600, 3200, 1024), (1600, 1000, 1024), (3200, 10000, 1024)]


def timed(q, gal, qn, gn, per_graph=20, reps=10):
    for _ in range(3):
        ops.match_topk(q, gal, 1, q_norms=qn, g_norms=gn)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode='thread_local'):
        for _ in range(per_graph):
            ops.match_topk(q, gal, 1, q_norms=qn, g_norms=gn)
    graph.replay(); torch.cuda.synchronize()
    out = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            graph.replay()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / (reps * per_graph))
    return sorted(out)[1]


for P, G, D in cases:
    g = torch.Generator().manual_seed(0)
    gal = torch.nn.functional.normalize(torch.randn(G, D, generator=g), dim=1).to(dev).to(torch.bfloat16)
    q = torch.nn.functional.normalize(torch.randn(P, D, generator=torch.Generator().manual_seed(1)), dim=1).to(dev).to(torch.bfloat16)
    gn, qn = ops.row_norms(gal), ops.row_norms(q)
    res = {}
    ref = None
    for name, core, nq in (('auto', 0, 0), ('small', 1, 0), ('big2', 2, 2), ('big3', 2, 3), ('big4', 2, 4), ('big5', 2, 5)):
        assert lib.cvpce_match_set_core(core, nq) == 0
        idx, dist = ops.match_topk(q, gal, 3, q_norms=qn, g_norms=gn, return_distance=True)
        torch.cuda.synchronize()
        if ref is None:
            ref = (idx.clone(), dist.clone())
        same = bool(torch.equal(idx, ref[0]) and torch.equal(dist, ref[1]))
        us = timed(q, gal, qn, gn)
        res[name] = (round(us, 2), round(2.0 * P * G * D / us / 1e6, 1), 'same' if same else 'DIFFERENT')
    lib.cvpce_match_set_core(0, 0)
    print(f'{P} x {G} x {D}: ' + '  '.join(f'{k} {v[0]} us {v[1]} TF {v[2]}' for k, v in res.items()), flush=True)
