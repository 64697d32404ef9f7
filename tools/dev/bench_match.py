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


SIDE = torch.cuda.Stream()        # warm-up and capture on ONE stream: the one-launch form's state block is per stream and is never created inside a capture


def timed(q, gal, qn, gn, per_graph=20, reps=10):
    SIDE.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(SIDE):
        for _ in range(3):
            ops.match_topk(q, gal, 1, q_norms=qn, g_norms=gn)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=SIDE, capture_error_mode='thread_local'):
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
    ref = ref1 = None
    variants = [('auto', 0, 0, 0, 0), ('small', 1, 0, 0, 0)]
    for mg in (1, 2):
        for nq in (2, 3, 4, 5):
            if mg == 1 and nq == 5:
                continue
            variants += [(f'b{nq}x{mg}', 2, nq, mg, 0), (f'b{nq}x{mg}-1L', 2, nq, mg, 1)]
    for name, core, nq, mg, one in variants:
        assert ops.match_set_core(core, nq, mg, one) == 0
        idx, dist = ops.match_topk(q, gal, 3, q_norms=qn, g_norms=gn, return_distance=True)
        idx1, dist1 = ops.match_topk(q, gal, 1, q_norms=qn, g_norms=gn, return_distance=True)      # (k = 1: the one-launch form where enabled)
        idx1b, dist1b = ops.match_topk(q, gal, 1, q_norms=qn, g_norms=gn, return_distance=True)    # ... and again on the restored state block
        torch.cuda.synchronize()
        if ref is None:
            ref, ref1 = (idx.clone(), dist.clone()), (idx1.clone(), dist1.clone())
        same = bool(torch.equal(idx, ref[0]) and torch.equal(dist, ref[1]) and torch.equal(idx1, ref1[0]) and torch.equal(dist1, ref1[1])
                    and torch.equal(idx1b, ref1[0]) and torch.equal(dist1b, ref1[1]) and torch.equal(idx1, idx[:, :1]) and torch.equal(dist1, dist[:, :1]))
        us = timed(q, gal, qn, gn)
        res[name] = (round(us, 2), round(2.0 * P * G * D / us / 1e6, 1), 'same' if same else 'DIFFERENT')
    ops.match_set_core(0, 0, 0, 0)
    bad = [k for k, v in res.items() if v[2] != 'same']
    print(f'{P} x {G} x {D}: ' + '  '.join(f'{k} {v[0]}' for k, v in res.items()) + (f'  DIFFERENT: {bad}' if bad else '  (all identical)'), flush=True)
