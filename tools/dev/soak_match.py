"""dev tool: race screen of the distance-GEMM kernels -- every variant (core, tile, one-/two-launch), several shapes, `reps` launches each under
memory load from a second stream, every result compared bit for bit with the first (LDS-DMA hand-offs are placed by vmcnt / barrier counts:
a misplaced read passes single runs and fails rarely).  usage: soak_match.py [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cvpce_amd import ops
from cvpce_amd._lib import lib
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device('cuda:0')
side = torch.cuda.Stream()
noise_a = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
noise_b = torch.empty_like(noise_a)
bad = 0
for P, G, D in ((200, 10000, 512), (1600, 10000, 1024), (777, 3333, 256), (320, 2049, 1024)):
    g = torch.Generator().manual_seed(P)
    gal = torch.nn.functional.normalize(torch.randn(G, D, generator=g), dim=1).to(dev).to(torch.bfloat16)
    q = torch.nn.functional.normalize(torch.randn(P, D, generator=g), dim=1).to(dev).to(torch.bfloat16)
    gn, qn = ops.row_norms(gal), ops.row_norms(q)
    ref = None
    for core, nq, mg, one in [(1, 0, 0, 0)] + [(2, n_, m_, o_) for m_ in (1, 2) for n_ in (2, 3, 4, 5) if not (m_ == 1 and n_ == 5) for o_ in (0, 1)]:
        assert ops.match_set_core(core, nq, mg, one) == 0
        for k in (1, 3):
            first = None
            for r in range(reps):
                if r % 8 == 0:
                    with torch.cuda.stream(side):
                        noise_b.copy_(noise_a)                      # HBM / L2 traffic beside the launches
                idx, dist = ops.match_topk(q, gal, k, q_norms=qn, g_norms=gn, return_distance=True)
                if first is None:
                    first = (idx.clone(), dist.clone())
                    if ref is None and k == 3:
                        ref = first
                elif not (torch.equal(idx, first[0]) and torch.equal(dist.view(torch.int32), first[1].view(torch.int32))):
                    bad += 1
            if k == 3 and ref is not None and not (torch.equal(first[0], ref[0]) and torch.equal(first[1].view(torch.int32), ref[1].view(torch.int32))):
                bad += 1
    torch.cuda.synchronize()
    print(f'{P} x {G} x {D}: 15 variants x 2 k x {reps} launches, mismatches so far {bad}', flush=True)
ops.match_set_core(0, 0, 0, 0)
print('soak_match:', 'OK' if bad == 0 else f'{bad} MISMATCHES')
