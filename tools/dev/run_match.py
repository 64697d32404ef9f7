"""dev tool: N eager launches of the bf16 distance GEMM + top-1 (cost-model choice of core and tile) for rocprofv3 traces / counters.
usage: run_match.py <P> <G> <D> [launches]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cvpce_amd import ops
P, G, D = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 30
dev = torch.device('cuda')
gal = torch.nn.functional.normalize(torch.randn(G, D, generator=torch.Generator().manual_seed(0)), dim=1).to(dev).to(torch.bfloat16)
q = torch.nn.functional.normalize(torch.randn(P, D, generator=torch.Generator().manual_seed(1)), dim=1).to(dev).to(torch.bfloat16)
gn, qn = ops.row_norms(gal), ops.row_norms(q)
for _ in range(n):
    ops.match_topk(q, gal, 1, q_norms=qn, g_norms=gn)
torch.cuda.synchronize()
print(f'{P} x {G} x {D}: {n} launches', flush=True)
