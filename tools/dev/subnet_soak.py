"""dev: race screen of the fused Gaussian subnet (private LDS rings, LDS-DMA, counted vmcnt waits): 300 launches per shape under copy traffic
from a second stream, every result bit-identical to the first; shapes: the detector's 200 x 200 (N = 1, 4, 8) and a portrait map."""
import sys, os, math
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from cvpce_amd import ops
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(5)
shapes = [(32, 64, 3), (32, 32, 3), (16, 32, 3), (16, 16, 1), (1, 16, 1)]
ws = [torch.randn(co, ci, k, k, generator=g) * math.sqrt(2.0 / (k * k * ci)) for co, ci, k in shapes]
bs = [torch.randn(co, generator=g) * 0.1 for co, _, _ in shapes]
side = torch.cuda.Stream()
src = torch.empty(1 << 28, dtype=torch.uint8, device=dev)
dst = torch.empty_like(src)
bad = 0
for dt in (torch.float16, torch.bfloat16):
    convs = [ops.PackedConv(w, b, 1, 1 if w.shape[-1] == 3 else 0, device=dev, dtype=dt) for w, b in zip(ws, bs)]
    for n, hs, ws_ in ((1, 200, 200), (4, 200, 200), (8, 200, 200), (3, 272, 200)):
        x = torch.randn(n, hs, ws_, 64, generator=g).relu().to(dt).to(dev)
        ref = ops.gauss_subnet(x, convs, 2).clone()
        for it in range(300):
            if it % 10 == 0:
                with torch.cuda.stream(side):
                    dst.copy_(src, non_blocking=True)
            out = ops.gauss_subnet(x, convs, 2)
            if not torch.equal(out, ref):
                bad += 1
        torch.cuda.synchronize()
        print(dt, (n, hs, ws_), 'mismatching launches so far:', bad, flush=True)
print('subnet soak:', 'OK' if bad == 0 else f'{bad} MISMATCHES')
sys.exit(1 if bad else 0)
