"""dev: time the fused Gaussian subnet against the per-layer launches (graph-replayed)."""
import sys, os, math
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from cvpce_amd import ops
cuda = torch.device('cuda:0')
g = torch.Generator().manual_seed(5)
shapes = [(32, 64, 3), (32, 32, 3), (16, 32, 3), (16, 16, 1), (1, 16, 1)]
ws = [torch.randn(co, ci, k, k, generator=g) * math.sqrt(2.0 / (k * k * ci)) for co, ci, k in shapes]
bs = [torch.randn(co, generator=g) * 0.1 for co, _, _ in shapes]
convs = [ops.PackedConv(w, b, 1, 1 if w.shape[-1] == 3 else 0, device=cuda, dtype=torch.float16) for w, b in zip(ws, bs)]
def per_layer(x):
    t = ops.conv2d(x, convs[0], act=1, in_up_shift=1)
    t = ops.conv2d(ops.conv2d(t, convs[1], act=1), convs[2], act=1)
    return ops.gauss_tail(t, convs[3], convs[4], 2)
def timeit(fn, x, reps=20):
    for _ in range(3): fn(x)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps): fn(x)
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for n in (1, 4, 8):
    x = torch.randn(n, 200, 200, 64, generator=g).relu().to(torch.float16).to(cuda)
    a = timeit(lambda t: ops.gauss_subnet(t, convs, 2), x)
    b = timeit(per_layer, x)
    fl = 2.0 * n * 160000 * (9 * 64 * 32 + 9 * 32 * 32 + 9 * 32 * 16 + 272)
    print(f'N={n}: fused {a:.1f} us ({fl / a / 1e6:.0f} TFLOP/s)  per-layer {b:.1f} us', flush=True)
