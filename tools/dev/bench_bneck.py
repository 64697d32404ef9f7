"""dev tool: one ResNet bottleneck block, fused kernel vs the three-launch schedule, per layer shape and batch size (graph-replayed)."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cvpce_amd import ops
dev = torch.device('cuda')
ops.FUSED_BOTTLENECK_MAX_PLANES = 256          # (the product fuses P = 64 only: this tool measures all three widths)
dt = torch.float16 if (len(sys.argv) > 1 and sys.argv[1] == 'fp16') else torch.bfloat16
g = torch.Generator().manual_seed(0)
mk = lambda co, ci, k: ops.PackedConv(torch.randn(co, ci, k, k, generator=g) / math.sqrt(ci * k * k), torch.randn(co, generator=g) * 0.1, 1, k // 2, device=dev, dtype=dt)


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(iters):
            fn()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for p, hw in ((64, 200), (128, 100), (256, 50)):
    cs = (mk(p, 4 * p, 1), mk(p, p, 3), mk(4 * p, p, 1))
    for n in (1, 4, 8):
        x = torch.randn(n, hw, hw, 4 * p, generator=g).to(dt).to(dev).relu()
        fused = timeit(lambda: ops.bottleneck(x, *cs, x))
        def unf():
            ops.USE_FUSED_BOTTLENECK = False
            y = ops.conv2d(ops.conv2d(ops.conv2d(x, cs[0], act=1), cs[1], act=1), cs[2], act=1, residual=x)
            ops.USE_FUSED_BOTTLENECK = True
            return y
        three = timeit(unf)
        gf = 2.0 * n * hw * hw * (4 * p * p + 9 * p * p + 4 * p * p) / 1e9
        print(f'P={p:3d} {hw}x{hw} N={n}: fused {fused:7.1f} us ({gf / fused * 1e3:6.1f} TF)   three launches {three:7.1f} us ({gf / three * 1e3:6.1f} TF)', flush=True)
