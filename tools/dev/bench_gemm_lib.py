"""dev tool: the detector's 1x1 conv shapes as plain library GEMMs (torch -> hipBLASLt / rocBLAS), for comparison with tools/dev/bench_1x1.py."""
import torch
dev = torch.device('cuda')
SHAPES = [(50, 256, 1024, True), (100, 128, 512, True), (50, 1024, 256, False), (200, 256, 256, False), (25, 512, 2048, True), (100, 512, 128, False),
          (100, 512, 256, False), (25, 2048, 512, False), (200, 64, 256, False)]
out = []
for hw, cin, cout, res in SHAPES:
    m = 8 * hw * hw
    x = torch.randn(m, cin, device=dev, dtype=torch.bfloat16)
    w = torch.randn(cout, cin, device=dev, dtype=torch.bfloat16)
    b = torch.randn(cout, device=dev, dtype=torch.bfloat16)
    r = torch.randn(m, cout, device=dev, dtype=torch.bfloat16) if res else None
    f = (lambda: torch.addmm(r, x, w.t())) if res else (lambda: torch.nn.functional.linear(x, w, b))
    for _ in range(5):
        y = f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        y = f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 30 * 1e3
    mb = (x.numel() + y.numel() + (r.numel() if res else 0)) * 2 / 1e6
    out.append(f'{hw}^2 {cin}->{cout}{"+r" if res else ""}: {us:5.1f} us {mb / us:4.2f} TB/s')
print('library   ' + ' | '.join(out), flush=True)
