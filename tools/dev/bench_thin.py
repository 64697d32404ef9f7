"""dev tool: the Gaussian subnet's thin 3x3 layers (8 x 400 x 400 x 32 -> 32 | 16) through the thin kernel and the implicit GEMM."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cvpce_amd import ops
dev = torch.device('cuda')
g = torch.Generator().manual_seed(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
x = torch.randn(n, 400, 400, 32, generator=g).relu().to(torch.bfloat16).to(dev)
for cout in (32, 16):
    pc = ops.PackedConv(torch.randn(cout, 32, 3, 3, generator=g) / 17, torch.randn(cout, generator=g) * 0.1, 1, 1, device=dev)
    for thin in (True, False, True, False):
        ops.USE_THIN_3X3 = thin
        for _ in range(3):
            y = ops.conv2d(x, pc, act=1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            y = ops.conv2d(x, pc, act=1)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f'32 -> {cout}  thin={thin}: {us:6.1f} us  {(x.numel() + y.numel()) * 2 / us / 1e6:.2f} TB/s', flush=True)
