"""dev tool: the Gaussian layer's lateral conv (1x1 256 -> 256 at 200 x 200 + up2(P3) residual) through the ring kernel and the pointwise kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cvpce_amd import ops
dev = torch.device('cuda')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
g = torch.Generator().manual_seed(0)
x = torch.randn(n, 200, 200, 256, generator=g).to(torch.bfloat16).to(dev)
p3 = torch.randn(n, 100, 100, 256, generator=g).to(torch.bfloat16).to(dev)
pc = ops.PackedConv(torch.randn(256, 256, 1, 1, generator=g) / 16, torch.randn(256, generator=g) * 0.1, 1, 0, device=dev)


def t(tag):
    for _ in range(3):
        y = ops.conv2d(x, pc, residual=p3, res_mode=2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        y = ops.conv2d(x, pc, residual=p3, res_mode=2)
    e1.record(); torch.cuda.synchronize()
    print(f'{tag:12s} {e0.elapsed_time(e1) / 20 * 1e3:.1f} us', flush=True)
    return y


a = t('ring')
ops.CONV1X1_ANY_SHAPE = True
b = t('pointwise')
print('max abs diff', (a.float() - b.float()).abs().max().item())
