"""dev tool: the detector's 1x1 conv shapes (8 images) one at a time: pointwise kernel | ring kernels (default dispatch) | 64-pixel ring tiles
(CVPCE_DBG_RING64, if the library has the experiment); CVPCE_LIB selects an ablation build."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cvpce_amd import ops
dev = torch.device('cuda')
g = torch.Generator().manual_seed(0)
SHAPES = [(50, 256, 1024, True), (100, 128, 512, True), (50, 1024, 256, False), (200, 256, 256, False), (25, 512, 2048, True), (100, 512, 128, False),
          (100, 512, 256, False), (25, 2048, 512, False), (200, 64, 256, False)]
modes = sys.argv[1:] or ['pointwise', 'ring']
for mode in modes:
    ops.CONV1X1_ANY_SHAPE = mode == 'pointwise'
    ops.USE_CONV1X1 = mode == 'pointwise'
    if mode == 'ring64':
        os.environ['CVPCE_DBG_RING64'] = '1'
    else:
        os.environ.pop('CVPCE_DBG_RING64', None)
    out = []
    for hw, cin, cout, res in SHAPES:
        x = torch.randn(8, hw, hw, cin, generator=g).to(torch.bfloat16).to(dev)
        r = torch.randn(8, hw, hw, cout, generator=g).to(torch.bfloat16).to(dev) if res else None
        pc = ops.PackedConv(torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5, torch.randn(cout, generator=g) * 0.1, 1, 0, device=dev)
        for _ in range(3):
            y = ops.conv2d(x, pc, act=1, residual=r)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            y = ops.conv2d(x, pc, act=1, residual=r)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 30 * 1e3
        mb = (x.numel() + y.numel() + (r.numel() if res else 0)) * 2 / 1e6
        out.append(f'{hw}^2 {cin}->{cout}{"+r" if res else ""}: {us:5.1f} us {mb / us:4.2f} TB/s')
    print(f'{mode:9s} ' + ' | '.join(out), flush=True)
