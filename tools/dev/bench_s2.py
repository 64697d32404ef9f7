"""dev tool: the stride-2 3x3 convs that open layer2 / layer3 (and the Gaussian branch's 128 -> 64) under the library's kernels.  usage: bench_s2.py [images]"""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cvpce_amd import ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device('cuda:0')
for cin, h, w, cout, k, stride in ((128, 200, 200, 128, 3, 2), (256, 100, 100, 256, 3, 2), (128, 200, 200, 64, 3, 1), (128, 100, 100, 128, 3, 1), (128, 200, 200, 128, 3, 1)):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, h, w, cin, generator=g).to(torch.float16).to(dev)
    pc = ops.PackedConv(torch.randn(cout, cin, k, k, generator=g) / math.sqrt(k * k * cin), torch.zeros(cout), stride, k // 2, device=dev, dtype=torch.float16)
    row = []
    for fg, halo in ((False, True), (False, False), (1, False), (2, False)):
        ops.FORCE_GENERIC_CONV = fg
        ops.USE_HALO_3X3 = halo
        ops.PROFILE = ops.ConvProfile()
        ops.conv2d(x, pc, act=1)
        name = ops.PROFILE.layer_records[-1][0].split(' ')[0]
        ops.PROFILE = None
        for _ in range(5):
            ops.conv2d(x, pc, act=1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            ops.conv2d(x, pc, act=1)
        e1.record(); torch.cuda.synchronize()
        row.append(f'{name}(fg={fg}): {e0.elapsed_time(e1) * 20:.1f} us')
    ops.FORCE_GENERIC_CONV = False; ops.USE_HALO_3X3 = True
    fl = 2.0 * n * (h // stride) * (w // stride) * cout * k * k * cin
    print(f'{n} x {h}x{w} {cin}->{cout} k{k} s{stride} ({fl / 1e9:.1f} GFLOP): ' + '  '.join(row), flush=True)
