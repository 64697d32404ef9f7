"""dev tool: the detector's small-map long-K convs one by one, unsplit vs split-K 2 / 4 / 8 (eager launches, 50 back to back).  usage: bench_splitk.py [images]"""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cvpce_amd import ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device('cuda:0')
orig = ops.splitk_factor
for cin, h, w, cout, k, stride in ((512, 25, 25, 512, 3, 1), (512, 50, 50, 512, 3, 2), (256, 100, 100, 256, 3, 2), (256, 25, 25, 256, 3, 1),
                                   (256, 25, 25, 256, 3, 2), (256, 13, 13, 256, 3, 2)):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, h, w, cin, generator=g).to(torch.float16).to(dev)
    pc = ops.PackedConv(torch.randn(cout, cin, k, k, generator=g) / math.sqrt(k * k * cin), torch.zeros(cout), stride, k // 2, device=dev, dtype=torch.float16)
    row = []
    for ks in (0, 2, 4, 8):
        ops.splitk_factor = (lambda pc_, ho, wo, ks=ks: ks)
        for _ in range(5):
            ops.conv2d(x, pc, act=1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            ops.conv2d(x, pc, act=1)
        e1.record(); torch.cuda.synchronize()
        row.append(f'S={ks}: {e0.elapsed_time(e1) * 20:.1f} us')
    print(f'{n} x {h}x{w} {cin}->{cout} k{k} s{stride}: ' + '  '.join(row), flush=True)
ops.splitk_factor = orig
