import sys, os, math, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from cvpce_amd import ops, _lib
cuda = torch.device('cuda:0')
g = torch.Generator().manual_seed(5)
shapes = [(32, 64, 3), (32, 32, 3), (16, 32, 3), (16, 16, 1), (1, 16, 1)]
ws = [torch.randn(co, ci, k, k, generator=g) * math.sqrt(2.0 / (k * k * ci)) for co, ci, k in shapes]
bs = [torch.randn(co, generator=g) * 0.1 for co, _, _ in shapes]
convs = [ops.PackedConv(w, b, 1, 1 if w.shape[-1] == 3 else 0, device=cuda, dtype=torch.float16) for w, b in zip(ws, bs)]
x = torch.randn(8, 200, 200, 64, generator=g).relu().to(torch.float16).to(cuda)
for _ in range(5): ops.gauss_subnet(x, convs, 2)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 8)()
_lib.lib.cvpce_debug_gauss_subnet_stamps.restype = ctypes.c_int
assert _lib.lib.cvpce_debug_gauss_subnet_stamps(buf) == 0
v = [int(b) for b in buf]
it = max(1, v[6])
print('iterations', it, 'cycles per iteration by part [top, r1, r2, r3, r4, r5]:', [round(c / it) for c in v[:6]], 'sum', round(sum(v[:6]) / it))
