"""dev experiment: the wide-tile kernel (16x32 px x 128 couts, halo3) on the Cout >= 256 layers vs halo2 (16x16 px x 256 couts)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cvpce_amd import ops
from cvpce_amd._lib import lib, check
dev = torch.device('cuda')
L = {'vgg3_1': (256, 128, 64, 64, 256), 'vgg3_2': (256, 256, 64, 64, 256), 'vgg4_1': (256, 256, 32, 32, 512), 'vgg4_2': (256, 512, 32, 32, 512), 'vgg5_1': (256, 512, 16, 16, 512)}
for name, (n, cin, h, w, cout) in L.items():
    g = torch.Generator().manual_seed(0)
    pc = ops.PackedConv(torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5, torch.zeros(cout), 1, 1, device=dev)
    x = torch.relu(torch.randn(n, h, w, cin, generator=g)).to(torch.bfloat16).to(dev)
    out = torch.empty(n, h, w, cout, dtype=torch.bfloat16, device=dev)
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    res = {}
    for kind, fn in (('halo2', lib.cvpce_conv3x3_halo), ('wide', lib.cvpce_conv3x3_halo_wide)):
        call = lambda: check(fn(p(x), p(pc.weight), p(pc.bias), p(out), n, h, w, cin, cout, pc.k_pad, pc.cout_pad, 1, 0, s), kind)
        for _ in range(3): call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): call()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        res[kind] = (ms, out.float().sum().item())
    fl = 2.0 * n * h * w * cout * 9 * cin
    print(f'{name}: halo2 {res["halo2"][0]:.3f} ms {fl / res["halo2"][0] / 1e9:7.1f} TF | wide {res["wide"][0]:.3f} ms {fl / res["wide"][0] / 1e9:7.1f} TF | checksums {res["halo2"][1]:.1f} {res["wide"][1]:.1f}')
