"""dev tool: which detector pieces run beside the embedder on CU-masked streams (224 / 32 CUs)?"""
import sys, os, time, math, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cvpce_amd import synthetic, _lib, ops
import cvpce_amd.models.proposals as P
dev = torch.device('cuda')


def masked_stream(cu_bits):
    hip = ctypes.CDLL('libamdhip64.so')
    words = (ctypes.c_uint32 * 8)(*[sum(1 << b for b in range(32) if (32 * w + b) in cu_bits) for w in range(8)])
    st = ctypes.c_void_p()
    assert hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words) == 0
    return torch.cuda.ExternalStream(st.value)


enc = synthetic.synthetic_macvgg(seed=1).to(dev)
eng = enc.engine()
x = (torch.rand(1600, 256, 256, 4, device=dev) - 0.5).to(torch.bfloat16); x[..., 3] = 0
bits_b = {i for i in range(256) if i % 8 == 7}
sa, sb = masked_stream(set(range(256)) - bits_b), masked_stream(bits_b)
_lib.lib.cvpce_set_persistent_workgroups(224)
g = torch.Generator().manual_seed(0)
mk = lambda co, ci, k: ops.PackedConv(torch.randn(co, ci, k, k, generator=g) / math.sqrt(ci * k * k), torch.randn(co, generator=g) * 0.1, 1, k // 2, device=dev)
c1, c3, ch = mk(256, 64, 1), mk(64, 64, 3), mk(256, 256, 3)
a200 = torch.randn(8, 200, 200, 64, generator=g).to(torch.bfloat16).to(dev)
a100 = torch.randn(8, 100, 100, 256, generator=g).to(torch.bfloat16).to(dev)
imgs = [synthetic.shelf_image(i, 2048, 2048).to(dev) for i in range(8)]
batch = torch.empty(8, 800, 800, 8, dtype=torch.bfloat16, device=dev)
pieces = {
    'transform x200': lambda: [ops.gln_transform_batch(imgs, batch, [(800, 800)] * 8, P.IMAGE_MEAN, P.IMAGE_STD) for _ in range(200)],
    'conv1x1 64->256 @200^2 x200': lambda: [ops.conv2d(a200, c1, act=1) for _ in range(200)],
    'conv3x3 64->64 @200^2 (LDS ring) x200': lambda: [ops.conv2d(a200, c3, act=1) for _ in range(200)],
    'conv3x3 256->256 @100^2 (halo2) x100': lambda: [ops.conv2d(a100, ch, act=1) for _ in range(100)],
    'torch fill 64 MB x200': lambda: [batch.zero_() for _ in range(200)],
}
emb = lambda: eng.embed_packed(x)


def wall(fa, fb):
    torch.cuda.synchronize(); t = time.perf_counter()
    if fa:
        with torch.cuda.stream(sa): fa()
    if fb:
        with torch.cuda.stream(sb): fb()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) * 1e3


wall(emb, None); a = min(wall(emb, None) for _ in range(2))
for name, f in pieces.items():
    wall(None, f)
    b = min(wall(None, f) for _ in range(2))
    both = min(wall(emb, f) for _ in range(2))
    print(f'{name}: embed {a:.1f} ms, piece {b:.1f} ms, together {both:.1f} ms', flush=True)
