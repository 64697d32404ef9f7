"""dev tool: does CU masking pay?  Embedder alone on n CUs (persistent grids bounded to n), detector alone on m CUs."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvpce_amd import ops, synthetic, _lib
import cvpce_amd.models.proposals as P
dev = torch.device('cuda')
hip = ctypes.CDLL('libamdhip64.so')


def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[sum(1 << b for b in range(32) if (32 * w + b) in bits) for w in range(8)])
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)


enc = synthetic.synthetic_macvgg(seed=1).to(dev)
eng = enc.engine()
x = (torch.rand(1600, 256, 256, 8, device=dev) - 0.5).to(torch.bfloat16)
det = synthetic.synthetic_gln(seed=0, detections_per_img=200).to(dev)
P.N_SIDE_STREAMS = 0
deng = P.GLNEngine(det, dev)
imgs = [synthetic.shelf_image(i, 2048, 2048).to(dev) for i in range(8)]


def timeit(fn, stream, n=3):
    with torch.cuda.stream(stream):
        fn(); fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(n): fn()
        e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, pick in (('all 256', set(range(256))), ('224: drop every 8th', {i for i in range(256) if i % 8 != 7}),
                   ('224: drop the last 32', set(range(224))), ('240: drop every 16th', {i for i in range(256) if i % 16 != 15})):
    st = masked_stream(pick)
    _lib.lib.cvpce_set_persistent_workgroups(len(pick))
    print(f'embed 1600 crops on {name:24s}: {timeit(lambda: eng.embed_packed(x), st):8.3f} ms')
_lib.lib.cvpce_set_persistent_workgroups(256)
for name, pick in (('all 256', set(range(256))), ('32: every 8th', {i for i in range(256) if i % 8 == 7}), ('32: the last 32', set(range(224, 256))),
                   ('16: every 16th', {i for i in range(256) if i % 16 == 15})):
    st = masked_stream(pick)
    _lib.lib.cvpce_set_persistent_workgroups(len(pick))
    print(f'detect 8 images on {name:24s}: {timeit(lambda: deng.detect(imgs, 1, 200), st):8.3f} ms')
