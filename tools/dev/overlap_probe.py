"""dev tool: do the detector and the embedder overlap at all when they sit on different streams?  wall time of {embed on A, 6 x detect on B}
launched together vs each alone, for plain streams and for CU-masked streams (1 XCD / 7 XCDs)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cvpce_amd import synthetic, _lib
import ctypes


def _masked_stream(cu_bits):
    hip = ctypes.CDLL('libamdhip64.so')
    words = (ctypes.c_uint32 * 8)(*[sum(1 << b for b in range(32) if (32 * w + b) in cu_bits) for w in range(8)])
    st = ctypes.c_void_p()
    assert hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words) == 0
    return torch.cuda.ExternalStream(st.value)


dev = torch.device('cuda')
enc = synthetic.synthetic_macvgg(seed=1).to(dev)
eng = enc.engine()
x = (torch.rand(1600, 256, 256, 4, device=dev) - 0.5).to(torch.bfloat16)
x[..., 3] = 0
det = synthetic.synthetic_gln(seed=0, detections_per_img=200).to(dev)
deng = det.engine()
imgs = [synthetic.shelf_image(i, 2048, 2048).to(dev) for i in range(8)]
for _ in range(3):
    deng.detect(imgs, 1, 200)
    eng.embed_packed(x)
torch.cuda.synchronize()


def wall(fa, sa, fb, sb):
    torch.cuda.synchronize(); t = time.perf_counter()
    if fa:
        with torch.cuda.stream(sa): fa()
    if fb:
        with torch.cuda.stream(sb): fb()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) * 1e3


emb = lambda: eng.embed_packed(x)
dets = lambda: [deng.detect(imgs, 1, 200) for _ in range(6)]
for name, sa, sb, wgs in (('plain streams', torch.cuda.Stream(), torch.cuda.Stream(priority=-1), 256),
                          ('masked 224 / 32 (one XCD)', _masked_stream({i for i in range(256) if i % 8 != 7}), _masked_stream({i for i in range(256) if i % 8 == 7}), 224)):
    _lib.lib.cvpce_set_persistent_workgroups(wgs)
    if wgs != 256:          # the detector's internal side branches must live on the detector's CUs too
        deng.side_streams = [_masked_stream({i for i in range(256) if i % 8 == 7}) for _ in deng.side_streams]
        deng.__dict__.pop('_graphs', None); deng.__dict__.pop('_sights', None)
        with torch.cuda.stream(sb):
            for _ in range(3):
                deng.detect(imgs, 1, 200)
        torch.cuda.synchronize()
    for _ in range(2):
        wall(emb, sa, dets, sb)
    a = min(wall(emb, sa, None, None) for _ in range(3))
    b = min(wall(None, None, dets, sb) for _ in range(3))
    both = min(wall(emb, sa, dets, sb) for _ in range(3))
    rev = min(wall(dets, sb, emb, sa) for _ in range(3))
    print(f'{name}: embed alone {a:.2f} ms, 6 x detect alone {b:.2f} ms, together {both:.2f} ms (detector queued first: {rev:.2f} ms)', flush=True)
