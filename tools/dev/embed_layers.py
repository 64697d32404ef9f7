"""dev tool: the embedder launch by launch on bench-shaped crops (wide boxes, short / long = 0.385 by default) with the work lists:
ms, algorithmic and executed TFLOP/s per launch.   python tools/dev/embed_layers.py [short_over_long] [tall]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cvpce_amd import ops, synthetic
from cvpce_amd.models import classification as C

ratio = float(sys.argv[1]) if len(sys.argv) > 1 else 0.385
tall = len(sys.argv) > 2
dev = torch.device('cuda')
enc = synthetic.synthetic_macvgg(seed=1).to(dev)
eng = enc.engine()
img = synthetic.shelf_image(0, 2048, 2048).to(dev)
g = torch.Generator().manual_seed(3)
n = 1600
long_ = 150 + 100 * torch.rand(n, generator=g)
short = long_ * ratio
w, h = (short, long_) if tall else (long_, short)
x1, y1 = torch.rand(n, generator=g) * (2048 - w), torch.rand(n, generator=g) * (2048 - h)
boxes = torch.stack((x1, y1, x1 + w, y1 + h), 1).to(dev)
crops = ops.crop_resize(img, boxes, 256, mode=2, mean=C.TANH_MEAN, std=C.TANH_STD)
ext = ops.crop_extents(boxes, None, 2048, 2048, 256)
const = eng.const_crop(C.TANH_MEAN, C.TANH_STD, 4, 256)
for skip in (True, False):
    kw = dict(ext=ext, const_in=const) if skip else {}
    for _ in range(2):
        eng.embed_packed(crops, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        eng.embed_packed(crops, **kw)
    e1.record(); torch.cuda.synchronize()
    print(f'skip={skip}: {e0.elapsed_time(e1) / 5:.3f} ms per 1600 crops')
    prof = ops.PROFILE = ops.ConvProfile()
    eng.embed_packed(crops, **kw)
    torch.cuda.synchronize()
    ops.PROFILE = None
    agg, order = {}, []
    for i, rec in enumerate(prof.records):            # work-list records carry a tag (layer shape); plain ones are numbered
        nm = rec[6] if len(rec) > 6 else f'launch{i % 12:02d}'
        ms = rec[2].elapsed_time(rec[3])
        exe = (float(rec[4].item()) * rec[5] if rec[4] is not None else 0.0) if len(rec) > 4 else rec[1]
        if nm not in agg:
            order.append(nm)
        a = agg.setdefault(nm, [0.0, 0.0, 0.0]); a[0] += ms; a[1] += rec[1]; a[2] += exe
    for nm in order:
        ms, alg, exe = agg[nm]
        print(f'  {nm:16s} {ms:7.3f} ms  alg {alg / 1e12:6.2f} TF  exe {exe / 1e12:6.2f} TF ({exe / max(alg, 1):5.2f})  {exe / ms / 1e9:7.1f} TFLOP/s executed')
