"""dev tool: per-launch conv timing of one detector pass (8 x 2048^2 images), sorted by time."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvpce_amd import ops, synthetic
import cvpce_amd.models.proposals as P
P.N_SIDE_STREAMS = 0
P.USE_DETECT_GRAPH = False
dev = torch.device('cuda')
det = synthetic.synthetic_gln(seed=0, detections_per_img=200).to(dev)
imgs = [synthetic.shelf_image(i, 2048, 2048).to(dev) for i in range(8)]
eng = det.engine()
for _ in range(2): eng.detect(imgs, 1, 200)
torch.cuda.synchronize()
# wrap conv2d to record shapes
recs = []
orig = ops.conv2d
def wrapped(x, pc, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); y = orig(x, pc, **kw); e1.record()
    recs.append((tuple(x.shape), pc.cout, pc.kh, pc.stride, e0, e1, 2.0 * y.shape[0] * y.shape[1] * y.shape[2] * pc.cout * pc.kh * pc.kw * pc.cin * (4 if kw.get('pool') else 1)))
    return y
ops.conv2d = wrapped
orig_atlas = ops.conv3x3_atlas
def wrapped_atlas(x, pc, mask, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); y = orig_atlas(x, pc, mask, **kw); e1.record()
    recs.append((('atlas',) + tuple(x.shape[1:]), pc.cout, pc.kh, pc.stride, e0, e1, 2.0 * float(mask.sum()) * x.shape[0] * pc.cout * 9 * pc.cin))
    return y
ops.conv3x3_atlas = wrapped_atlas
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); eng.detect(imgs, 1, 200); e1.record()
torch.cuda.synchronize()
print('detector pass ms', e0.elapsed_time(e1))
tot = 0
agg = {}
for shp, cout, k, s, a, b, fl in recs:
    ms = a.elapsed_time(b); tot += ms
    key = (shp[1:] if shp[0] != 'atlas' else shp, cout, k, s)
    d = agg.setdefault(key, [0, 0.0, 0.0]); d[0] += 1; d[1] += ms; d[2] += fl
print('sum conv ms', tot)
for key, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f'{str(key):44s} x{n:2d} {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TF')
