"""dev tool: the detector's post-processing (decode_topk + NMS) alone, on the logits of the bench's random-weight detector.
usage: bench_post.py [images] [detections_per_img] [reps]   (run under rocprofv3 --kernel-trace --stats for per-kernel times)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cvpce_amd import synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dpi = int(sys.argv[2]) if len(sys.argv) > 2 else 200
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev = torch.device('cuda')
det = synthetic.synthetic_gln(seed=0, detections_per_img=dpi).to(dev)
imgs = [synthetic.shelf_image(i, 2048, 2048).to(dev) for i in range(n)]
eng = det.engine()
out, mid = eng.detect(imgs, 1, dpi, want_intermediates=True)
cls, reg, batch = mid['cls'], mid['reg'], mid['batch']
original, resized, padded = eng.batch_geometry(imgs)
for c in cls:
    s = torch.sigmoid(c.float())
    print('level', tuple(c.shape), 'candidates > 0.05 per image:', (s > 0.05).flatten(1).sum(1).tolist(), flush=True)
args = (cls, reg, tuple(batch.shape[1:3]), resized, original, 1, dpi, 0.5)
for _ in range(3):
    ref = eng.postprocess(*args)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    o = eng.postprocess(*args)
e1.record(); torch.cuda.synchronize()
print(f'postprocess {n} images dpi {dpi}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us per call', flush=True)
import hashlib
h = hashlib.sha256()
for t in o[:5]:
    h.update(t.cpu().numpy().tobytes())
print('digest', h.hexdigest()[:16], 'count', o[3].tolist(), flush=True)
