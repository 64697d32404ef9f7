"""temporary: does an image's detection depend on the batch it is in?"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from cvpce_amd import synthetic
dev = torch.device('cuda:0')
det = synthetic.synthetic_gln(seed=0, detections_per_img=200).to(dev)
eng = det.engine()
imgs = [synthetic.shelf_image(g, 2048, 2048).to(dev) for g in range(8)]
ref = None
for nb in (1, 2, 4, 8):
    out = eng.detect(imgs[:nb], 1, 200, 0.5)
    torch.cuda.synchronize()
    cur = [t[0].clone() for t in out[:4]] + [out[5][0].clone()]
    if ref is None: ref = cur
    print(nb, [bool(torch.equal(a, b)) for a, b in zip(cur, ref)], flush=True)
