"""dev A/B in ONE process: the bench's pipeline step (8 x 2048^2, G = 3200) under module switches, alternating."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cvpce_amd import production, synthetic
from cvpce_amd.models import classification as C
dev = torch.device('cuda')
det = synthetic.synthetic_gln(seed=0, detections_per_img=200).to(dev)
enc = synthetic.synthetic_macvgg(seed=1).to(dev)
gal = torch.nn.functional.normalize(torch.rand(3200, 1024, generator=torch.Generator().manual_seed(1)), dim=1).to(dev)
clf = production.Classifier.from_embedding(enc, gal, [str(i) for i in range(3200)], device=dev, emb_device=dev, match_dtype=torch.bfloat16)
pipe = production.BatchedPipeline(det, clf, 0.5)
imgs = [synthetic.shelf_image(i, 2048, 2048).to(dev) for i in range(8)]


def t(steps=10):
    for _ in range(2): pipe.run(imgs)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): pipe.run(imgs)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps * 1e3


cases = {'base': {}, 'late layers in passes': {'LATE_EMBED_MAX': 959}}
for rep in range(3):
    for name, sw in cases.items():
        C.LATE_EMBED_MAX = sw.get('LATE_EMBED_MAX', 2000)
        print(f'{name:24s} {t():.3f} ms', flush=True)
