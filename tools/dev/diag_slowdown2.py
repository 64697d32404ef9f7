"""dev diagnostic: which step of bench.py's sequence inflates the stage-timing leg?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cvpce_amd import production, synthetic, ops
dev = torch.device('cuda:0')
torch.cuda.set_device(dev)
det = synthetic.synthetic_gln(seed=0, detections_per_img=200).to(dev)
enc = synthetic.synthetic_macvgg(seed=1).to(dev)
gal = enc(synthetic.gallery_images(256, seed=100).to(dev))
clf = production.Classifier.from_embedding(enc, torch.cat([gal] * 13)[:3200], list(range(3200)), device=dev, emb_device=dev, k=1, match_dtype=torch.bfloat16)
imgs = [synthetic.shelf_image(g, 2048, 2048).to(dev) for g in range(8)]


def stage(pipe, tag):
    ev = []
    for _ in range(10):
        pipe.run(imgs, ev)
    torch.cuda.synchronize()
    print(f'{tag}: stage leg', {nm: round(sum(a.elapsed_time(b) for n_, a, b in ev if n_ == nm) / 10, 3) for nm in ('detect', 'crop', 'embed', 'match')}, flush=True)


pipe = production.BatchedPipeline(det, clf, 0.5, overlap_detector=True)
for _ in range(12):
    pipe.run(imgs, inputs_ready=True)
torch.cuda.synchronize()
pipe.overlap_detector = False
stage(pipe, 'after overlapped steps (graph captured on the side stream)')
ops.PROFILE = ops.ConvProfile()
for _ in range(10):
    pipe.run(imgs)
summ = ops.PROFILE.summary()
ops.PROFILE = None
d = summ['conv3x3_halo2_kernel']
print('ConvProfile leg: halo2 avg us', round(d['ms'] * 1e3 / d['launches'], 1), flush=True)
stage(pipe, 'after the ConvProfile leg')
det.engine().__dict__.pop('_graphs', None); det.engine().__dict__.pop('_sights', None)
for _ in range(3):
    pipe.run(imgs)
stage(pipe, 'graph re-captured on the main stream')
