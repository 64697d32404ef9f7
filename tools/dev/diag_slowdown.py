"""dev diagnostic: why are the detector-only workload and the stage-timing leg slower inside bench.py's process than alone?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from cvpce_amd import production, synthetic
dev = torch.device('cuda:0')
torch.cuda.set_device(dev)


def det_only(tag):
    w = bench.detector_workload(dev, 4, 1000, 2048, 20, 5, 'bf16', collective=False)
    print(f'{tag}: detector 4 x dpi 1000: {w["ms_per_step"]} ms', flush=True)


det_only('fresh process')
det = synthetic.synthetic_gln(seed=0, detections_per_img=200).to(dev)
enc = synthetic.synthetic_macvgg(seed=1).to(dev)
gal = enc(synthetic.gallery_images(256, seed=100).to(dev))
clf = production.Classifier.from_embedding(enc, torch.cat([gal] * 13)[:3200], list(range(3200)), device=dev, emb_device=dev, k=1, match_dtype=torch.bfloat16)
imgs = [synthetic.shelf_image(g, 2048, 2048).to(dev) for g in range(8)]
for overlap in (False, True):
    pipe = production.BatchedPipeline(det, clf, 0.5, overlap_detector=overlap)
    for _ in range(3):
        pipe.run(imgs, inputs_ready=True)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10):
        pipe.run(imgs, inputs_ready=True)
    torch.cuda.synchronize()
    print(f'overlap={overlap}: {(time.perf_counter() - t) * 100:.2f} ms per step', flush=True)
    ev = []
    for _ in range(10):
        pipe.run(imgs, ev)
    torch.cuda.synchronize()
    print('   stage leg:', {nm: round(sum(a.elapsed_time(b) for n_, a, b in ev if n_ == nm) / 10, 3) for nm in ('detect', 'crop', 'embed', 'match')}, flush=True)
    det_only(f'after pipeline overlap={overlap}')
del pipe
torch.cuda.empty_cache()
det_only('after empty_cache')
