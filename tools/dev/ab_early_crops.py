"""dev A/B (one process, alternating): BatchedPipeline.run with the host sync beside the crop kernels vs before them."""
import sys, os, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cvpce_amd import production, synthetic
dev = torch.device('cuda')
det = synthetic.synthetic_gln(seed=0, detections_per_img=200).to(dev)
enc = synthetic.synthetic_macvgg(seed=1).to(dev)
gal = enc(synthetic.gallery_images(256, seed=100).to(dev))
clf = production.Classifier.from_embedding(enc, gal, list(range(256)), device=dev, emb_device=dev, match_dtype=torch.bfloat16)
pipe = production.BatchedPipeline(det, clf, 0.5)
imgs = [synthetic.shelf_image(i, 2048, 2048).to(dev) for i in range(8)]
def run_old(self, images):
    d = self.detector
    det_out = d.engine().detect(images, d.num_classes, d.detections_per_img, self.confidence_threshold)
    counts = det_out[4].tolist()
    crops, valid, sel = self._crop_embed_match(images, det_out, counts)
    emb = self.classifier.encoder.engine().embed_packed(valid)
    idx = self.classifier.match(emb)
    return self._finish(images, det_out, counts, emb, idx, sel)
for _ in range(3): pipe.run(imgs); run_old(pipe, imgs)
for rep in range(3):
    for name, fn in (('new', lambda: pipe.run(imgs)), ('old', lambda: run_old(pipe, imgs))):
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): fn()
        torch.cuda.synchronize(); ms = (time.perf_counter() - t) * 100
        print(name, round(ms, 3), 'ms/step', round(8000 / ms, 2), 'img/s', flush=True)
