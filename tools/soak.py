"""dev tool: 150 pipeline passes over the same 8 images; every pass must reproduce the first one bit for bit (races in the
team barriers, counted waits or side streams would show up as mismatches)."""
import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvpce_amd import production, synthetic
dev = torch.device('cuda')
det = synthetic.synthetic_gln(seed=0, detections_per_img=200).to(dev)
enc = synthetic.synthetic_macvgg(seed=1).to(dev)
gal = synthetic.gallery_images(512, seed=100)
clf = production.Classifier(enc, synthetic.TensorGallery(gal), device=dev, emb_device=dev, batch_size=128, k=1)
pipe = production.BatchedPipeline(det, clf, 0.5)
imgs = [synthetic.shelf_image(1000 + i, 2048, 2048).to(dev) for i in range(8)]
ref = pipe.run(imgs)
bad = 0
for it in range(150):
    out = pipe.run(imgs)
    for k in ('boxes', 'scores', 'indices', 'embeddings'):
        if not torch.equal(out[k], ref[k]):
            bad += 1; print('MISMATCH iter', it, k, (out[k].float() - ref[k].float()).abs().max().item())
print('soak done, mismatches:', bad)
