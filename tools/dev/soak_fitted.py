"""dev tool: the pipeline on structured scenes with the fitted detector head (ragged confident counts: the compaction path), 60 passes,
every pass bit-identical to the first; and the same results with the work lists off."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import accuracy
from cvpce_amd import production, synthetic
from cvpce_amd.models import classification as C
dev = torch.device('cuda')
det = accuracy.fitted_detector(200, 'fp16')[0].to(dev)
enc = synthetic.synthetic_macvgg(seed=1).to(dev)
clf = production.Classifier(enc, synthetic.TensorGallery(synthetic.gallery_images(256, seed=100)), device=dev, emb_device=dev, batch_size=128, k=1)
products = synthetic.product_images(256, seed=200)
imgs = [synthetic.structured_shelf(i, 2048, 2048, products)[0].to(dev) for i in range(8)]
pipe = production.BatchedPipeline(det, clf, 0.5)
ref = pipe.run(imgs)
print('confident per image', ref['counts_host'])
bad = 0
for it in range(60):
    out = pipe.run(imgs)
    for k in ('boxes', 'scores', 'indices', 'embeddings'):
        if not torch.equal(out[k], ref[k]):
            bad += 1; print('MISMATCH iter', it, k)
C.SKIP_PADDING = False
off = pipe.run(imgs)
for k in ('boxes', 'scores', 'indices', 'embeddings'):
    if not torch.equal(off[k], ref[k]):
        bad += 1; print('MISMATCH lists off', k)
print('soak done, mismatches:', bad)
