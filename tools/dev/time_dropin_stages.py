"""dev tool: where the per-image drop-in API (production.py:118-129) spends its time: wall-clock per stage with a device
synchronise after each (so stage times add up to more than the un-instrumented loop), plus the plain loop."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cvpce_amd import production, synthetic
dev = torch.device('cuda')
det = synthetic.synthetic_gln(seed=0, detections_per_img=200).to(dev)
enc = synthetic.synthetic_macvgg(seed=1).to(dev)
G = 1000
gal = enc(synthetic.gallery_images(256, seed=100).to(dev))
gal = torch.cat([gal] * 4)[:G]
clf = production.Classifier.from_embedding(enc, gal, [f'p{i}' for i in range(G)], device=dev, emb_device=dev, batch_size=8)
pg = production.ProposalGenerator(det, device=dev, confidence_threshold=0.5)
imgs = [synthetic.shelf_image(i, 2048, 2048).to(dev) for i in range(8)]
sync = torch.cuda.synchronize
for rep in range(3):
    acc = {'detect': 0.0, 'crops': 0.0, 'classify': 0.0}
    for im in imgs:
        sync(); t = time.perf_counter()
        boxes = pg.generate_proposals(im)
        sync(); acc['detect'] += time.perf_counter() - t; t = time.perf_counter()
        boxes, crops = pg.generate_proposals_and_images(im) if False else (boxes, None)
        from cvpce_amd import ops, datautils
        keep = production._nondegenerate(boxes)
        boxes = boxes[keep]
        crops = ops.crop_resize(im.contiguous(), boxes, datautils.CLASSIFICATION_IMAGE_SIZE, mode=0)
        sync(); acc['crops'] += time.perf_counter() - t; t = time.perf_counter()
        labels = clf.classify(crops)
        sync(); acc['classify'] += time.perf_counter() - t
    print('staged (ms/image):', {k: round(v / len(imgs) * 1e3, 2) for k, v in acc.items()}, flush=True)
for rep in range(3):
    sync(); t = time.perf_counter()
    for im in imgs:
        boxes, crops = pg.generate_proposals_and_images(im)
        labels = clf.classify(crops)
    sync(); dt = time.perf_counter() - t
    print(f'per-image API: {len(imgs) / dt:.1f} images/s ({dt / len(imgs) * 1e3:.2f} ms/image)', flush=True)
# inside classify: pack / embed / match
from cvpce_amd.models.classification import TANH_MEAN, TANH_STD
boxes, crops = pg.generate_proposals_and_images(imgs[0])
for rep in range(3):
    sync(); t = time.perf_counter()
    packed = ops.pack_embed_input(crops, True, TANH_MEAN, TANH_STD)
    sync(); t1 = time.perf_counter()
    emb = enc.engine().embed_packed(packed)
    sync(); t2 = time.perf_counter()
    idx = clf.match(emb).tolist()
    sync(); t3 = time.perf_counter()
    print(f'classify parts: pack {1e3 * (t1 - t):.2f} embed {1e3 * (t2 - t1):.2f} match+tolist {1e3 * (t3 - t2):.2f} ms ({len(crops)} crops)', flush=True)
