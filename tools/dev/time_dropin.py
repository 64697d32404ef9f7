"""dev tool: throughput of the reference-shaped per-image API (production.py:118-129: ProposalGenerator.generate_proposals_and_images
-> Classifier.classify -> labels) against BatchedPipeline on the same images; batch_size as cvpce/cli/eval.py passes it (8) and the
class default (32)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cvpce_amd import production, synthetic
dev = torch.device('cuda')
det = synthetic.synthetic_gln(seed=0, detections_per_img=200).to(dev)
enc = synthetic.synthetic_macvgg(seed=1).to(dev)
G = 1000
t0 = time.time()
gal_imgs = synthetic.gallery_images(G, seed=100)


class Gal:
    def __len__(self): return G
    def __getitem__(self, i): return gal_imgs[i], None, None, f'p{i}'


imgs = [synthetic.shelf_image(i, 2048, 2048) for i in range(8)]
for bs in (8, 32):
    torch.cuda.synchronize(); t = time.time()
    clf = production.Classifier(enc, Gal(), device=dev, emb_device=dev, batch_size=bs, num_workers=0)
    torch.cuda.synchronize(); tb = time.time() - t
    pg = production.ProposalGenerator(det, device=dev, confidence_threshold=0.5)
    dimgs = [im.to(dev) for im in imgs]
    for rep in range(2):
        torch.cuda.synchronize(); t = time.time(); n = 0
        for im in dimgs:
            boxes, crops = pg.generate_proposals_and_images(im)
            labels = clf.classify(crops)
            n += len(labels)
        torch.cuda.synchronize(); dt = time.time() - t
    print(f'batch_size {bs}: build_index({G}) {tb:.2f} s = {G / tb:.0f} gallery images/s; per-image API {len(imgs) / dt:.1f} images/s ({n} crops, {dt / len(imgs) * 1e3:.1f} ms/image)', flush=True)
clf = production.Classifier.from_embedding(enc, clf.embedding, clf.annotations, device=dev, emb_device=dev, match_dtype=torch.bfloat16)
pipe = production.BatchedPipeline(det, clf, 0.5)
for rep in range(3):
    torch.cuda.synchronize(); t = time.time()
    out = pipe.run(dimgs)
    torch.cuda.synchronize(); dt = time.time() - t
print(f'BatchedPipeline: {len(imgs) / dt:.1f} images/s')
# the batched drop-in evaluation (PlanogramEvaluator.detect_and_classify_batch: what `cvpce eval-planograms` runs per window of 8 images);
# images arrive as HOST tensors, like a dataset's (the upload is inside the timed region)
clf32 = production.Classifier.from_embedding(enc, clf.embedding, clf.annotations, device=dev, emb_device=dev)      # f32 matcher: the Classifier default
ev = production.PlanogramEvaluator(production.ProposalGenerator(det, device=dev, confidence_threshold=0.5), clf32, production.PlanogramComparator())
for src, what in ((imgs, 'host images'), (dimgs, 'device images')):
    for rep in range(3):
        torch.cuda.synchronize(); t = time.time()
        res = ev.detect_and_classify_batch(src)
        torch.cuda.synchronize(); dt = time.time() - t
    print(f'detect_and_classify_batch ({what}): {len(imgs) / dt:.1f} images/s ({sum(len(b) for b, _ in res)} boxes)')
# the look-ahead iterator over a stream of (image, planogram) pairs (PlanogramEvaluator.evaluate_iter): the per-image calling pattern with the
# detector / embedder passes shared by up to `lookahead` consecutive images.  The comparator here is a stub (the real one is CPU graph
# matching -- ~0.3 s per image on 200 random boxes -- and would be all that is measured)
class CountingComparator:
    def compare(self, expected, actual, image=None, classifier=None):
        return len(actual['labels'])


ev2 = production.PlanogramEvaluator(production.ProposalGenerator(det, device=dev, confidence_threshold=0.5), clf32, CountingComparator())
for rep in range(3):
    torch.cuda.synchronize(); t = time.time()
    out = [ev2.evaluate(im, None) for im in dimgs]
    torch.cuda.synchronize(); dt = time.time() - t
print(f'evaluate() per image, stub comparator: {len(imgs) / dt:.1f} images/s')
for la in (1, 2, 4, 8):
    for rep in range(3):
        torch.cuda.synchronize(); t = time.time()
        out = list(ev2.evaluate_iter(((im, None) for im in dimgs), lookahead=la))
        torch.cuda.synchronize(); dt = time.time() - t
    print(f'evaluate_iter(lookahead={la}), stub comparator: {len(imgs) / dt:.1f} images/s ({sum(out)} labels)')
