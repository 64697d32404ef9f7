"""Eval harness counterparts on the GPU: the same per-image data reaches the metric / counting
code as in the reference's loops, checked end to end on synthetic data against the oracle."""
import pytest
import torch

pytestmark = pytest.mark.gpu


class _GP180Like:
    def __init__(self, items, anns):
        self.items = items
        self.int_to_ann = list(anns)
        self.ann_to_int = {a: i for i, a in enumerate(anns)}

    def __iter__(self):
        return iter(self.items)

    def __len__(self):
        return len(self.items)


def test_eval_dihe_counts_topk(cuda):
    """classification_eval.py:6-56: paste gallery products into a shelf image, crop the ground-truth boxes, top-k accuracy."""
    from cvpce_amd import classification_eval, synthetic
    enc = synthetic.synthetic_macvgg(seed=1).to(cuda)
    gal = synthetic.gallery_images(12, seed=5)
    anns = [f'p{i}' for i in range(12)]
    shelf = torch.full((3, 300, 1100), 0.3)
    boxes, targets = [], []
    for j, i in enumerate((3, 7, 0, 11)):
        shelf[:, 20:276, 10 + j * 270:266 + j * 270] = (gal[i] + 1) / 2      # exact 256x256 paste -> identity crop
        boxes.append([10 + j * 270, 20, 266 + j * 270, 276]); targets.append(anns[i])
    testset = [(shelf, targets, torch.tensor(boxes))]
    acc = classification_eval.eval_dihe(enc, synthetic.TensorGallery(gal, anns), testset, batch_size=8, k=(1, 3))
    assert acc == {1: 1.0, 3: 1.0}
    # a wrong annotation is counted as a miss for k=1
    testset = [(shelf, ['p3', 'p7', 'p0', 'p1'], torch.tensor(boxes))]
    acc = classification_eval.eval_dihe(enc, synthetic.TensorGallery(gal, anns), testset, batch_size=8, k=(1,))
    assert acc == {1: 0.75}


def test_evaluate_gln_sync_matches_oracle_harness(cuda):
    """proposals_eval.py:19-48: GPU detections vs targets through calculate_metrics == oracle detections through it."""
    from cvpce_amd import metrics, proposals_eval, synthetic
    from oracle import gln as og
    model = synthetic.synthetic_gln(seed=0, detections_per_img=60)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to(cuda)
    imgs = [synthetic.shelf_image(s, 512, 512) for s in (21, 22)]
    ref = [og.gln_forward([i], sd, detections_per_img=60)[0] for i in imgs]
    # ground truth := the oracle's 30 best boxes per image
    dataset = [(i, {'boxes': r['boxes'][:30]}) for i, r in zip(imgs, ref)]
    got = proposals_eval.evaluate_gln_sync(model, dataset, thresholds=(0.5, 0.75), batch_size=1)
    want = metrics.calculate_metrics([d[1]['boxes'] for d in dataset], [r['boxes'] for r in ref], [r['scores'] for r in ref], (0.5, 0.75))
    for thr in (0.5, 0.75):
        assert 'raw' not in got[thr]
        # "mAP within tolerance of the oracle path": the reference's 11-point AP (cvpce/metrics.py:66-73) moves in steps of 1/11 --
        # missing ONE of the 60 ground-truth boxes (recall < 1.0 zeroes the last sample point) costs exactly 0.0909 -- so the
        # tolerance is one step; AR@300 below is the continuous measure
        assert abs(float(got[thr]['ap']) - float(want[thr]['ap'])) < 1 / 11 + 0.01
        assert abs(float(got[thr]['ar_300']) - float(want[thr]['ar_300'])) < 0.05
        assert float(want[thr]['ap']) > 0.5


def test_evaluate_detections_class_bucketing(cuda):
    """detection_eval.py:6-49: no confidence filter, unknown labels -> class -1 (excluded per class, kept in 'all')."""
    from cvpce_amd import detection_eval, synthetic
    det = synthetic.synthetic_gln(seed=0, detections_per_img=12).to(cuda)
    enc = synthetic.synthetic_macvgg(seed=1).to(cuda)
    gal = synthetic.gallery_images(6, seed=9)
    gallery_anns = ['a', 'b', 'c', 'zz_unknown', 'a', 'b']              # 'zz_unknown' is not a test-set class
    img = synthetic.shelf_image(31, 512, 512)
    pred = det([img.to(cuda)])[0]
    tset = _GP180Like([(img, {'boxes': pred['boxes'][:6].cpu(), 'labels': torch.tensor([0, 1, 2, 0, 1, 2])})], ['a', 'b', 'c'])
    per_class, overall = detection_eval.evaluate_detections(det, enc, tset, synthetic.TensorGallery(gal, gallery_anns),
                                                            proposal_batch_size=1, classification_batch_size=8)
    assert set(per_class) <= {0, 1, 2} and 0.5 in overall and 'ap' in overall[0.5]
    m = detection_eval.mean_average_metrics(per_class, (0.5,))
    assert 0.0 <= float(m[0.5]['map']) <= 1.0


def test_planogram_evaluator_end_to_end(cuda):
    """production.py:118-129 + :76-116: proposals -> classify -> graph match -> homography -> the product the detector
    missed is re-classified from the image at its projected planogram position (second trip through K9-K11)."""
    from cvpce_amd import ops, production, synthetic
    enc = synthetic.synthetic_macvgg(seed=1).to(cuda)
    gal = synthetic.gallery_images(8, seed=17)
    anns = [f'sku{i}' for i in range(8)]
    clf = production.Classifier(enc, synthetic.TensorGallery(gal, anns), device=cuda, emb_device=cuda, batch_size=8,
                                match_dtype=torch.float32)
    rows, cols, size, gap = 2, 4, 256, 16      # pasted at gallery resolution: a random-weight embedder is not scale invariant
    shelf = torch.full((3, rows * (size + gap) + gap, cols * (size + gap) + gap), 0.5)
    boxes, labels = [], []
    for r in range(rows):
        for c in range(cols):
            i = r * cols + c
            x, y = gap + c * (size + gap), gap + r * (size + gap)
            shelf[:, y:y + size, x:x + size] = (gal[i] + 1) / 2
            boxes.append([float(x), float(y), float(x + size), float(y + size)]); labels.append(anns[i])
    boxes = torch.tensor(boxes)
    planogram = {'boxes': boxes / 2.0, 'labels': labels}         # planogram drawn at half scale

    class _Detector:                                              # stands in for GLN: every product except #5
        device = cuda
        def generate_proposals_and_images(self, image):
            keep = torch.tensor([i for i in range(len(boxes)) if i != 5])
            b = boxes[keep].to(cuda)
            return b, ops.crop_resize(image.to(cuda).contiguous(), b, 256, mode=0)

    ev = production.PlanogramEvaluator(_Detector(), clf, production.PlanogramComparator())
    assert float(ev.evaluate(shelf, planogram)) == 1.0            # missing detection recovered by re-classification
    blank = shelf.clone(); blank[:, int(boxes[5, 1]):int(boxes[5, 3]), int(boxes[5, 0]):int(boxes[5, 2])] = 0.5
    assert abs(float(ev.evaluate(blank, planogram)) - 7 / 8) < 1e-6    # product really absent -> non-compliant slot


def test_evaluate_batch_equals_per_image(cuda):
    """The batched drop-in evaluation (PlanogramEvaluator.evaluate_batch, what `cvpce eval-planograms` runs over windows of 8
    images) hands the comparator exactly what the reference-shaped per-image `evaluate` does (production.py:123-129): same
    boxes, same labels, same verdicts -- images of one size share a detector / embedder / matcher pass, other sizes do not."""
    from cvpce_amd import production, synthetic
    det = synthetic.synthetic_gln(seed=0, detections_per_img=40).to(cuda)
    enc = synthetic.synthetic_macvgg(seed=1).to(cuda)
    gal = synthetic.gallery_images(24, seed=17)
    clf = production.Classifier(enc, synthetic.TensorGallery(gal, [f'sku{i % 6}' for i in range(24)]), device=cuda, emb_device=cuda,
                                batch_size=8)
    pg = production.ProposalGenerator(det, device=cuda, confidence_threshold=0.5)
    ev = production.PlanogramEvaluator(pg, clf, production.PlanogramComparator())
    imgs = [synthetic.shelf_image(60, 512, 640), synthetic.shelf_image(61, 512, 640), synthetic.shelf_image(62, 448, 512),
            synthetic.shelf_image(63, 512, 640)]
    single = []
    for im in imgs:
        boxes, crops = pg.generate_proposals_and_images(im)
        single.append((boxes.cpu(), [a[0] for a in clf.classify(crops)]))
    batch = ev.detect_and_classify_batch(imgs)
    assert sum(len(b) for b, _ in single) > 20
    for (b1, l1), (b2, l2) in zip(single, batch):
        assert torch.equal(b1, b2) and l1 == l2
    # planograms: each image's own detections, a third of them dropped and a few labels changed -> partial compliance
    planos = []
    for k, (b, l) in enumerate(single):
        keep = [i for i in range(len(b)) if i % 3 != k % 3]
        planos.append({'boxes': b[keep] * 0.5, 'labels': [('other' if i % 7 == 0 else l[i]) for i in keep]})
    want = [float(ev.evaluate(im, p)) for im, p in zip(imgs, planos)]
    got = [float(v) for v in ev.evaluate_batch(imgs, planos)]
    assert got == want and any(0.0 < v < 1.0 for v in want), (got, want)
    # the look-ahead iterator (round 5): the per-image calling pattern, pairs from a generator, windows cut at a size change and at
    # `lookahead` -- every yielded verdict is the serial call's
    for la in (1, 2, 4):
        gen = ev.evaluate_iter(((im, p) for im, p in zip(imgs, planos)), lookahead=la)
        assert [float(v) for v in gen] == want, la
    assert list(ev.evaluate_iter(iter(()))) == []
