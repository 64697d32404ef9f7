"""Detect -> crop -> embed -> match -> per-class mAP harness -- counterpart of
/root/reference/cvpce/detection_eval.py:6-55.  `testset` yields `(image, target dict with 'boxes','labels')` and
has `ann_to_int` / `int_to_ann` like the reference's GP-180 dataset.  As in the reference there is NO confidence
filter here: every kept detection is embedded (detection_eval.py:29); unknown labels map to class -1 (:31)."""
import torch

from . import metrics, ops, production, datautils
from .proposals_eval import _batches


@torch.no_grad()
def evaluate_detections(p_model, c_model, testset, trainset, thresholds=(0.5,), proposal_batch_size=2,
                        classification_batch_size=16, num_workers=0, load_classifier_index=None, verbose=False):
    classifier = production.Classifier(c_model, trainset, batch_size=classification_batch_size,
                                       num_workers=num_workers, load=load_classifier_index)
    n_cls = len(testset.int_to_ann)
    predictions = {c: [] for c in range(n_cls)}
    targets = {c: [] for c in range(n_cls)}
    confidences = {c: [] for c in range(n_cls)}
    all_p, all_t, all_c = [], [], []
    # detect -> crop -> embed -> match of a whole proposal batch in one pass of the kernels (the reference crops and classifies image
    # by image, detection_eval.py:27-30); confidence threshold -1: every kept detection is embedded, as in the reference
    pipe = production.BatchedPipeline(p_model, classifier, confidence_threshold=-1.0) if hasattr(p_model, 'engine') else None
    for i, batch in enumerate(_batches(testset, proposal_batch_size)):
        if verbose and i % 10 == 0:
            print(f'{i}...')
        images = [img.cuda(non_blocking=True).to(torch.float32).contiguous() for img, _ in batch]
        if pipe is not None:
            out = pipe.run(images)
            ob, osc, oix = out['boxes'].cpu(), out['scores'].cpu(), out['indices'][:, :, 0].cpu()
            results = [{'boxes': ob[j, :c], 'scores': osc[j, :c], 'idx': oix[j, :c]} for j, c in enumerate(out['counts_host'])]
        else:
            results = p_model(images)
        for img, r, (_, t) in zip(images, results, batch):
            boxes = r['boxes']
            keep = production._nondegenerate(boxes) if len(boxes) else torch.zeros(0, dtype=torch.bool, device=boxes.device)
            boxes, scores = boxes[keep], r['scores'][keep]
            if 'idx' in r:
                classes = [[classifier.annotations[k]] for k in r['idx'][keep].tolist()]
            elif len(boxes):
                crops = ops.crop_resize(img.contiguous(), boxes, datautils.CLASSIFICATION_IMAGE_SIZE, mode=0)
                classes = classifier.classify(crops)
            else:
                classes = []
            cls = torch.tensor([testset.ann_to_int.get(a[0], -1) for a in classes], dtype=torch.long)
            boxes, scores = boxes.detach().cpu(), scores.detach().cpu()
            t_boxes, t_labels = t['boxes'].detach().cpu(), t['labels'].detach().cpu()
            for c in {int(x) for x in cls} | {int(x) for x in t_labels}:
                b, s, tb = boxes[cls == c], scores[cls == c], t_boxes[t_labels == c]
                all_p.append(b); all_c.append(s); all_t.append(tb)
                if c != -1:
                    predictions[c].append(b); confidences[c].append(s); targets[c].append(tb)
    strip = lambda res: {thr: {k: v for k, v in itm.items() if k != 'raw'} for thr, itm in res.items()}
    per_class = {c: strip(metrics.calculate_metrics(targets[c], predictions[c], confidences[c], thresholds))
                 for c in range(n_cls) if targets[c] or predictions[c]}
    return per_class, strip(metrics.calculate_metrics(all_t, all_p, all_c, thresholds))


def mean_average_metrics(metrics_per_class, thresholds):
    return {t: {'map': sum(d[t]['ap'] for d in metrics_per_class.values()) / len(metrics_per_class),
                'mar300': sum(d[t]['ar_300'] for d in metrics_per_class.values()) / len(metrics_per_class)}
            for t in thresholds}
