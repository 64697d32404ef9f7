"""Detection metrics of the eval harnesses (post-hoc host logic, CPU tensors) -- counterpart of
/root/reference/cvpce/metrics.py:11-138 (the multiprocess variant :140-175 and the plotting :177-204 are out of
scope).  Same function names, argument meaning and result dictionary as the reference; pinned by the reference's
known-answer tests (test/metrics_test.py) and by golden outputs made with the reference's own implementation
(tests/golden/metrics.pt).

Semantics kept on purpose: predictions are visited in descending confidence; a prediction is a true positive when at
least one target with IoU >= threshold is still unused, and -- like the reference's inner loop, which has no `break`
after a match (cvpce/metrics.py:22-26) -- it then marks EVERY still-unused target at or above the threshold as used,
not only the best one; AP is the 11-point interpolated AP; AR@300 is the recall after the first 300 predictions per
image.
"""
import torch


def box_iou(a, b):
    """(P,4),(T,4) xyxy -> (P,T); torchvision.ops.box_iou semantics (no +1, inter / union)."""
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = torch.max(a[:, None, :2], b[None, :, :2])
    rb = torch.min(a[:, None, 2:], b[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    return inter / (area_a[:, None] + area_b[None, :] - inter)


def iou_matrices(targets, sorted_predictions):
    """-> (ious sorted descending along targets, matching target indices), one row per prediction."""
    return torch.sort(box_iou(sorted_predictions, targets), dim=1, descending=True)


def check_matches(sorted_ious, indices, iou_threshold=0.5):
    n_pred, n_tgt = sorted_ious.shape
    used = torch.zeros(n_tgt, dtype=torch.bool)
    tp = torch.zeros(n_pred)
    ok = sorted_ious >= iou_threshold
    for p in range(n_pred):
        cand = indices[p][ok[p]]                 # targets in descending-IoU order, cut at the threshold
        free = cand[~used[cand]]
        if free.numel():
            used[free] = True                    # every free target above the threshold (no `break` in the reference)
            tp[p] = 1
    return tp, 1 - tp


def merge_matches(matches, confidences):
    conf, order = torch.sort(torch.cat(confidences), descending=True)
    merged = {}
    for thr, d in matches.items():
        merged[thr] = {'true_positives': torch.cat(d['true_positives'])[order],
                       'false_positives': torch.cat(d['false_positives'])[order],
                       'ar_300': sum(d['recall_300']) / len(d['recall_300'])}
    return merged, conf


def get_merge_index(c1, c2):
    return torch.sort(torch.cat((c1, c2)), descending=True)


def precision_and_recall(true_positives, false_positives, total_targets):
    tp, fp = true_positives.cumsum(0), false_positives.cumsum(0)
    precision = tp / (tp + fp)
    precision[torch.isnan(precision)] = 0
    recall = tp / total_targets if total_targets > 0 else torch.zeros_like(tp)
    return precision, recall


def f_score(precision, recall):
    f = 2 * precision * recall / (precision + recall)
    f[torch.isnan(f)] = 0
    return f


def average_precision(precision, recall):
    vals = torch.zeros(11, dtype=torch.float)
    for i, r in enumerate(torch.linspace(0, 1, 11)):
        reach = precision[recall >= r]
        if not len(reach):
            break
        vals[i] = reach.max()
    return vals.mean()


def _process_one(target, prediction, confidence, iou_thresholds):
    confidence, order = torch.sort(confidence, descending=True)
    ious, idx = iou_matrices(target, prediction[order])
    out = {}
    for thr in iou_thresholds:
        tp, fp = check_matches(ious, idx, thr)
        _, rec = precision_and_recall(tp, fp, len(target))
        out[thr] = {'true_positives': tp, 'false_positives': fp, 'recall_300': rec[:300][-1] if len(rec) > 0 else 0}
    return out, confidence, target.shape[0]


def _do_calculate(iou_thresholds, matches_for_threshold, sorted_confidences, total_targets):
    merged, conf = merge_matches(matches_for_threshold, sorted_confidences)
    res = {}
    for thr in iou_thresholds:
        p, r = precision_and_recall(merged[thr]['true_positives'], merged[thr]['false_positives'], total_targets)
        f = f_score(p, r)
        if len(f) > 0:
            best_f, at = f.max(0)
            best = (best_f, p[at], r[at], conf[at])
        else:
            best = (0.0, 0.0, 0.0, 0.0)
        res[thr] = {'raw': {'p': p, 'r': r, 'f': f, 'c': conf}, 'f': best[0], 'p': best[1], 'r': best[2], 'c': best[3],
                    'ap': average_precision(p, r), 'ar_300': merged[thr]['ar_300']}
    return res


def calculate_metrics(targets, predictions, confidences, iou_thresholds=(0.5,)):
    per_thr = {t: {'true_positives': [], 'false_positives': [], 'recall_300': []} for t in iou_thresholds}
    confs, total = [], 0
    for target, prediction, confidence in zip(targets, predictions, confidences):
        m, c, n = _process_one(target, prediction, confidence, iou_thresholds)
        confs.append(c)
        total += n
        for t in iou_thresholds:
            for key in per_thr[t]:
                per_thr[t][key].append(m[t][key])
    return _do_calculate(iou_thresholds, per_thr, confs, total)
