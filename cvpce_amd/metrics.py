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


# ---------------------------------------------------------------------------------------------------------------------
# Multiprocess variant (cvpce/metrics.py:140-175): per-image matching in worker processes while the GPU keeps detecting.
# Same protocol as the reference -- a JoinableQueue of (targets, predictions, confidences) triples, one `None` per worker
# to stop them, an output queue drained by one collector that sends the result dictionary through a pipe -- so
# proposals_eval.evaluate_gln_async reads like the reference's.  Workers are SPAWNED (not forked): the parent has the GPU
# open, the children import only this host-only module.  The collector re-orders results by image number before merging,
# so the result is identical to calculate_metrics (the reference merges in completion order, which only permutes ties).
# ---------------------------------------------------------------------------------------------------------------------
def _image_processer(input_queue, output_queue, iou_thresholds):
    torch.set_num_threads(1)
    for seq, target, prediction, confidence in iter(input_queue.get, None):
        m, c, n = _process_one(torch.from_numpy(target), torch.from_numpy(prediction), torch.from_numpy(confidence), iou_thresholds)
        # results cross the process boundary BY VALUE (numpy): torch tensors would be shared through file descriptors served
        # by the producing process, which may have exited before the collector reads them
        m = {t: {'true_positives': d['true_positives'].numpy(), 'false_positives': d['false_positives'].numpy(),
                 'recall_300': float(d['recall_300'])} for t, d in m.items()}
        output_queue.put((seq, m, c.numpy(), n))
        input_queue.task_done()
    input_queue.task_done()


def _metric_calculator(output_queue, pipe, iou_thresholds):
    results = []
    for item in iter(output_queue.get, None):
        results.append(item)
        output_queue.task_done()
    results.sort(key=lambda r: r[0])
    per_thr = {t: {'true_positives': [], 'false_positives': [], 'recall_300': []} for t in iou_thresholds}
    confs, total = [], 0
    for _, m, c, n in results:
        confs.append(torch.from_numpy(c))
        total += n
        for t in iou_thresholds:
            per_thr[t]['true_positives'].append(torch.from_numpy(m[t]['true_positives']))
            per_thr[t]['false_positives'].append(torch.from_numpy(m[t]['false_positives']))
            per_thr[t]['recall_300'].append(torch.tensor(m[t]['recall_300']) if len(m[t]['true_positives']) else 0)
    res = _do_calculate(iou_thresholds, per_thr, confs, total) if results else None
    if res is not None:       # the result dictionary goes through the pipe by value as well
        res = {t: {k: ({kk: vv.numpy() for kk, vv in v.items()} if k == 'raw' else (v.item() if torch.is_tensor(v) else v))
                   for k, v in d.items()} for t, d in res.items()}
    pipe.send(res)
    output_queue.task_done()


def _result_from_pipe(res):
    if res is None:
        return None
    return {t: {k: ({kk: torch.from_numpy(vv) for kk, vv in v.items()} if k == 'raw' else torch.tensor(v)) for k, v in d.items()}
            for t, d in res.items()}


class _ResultPipe:
    def __init__(self, conn):
        self._c = conn

    def recv(self):
        return _result_from_pipe(self._c.recv())


class _SequencedQueue:
    """The reference's `queue.put((targets, predictions, confidences))` interface over a JoinableQueue whose items also
    carry the image's sequence number."""

    def __init__(self, queue):
        self._q, self._n = queue, 0

    def put(self, item):
        if item is None:
            self._q.put(None)
        else:
            self._q.put((self._n,) + tuple(t.detach().cpu().contiguous().numpy() for t in item))
            self._n += 1

    def join(self):
        self._q.join()


def calculate_metrics_async(processes=4, iou_thresholds=(0.5,)):
    """-> (input queue, output queue, result pipe), used exactly like the reference's (cvpce/metrics.py:163-175)."""
    import multiprocessing as mp
    ctx = mp.get_context('spawn')
    iou_thresholds = tuple(iou_thresholds)
    input_queue, output_queue = ctx.JoinableQueue(), ctx.JoinableQueue()
    out_pipe, in_pipe = ctx.Pipe()
    for _ in range(processes):
        ctx.Process(target=_image_processer, args=(input_queue, output_queue, iou_thresholds), daemon=True).start()
    ctx.Process(target=_metric_calculator, args=(output_queue, in_pipe, iou_thresholds), daemon=True).start()
    return _SequencedQueue(input_queue), output_queue, _ResultPipe(out_pipe)
