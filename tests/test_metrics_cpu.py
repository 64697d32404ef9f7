"""cvpce_amd.metrics against the reference's known-answer tests (test/metrics_test.py:5-128, re-expressed against
this package) and against outputs of the reference's own implementation (tests/golden/metrics.pt)."""
import os

import pytest
import torch

from cvpce_amd import metrics

TARGETS = [
    torch.tensor([[0, 0, 1, 1], [1, 0, 2, 1], [1, 1, 2, 2]], dtype=torch.float),
    torch.tensor([[1, 1, 2, 2], [3, 1, 4, 2], [5, 1, 6, 2], [7, 1, 8, 2]], dtype=torch.float),
    torch.tensor([[0, 0, 5, 5], [5, 5, 10, 10]], dtype=torch.float),
]
PREDICTIONS = [
    torch.tensor([[0, 0, .9, .9], [1.1, 0.1, 1.9, 0.9], [0, 0, 1, 1], [0.9, 0.9, 2.1, 2.1], [3, 3, 4, 4]], dtype=torch.float),
    torch.tensor([[1, 0, 2, 1], [1, 1, 2, 2], [5, 1, 6, 2], [7, 1.1, 8, 1.9], [9, 9, 10, 10]], dtype=torch.float),
    torch.tensor([[0, 0, 1, 1], [1, 1, 3, 3], [0.5, 0.5, 4.5, 4.5], [0, 0, 6, 6], [6, 6, 9, 9]], dtype=torch.float),
]
CONFIDENCES = [
    torch.tensor([1, 0.8, 0.6, 0.4, 0.2], dtype=torch.float),
    torch.tensor([0.9, 0.8, 0.7, 0.65, 0.5], dtype=torch.float),
    torch.tensor([0.85, 0.6, 0.4, 0.2, 0.1], dtype=torch.float),
]


def test_iou_matrices_kat():
    ious, idx = metrics.iou_matrices(TARGETS[2], PREDICTIONS[2])
    want = torch.tensor([[0.04, 0], [0.16, 0], [0.64, 0], [25 / 36, 1 / (25 + 36 - 1)], [0.36, 0]])
    assert idx.equal(torch.tensor([[0, 1], [0, 1], [0, 1], [0, 1], [1, 0]])) and want.allclose(ious)
    ious, idx = metrics.iou_matrices(TARGETS[0], PREDICTIONS[0])
    want = torch.tensor([[0.81, 0, 0], [0.64, 0, 0], [1, 0, 0], [1 / 1.44, 0.1 / 2.34, 0.01 / 2.43], [0, 0, 0]])
    assert idx.equal(torch.tensor([[0, 1, 2], [1, 0, 2], [0, 1, 2], [2, 1, 0], [0, 1, 2]])) and want.allclose(ious)


def test_check_matches_kat():
    ious, idx = metrics.iou_matrices(TARGETS[0], PREDICTIONS[0])
    tp, fp = metrics.check_matches(ious, idx, iou_threshold=0.65)
    assert tp.tolist() == [1, 0, 0, 1, 0] and fp.tolist() == [0, 1, 1, 0, 1]


def _tps_fps():
    tps, fps = [], []
    for t, p in zip(TARGETS, PREDICTIONS):
        tp, fp = metrics.check_matches(*metrics.iou_matrices(t, p))
        tps.append(tp); fps.append(fp)
    return {0.5: {'true_positives': tps, 'false_positives': fps, 'recall_300': [1, 3 / 4, 1 / 2]}}


def test_merge_pr_ap_kat():
    m, conf = metrics.merge_matches(_tps_fps(), CONFIDENCES)
    tp, fp = m[0.5]['true_positives'], m[0.5]['false_positives']
    assert tp.tolist() == [1, 0, 0, 1, 1, 1, 1, 0, 0, 0, 1, 1, 0, 0, 0] and (tp + fp == 1).all()
    assert conf.allclose(torch.tensor([1, 0.9, 0.85, 0.8, 0.8, 0.7, 0.65, 0.6, 0.6, 0.5, 0.4, 0.4, 0.2, 0.2, 0.1]))
    p, r = metrics.precision_and_recall(tp, fp, 9)
    assert p.allclose(torch.tensor([1, 1/2, 1/3, 2/4, 3/5, 4/6, 5/7, 5/8, 5/9, 5/10, 6/11, 7/12, 7/13, 7/14, 7/15]))
    assert r.allclose(torch.tensor([1, 1, 1, 2, 3, 4, 5, 5, 5, 5, 6, 7, 7, 7, 7]) / 9)
    assert metrics.average_precision(p, r).isclose(torch.tensor((1 + 1 + 4 * 5 / 7 + 2 * 7 / 12) / 11))


def test_calculate_metrics_kat():
    res = metrics.calculate_metrics(TARGETS, PREDICTIONS, CONFIDENCES)[0.5]
    ep, er = torch.tensor(7 / 12), torch.tensor(7 / 9)
    assert torch.isclose(res['ap'], torch.tensor((1 + 1 + 4 * 5 / 7 + 2 * 7 / 12) / 11))
    assert torch.isclose(torch.as_tensor(res['ar_300']), torch.tensor((1 + 3 / 4 + 1 / 2) / 3))
    assert torch.isclose(res['p'], ep) and torch.isclose(res['r'], er) and torch.isclose(res['f'], 2 * ep * er / (ep + er))


def test_against_reference_outputs(golden_dir):
    g = torch.load(os.path.join(golden_dir, 'metrics.pt'), weights_only=False)
    res = metrics.calculate_metrics(g['targets'], g['predictions'], g['confidences'])[0.5]
    for k in ('ap', 'ar_300', 'p', 'r', 'f', 'c'):
        assert torch.isclose(torch.as_tensor(res[k], dtype=torch.float), torch.as_tensor(g['kat_result'][k], dtype=torch.float))
    for case in g['random']:
        got = metrics.calculate_metrics(case['targets'], case['predictions'], case['confidences'], iou_thresholds=(0.5, 0.75))
        for thr, want in case['result'].items():
            for k in ('ap', 'ar_300', 'p', 'r', 'f', 'c'):
                assert torch.isclose(torch.as_tensor(got[thr][k], dtype=torch.float), torch.as_tensor(want[k], dtype=torch.float)), (thr, k)
            for k in ('p', 'r', 'f', 'c'):
                assert torch.allclose(got[thr]['raw'][k], want['raw'][k])


def test_one_prediction_overlapping_several_targets(golden_dir):
    """The reference's matching loop has no `break` (cvpce/metrics.py:22-26): a matched prediction uses up EVERY free target
    at or above the threshold.  Reference-made outputs on dense, mutually overlapping targets."""
    g = torch.load(os.path.join(golden_dir, 'metrics.pt'), weights_only=False)
    two = g['two_targets']
    tp, fp = metrics.check_matches(*metrics.iou_matrices(two['targets'], two['predictions']))
    assert tp.tolist() == two['tp'].tolist() == [1, 0] and fp.tolist() == two['fp'].tolist()
    for case in g['dense']:
        got = metrics.calculate_metrics(case['targets'], case['predictions'], case['confidences'], iou_thresholds=(0.5, 0.75))
        for thr, want in case['result'].items():
            for k in ('ap', 'ar_300', 'p', 'r', 'f', 'c'):
                assert torch.isclose(torch.as_tensor(got[thr][k], dtype=torch.float), torch.as_tensor(want[k], dtype=torch.float)), (thr, k)
            for k in ('p', 'r', 'f', 'c'):
                assert torch.allclose(got[thr]['raw'][k], want['raw'][k])


def test_edge_cases():
    empty = metrics.calculate_metrics([torch.zeros(0, 4)], [torch.zeros(0, 4)], [torch.zeros(0)])[0.5]
    assert empty['f'] == 0.0 and empty['ap'] == 0
    none_found = metrics.calculate_metrics([TARGETS[0]], [torch.zeros(0, 4)], [torch.zeros(0)])[0.5]
    assert none_found['ar_300'] == 0
    from cvpce_amd.detection_eval import mean_average_metrics
    m = mean_average_metrics({0: {0.5: {'ap': 0.5, 'ar_300': 0.2}}, 1: {0.5: {'ap': 1.0, 'ar_300': 0.4}}}, (0.5,))
    assert m[0.5]['map'] == 0.75 and abs(m[0.5]['mar300'] - 0.3) < 1e-9


def test_calculate_metrics_async_equals_sync(golden_dir):
    """cvpce/metrics.py:140-175 (multiprocess matching): same protocol as the reference's evaluate_gln_async loop, same result as
    the synchronous routine -- checked on the reference-made dense case."""
    g = torch.load(os.path.join(golden_dir, 'metrics.pt'), weights_only=False)
    case = g['dense'][1]
    thr = (0.5, 0.75)
    queue, mqueue, pipe = metrics.calculate_metrics_async(processes=2, iou_thresholds=thr)
    for t, p, c in zip(case['targets'], case['predictions'], case['confidences']):
        queue.put((t, p, c))
    queue.join()
    for _ in range(2):
        queue.put(None)
    queue.join()
    mqueue.join()
    mqueue.put(None)
    res = pipe.recv()
    mqueue.join()
    want = metrics.calculate_metrics(case['targets'], case['predictions'], case['confidences'], iou_thresholds=thr)
    for t in thr:
        for k in ('ap', 'ar_300', 'p', 'r', 'f', 'c'):
            assert torch.equal(torch.as_tensor(res[t][k], dtype=torch.float), torch.as_tensor(want[t][k], dtype=torch.float)), (t, k)
        for k in ('p', 'r', 'f', 'c'):
            assert torch.equal(res[t]['raw'][k], want[t]['raw'][k])
