"""End-to-end accuracy at the north-star tolerance: the whole HIP pipeline with the product defaults of bench.py (bf16
activations, bf16 distance GEMM) against the whole fp32 oracle on structured shelf images -- tests/accuracy.py, the same
code bench.py uses for its `parity` object.  Thresholds are the figures MEASURED on MI355X for the full-size run
(profiles/r02_accuracy.json, 32 images of 2048^2) with head-room for the smaller sample used here; they replace the
80 % / 5-point bounds of round 1."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def report(cuda):
    import accuracy                      # tests/accuracy.py
    import os
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    return accuracy.run(n_images=8, image_size=1024, galleries=(256,), dpi=200, queries=96, oracle_device='cpu',
                        match_dtypes=('bf16', 'f32'), images_per_batch=8)


def test_detection_agreement(report):
    d = report['detection']
    assert d['oracle_boxes'] >= 8 * 150
    # the oracle's detections as ground truth: AP / AR300 of the HIP detections (cvpce/proposals_eval.py:19-48 metric code)
    assert d['ap50_vs_oracle'] > 0.95, d
    assert d['ar300_vs_oracle'] > 0.95, d
    assert d['frac_oracle_boxes_iou90'] > 0.90, d
    assert d['paired_abs_score_diff_mean'] < 5e-3 and d['paired_box_diff_px_mean'] < 1.0, d
    # the "mAP delta" of north_star: both detectors scored against the same pseudo ground truth
    assert abs(d['pseudo_gt']['delta_pt']) <= 1.0, d['pseudo_gt']
    assert abs(d['pseudo_gt']['delta_ar300_pt']) <= 1.0, d['pseudo_gt']
    assert abs(d['count_hip'] - d['count_oracle']) <= 0.02 * d['count_oracle']


def test_matching_agreement(report):
    assert report['gallery_embedding_cosine_min'] > 0.999 and report['embedding_cosine_min_gt_crops'] > 0.999
    for key, m in report['matching'].items():
        assert m['pairs']['n'] >= 64 and m['gt_boxes']['n'] >= 64, (key, m)
        assert m['gt_boxes']['top1_agree'] >= 0.97, (key, m)            # same crops through both paths
        assert abs(m['gt_boxes']['delta_pt']) <= 2.0, (key, m)          # top-1 accuracy vs the true product id, HIP - oracle
        assert m['pairs']['top1_agree'] >= 0.95, (key, m)               # whole pipeline vs whole oracle on paired detections
    # bf16 distance GEMM vs exact-f32 distance GEMM on the same HIP embeddings: the rounding of gallery + queries to bf16
    bf, f32 = report['matching']['G256_bf16'], report['matching']['G256_f32']
    assert abs(bf['gt_boxes']['acc_hip'] - f32['gt_boxes']['acc_hip']) <= 0.02
