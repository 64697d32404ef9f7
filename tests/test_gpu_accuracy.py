"""End-to-end accuracy at the north-star tolerance: the whole HIP pipeline with the product defaults of bench.py (detector storing
fp16 -- the default since round 5, the mode that meets the 0.1 pt tolerance; bf16 embedder, bf16 distance GEMM) against the whole fp32
oracle on structured shelf images -- tests/accuracy.py, the same code bench.py uses for its `parity` object.  The report's top level
is the DEFAULT mode; the opt-in bf16 detector storage is measured beside it (`by_precision`).  Thresholds are the figures MEASURED on MI355X (full-size run:
profiles/r02_accuracy.json) with head-room for the smaller sample used here; they replace round 1's 80 % / 5-point bounds.

What the measurement shows (DESIGN.md "Accuracy"): the embed + match half of the path agrees with the fp32 oracle to the
index (same crops -> same top-1, accuracy delta 0.0 pt).  The detector's confident boxes agree in score to ~2e-4, but with
RANDOM-INIT weights its score field is dense and unstructured, so the top-200 / NMS decisions sit on near-ties that the
bf16 storage of weights and activations (~1.8 % of the logit spread after 60 layers) moves: ~7 % of the kept boxes differ.
The control shows this is the floor of the stated numerics, not a kernel defect: a CPU emulation of the same rounding
points (oracle/bf16_model.py) deviates from the fp32 oracle just as much as the HIP path does -- and the HIP path from that
emulation (the head logits of any two of the three differ by 1-2 % rms: tools/dev/diag_noise.py)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def report(cuda):
    import accuracy                      # tests/accuracy.py
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    return accuracy.run(n_images=8, image_size=1024, galleries=(256,), dpi=200, queries=96, oracle_device='cpu',
                        match_dtypes=('bf16', 'f32'), images_per_batch=8, control_images=4, precisions=('fp16', 'bf16'))


def test_default_mode_is_the_accuracy_mode(report):
    """The product default (`gln()` without `precision`, bench.py without `--detector-precision`) is fp16 storage, and the report's top
    level is that mode."""
    from cvpce_amd import ops, synthetic
    assert ops.DEFAULT_DETECTOR_PRECISION == 'fp16' and synthetic.synthetic_gln(seed=0, calibrate=False).precision == 'fp16'
    assert report['detector_precisions'][0] == 'fp16' and report['detection'] is report['by_precision']['fp16']['detection']


def test_detection_agreement(report):
    """The opt-in bf16 detector storage on the random-weight detector: the floor of 7 mantissa bits, with its CPU-emulation control."""
    d = report['by_precision']['bf16']['detection']
    assert d['oracle_boxes'] >= 8 * 150
    # the oracle's detections as ground truth: AP / AR300 of the HIP detections (cvpce/proposals_eval.py:19-48 metric code)
    assert d['ap50_vs_oracle'] > 0.84, d
    assert d['ar300_vs_oracle'] > 0.93, d
    assert d['frac_oracle_boxes_iou90'] > 0.88, d
    assert d['paired_abs_score_diff_mean'] < 1e-3 and d['paired_box_diff_px_mean'] < 1.5, d
    assert abs(d['count_hip'] - d['count_oracle']) <= 0.02 * d['count_oracle']
    # against the products' true boxes (what north_star's "mAP within 0.1 pt" is quoted on): both detectors score alike
    assert abs(d['gt']['delta_pt']) <= 0.1, d['gt']
    c = d['control_bf16_emulation']
    # Control: a CPU emulation of the SAME rounding points (oracle/bf16_model.py) is no closer to the fp32 oracle than the HIP
    # path is, and the HIP path is as close to that emulation as to the oracle: three computations that differ only by
    # rounding (fp32 / bf16 storage in two summation orders) disagree pairwise on the same ~10 % of near-tie selections --
    # the deviation is the floor of bf16 storage on this random-weight detector, not a kernel defect.
    e_o, h_o, h_e = c['emulation_vs_oracle'], c['hip_vs_oracle_same_images'], c['hip_vs_emulation']
    assert abs(h_o['frac_boxes_iou90'] - e_o['frac_boxes_iou90']) <= 0.04, c
    assert abs(h_o['ap50'] - e_o['ap50']) <= 0.05, c
    assert h_e['frac_boxes_iou90'] >= e_o['frac_boxes_iou90'] - 0.04 and h_e['ap50'] >= e_o['ap50'] - 0.05, c


def test_fp16_accuracy_mode_closes_the_detector_gap(report):
    """`gln(..., precision='fp16')`: the same kernels on fp16 storage (10 mantissa bits instead of 7, same MFMA rate).  Thresholds
    = the round-3 review's targets (>= 98 % of the oracle's boxes at IoU > 0.9, AP50-vs-oracle >= 0.97 -- on the all-point AP:
    the reference's 11-point AP is capped at 10/11 unless recall is exactly 1.0) with head-room for this 8-image sample;
    full-size figures: profiles/r03_accuracy.json."""
    b, f = report['by_precision']['bf16']['detection'], report['by_precision']['fp16']['detection']
    assert f['oracle_boxes'] >= 8 * 150
    assert f['frac_oracle_boxes_iou90'] >= 0.975, f
    assert f['ap50_area_vs_oracle'] >= 0.97 and f['ar300_vs_oracle'] >= 0.985, f
    assert f['ap50_vs_oracle'] >= 0.90, f                                  # the 11-point form, at its 10/11 ceiling
    assert f['paired_box_diff_px_mean'] < 0.3 and f['paired_abs_score_diff_mean'] < 2e-4, f
    assert abs(f['pseudo_gt']['delta_area_pt']) <= 1.5 < abs(b['pseudo_gt']['delta_area_pt']), (f['pseudo_gt'], b['pseudo_gt'])   # full size: -0.8 vs -4.7 pt
    # and it is an improvement over the bf16 storage on every agreement figure
    assert f['frac_oracle_boxes_iou90'] > b['frac_oracle_boxes_iou90'] + 0.03
    assert f['paired_box_diff_px_mean'] < 0.5 * b['paired_box_diff_px_mean']
    pb, pf = report['by_precision']['bf16']['matching_pairs'], report['by_precision']['fp16']['matching_pairs']
    for key in pf:
        assert pf[key]['top1_agree'] >= pb[key]['top1_agree'] - 0.02, (key, pf[key], pb[key])
        assert pf[key]['top1_agree'] >= 0.93, (key, pf[key])              # whole-pipeline index agreement on paired detections


def test_matching_agreement(report):
    assert report['gallery_embedding_cosine_min'] > 0.9999 and report['embedding_cosine_min_gt_crops'] > 0.9999
    for key, m in report['matching'].items():
        assert m['pairs']['n'] >= 64 and m['gt_boxes']['n'] >= 64, (key, m)
        assert m['gt_boxes']['top1_agree'] >= 0.98, (key, m)            # same crops through both paths: same matched index
        assert abs(m['gt_boxes']['delta_pt']) <= 1.1, (key, m)          # top-1 accuracy vs the true product id, HIP - oracle (1 of 96 = 1.04 pt)
        assert m['pairs']['top1_agree'] >= 0.80, (key, m)               # paired detections: crops differ by the boxes' sub-pixel drift
    # bf16 distance GEMM vs exact-f32 distance GEMM on the same HIP embeddings: the rounding of gallery + queries to bf16
    bf, f32 = report['matching']['G256_bf16'], report['matching']['G256_f32']
    assert abs(bf['gt_boxes']['acc_hip'] - f32['gt_boxes']['acc_hip']) <= 0.011


# ---------------------------------------------------------------------------------------------------------------------
# Round 4: the north-star tolerance read against TRUE boxes.  The random-init detector above finds no product (AP50 against the
# pasted products' boxes ~ 0.001 for every implementation), so "mAP within 0.1 pt" could only be read as agreement on a dense
# noise score field.  tests/golden/fitted_head.pt holds a RetinaNet head FITTED on structured shelf scenes (tests/golden/
# fit_head.py: focal + L1 loss of torchvision's RetinaNet over the frozen, seeded backbone): the oracle's AP50 against the true
# boxes is 0.80-0.91, the score field is bimodal like a trained detector's.  Full-size figures: profiles/r04_accuracy.json.
# ---------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope='module')
def fitted(cuda):
    import accuracy
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    return accuracy.run(n_images=8, image_size=1024, galleries=(256,), dpi=200, queries=96, oracle_device='cpu',
                        match_dtypes=('bf16', 'f32'), images_per_batch=8, control_images=0, precisions=('fp16', 'bf16'), detector='fitted')


def test_fitted_detector_map_against_true_boxes(fitted):
    """AP50 / AP75 / AR300 of the reference's metric code (cvpce/proposals_eval.py:19-48, cvpce/metrics.py:66-73,116-138) for the
    HIP detector and for the fp32 oracle, both against the pasted products' true boxes: |delta| <= 0.1 pt in the DEFAULT mode
    (fp16 storage; north_star's tolerance) on AP50, AP75, AR300 and the all-point AP50; the opt-in bf16 storage is within half a point
    (measured +0.21 / -0.19 / 0.0 pt here, +0.01 / +0.14 / +0.03 pt on the 32 full-size scenes)."""
    b, f = fitted['by_precision']['bf16']['detection'], fitted['by_precision']['fp16']['detection']
    assert fitted['detection'] is f                                       # the report's top level = the default mode
    g = f['gt']
    assert g['true_boxes'] >= 200 and g['ap50_oracle'] >= 0.7 and g['ar300_oracle'] >= 0.9, g      # non-vacuous: the detector finds the products
    assert abs(g['delta_pt']) <= 0.1 and abs(g['delta75_pt']) <= 0.1 and abs(g['delta_ar300_pt']) <= 0.1, g
    assert abs(100 * (g['ap50_area_hip'] - g['ap50_area_oracle'])) <= 0.1, g
    gb = b['gt']
    # (the opt-in bf16 storage: 7 mantissa bits move boxes across the IoU 0.75 line -- on this 8-scene sample of ~400 boxes one box is 0.25 pt of
    #  AP75; measured over the round's builds: -0.19 ... +0.60 pt here, +0.14 pt on the 32 full-size scenes of profiles/r05_accuracy.json)
    assert abs(gb['delta_pt']) <= 0.5 and abs(gb['delta75_pt']) <= 1.0 and abs(gb['delta_ar300_pt']) <= 0.5, gb
    # agreement with the oracle's own detections: a structured score field has no dense near-ties for 16-bit storage to flip
    assert f['frac_oracle_boxes_iou90'] >= 0.99 and f['paired_box_diff_px_mean'] < 0.1, f
    assert b['frac_oracle_boxes_iou90'] >= 0.93 and b['paired_box_diff_px_mean'] < 0.6, b
    assert abs(f['confident_hip'] - f['confident_oracle']) <= 2 and abs(b['confident_hip'] - b['confident_oracle']) <= 4


def test_fitted_pipeline_top1_against_true_products(fitted):
    """End to end: own box -> own crop -> own embedding -> own nearest neighbour, against the id of the pasted product under the
    detection, HIP pipeline vs fp32 oracle path.  ~70 samples here (one flip = 1.5 pt): at most one flip apart; the full-size
    report resolves it (861-879 samples: +0.0 ... +0.34 pt)."""
    for prec in ('bf16', 'fp16'):
        for key, m in fitted['by_precision'][prec]['matching_pairs'].items():
            v = m['vs_true_product']
            assert v['n'] >= 50 and v['top1_acc_oracle'] >= 0.6, (prec, key, v)
            assert abs(v['delta_pt']) <= 100.0 / v['n'] + 1e-6, (prec, key, v)
            assert m['top1_agree'] >= (0.93 if prec == 'fp16' else 0.9), (prec, key, m)
    for key, m in fitted['matching'].items():                          # the true boxes cropped and matched by both paths:
        # top-1 accuracy against the pasted product's id within 0.1 pt (measured 0.0 here; 0.00 / +-0.02 pt on 4 096 crops at full size)
        assert abs(m['gt_boxes']['delta_pt']) <= 0.1 and m['gt_boxes']['top1_agree'] >= 0.98, (key, m['gt_boxes'])
