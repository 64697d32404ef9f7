"""The oracle against golden vectors produced by the reference's own code
(tests/golden/make_golden.py) and against the reference's known-answer tests."""
import os

import pytest
import torch

from oracle import gln as ogln
from oracle import match as omatch


def _load(golden_dir, name):
    return torch.load(os.path.join(golden_dir, name), weights_only=False)


@pytest.mark.parametrize('tanh', [False, True])
def test_gaussian_head_matches_reference(golden_dir, tanh):
    g = _load(golden_dir, 'gaussian_head.pt')[f'tanh_{tanh}']
    sd = {f'backbone.gaussian_layer.{k}': v for k, v in g['layer_state'].items()}
    sd.update({f'backbone.gaussian_subnet.{k}': v for k, v in g['subnet_state'].items()})
    feat = ogln.gaussian_layer(g['c2'], g['p3'], sd)
    out = ogln.gaussian_subnet(feat, sd, tanh)
    assert feat.shape == g['features'].shape and out.shape == g['gaussians'].shape
    torch.testing.assert_close(feat, g['features'], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(out, g['gaussians'], rtol=1e-5, atol=1e-5)


def test_nearest_neighbors_reference_kat(golden_dir):
    """test/models/classification_test.py:8-25"""
    kat = _load(golden_dir, 'nearest.pt')['kat']
    for fn in (omatch.nearest_neighbors_literal, omatch.nearest_neighbors):
        assert kat['expected'].equal(fn(kat['anchors'], kat['queries'])[:, 0])


def test_nearest_neighbors_matches_reference(golden_dir):
    for case in _load(golden_dir, 'nearest.pt')['cases']:
        a, q, k = case['anchors'], case['queries'], case['k']
        lit = omatch.nearest_neighbors_literal(a, q, k)
        assert lit.equal(case['indices'])  # same algorithm -> bit-identical distances -> identical order
        d = omatch.cosine_distance_matrix(a, q)
        torch.testing.assert_close(d, case['distances'], rtol=0, atol=2e-6)
        gemm = omatch.nearest_neighbors(a, q, k)
        # GEMM restatement: identical wherever the reference's own ordering is not a near-tie
        srt = case['distances'].sort(dim=-1).values[:, :k + 1]
        safe = (srt[:, 1:] - srt[:, :-1]).min(dim=1).values > 1e-5
        assert gemm[safe].equal(case['indices'][safe])
        assert safe.float().mean() > 0.8


def test_distance_matches_reference(golden_dir):
    case = _load(golden_dir, 'nearest.pt')['cases'][1]
    a, q = case['anchors'], case['queries']
    d = omatch.distance(a[None].expand(len(q), -1, -1), q[:, None].expand(-1, len(a), -1), dim=-1)
    assert d.equal(case['distances'])


# ---- round 3: reference-made fixtures for members that run without torchvision (tests/golden/members.pt) -----------------
def _macresnet_state(m, case):
    return {k: m['source_state'][src] for k, src in case['state_key_to_source_key'].items()}


def test_macresnet_forward_matches_reference(golden_dir):
    """cvpce/models/classification.py:53-85 executed by the reference over a hand-built resnet-like source: Sequential key
    nesting, per-block amax, concatenation order, L2 normalisation."""
    from oracle import macresnet as ores
    m = _load(golden_dir, 'members.pt')['macresnet']
    for case in m['cases']:
        sd = _macresnet_state(m, case)
        out = ores.macresnet_forward(case['input'], sd, tuple(case['descriptor_layers']), layers=tuple(m['layers']))
        assert out.shape == case['output'].shape
        torch.testing.assert_close(out, case['output'], rtol=1e-5, atol=1e-6)
        zero = ores.macresnet_forward(torch.zeros(1, 3, 64, 64), sd, tuple(case['descriptor_layers']), layers=tuple(m['layers']))
        torch.testing.assert_close(zero, case['output_zero_input'], rtol=1e-5, atol=1e-6)
    # the key nesting the reference produces for the default descriptor layers [2, 3]
    keys = set(m['cases'][0]['state_key_to_source_key'])
    assert {'blocks.0.0.0.weight', 'blocks.0.0.1.running_var', 'blocks.0.1.0.conv1.weight', 'blocks.0.2.1.bn3.bias',
            'blocks.1.0.0.downsample.0.weight'} <= keys and not any(k.startswith('blocks.2') for k in keys)


def test_mean_average_metrics_matches_reference(golden_dir):
    """cvpce/detection_eval.py:51-55 on per-class results of the reference's calculate_metrics."""
    from cvpce_amd import metrics
    m = _load(golden_dir, 'members.pt')['mean_average_metrics']
    per_class = {}
    for c, inp in m['inputs'].items():
        r = metrics.calculate_metrics(inp['targets'], inp['predictions'], inp['confidences'], (0.5, 0.75))
        per_class[c] = {t: {k: v for k, v in d.items() if k != 'raw'} for t, d in r.items()}
        for t in (0.5, 0.75):
            assert float(per_class[c][t]['ap']) == pytest.approx(float(m['per_class'][c][t]['ap']), abs=1e-6)
            assert float(per_class[c][t]['ar_300']) == pytest.approx(float(m['per_class'][c][t]['ar_300']), abs=1e-6)
    # the function under test is host-only arithmetic; cvpce_amd.detection_eval needs the built HIP library to import
    from cvpce_amd import detection_eval
    got = detection_eval.mean_average_metrics(per_class, (0.5, 0.75))
    for t in (0.5, 0.75):
        assert float(got[t]['map']) == pytest.approx(m['result'][t]['map'], abs=1e-6)
        assert float(got[t]['mar300']) == pytest.approx(m['result'][t]['mar300'], abs=1e-6)


def test_planogram_comparator_early_outs_match_reference(golden_dir):
    """cvpce/production.py:79-90: nothing detected -> 0 (1 when nothing was expected either), no common subgraph -> 0."""
    from cvpce_amd import production
    cmp_ = production.PlanogramComparator()
    for e in _load(golden_dir, 'members.pt')['comparator_early_outs']:
        img = torch.zeros(3, *e['image_hw']) if 'image_hw' in e else None
        assert float(cmp_.compare(e['expected'], e['actual'], img)) == e['result'], e['name']


def test_resnet_body_matches_third_party_implementation(golden_dir):
    """The ResNet-50 body of the oracle (stem, 3-4-6-3 bottlenecks with the stride on the 3x3, FrozenBatchNorm) against a THIRD
    implementation of the same architecture -- Hugging Face transformers.ResNetModel at reduced width, eval-mode BatchNorm with
    random statistics (tests/golden/make_thirdparty.py).  torchvision, which the reference takes this body from
    (/root/reference/cvpce/models/proposals.py:176-181), is not in the image and the reference has no fixture for it: this pins
    the restatement to an independent implementation instead (weights are stored under torchvision's key names)."""
    fx = _load(golden_dir, 'resnet_body_hf.pt')
    feats = ogln.resnet_body(fx['x'], fx['state_dict'], prefix='backbone.body')
    assert list(feats) == ['0', '1', '2', '3']
    for got, ref in zip(feats.values(), fx['stages']):
        assert got.shape == ref.shape
        assert float((got - ref).abs().max()) <= 1e-5 * float(ref.abs().max())


def test_frozen_bn_and_box_iou_match_torchvision_derived_third_party_code():
    """Two more items of the torchvision-resident arithmetic pinned against code that is DERIVED FROM torchvision and ships in this image
    (Hugging Face transformers): `DetrFrozenBatchNorm2d` ("copy-paste from torchvision.misc.ops with added eps before rsqrt", eps 1e-5) against
    the oracle's `frozen_bn` (FrozenBatchNorm2d eps = 1e-5), and `transformers.loss.loss_for_object_detection.box_iou`
    ("modified from torchvision to also return the union") against the IoU the oracle's NMS uses.  Not a fixture: both run here."""
    pytest.importorskip('transformers')
    from transformers.models.detr.modeling_detr import DetrFrozenBatchNorm2d
    from transformers.loss.loss_for_object_detection import box_iou as hf_box_iou
    g = torch.Generator().manual_seed(11)
    c = 24
    bn = DetrFrozenBatchNorm2d(c)
    bn.weight.copy_(torch.rand(c, generator=g) + 0.5); bn.bias.copy_(torch.randn(c, generator=g))
    bn.running_mean.copy_(torch.randn(c, generator=g)); bn.running_var.copy_(torch.rand(c, generator=g) * 1e-4 + 1e-6)   # (tiny variances: eps matters)
    x = torch.randn(2, c, 5, 7, generator=g)
    sd = {'p.weight': bn.weight, 'p.bias': bn.bias, 'p.running_mean': bn.running_mean, 'p.running_var': bn.running_var}
    assert torch.equal(ogln.frozen_bn(x, sd, 'p'), bn(x))
    assert ogln.FROZEN_BN_EPS == 1e-5
    xy = torch.rand(40, 2, generator=g) * 100
    wh = torch.rand(40, 2, generator=g) * 50 + 1
    a = torch.cat((xy, xy + wh), 1)
    b = a[torch.randperm(40, generator=g)] + torch.randn(40, 4, generator=g)
    b[:, 2:] = torch.maximum(b[:, 2:], b[:, :2] + 0.5)
    ref, _ = hf_box_iou(a, b)
    from cvpce_amd import metrics
    torch.testing.assert_close(metrics.box_iou(a, b), ref, rtol=1e-6, atol=1e-7)
    # the oracle's greedy NMS, judged by that IoU: no two kept boxes overlap by more than the threshold, and every suppressed box does
    # overlap an earlier-kept, higher-scoring one by more than it (strict `>`, areas without +1: torchvision's definition)
    boxes = torch.cat((a, a + torch.randn(40, 4, generator=g) * 3))
    boxes[:, 2:] = torch.maximum(boxes[:, 2:], boxes[:, :2] + 0.5)
    scores = torch.rand(80, generator=g)
    keep = ogln.nms(boxes, scores, 0.5)
    assert bool((scores[keep][:-1] >= scores[keep][1:]).all()) and 0 < len(keep) < 80
    iou, _ = hf_box_iou(boxes, boxes)
    kk = iou[keep][:, keep]
    assert float((kk - torch.eye(len(keep))).max()) <= 0.5
    kept = set(keep.tolist())
    for j in range(80):
        if j not in kept:
            assert any(float(iou[i, j]) > 0.5 and float(scores[i]) >= float(scores[j]) for i in kept), j
