"""cvpce_amd.planograms against outputs of the reference's own build_graph / build_hypotheses /
large_common_subgraph (tests/golden/planograms.pt), plus property tests of the from-scratch RANSAC homography
(the reference delegates that step to OpenCV, absent here: parity unpinned)."""
import os

import pytest
import torch

from cvpce_amd import planograms


@pytest.fixture(scope='module')
def cases(golden_dir):
    return torch.load(os.path.join(golden_dir, 'planograms.pt'), weights_only=False)


def _edges(g):
    return sorted((int(a), int(b), d['dir'], float(d['weight'])) for a, b, d in g.edges(data=True))


def test_build_graph_matches_reference(cases):
    for c in cases:
        ge = planograms.build_graph(c['expected_boxes'], c['expected_labels'], 0.5)
        ga = planograms.build_graph(c['actual_boxes'], c['actual_labels'], 0.5)
        assert _edges(ge) == c['expected_edges']
        assert _edges(ga) == c['actual_edges']
        assert [ge.nodes[i]['label'] for i in ge] == c['expected_labels']


def test_hypotheses_and_matching_match_reference(cases):
    for c in cases:
        ge = planograms.build_graph(c['expected_boxes'], c['expected_labels'], 0.5)
        ga = planograms.build_graph(c['actual_boxes'], c['actual_labels'], 0.5)
        hyp = planograms.build_hypotheses(ge, ga)
        assert [(float(s), a, b) for s, a, b in hyp] == c['hypotheses']
        assert sorted(planograms.large_common_subgraph(ge, ga)) == c['matching']


def test_graph_properties():
    boxes = torch.tensor([[0., 0, 10, 10], [20, 0, 30, 10], [0, 20, 10, 30], [20, 20, 30, 30]])
    g = planograms.build_graph(boxes, ['a', 'b', 'c', 'd'], 1.0)
    assert g[0][1]['dir'] == 'E' and g[1][0]['dir'] == 'W'
    assert g[0][2]['dir'] == 'N' and g[2][0]['dir'] == 'S'      # +y (image down) is labelled N: acos convention of the reference
    assert g[0][3]['dir'] == 'NE'
    for a, b, d in g.edges(data=True):                           # every edge has its mirror with the opposite direction
        assert g[b][a]['dir'] == planograms._OPPOSITE[d['dir']] and g[b][a]['weight'] == d['weight']


def test_find_homography_recovers_known_transform():
    g = torch.Generator().manual_seed(3)
    h_true = torch.tensor([[1.7, 0.05, 30.0], [-0.03, 1.6, 12.0], [1e-5, 2e-5, 1.0]], dtype=torch.float64)
    src = torch.rand(60, 2, generator=g, dtype=torch.float64) * 800
    p = torch.cat((src, torch.ones(60, 1, dtype=torch.float64)), 1) @ h_true.T
    dst = p[:, :2] / p[:, 2:]
    dst_noisy = dst + torch.randn(60, 2, generator=g, dtype=torch.float64) * 0.3
    dst_noisy[:12] += torch.rand(12, 2, generator=g, dtype=torch.float64) * 300 + 50          # 20 % gross outliers
    h, inl = planograms.find_homography(src, dst_noisy, reproj_threshold=3.0)
    assert h is not None and abs(float(h[2, 2]) - 1) < 1e-9
    assert not inl[:12].any() and inl[12:].float().mean() > 0.9
    q = torch.cat((src, torch.ones(60, 1, dtype=torch.float64)), 1) @ h.T
    assert ((q[:, :2] / q[:, 2:]) - dst)[12:].norm(dim=1).max() < 1.5
    assert planograms.find_homography(src[:3], dst[:3])[0] is None                                # < 4 points
    assert planograms.find_homography(torch.zeros(8, 2), torch.zeros(8, 2))[0] is None             # degenerate


def test_finalize_and_comparator_verdict(cases):
    from cvpce_amd import production
    c = cases[1]          # 4x6 planogram, one product missing in the detections, detections scaled by 1.7 and shifted
    expected = {'boxes': c['expected_boxes'], 'labels': c['expected_labels']}
    actual = {'boxes': c['actual_boxes'], 'labels': c['actual_labels']}
    comp = production.PlanogramComparator()
    verdict = float(comp.compare(expected, actual))
    assert abs(verdict - 23 / 24) < 1e-6                    # exactly the one dropped product is reported missing
    found, missing_idx, missing_pos, missing_lbl = planograms.finalize_via_ransac(
        [tuple(m) for m in c['matching']], c['expected_boxes'], c['actual_boxes'], c['expected_labels'], c['actual_labels'])
    assert int((~found).sum()) == 1 and missing_lbl == [c['expected_labels'][int(missing_idx[0])]]
    # the projected position of the missing product lands where the product would be (1.7x + offset)
    want = c['expected_boxes'][int(missing_idx[0])] * 1.7 + torch.tensor([30.0, 12.0, 30.0, 12.0])
    assert (missing_pos[0] - want).abs().max() < 8.0
    assert comp.compare(expected, {'boxes': torch.zeros(0, 4), 'labels': []}) == 0
    assert comp.compare({'boxes': torch.zeros(0, 4), 'labels': []}, {'boxes': torch.zeros(0, 4), 'labels': []}) == 1
    wrong = dict(actual, labels=['nope'] * len(actual['labels']))
    assert comp.compare(expected, wrong) == 0               # no common subgraph
