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
