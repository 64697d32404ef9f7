"""Oracle: cosine distance + nearest neighbours (the K11 stage).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows
/root/reference/cvpce/models/classification.py:87-95.  Pinned by
tests/golden/nearest.pt (reference KAT + seeded cases made by the reference).

The reference's argsort is unstable, so ties are unspecified there; the oracle
(and the HIP path) break ties lowest-index-first.
"""
import torch
import torch.nn.functional as F


def distance(emb1, emb2, dim=1):
    return 1 - F.cosine_similarity(emb1, emb2, dim=dim)


def nearest_neighbors_literal(anchors, queries, k=1):
    """The reference algorithm verbatim in structure: materialised (Q,A,D) gathers + argsort."""
    a_idx = torch.arange(len(anchors))
    q_idx = torch.arange(len(queries))
    q_mesh, a_mesh = torch.meshgrid(q_idx, a_idx, indexing='ij')
    d = distance(anchors[a_mesh], queries[q_mesh], dim=-1)
    return torch.sort(d, dim=-1, stable=True).indices[:, :k]


def cosine_distance_matrix(anchors, queries, eps=1e-8):
    """(Q,A) distances by one GEMM -- same maths as `distance` over the mesh."""
    an = anchors.norm(dim=1).clamp(min=eps)
    qn = queries.norm(dim=1).clamp(min=eps)
    return 1 - (queries @ anchors.t()) / (qn[:, None] * an[None, :])


def nearest_neighbors(anchors, queries, k=1):
    d = cosine_distance_matrix(anchors, queries)
    return torch.sort(d, dim=-1, stable=True).indices[:, :k]
