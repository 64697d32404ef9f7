"""Planogram graph logic -> compliance verdict (host logic, CPU tensors + networkx) -- counterpart of
/root/reference/cvpce/planograms.py:12-132,171-245.

  build_graph             8-direction nearest-neighbour DiGraph over box centres (planograms.py:30-71)
  build_hypotheses /
  large_common_subgraph   hypothesis-seeded breadth-first common subgraph (planograms.py:73-132)
  finalize_via_ransac     homography planogram -> image from 3 points per matched box pair, projection of every
                          expected box, per-label IoU matching (planograms.py:171-245)

`build_graph` / `build_hypotheses` / `large_common_subgraph` are pinned by golden outputs of the reference's own
functions (tests/golden/planograms.pt): graphs are mutated in the same order as the reference mutates them, because the
adjacency order of the DiGraph decides the BFS order and therefore the matching.
The reference obtains the homography from `cv2.findHomography(RANSAC)` (OpenCV is not installable here): PARITY
UNPINNED for that step -- `find_homography` below is a from-scratch seeded RANSAC + normalised DLT refit.
"""
from math import pi

import networkx as nx
import torch

from . import metrics

CARDINALS = ['E', 'NE', 'N', 'NW', 'W', 'SW', 'S', 'SE']
_OPPOSITE = {d: CARDINALS[(i + 4) % 8] for i, d in enumerate(CARDINALS)}


def _direction_matrix(centres, dists):
    """(n,n) int sector index of the direction i -> j (y axis as given, i.e. image coordinates), -1 where undefined."""
    n = len(centres)
    vec = (centres[None, :, :] - centres[:, None, :]) / dists.reshape(n, n, 1)
    ang = torch.acos(vec[:, :, 0].clamp(-1, 1))
    below = vec[:, :, 1] < 0
    ang[below] = 2 * pi - ang[below]
    sector = torch.full((n, n), -1, dtype=torch.long)
    sector[(ang > 15 * pi / 8) | (ang <= pi / 8)] = 0
    for i in range(7):
        sector[(ang > (1 + 2 * i) * pi / 8) & (ang <= (1 + 2 * (i + 1)) * pi / 8)] = i + 1
    return sector


def build_graph(boxes, labels, thresh_size=0.5):
    span = (torch.amax(boxes[:, 2]) - torch.amin(boxes[:, 0]) + torch.amax(boxes[:, 3]) - torch.amin(boxes[:, 1])) / 2
    thresh = thresh_size * span
    centres = torch.stack(((boxes[:, 0] + boxes[:, 2]) / 2, (boxes[:, 1] + boxes[:, 3]) / 2), dim=1)
    dists = torch.cdist(centres[None], centres[None])[0]
    sector = _direction_matrix(centres, dists)

    g = nx.DiGraph()
    g.add_nodes_from((i, {'label': labels[i]}) for i in range(len(centres)))
    by_dist, order = dists.sort(dim=1)
    for i in range(len(centres)):
        wanted = set(CARDINALS) - {g[i][nb]['dir'] for nb in g[i]}
        for d, j in zip(by_dist[i], order[i].tolist()):
            if d > thresh or not wanted:
                break
            if j == i or sector[i, j] < 0:
                continue
            direction = CARDINALS[int(sector[i, j])]
            if direction not in wanted:
                continue
            back = _OPPOSITE[direction]
            rival = next((k for k in g[j] if g[j][k]['dir'] == back), None)   # j's current neighbour on i's side
            if rival is not None:
                if g[j][rival]['weight'] <= d:
                    continue                       # j already has a closer neighbour in that direction
                g.remove_edge(j, rival)
                g.remove_edge(rival, j)
            g.add_edge(i, j, dir=direction, weight=d)
            g.add_edge(j, i, dir=back, weight=d)
            wanted.remove(direction)
    return g


def _neighbour_labels(g, n, edge_label):
    return {g[n][nb][edge_label]: g.nodes[nb] for nb in g[n]}


def build_hypotheses(g1, g2, edge_label='dir'):
    """Seed pairs (equal node attributes), best first: (-fraction of the 8 directions whose neighbours agree, n1, n2)."""
    out = []
    for n1 in g1:
        for n2 in g2:
            if g1.nodes[n1] != g2.nodes[n2]:
                continue
            a, b = _neighbour_labels(g1, n1, edge_label), _neighbour_labels(g2, n2, edge_label)
            agree = sum(a[k] == b[k] for k in a if k in b)
            out.append((-(agree / len(CARDINALS)), n1, n2))
    return sorted(out)


def _compatible_steps(g1, g2, n1, n2, edge_label):
    return [(e1, e2) for e1 in g1[n1] for e2 in g2[n2]
            if g1[n1][e1][edge_label] == g2[n2][e2][edge_label] and g1.nodes[e1] == g2.nodes[e2]]


def large_common_subgraph(g1, g2, edge_label='dir', min_score=-0.2, stop_at_fraction=1 / 2):
    best = set()
    enough = min(len(g1), len(g2)) * stop_at_fraction
    for score, s1, s2 in build_hypotheses(g1, g2, edge_label):
        if score > min_score and best:
            return best
        grown, used1, used2 = {(s1, s2)}, {s1}, {s2}
        frontier = _compatible_steps(g1, g2, s1, s2, edge_label)
        while frontier:
            n1, n2 = frontier.pop(0)
            if n1 in used1 or n2 in used2:
                continue
            frontier += _compatible_steps(g1, g2, n1, n2, edge_label)
            grown.add((n1, n2)); used1.add(n1); used2.add(n2)
        if len(grown) > enough:
            return grown
        if len(grown) > len(best):
            best = grown
    return best


# ---------------------------------------------------------------------------
# homography (replaces cv2.findHomography(points1, points2, RANSAC, reproj_threshold))
# ---------------------------------------------------------------------------
def _dlt(src, dst):
    """Normalised DLT, (n,2) double -> 3x3 or None."""
    def norm(p):
        c = p.mean(0)
        s = (p - c).norm(dim=1).mean()
        if s < 1e-12:
            return None, None
        k = (2 ** 0.5) / s
        t = torch.tensor([[k, 0, -k * c[0]], [0, k, -k * c[1]], [0, 0, 1]], dtype=torch.float64)
        return (p - c) * k, t
    a, ta = norm(src)
    b, tb = norm(dst)
    if a is None or b is None:
        return None
    rows = []
    for (x, y), (u, v) in zip(a.tolist(), b.tolist()):
        rows.append([-x, -y, -1, 0, 0, 0, u * x, u * y, u])
        rows.append([0, 0, 0, -x, -y, -1, v * x, v * y, v])
    m = torch.tensor(rows, dtype=torch.float64)
    try:
        _, s, vh = torch.linalg.svd(m)
    except RuntimeError:
        return None
    h = vh[-1].reshape(3, 3)
    h = torch.linalg.inv(tb) @ h @ ta
    if abs(float(h[2, 2])) < 1e-12:
        return None
    return h / h[2, 2]


def _reproj_err(h, src, dst):
    p = torch.cat((src, torch.ones(len(src), 1, dtype=src.dtype)), dim=1) @ h.T
    return (p[:, :2] / p[:, 2:3] - dst).norm(dim=1)


def find_homography(points1, points2, reproj_threshold=3.0, max_iters=2000, confidence=0.995, seed=0):
    """-> (3x3 float64 H with H[2,2] = 1 mapping points1 -> points2, bool inlier mask) or (None, mask of False)."""
    src, dst = points1.to(torch.float64), points2.to(torch.float64)
    n = len(src)
    none = (None, torch.zeros(n, dtype=torch.bool))
    if n < 4:
        return none
    gen = torch.Generator().manual_seed(seed)
    best_mask, best_count, iters, it = None, 0, max_iters, 0
    while it < iters:
        it += 1
        pick = torch.randperm(n, generator=gen)[:4]
        h = _dlt(src[pick], dst[pick])
        if h is None:
            continue
        err = _reproj_err(h, src, dst)
        mask = err < reproj_threshold
        count = int(mask.sum())
        if count > best_count:
            best_mask, best_count = mask, count
            w = count / n
            if w >= 1.0:
                break
            denom = torch.log1p(torch.tensor(-(w ** 4)))
            iters = min(max_iters, int(torch.ceil(torch.log(torch.tensor(1 - confidence)) / denom).item()) + 1)
    if best_mask is None or best_count < 4:
        return none
    h = _dlt(src[best_mask], dst[best_mask])
    if h is None:
        return none
    final = _reproj_err(h, src, dst) < reproj_threshold
    return h, final


# ---------------------------------------------------------------------------
def _three_points(boxes):
    centres = torch.stack(((boxes[:, 0] + boxes[:, 2]) / 2, (boxes[:, 1] + boxes[:, 3]) / 2), dim=1)
    return torch.cat((boxes[:, :2], boxes[:, 2:], centres))


def labels_to_tensors(l1, *ln):
    """cvpce/utils.py:290-294"""
    key = list(set(l1).union(*ln))
    lut = {l: i for i, l in enumerate(key)}
    return (*(torch.tensor([lut[l] for l in ls], dtype=torch.long) for ls in (l1, *ln)), key)


def finalize_via_ransac(solution, b1, b2, l1, l2, reproj_threshold=10, iou_threshold=0.5,
                        return_matched_actual=False, report_accuracy=False, return_expected_positions=False):
    nodes1, nodes2 = (list(x) for x in zip(*solution))
    boxes1, boxes2 = b1[nodes1], b2[nodes2]
    p1, p2 = _three_points(boxes1), _three_points(boxes2)
    if len(solution) < 2:   # too few correspondences: add the other two corners (planograms.py:194-196)
        p1 = torch.cat((p1, boxes1[:, (2, 1)], boxes1[:, (0, 3)]))
        p2 = torch.cat((p2, boxes2[:, (2, 1)], boxes2[:, (0, 3)]))
    h, inliers = find_homography(p1, p2, reproj_threshold)
    if report_accuracy:
        print(f'Homography accuracy: {inliers.sum() / len(inliers)}')
    if h is None:
        return (None,) * 5 if return_matched_actual else (None,) * 4
    h = h.to(torch.float)

    def project(x, y):
        r = h @ torch.tensor([x, y, 1], dtype=torch.float)
        return r[:2] / r[2]

    expected_positions = torch.stack([torch.cat((project(x1, y1), project(x2, y2))) for x1, y1, x2, y2 in b1.tolist()])
    t1, t2, key = labels_to_tensors(l1, l2)
    matched_expected = torch.zeros(len(expected_positions), dtype=torch.bool)
    matched_actual = torch.zeros(len(b2), dtype=torch.bool)
    for lbl in range(len(key)):
        exp_idx, act_idx = torch.where(t1 == lbl)[0], torch.where(t2 == lbl)[0]
        if not len(exp_idx) or not len(act_idx):
            continue
        taken = torch.zeros(len(act_idx), dtype=torch.bool)
        ious, order = torch.sort(metrics.box_iou(expected_positions[exp_idx], b2[act_idx]), dim=1, descending=True)
        for i in range(len(exp_idx)):
            for iou, idx in zip(ious[i].tolist(), order[i].tolist()):
                if iou < iou_threshold:
                    break
                if taken[idx]:
                    continue
                # NB (reference behaviour, planograms.py:236-240): no break -- an expected box claims EVERY free
                # detection of its label that overlaps it by >= the threshold
                taken[idx] = True
                matched_expected[exp_idx[i]] = True
                matched_actual[act_idx[idx]] = True
    missing_expected = torch.where(~matched_expected)[0]
    missing_positions = expected_positions[missing_expected]
    missing_labels = [key[i] for i in t1[missing_expected]]
    if return_expected_positions and return_matched_actual:
        return matched_expected, matched_actual, expected_positions, missing_expected, missing_positions, missing_labels
    if return_expected_positions:
        return matched_expected, expected_positions, missing_expected, missing_positions, missing_labels
    if return_matched_actual:
        return matched_expected, matched_actual, missing_expected, missing_positions, missing_labels
    return matched_expected, missing_expected, missing_positions, missing_labels
