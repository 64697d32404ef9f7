"""Tonioni planogram JSON -> (boxes, labels, neighbour graph)   (reference: cvpce/planogram_adapters.py:17-122).

Host-side reader: pure Python + networkx, like the reference.  The planogram file describes a
shelf as a graph of facings (`graph`: per node the object index `ogg` and the neighbour node in each of the 8 compass
directions, -1 = none) plus the object sizes (`objects`: width, height, img_path).  The reader lays the facings out on a
common grid: every node belongs to one row (walk east from a node without a western neighbour) and one column (walk
along the flipped north/south axis from a node without a northern neighbour); column x-origins and row y-origins are
relaxed in two passes so that neighbours never overlap, and a facing's box hangs from its row's origin.  North and south
are swapped on the way in because detector coordinates grow downwards.
"""
import json

import networkx as nx
import torch


def _flip_ns(direction):
    d = direction.upper()
    if 'N' in d:
        return d.replace('N', 'S')
    if 'S' in d:
        return d.replace('S', 'N')
    return d


def _walk(graph, start, direction, what, source):
    """Nodes reached from `start` by following the unique `direction` edge until there is none."""
    chain, node = [], start
    while True:
        chain.append(node)
        onward = [m for m in graph[node] if graph[node][m]['dir'] == direction]
        if len(onward) > 1:
            raise RuntimeError(f'Multiple nodes {what} from {node}: {onward} (file: {source})')
        if not onward:
            return chain
        node = onward[0]


def _relax(chains, key_of, extent_of, origin):
    """Two passes over the chains (rows or columns): `origin[key]` is the running coordinate at which the cross-axis
    line `key` starts; a chain first aligns itself to a line that already has an origin, then pushes lines forward so
    that consecutive members do not overlap."""
    for chain in chains:
        offset, run = 0, 0
        for node in chain:
            known = origin[key_of(node)]
            if known > float('-inf'):
                offset = known - run
                break
            run += extent_of(node)
        run = offset
        for node in chain:
            k = key_of(node)
            origin[k] = max(run, origin[k])
            run += extent_of(node)
    for chain in chains:
        run = origin[key_of(chain[0])] + extent_of(chain[0])
        for node in chain[1:]:
            k = key_of(node)
            if run > origin[k]:
                origin[k] = run
            else:
                run = origin[k]
            run += extent_of(node)


def read_tonioni_planogram(planogram_path):
    """-> boxes (n,4) float [x1,y1,x2,y2], labels list[str] (img_path without extension), nx.DiGraph with node attribute
    'label' and edge attribute 'dir' (N/S flipped)."""
    with open(planogram_path, 'r') as f:
        planogram = json.load(f)
    nodes = planogram['graph']
    objects = planogram['objects']

    g = nx.DiGraph()
    # (sets, iterated as sets: the reference visits rows / columns in CPython's set order of the node ids)
    row_heads, col_heads = set(), set()
    for i, entry in enumerate(nodes):
        g.add_node(i, ogg=entry['ogg'])
        g.add_edges_from((i, j, {'dir': _flip_ns(k)}) for k, j in entry.items() if k != 'ogg' and j >= 0)
        if entry['w'] == -1:
            row_heads.add(i)
        if entry['n'] == -1:
            col_heads.add(i)

    row_of, col_of = {}, {}
    rows, cols = [], []
    for head in row_heads:
        chain = _walk(g, head, 'E', 'east', planogram_path)
        rows.append(chain)
        for node in chain:
            row_of[node] = head
    for head in col_heads:
        chain = _walk(g, head, 'N', 'north', planogram_path)
        cols.append(chain)
        for node in chain:
            col_of[node] = head

    width = lambda n: objects[nodes[n]['ogg']]['width']
    height = lambda n: objects[nodes[n]['ogg']]['height']
    col_x = {head: float('-inf') for head in col_heads}
    row_y = {head: float('-inf') for head in row_heads}
    _relax(rows, lambda n: col_of[n], width, col_x)
    _relax(cols, lambda n: row_of[n], height, row_y)

    boxes, labels = [], []
    for i in range(len(nodes)):
        x1, y2 = col_x[col_of[i]], row_y[row_of[i]]
        boxes.append((x1, y2 - height(i), x1 + width(i), y2))
        label = objects[nodes[i]['ogg']]['img_path'].split('.')[0]
        labels.append(label)
        del g.nodes[i]['ogg']
        g.nodes[i]['label'] = label
    return torch.tensor(boxes, dtype=torch.float), labels, g
