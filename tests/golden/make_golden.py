#!/usr/bin/env python3
"""Generate golden input/output vectors from the reference's own pure-torch code.

DEV-ONLY TOOL: runs in the build container where /root/reference is mounted.
It never travels to the GPU box and nothing in tests/, bench.py or the package
imports it.  The committed products are the small `*.pt` fixtures next to this
file (data only: inputs, weights, expected outputs).

torchvision / cv2 / ray / pix2pix are absent here, so every
hot-path module of the reference fails on import.  The members we capture are
pure torch; we therefore place inert stand-ins for the absent third-party
modules in sys.modules *for the import only* (base classes -> nn.Module).  No
stand-in is ever *executed* to make a golden value, with one exception that is
pinned by the reference's own KATs: `box_iou` (test/metrics_test.py:23-59).

Captured:
  gaussian_head.pt   GaussianLayer + GaussianSubnet (cvpce/models/proposals.py:51-107)
                     eval() with non-trivial BN running stats, tanh in {False, True}
  nearest.pt         distance / nearest_neighbors (cvpce/models/classification.py:87-95)
                     incl. the reference KAT (test/models/classification_test.py:8-25)
  metrics.pt         metrics KAT inputs + outputs of the reference implementation
                     (cvpce/metrics.py:11-138, test/metrics_test.py:5-21,116-128)
  members.pt         (round 3; `python tests/golden/make_golden.py members` writes only this file)
                     MACResNet.forward over a hand-built resnet-like source (cvpce/models/classification.py:53-85): state-dict
                     key nesting, MAC / concat / L2 norm; Classifier.classify with a toy encoder (cvpce/production.py:57-74):
                     batching, k, return_embedding, empty input, label lookup; PlanogramComparator.compare early-outs
                     (production.py:79-90); mean_average_metrics (cvpce/detection_eval.py:51-55)
"""
import os
import sys
import types
from math import sqrt

import torch
from torch import nn

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))


def _stub_modules():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Base(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    def box_iou(a, b):  # pinned by test/metrics_test.py:23-59 (checked below)
        area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
        area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
        lt = torch.max(a[:, None, :2], b[:, :2])
        rb = torch.min(a[:, None, 2:], b[:, 2:])
        wh = (rb - lt).clamp(min=0)
        inter = wh[:, :, 0] * wh[:, :, 1]
        return inter / (area_a[:, None] + area_b - inter)

    tv = mod('torchvision')
    tv.models = mod('torchvision.models', utils=mod('torchvision.models.utils'),
                    vgg=mod('torchvision.models.vgg'), resnet=mod('torchvision.models.resnet'))
    det = mod('torchvision.models.detection', RetinaNet=_Base)
    mod('torchvision.models.detection.backbone_utils', BackboneWithFPN=_Base)
    tv.ops = mod('torchvision.ops', box_iou=box_iou)
    mod('torchvision.ops.feature_pyramid_network', ExtraFPNBlock=_Base, LastLevelP6P7=_Base)
    mod('torchvision.ops.misc', FrozenBatchNorm2d=_Base)
    tv.transforms = mod('torchvision.transforms', functional=mod('torchvision.transforms.functional'))
    tv.utils = mod('torchvision.utils')
    tv.models.detection = det
    for name in ('cv2', 'squarify', 'skimage', 'skimage.draw', 'ray', 'ray.tune'):
        mod(name)
    mod('skimage.segmentation', flood=None)
    mod('skimage.filters', sobel=None)
    # un-vendored pix2pix submodule (cvpce/models/classification.py:8)
    mod('cvpce.models.pix2pix')
    mod('cvpce.models.pix2pix.models', networks=mod('cvpce.models.pix2pix.models.networks'))


class TinyBottleneck(nn.Module):
    """A torchvision-Bottleneck-shaped block (v1.5: stride on the 3x3; attribute names conv1..3 / bn1..3 / downsample) for the
    hand-built `source_resnet` of the MACResNet fixture.  Written here -- torchvision is absent -- and executed by the REFERENCE's
    MACResNet.forward through nn.Sequential; what the fixture pins is the reference's own code: the Sequential nesting of the
    blocks (state-dict keys), the per-block amax, the concatenation order and the L2 normalisation."""

    def __init__(self, inplanes, planes, stride, downsample):
        super().__init__()
        self.conv1, self.bn1 = nn.Conv2d(inplanes, planes, 1, bias=False), nn.BatchNorm2d(planes)
        self.conv2, self.bn2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False), nn.BatchNorm2d(planes)
        self.conv3, self.bn3 = nn.Conv2d(planes, planes * 4, 1, bias=False), nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride, bias=False), nn.BatchNorm2d(planes * 4)) if downsample else None

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        return self.relu(self.bn3(self.conv3(y)) + idt)


class TinyResNet(nn.Module):
    def __init__(self, stem=16, planes=(8, 16, 32, 64), layers=(2, 2, 2, 1)):
        super().__init__()
        self.conv1, self.bn1 = nn.Conv2d(3, stem, 7, 2, 3, bias=False), nn.BatchNorm2d(stem)
        self.relu, self.maxpool = nn.ReLU(inplace=True), nn.MaxPool2d(3, 2, 1)
        inpl = stem
        for li, (pl, nb) in enumerate(zip(planes, layers)):
            seq = []
            for bi in range(nb):
                seq.append(TinyBottleneck(inpl, pl, 2 if (bi == 0 and li > 0) else 1, bi == 0))
                inpl = pl * 4
            setattr(self, f'layer{li + 1}', nn.Sequential(*seq))


class ToyEncoder(nn.Module):
    """Encoder stand-in for the Classifier fixture: 4x4 average pool -> linear -> unit norm.  Any nn.Module with an
    `embedding_size` serves the reference's Classifier (production.py:57-74); what is pinned is Classifier.classify."""
    embedding_size = 64

    def __init__(self):
        super().__init__()
        self.proj = nn.Linear(48, 64, bias=False)

    def forward(self, x):
        v = self.proj(torch.nn.functional.adaptive_avg_pool2d(x, 4).flatten(1))
        return v / v.norm(dim=1, keepdim=True).clamp(min=1e-8)


def make_members():
    """members.pt -- see the module docstring."""
    from cvpce.models import classification as ref_cls
    from cvpce import metrics as ref_metrics
    out = {}
    # ---- (a) MACResNet.forward ------------------------------------------------------------------
    torch.manual_seed(31)
    src = TinyResNet()
    for m in src.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.6, 1.6); m.weight.data.uniform_(0.7, 1.3); m.bias.data.normal_(0, 0.1)
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
    cases = []
    for dl in ([2, 3], [1, 2, 4], [3]):
        model = ref_cls.MACResNet(src, dl).eval()
        x = torch.rand(3, 3, 64, 64, generator=torch.Generator().manual_seed(len(dl))) * 2 - 1
        with torch.no_grad():
            y = model(x)
            zero = model(torch.zeros(1, 3, 64, 64))          # eps path is not reached (biases), but pins a second input
        # the model's state dict holds the SAME tensors as the source under the reference's Sequential nesting: store the key map
        by_ptr = {v.data_ptr(): k for k, v in src.state_dict().items()}
        cases.append({'descriptor_layers': dl, 'state_key_to_source_key': {k: by_ptr[v.data_ptr()] for k, v in model.state_dict().items()},
                      'input': x, 'output': y, 'output_zero_input': zero, 'embedding_size_attr': model.embedding_size})
    out['macresnet'] = {'stem': 16, 'planes': (8, 16, 32, 64), 'layers': (2, 2, 2, 1), 'cases': cases,
                        'source_state': {k: v.clone() for k, v in src.state_dict().items()}}
    # ---- (b) Classifier.classify ------------------------------------------------------------------
    from cvpce import production as ref_prod
    torch.manual_seed(32)
    enc = ToyEncoder().eval()
    g = torch.Generator().manual_seed(33)
    gal_pat = torch.rand(40, 3, 4, 4, generator=g)            # gallery "products" as 4x4 colour patterns in [0,1]
    up = lambda p: torch.nn.functional.interpolate(p, size=(256, 256), mode='nearest')
    with torch.no_grad():
        gallery = enc(up(gal_pat) * 2 - 1)                     # gallery tensors live in [-1,1] (datautils.py:446)
    annotations = [f'Food/Cat{i % 7}/{100 + i}.jpg' for i in range(40)]
    pick = torch.randint(0, 40, (23,), generator=g)
    q_pat = (gal_pat[pick] + 0.02 * torch.randn(23, 3, 4, 4, generator=g)).clamp(0, 1)
    cls_cases = []
    for bs, k, n in ((8, 1, 23), (5, 3, 23), (32, 1, 7), (4, 2, 0), (1, 5, 3)):
        c = ref_prod.Classifier.__new__(ref_prod.Classifier)
        c.batch_size, c.num_workers, c.device, c.emb_device, c.k = bs, 0, torch.device('cpu'), torch.device('cpu'), k
        c.encoder, c.embedding, c.annotations = enc, gallery, annotations
        imgs = up(q_pat[:n]) if n else torch.empty(0, 3, 256, 256)
        with torch.no_grad():
            res = c.classify(imgs)
            res2, emb = c.classify(imgs, return_embedding=True)
        assert res == res2
        cls_cases.append({'batch_size': bs, 'k': k, 'n': n, 'labels': res, 'embedding': emb})
    out['classifier'] = {'encoder_state': {k_: v.clone() for k_, v in enc.state_dict().items()}, 'gallery_patterns': gal_pat,
                         'gallery_embedding': gallery, 'annotations': annotations, 'query_patterns': q_pat, 'picked': pick,
                         'recipe': 'image = F.interpolate(pattern, size=(256, 256), mode="nearest"); gallery images are image * 2 - 1',
                         'cases': cls_cases}
    # ---- (c) PlanogramComparator.compare early-outs (production.py:79-90) --------------------------------------------------
    cmp_ = ref_prod.PlanogramComparator()
    eb = torch.tensor([[0., 0., 10., 20.], [12., 0., 22., 20.], [24., 0., 34., 20.]])
    el = ['a', 'b', 'c']
    none = {'boxes': torch.empty(0, 4), 'labels': []}
    early = [
        {'name': 'nothing_detected', 'expected': {'boxes': eb, 'labels': el}, 'actual': none},
        {'name': 'nothing_expected_nothing_detected', 'expected': none, 'actual': none},
        {'name': 'no_common_label', 'expected': {'boxes': eb, 'labels': el}, 'actual': {'boxes': eb + 3.0, 'labels': ['x', 'y', 'z']}},
        {'name': 'nothing_detected_with_image', 'expected': {'boxes': eb, 'labels': el}, 'actual': none, 'image_hw': (40, 50)},
    ]
    for e in early:
        img = torch.zeros(3, *e['image_hw']) if 'image_hw' in e else None
        e['result'] = float(cmp_.compare(e['expected'], e['actual'], img))
    out['comparator_early_outs'] = early
    # ---- (d) mean_average_metrics (detection_eval.py:51-55) ---------------------------------------------------------------
    from cvpce import detection_eval as ref_de
    g = torch.Generator().manual_seed(34)
    per_class, inputs = {}, {}
    for c in range(4):
        tg, pr, cf = [], [], []
        for _ in range(3):
            nt = int(torch.randint(1, 8, (1,), generator=g))
            xy = torch.rand(nt, 2, generator=g) * 50
            t = torch.cat([xy, xy + torch.rand(nt, 2, generator=g) * 20 + 5], 1)
            npred = int(torch.randint(1, 12, (1,), generator=g))
            p_ = t[torch.randint(0, nt, (npred,), generator=g)] + torch.randn(npred, 4, generator=g) * 2
            p_[:, 2:] = torch.max(p_[:, 2:], p_[:, :2] + 1)
            tg.append(t); pr.append(p_); cf.append(torch.rand(npred, generator=g))
        inputs[c] = {'targets': tg, 'predictions': pr, 'confidences': cf}
        r = ref_metrics.calculate_metrics(tg, pr, cf, (0.5, 0.75))
        per_class[c] = {t: {k_: v for k_, v in d.items() if k_ != 'raw'} for t, d in r.items()}
    mam = ref_de.mean_average_metrics(per_class, (0.5, 0.75))
    out['mean_average_metrics'] = {'inputs': inputs, 'per_class': per_class,
                                   'result': {t: {k_: float(v) for k_, v in d.items()} for t, d in mam.items()}}
    torch.save(out, os.path.join(HERE, 'members.pt'))
    print('members.pt written:', os.path.getsize(os.path.join(HERE, 'members.pt')), 'bytes')


def main():
    _stub_modules()
    sys.path.insert(0, REF)
    import matplotlib
    matplotlib.use('Agg')
    if sys.argv[1:] == ['members']:
        import warnings
        warnings.simplefilter('ignore')
        sys.modules['cv2'].findHomography = None
        sys.modules['cv2'].RANSAC = 8
        return make_members()
    from cvpce.models import proposals as ref_prop
    from cvpce.models import classification as ref_cls
    from cvpce import metrics as ref_metrics

    # ---- 1. Gaussian head ------------------------------------------------
    out = {}
    for tanh in (False, True):
        torch.manual_seed(11 + int(tanh))
        c_ch, p_ch = 16, 16
        layer = ref_prop.GaussianLayer(c_ch, p_ch)
        subnet = ref_prop.GaussianSubnet(p_ch // 4, tanh)
        for m in list(layer.modules()) + list(subnet.modules()):
            if isinstance(m, nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.5)
                m.running_var.uniform_(0.5, 2.0)
                m.weight.data.uniform_(0.5, 1.5)
                m.bias.data.normal_(0, 0.2)
            if isinstance(m, nn.Conv2d):
                m.bias.data.normal_(0, 0.1)
        layer.eval(); subnet.eval()
        c2 = torch.randn(2, c_ch, 12, 20)
        p3 = torch.randn(2, p_ch, 6, 10)
        with torch.no_grad():
            feat = layer(c2, p3)
            g = subnet(feat)
        out[f'tanh_{tanh}'] = {
            'layer_state': {k: v.clone() for k, v in layer.state_dict().items()},
            'subnet_state': {k: v.clone() for k, v in subnet.state_dict().items()},
            'c2': c2, 'p3': p3, 'features': feat, 'gaussians': g,
        }
    torch.save(out, os.path.join(HERE, 'gaussian_head.pt'))

    # ---- 2. distance / nearest_neighbors ----------------------------------
    anchors = torch.tensor([
        [1, 0, 0],
        [1 / sqrt(3), 1 / sqrt(3), 1 / sqrt(3)],
        [-1 / sqrt(3), -1 / sqrt(3), -1 / sqrt(3)],
        [-1, 0, 0],
        [1 / sqrt(2), 0, 1 / sqrt(2)],
        [-1 / sqrt(2), 0, -1 / sqrt(2)],
    ], dtype=torch.float)
    queries = torch.tensor([
        [1 / sqrt(1.01), 0.1 / sqrt(1.01), 0],
        [0.9 / sqrt(2.02), 0, 1.1 / sqrt(2.02)],
        [-1, 0, 0],
        [1, 0, 0],
        [1 / sqrt(3), 1 / sqrt(3), 1 / sqrt(3)],
        [-1.1 / sqrt(2.02), 0, -0.9 / sqrt(2.02)],
        [-1, 0, 0],
    ])
    kat_expected = torch.tensor([0, 4, 3, 0, 1, 5, 3])
    import warnings
    warnings.simplefilter('ignore')
    got = ref_cls.nearest_neighbors(anchors, queries)[:, 0]
    assert kat_expected.equal(got), 'reference KAT failed under stubs'
    cases = []
    for seed, (q, a, d, k) in enumerate([(7, 50, 16, 1), (16, 120, 64, 4), (33, 200, 256, 4), (24, 150, 1024, 1), (20, 97, 1024, 8)]):
        g = torch.Generator().manual_seed(100 + seed)
        A = torch.nn.functional.normalize(torch.randn(a, d, generator=g), dim=1)
        Q = torch.nn.functional.normalize(torch.randn(q, d, generator=g), dim=1)
        dist = ref_cls.distance(A[None, :, :].expand(q, a, d), Q[:, None, :].expand(q, a, d), dim=-1)
        idx = ref_cls.nearest_neighbors(A, Q, k)
        # keep only tie-free cases (gap between consecutive sorted distances of the first k+1)
        srt = dist.sort(dim=-1).values[:, :k + 1]
        gap = (srt[:, 1:] - srt[:, :-1]).min().item()
        cases.append({'anchors': A, 'queries': Q, 'k': k, 'indices': idx, 'distances': dist, 'min_gap': gap})
    torch.save({'kat': {'anchors': anchors, 'queries': queries, 'expected': kat_expected}, 'cases': cases},
               os.path.join(HERE, 'nearest.pt'))

    # ---- 3. metrics ------------------------------------------------------
    T = [
        torch.tensor([[0, 0, 1, 1], [1, 0, 2, 1], [1, 1, 2, 2]], dtype=torch.float),
        torch.tensor([[1, 1, 2, 2], [3, 1, 4, 2], [5, 1, 6, 2], [7, 1, 8, 2]], dtype=torch.float),
        torch.tensor([[0, 0, 5, 5], [5, 5, 10, 10]], dtype=torch.float),
    ]
    P = [
        torch.tensor([[0, 0, .9, .9], [1.1, 0.1, 1.9, 0.9], [0, 0, 1, 1], [0.9, 0.9, 2.1, 2.1], [3, 3, 4, 4]], dtype=torch.float),
        torch.tensor([[1, 0, 2, 1], [1, 1, 2, 2], [5, 1, 6, 2], [7, 1.1, 8, 1.9], [9, 9, 10, 10]], dtype=torch.float),
        torch.tensor([[0, 0, 1, 1], [1, 1, 3, 3], [0.5, 0.5, 4.5, 4.5], [0, 0, 6, 6], [6, 6, 9, 9]], dtype=torch.float),
    ]
    C = [
        torch.tensor([1, 0.8, 0.6, 0.4, 0.2], dtype=torch.float),
        torch.tensor([0.9, 0.8, 0.7, 0.65, 0.5], dtype=torch.float),
        torch.tensor([0.85, 0.6, 0.4, 0.2, 0.1], dtype=torch.float),
    ]
    # validate the box_iou stand-in against the reference's own expected values
    ious, idx = ref_metrics.iou_matrices(T[2], P[2])
    exp = torch.tensor([[0.04, 0], [0.16, 0], [0.64, 0], [(5 * 5) / (6 * 6), 1 / (5 * 5 + 6 * 6 - 1)], [0.36, 0]])
    assert exp.allclose(ious)
    res = ref_metrics.calculate_metrics(T, P, C)
    g = torch.Generator().manual_seed(5)
    rnd = []
    for n_img in (3, 5):
        tg, pr, cf = [], [], []
        for _ in range(n_img):
            nt = int(torch.randint(1, 12, (1,), generator=g))
            xy = torch.rand(nt, 2, generator=g) * 50
            wh = torch.rand(nt, 2, generator=g) * 20 + 5
            t = torch.cat([xy, xy + wh], 1)
            npred = int(torch.randint(1, 20, (1,), generator=g))
            pick = torch.randint(0, nt, (npred,), generator=g)
            p = t[pick] + torch.randn(npred, 4, generator=g) * 3
            p[:, 2:] = torch.max(p[:, 2:], p[:, :2] + 1)
            tg.append(t); pr.append(p); cf.append(torch.rand(npred, generator=g))
        r = ref_metrics.calculate_metrics(tg, pr, cf, iou_thresholds=(0.5, 0.75))
        rnd.append({'targets': tg, 'predictions': pr, 'confidences': cf,
                    'result': {t: {k: v for k, v in d.items() if k != 'raw'} | {'raw': d['raw']} for t, d in r.items()}})
    # dense, mutually overlapping targets: one prediction reaches SEVERAL still-unused targets above the threshold, and
    # the reference's inner loop (cvpce/metrics.py:22-26, no break after a match) marks every one of them as used
    two = (torch.tensor([[0, 0, 10, 10], [0, 0, 10, 9]], dtype=torch.float),
           torch.tensor([[0, 0, 10, 10], [0, 0, 10, 9.5]], dtype=torch.float))
    tp2, fp2 = ref_metrics.check_matches(*ref_metrics.iou_matrices(*two))
    assert tp2.tolist() == [1, 0]
    dense = []
    for n_img in (2, 4):
        tg, pr, cf = [], [], []
        for _ in range(n_img):
            nt = int(torch.randint(6, 16, (1,), generator=g))
            xy = torch.rand(nt, 2, generator=g) * 12
            wh = torch.rand(nt, 2, generator=g) * 6 + 20
            t = torch.cat([xy, xy + wh], 1)
            npred = int(torch.randint(4, 24, (1,), generator=g))
            pick = torch.randint(0, nt, (npred,), generator=g)
            p = t[pick] + torch.randn(npred, 4, generator=g) * 1.5
            p[:, 2:] = torch.max(p[:, 2:], p[:, :2] + 1)
            tg.append(t); pr.append(p); cf.append(torch.rand(npred, generator=g))
        r = ref_metrics.calculate_metrics(tg, pr, cf, iou_thresholds=(0.5, 0.75))
        dense.append({'targets': tg, 'predictions': pr, 'confidences': cf,
                      'result': {t: {k: v for k, v in d.items() if k != 'raw'} | {'raw': d['raw']} for t, d in r.items()}})
    torch.save({'targets': T, 'predictions': P, 'confidences': C,
                'kat_result': {k: v for k, v in res[0.5].items()}, 'random': rnd,
                'two_targets': {'targets': two[0], 'predictions': two[1], 'tp': tp2, 'fp': fp2}, 'dense': dense},
               os.path.join(HERE, 'metrics.pt'))
    # ---- 4. planogram graph logic (cvpce/planograms.py:12-132; needs networkx only) --------------
    sys.modules['cv2'].findHomography = None
    sys.modules['cv2'].RANSAC = 8
    from cvpce import planograms as ref_plano

    def grid(rows, cols, seed, jitter=4.0, drop=(), w=60.0, h=90.0, gap=12.0):
        g = torch.Generator().manual_seed(seed)
        boxes, labels = [], []
        for r in range(rows):
            for c in range(cols):
                if (r, c) in drop:
                    continue
                x = c * (w + gap) + float(torch.randn(1, generator=g)) * jitter
                y = r * (h + gap) + float(torch.randn(1, generator=g)) * jitter
                boxes.append([x, y, x + w, y + h])
                labels.append(f'sku{(r * 3 + c * 5) % 7}')
        return torch.tensor(boxes), labels

    def edges(gr):
        return sorted((int(a), int(b), d['dir'], float(d['weight'])) for a, b, d in gr.edges(data=True))

    plano = []
    for seed, (rows, cols, drop) in enumerate([(3, 4, ()), (4, 6, ((1, 2),)), (2, 5, ((0, 0), (1, 4))), (5, 5, ())]):
        eb, el = grid(rows, cols, 40 + seed, jitter=0.0)                 # planogram: perfect grid
        ab, al = grid(rows, cols, 50 + seed, jitter=5.0, drop=drop)      # detections: jittered, some missing
        ab = ab * 1.7 + torch.tensor([30.0, 12.0, 30.0, 12.0])           # different scale / offset
        ge = ref_plano.build_graph(eb, el, 0.5)
        ga = ref_plano.build_graph(ab, al, 0.5)
        hyp = ref_plano.build_hypotheses(ge, ga)
        match = ref_plano.large_common_subgraph(ge, ga)
        plano.append({'expected_boxes': eb, 'expected_labels': el, 'actual_boxes': ab, 'actual_labels': al,
                      'expected_edges': edges(ge), 'actual_edges': edges(ga),
                      'hypotheses': [(float(sc), int(a), int(b)) for sc, a, b in hyp],
                      'matching': sorted((int(a), int(b)) for a, b in match)})
    torch.save(plano, os.path.join(HERE, 'planograms.pt'))

    # ---- 5. dataset readers (cvpce/datautils.py:142-165,195-220,320-387,637-670,709-739; planogram_adapters.py:17-122) ----
    # index builders only: pure csv / json / os / re / torch / networkx.  Inputs are tiny synthetic files written to a temp
    # dir; the fixture stores their text together with what the reference's builders return.
    import json, tempfile
    from cvpce import datautils as ref_du
    from cvpce import planogram_adapters as ref_pa
    files = {}

    def put(root, rel, text):
        full = os.path.join(root, rel)
        os.makedirs(os.path.dirname(full), exist_ok=True)
        with open(full, 'w') as f:
            f.write(text)
        files[rel] = text

    with tempfile.TemporaryDirectory() as root:
        # SKU-110K: 8 columns, rows of one image not contiguous, a skipped image, a malformed row
        put(root, 'sku/annotations.csv', '\n'.join([
            'test_0.jpg,10,20,110,220,object,1000,1500', 'test_1.jpg,5,6,50,60,object,800,600',
            'test_0.jpg,200,210,300,400,object,1000,1500', 'bad,row,only', 'train_882.jpg,1,2,3,4,object,10,10',
            'test_2.jpg,0,0,9,9,object,64,48', 'test_1.jpg,100,100,150,160,object,800,600']) + '\n')
        sku = ref_du.SKU110KDataset.build_index(None, os.path.join(root, 'sku/annotations.csv'), ['train_882.jpg'])
        # GP baseline: header + 6 columns
        put(root, 'base/gt.csv', '\n'.join([
            'image,x1,y1,x2,y2,class', 'store1_12.jpg,10,10,50,60,object', 'store2_3.jpg,1,2,30,40,object',
            'store1_12.jpg,100,110,150,160,object', 'notastore.jpg,1,1,2,2,object', 'short,row']) + '\n')
        base = ref_du.GPBaselineDataset.build_index(None, os.path.join(root, 'imgs'), os.path.join(root, 'base/gt.csv'))
        base = [{**e, 'image_path': os.path.relpath(e['image_path'], root)} for e in base]
        # GP-180: per-image csv, spaces after the commas
        put(root, 'ann/s1_15.csv', 'Food/Candy/12.jpg, 10, 20, 110, 220\nFood/Tea/3.jpg, 200, 20, 310, 230\nFood/Candy/12.jpg, 400, 20, 500, 225\n')
        put(root, 'ann/s2_3.csv', 'Food/Tea/3.jpg, 1, 2, 30, 40\nDrinks/Juice/7.jpg, 50, 2, 90, 44\nmalformed, row\n')
        put(root, 'ann/s3_111.csv', 'Food/Rice/9.jpg, 5, 5, 55, 75\n')
        put(root, 'ann/notes.txt', 'not an annotation file\n')
        ts = ref_du.GroceryProductsTestSet.__new__(ref_du.GroceryProductsTestSet)
        ts.image_dir = 'TESTIMGS'
        gp180 = {}
        for key, (only, skip) in {'all': (None, None), 'only': (['s2_3.csv', 's3_111.csv'], None), 'skip': (None, ['s1_15.csv'])}.items():
            idx = ts.build_index(os.path.join(root, 'ann'), only, skip)
            gp180[key] = sorted(({'id': e['id'], 'path': e['path'], 'anns': e['anns'], 'boxes': e['boxes']} for e in idx),
                                key=lambda e: e['id'])
        # GP training tree (+ TrainingFiles.txt)
        tree = ['Training/Food/Candy/1.jpg', 'Training/Food/Candy/2.png', 'Training/Food/Tea/Green/10.jpg',
                'Training/Food/Tea/Originals/11.jpg', 'Training/Food/Tea/original/12.jpg', 'Training/Background/b1.jpg',
                'Training/Drinks/Juice/7.jpg', 'Training/Drinks/.DS_Store', 'Training/Drinks/Juice/index.txt',
                'Training/Food/Thumbs.db', 'Training/Drinks/Juice/noextension']
        for rel in tree:
            put(root, 'gp/' + rel, 'x')
        put(root, 'gp/TrainingFiles.txt', '\n'.join(['Training/Food/Candy/1.jpg', 'Training/Food/Tea/Green/10.jpg',
                                                       'Training/Background/b1.jpg', 'Training/Drinks/Juice/7.jpg', '']) + '\n')
        import re
        default_skip = re.compile('|'.join(f'({s})' for s in [r'^Background.*$', r'^.*/[Oo]riginals?$']))
        gds = ref_du.GroceryProductsDataset.__new__(ref_du.GroceryProductsDataset)
        walk = {}
        for key, only in {'all': None, 'only_food': ['Food']}.items():
            try:
                p_, c_, a_ = gds.build_index([os.path.join(root, 'gp/Training')], default_skip, only, False)
            except AttributeError:      # 'noextension' makes the reference crash (None.group): captured as such
                p_ = c_ = a_ = None
            walk[key] = None if p_ is None else sorted(zip([os.path.relpath(x, root) for x in p_], c_, a_))
        os.remove(os.path.join(root, 'gp/Training/Drinks/Juice/noextension'))
        files.pop('gp/Training/Drinks/Juice/noextension')
        for key, only in {'all_clean': None, 'only_food_clean': ['Food']}.items():
            p_, c_, a_ = gds.build_index([os.path.join(root, 'gp/Training')], default_skip, only, False)
            walk[key] = sorted(zip([os.path.relpath(x, root) for x in p_], c_, a_))
        p_, c_, a_ = gds.build_index_from_file([os.path.join(root, 'gp')], default_skip, None)
        walk['from_file'] = list(zip([os.path.relpath(x, root) for x in p_], c_, a_))
        # Tonioni planograms: grids with 8-neighbourhood links, objects of different sizes, one ragged last row
        def tonioni(rows, cols, sizes, missing=()):
            cells = [(r, c) for r in range(rows) for c in range(cols) if (r, c) not in missing]
            ident = {rc: i for i, rc in enumerate(cells)}
            dirs = {'n': (-1, 0), 's': (1, 0), 'e': (0, 1), 'w': (0, -1), 'ne': (-1, 1), 'nw': (-1, -1), 'se': (1, 1), 'sw': (1, -1)}
            graph = [{'ogg': (r * 5 + c * 3) % len(sizes), **{k: ident.get((r + dr, c + dc), -1) for k, (dr, dc) in dirs.items()}}
                     for r, c in cells]
            objects = [{'width': w, 'height': h, 'img_path': f'Food/Cat{i}/{10 + i}.jpg'} for i, (w, h) in enumerate(sizes)]
            return {'graph': graph, 'objects': objects}
        planos = {}
        for name, spec in {'s1_15.json': (2, 3, [(10, 20), (14, 18)], ()), 's2_3.json': (3, 5, [(60, 90), (45, 100), (70, 80)], ()),
                           's3_111.json': (4, 7, [(30, 40), (36, 52), (25, 44), (41, 40)], ((3, 6), (3, 5)))}.items():
            put(root, 'plano/' + name, json.dumps(tonioni(*spec)))
            b_, l_, g_ = ref_pa.read_tonioni_planogram(os.path.join(root, 'plano/' + name))
            planos[name] = {'boxes': b_, 'labels': l_, 'nodes': sorted((int(n), d['label']) for n, d in g_.nodes(data=True)),
                            'edges': sorted((int(a), int(b), d['dir']) for a, b, d in g_.edges(data=True))}
        # internal planogram set
        put(root, 'internal/index.json', json.dumps([{'image': 'a.png', 'planogram': 'a.json', 'correct': 7, 'facings': 9},
                                                     {'image': 'b.png', 'planogram': 'b.json', 'correct': 4, 'facings': 4}]))
        put(root, 'internal/a.json', json.dumps([{'code': '570', 'box': [0, 0, 10, 20]}, {'code': '571', 'box': [12, 0, 22, 30]},
                                                 {'code': '570', 'box': [0, 35, 10, 50]}]))
        put(root, 'internal/b.json', json.dumps([{'code': '9', 'box': [1.5, 2.5, 3.5, 8.0]}]))
        internal = ref_du.InternalPlanoSet.build_index(None, os.path.join(root, 'internal'))
        internal = [{**e, 'img': os.path.relpath(e['img'], root)} for e in internal]
    torch.save({'files': files, 'sku110k': sku, 'gpbaseline': base, 'gp180': gp180, 'gp_walk': walk, 'tonioni': planos,
                'internal': internal}, os.path.join(HERE, 'datasets.pt'))
    print('golden fixtures written to', HERE)


if __name__ == '__main__':
    main()
