"""Round-4 host logic on the CPU: the embedder's constant-padding work lists (cvpce_amd/models/classification.py `MACVGGEngine.skip_plan`,
include/cvpce_amd.h `cvpce_skip_layer`), checked WITHOUT a GPU:

* the schedule of VGG16 cfg 'D' (`/root/reference/cvpce/models/classification.py:27-37`: conv / ReLU / MaxPool2d stack up to
  relu5_3) -- layers, tile shapes, op chain, which tensor every layer reads and writes;
* the extent rule itself against a brute-force experiment on the CPU: a crop padded to a square with a constant
  (`/root/reference/cvpce/datautils.py:232-239`) goes through a stack of random 3x3 convs and 2x2 pools; beyond the extent the
  rule predicts, every tensor of the stack equals the all-padding crop's tensor EXACTLY, and the rule is tight (the last row
  inside the extent does differ)."""
import torch
import torch.nn.functional as F


def _extent(e0, size0, pool_mask, nops, size):
    """include/cvpce_amd.h: the crop's content extent through the first `nops` ops of the pass (conv: +1, pool: halve upwards)."""
    if e0 >= size0:
        return size
    e = e0
    for i in range(nops):
        e = (e + 1) // 2 if (pool_mask >> i) & 1 else e + 1
    return min(size, e)


def test_vgg16_schedule():
    from cvpce_amd import synthetic
    from cvpce_amd.models import classification as C
    enc = synthetic.synthetic_macvgg(seed=1)
    eng = C.MACVGGEngine(enc, torch.device('cpu'))              # (packing weights needs no GPU; nothing is launched)
    steps, layers, pool_mask = eng.skip_plan(256)
    assert [st[0] for st in steps] == ['stem'] + ['conv'] * 11
    # conv / pool chain of VGG16 up to relu5_3
    assert [(pool_mask >> i) & 1 for i in range(17)] == [0, 0, 1, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0]
    names = ('H', 'W', 'tile_h', 'tile_w', 'out_ops', 'in_H', 'in_W', 'in_ops', 'skip')
    L = [dict(zip(names, l)) for l in layers]
    assert [l['in_ops'] for l in L] == [0, 3, 4, 6, 7, 8, 10, 11, 12, 14, 15, 16]
    assert [l['out_ops'] for l in L] == [3, 4, 6, 7, 8, 10, 11, 12, 14, 15, 16, 17]
    assert [l['in_H'] for l in L] == [256, 128, 128, 64, 64, 64, 32, 32, 32, 16, 16, 16]
    assert [l['H'] for l in L] == [128, 128, 64, 64, 64, 32, 32, 32, 16, 16, 16, 16]        # output tensors (after a fused pool; conv5_3 stores nothing)
    # tiles in OUTPUT pixels: stem 16x16 conv pixels = 8x8 pooled; wide-tile kernel 16x32 (pooled 8x16); halo2 16x16 (pooled 8x8)
    assert [(l['tile_h'], l['tile_w']) for l in L] == [(8, 8), (16, 32), (8, 16), (16, 16), (16, 16), (8, 8), (16, 16), (16, 16), (8, 8), (16, 16), (16, 16), (16, 16)]
    # every tensor is produced and consumed under the same name (number of ops before it)
    for a, b in zip(L, L[1:]):
        assert a['out_ops'] == b['in_ops'] and a['H'] == b['in_H']
    # MAC layers: conv4_3 (pooled output feeds conv5_1) and conv5_3 (no stored output)
    assert [st[3] for st in steps[1:]] == [False] * 7 + [True, False, False, True]
    assert steps[8][2] and steps[8][4] and not steps[11][4]
    # an engine whose plan is not stem + halo convolutions has no work-list schedule
    assert eng.skip_plan(250) is None                            # (not a multiple of the tile)


def test_extent_rule_matches_a_brute_force_experiment():
    g = torch.Generator().manual_seed(0)
    S, C = 64, 3
    chain = [0, 0, 1, 0, 0, 1, 0, 0, 0]                          # conv conv pool conv conv pool conv conv conv
    pool_mask = sum(1 << i for i, p in enumerate(chain) if p)
    # positive weights and content brighter than the padding: a window that touches content always yields a larger value than the
    # constant crop's (no dead ReLU, no max-pool picking the padding), so the tightness check below cannot pass by accident
    ws = [torch.rand(C, C, 3, 3, generator=g) * 0.2 + 0.01 for p in chain if not p]
    pad_value = 0.37

    def run(x):
        outs, wi = [], 0
        for p in chain:
            if p:
                x = F.max_pool2d(x, 2, 2)
            else:
                x = F.relu(F.conv2d(x, ws[wi], padding=1) + 0.1)
                wi += 1
            outs.append(x)
        return outs

    const = run(torch.full((1, C, S, S), pad_value))
    for ey0, ex0 in ((23, S), (S, 17), (40, S), (1, S), (S, S), (S, 63)):
        x = torch.full((1, C, S, S), pad_value)
        x[:, :, :ey0, :ex0] = torch.rand(1, C, ey0, ex0, generator=g) + 1.0  # content top-left, constant padding below / right
        outs = run(x)
        for k, (t, c) in enumerate(zip(outs, const)):
            size = t.shape[-1]
            ey, ex = _extent(ey0, S, pool_mask, k + 1, size), _extent(ex0, S, pool_mask, k + 1, size)
            assert torch.equal(t[..., ey:, :], c[..., ey:, :]) and torch.equal(t[..., :, ex:], c[..., :, ex:]), (ey0, ex0, k)
            if 0 < ey < size:                                    # tight: the last row inside the extent is not the constant crop's
                assert not torch.equal(t[..., ey - 1, :], c[..., ey - 1, :]), (ey0, k)
            if 0 < ex < size:
                assert not torch.equal(t[..., :, ex - 1], c[..., :, ex - 1]), (ex0, k)


def test_fitted_head_fixture_detects_products_in_the_oracle():
    """tests/golden/fitted_head.pt (made by tests/golden/fit_head.py) over the seeded base: the fp32 ORACLE finds the pasted products
    of unseen structured scenes -- the premise of the accuracy tests that read mAP against true boxes (tests/test_gpu_accuracy.py)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import accuracy
    from cvpce_amd import metrics, synthetic
    from oracle import gln as og
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    det, sd, recipe = accuracy.fitted_detector(200)
    assert recipe['residual_gain'] == 0.25 and len(recipe['fit_keys']) == 10
    # reference-format keys (proposals.py:162-168 via torchvision RetinaNetHead), all towers + the two output convs
    assert all(k.startswith('head.classification_head.') or k.startswith('head.regression_head.') for k in recipe['fit_keys'])
    products = synthetic.product_images(128, seed=200)
    scenes = [synthetic.structured_shelf(i, 1024, 1024, products) for i in range(2)]
    res = [og.gln_forward([sc[0]], sd, detections_per_img=200)[0] for sc in scenes]
    m = metrics.calculate_metrics([sc[1] for sc in scenes], [r['boxes'] for r in res], [r['scores'] for r in res], iou_thresholds=(0.5,))[0.5]
    assert float(m['ap']) >= 0.7 and float(m['ar_300']) >= 0.9, m
    conf = [int((r['scores'] > 0.5).sum()) for r in res]
    assert all(0.5 * len(sc[1]) <= c <= 3 * len(sc[1]) for c, sc in zip(conf, scenes)), (conf, [len(sc[1]) for sc in scenes])   # a bimodal score field


def test_bottleneck_fragment_major_packer():
    """`ops.pack_bottleneck_weights` (round 5: the fused bottleneck reads fragment-major weights) against the index formulas of
    include/cvpce_amd.h cvpce_bottleneck_fused_fm, for the three widths the kernel is instantiated for: every weight exactly once, and the
    element at a hand-computed (fragment, lane, e) position is the weight the formula names."""
    import torch
    from cvpce_amd import ops
    g = torch.Generator().manual_seed(0)
    for p, cin in ((64, 64), (64, 256), (128, 512), (256, 1024)):
        c1 = ops.PackedConv(torch.randn(p, cin, 1, 1, generator=g), torch.zeros(p), 1, 0, device='cpu')
        c2 = ops.PackedConv(torch.randn(p, p, 3, 3, generator=g), torch.zeros(p), 1, 1, device='cpu')
        c3 = ops.PackedConv(torch.randn(4 * p, p, 1, 1, generator=g), torch.zeros(4 * p), 1, 0, device='cpu')
        w1f, w2f, w3f = ops.pack_bottleneck_weights(c1, c2, c3)
        assert ops.pack_bottleneck_weights(c1, c2, c3)[0] is w1f                       # cached on the block
        for f, c, rows, k in ((w1f, c1, p, cin), (w2f, c2, p, 9 * p), (w3f, c3, 4 * p, p)):
            assert f.numel() == rows * k
            assert torch.equal(f.view(torch.int16).sort().values, c.weight[:rows, :k].contiguous().view(torch.int16).reshape(-1).sort().values)
        ks, b, lq, l16, e = 1, 3, 2, 5, 3
        assert w1f[(((ks * (p // 16) + b) * 64 + 16 * lq + l16) * 8) + e] == c1.weight[32 * (b >> 1) + 8 * (l16 >> 2) + (l16 & 3) + 4 * (b & 1), 32 * ks + 8 * lq + e]
        gg, ks, h, lq, l16, e = 2, 1, 1, 3, 9, 7
        assert w3f[((((gg * (p // 32) + ks) * 2 + h) * 64 + 16 * lq + l16) * 8) + e] == c3.weight[32 * gg + 8 * (l16 >> 2) + (l16 & 3) + 4 * h, 32 * ks + 8 * lq + e]
        cw = 16 if p == 64 else 32
        ncb, ns = cw // 16, (p // 64) * 6
        cg, st, kh, h, lq, l16, e = (p // cw) - 1, ns - 1, 2, ncb - 1, 1, 6, 2
        c64, kw, hf = st // 6, (st % 6) >> 1, st & 1
        cout = cw * cg + ((8 * (l16 >> 2) + (l16 & 3) + 4 * h) if ncb == 2 else l16)
        assert w2f[((((((cg * ns + st) * 3 + kh) * ncb + h) * 64) + 16 * lq + l16) * 8) + e] == c2.weight[cout, ((3 * c64 + kh) * 3 + kw) * 64 + 32 * hf + 8 * lq + e]
