"""Model-level parity on the GPU: the HIP schedule of the whole detector / embedder / pipeline
against the fp32 CPU oracle, on seeded synthetic weights and images (SURVEY.md 8d configs 1-3,
scaled so the oracle finishes in seconds).

Float tolerance (stated, per north_star): activations are stored in bf16 between layers
(8 mantissa bits), so feature maps agree with the fp32 oracle to ~1-2 % of their dynamic range
after 50+ layers; scores within 0.02; boxes of matched detections within 2 px at 2048 px scale
(1e-3 relative); embeddings cosine > 0.999.  Index outputs are exact whenever the oracle is fed
the same fp32 head outputs / the same embeddings (stage-isolated), see the tests below.
"""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def nchw(y):
    return y.float().permute(0, 3, 1, 2).cpu()


def rel(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp(min=1e-6)).item()


def l2rel(a, b):
    return ((a - b).norm() / b.norm().clamp(min=1e-12)).item()


def box_iou(a, b):
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = torch.max(a[:, None, :2], b[:, :2]); rb = torch.min(a[:, None, 2:], b[:, 2:])
    inter = (rb - lt).clamp(min=0).prod(dim=2)
    return inter / (area_a[:, None] + area_b - inter)


@pytest.fixture(scope='module')
def gln_model(cuda):
    from cvpce_amd import synthetic
    m = synthetic.synthetic_gln(seed=0, detections_per_img=200)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    return m.to(cuda), sd


@pytest.fixture(scope='module')
def vgg_model(cuda):
    from cvpce_amd import synthetic
    m = synthetic.synthetic_macvgg(seed=1)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    return m.to(cuda), sd


def test_macvgg_parity(cuda, vgg_model):
    from oracle import macvgg as ovgg
    model, sd = vgg_model
    x = torch.rand(3, 3, 256, 256, generator=torch.Generator().manual_seed(5)) * 2 - 1
    ref, ref_desc = ovgg.macvgg_forward(x, sd, return_descs=True)
    got = model(x.to(cuda)).cpu()
    assert got.shape == (3, 1024) and got.dtype == torch.float32
    assert (got >= 0).all()
    torch.testing.assert_close(got.norm(dim=1), torch.ones(3), rtol=0, atol=1e-5)
    cos = F.cosine_similarity(got, ref, dim=1)
    assert cos.min() > 0.999, cos
    assert (got - ref).abs().max() < 5e-3


def test_macvgg_bn_parity(cuda):
    """`macvgg_embedder('vgg16_bn')` -- the reference's default embedder (classification.py:97): eval-mode BatchNorm folded
    into the packed conv weights (fused stem included) against the oracle's literal conv -> F.batch_norm -> relu chain."""
    from cvpce_amd import synthetic
    from oracle import macvgg as ovgg
    m = synthetic.synthetic_macvgg(seed=2, batch_norm=True)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    assert 'block1.1.running_mean' in sd and 'block2.41.weight' in sd and 'block1.30.weight' in sd
    x = torch.rand(4, 3, 256, 256, generator=torch.Generator().manual_seed(6)) * 2 - 1
    ref = ovgg.macvgg_forward(x, sd)
    got = m.to(cuda)(x.to(cuda)).cpu()
    assert got.shape == (4, 1024)
    torch.testing.assert_close(got.norm(dim=1), torch.ones(4), rtol=0, atol=1e-5)
    cos = F.cosine_similarity(got, ref, dim=1)
    assert cos.min() > 0.999, cos
    assert (got - ref).abs().max() < 5e-3


def test_config1_plumbing_and_parity(cuda, gln_model):
    """BASELINE config 1: one 640x640 random image through gln() + ProposalGenerator."""
    from cvpce_amd import production
    from oracle import gln as og
    model, sd = gln_model
    img = torch.rand(3, 640, 640, generator=torch.Generator().manual_seed(0))
    res = model([img.to(cuda)])
    assert isinstance(res, list) and len(res) == 1
    r = res[0]
    assert set(r) == {'boxes', 'scores', 'labels', 'gaussians'}
    assert r['boxes'].dtype == torch.float32 and r['boxes'].shape[1] == 4
    assert r['labels'].dtype == torch.int64 and (r['labels'] == 0).all()
    assert r['gaussians'].shape == (1, 400, 400) and r['gaussians'].dtype == torch.float32
    assert len(r['boxes']) == len(r['scores']) <= 200
    assert (r['scores'][:-1] >= r['scores'][1:]).all()
    pg = production.ProposalGenerator(model, device=cuda)
    boxes, crops = pg.generate_proposals_and_images(img)
    assert crops.shape[1:] == (3, 256, 256) and len(boxes) == len(crops) == int((r['scores'] > 0.5).sum())
    # oracle
    ref = og.gln_forward([img], sd, detections_per_img=200)[0]
    g = r['gaussians'].cpu()
    # the product default (fp16 storage) measures l2rel 0.013 and 99 % of the oracle's boxes at IoU > 0.9 here: the gates sit just
    # above that, so a regression to bf16-level agreement (0.11, 93 %) fails
    assert model.precision == 'fp16'
    assert l2rel(g, ref['gaussians']) < 0.05, l2rel(g, ref['gaussians'])
    _compare_detections(r, ref, min_frac=0.95)
    # the opt-in bf16 storage keeps the loose bounds (7 mantissa bits through 50+ random-weight layers)
    model.set_precision('bf16')
    try:
        rb = model([img.to(cuda)])[0]
    finally:
        model.set_precision('fp16')
    assert l2rel(rb['gaussians'].cpu(), ref['gaussians']) < 0.25
    _compare_detections(rb, ref, min_frac=0.8)


def _compare_detections(r, ref, min_frac=0.8):
    gb, gs = r['boxes'].cpu(), r['scores'].cpu()
    rb, rs = ref['boxes'], ref['scores']
    assert abs(len(gb) - len(rb)) <= max(3, 0.05 * len(rb))
    if len(rb) == 0:
        return
    iou = box_iou(gb, rb)
    best, arg = iou.max(dim=1)
    matched = best > 0.9
    assert matched.float().mean() > min_frac, matched.float().mean()
    assert (gs[matched] - rs[arg[matched]]).abs().max() < 0.02
    scale = rb.abs().max()
    assert (gb[matched] - rb[arg[matched]]).abs().max() < 2e-3 * scale + 0.5


@pytest.mark.parametrize('precision', ['fp16', 'bf16'])
def test_gln_intermediates_and_stage_exact(cuda, gln_model, precision):
    """Two images of different shapes in one batch (ragged -> padded), all intermediate tensors
    against the oracle, then K6-K8 re-run by the oracle on the GPU's own fp32 head outputs: exact.
    Both storage modes of the detector: fp16 (the default) and the opt-in bf16, each against the CPU model of ITS rounding points."""
    from oracle import gln as og
    model, sd = gln_model
    model.set_precision(precision)
    try:
        _gln_intermediates_and_stage_exact(cuda, model, sd, og)
    finally:
        model.set_precision('fp16')


def _gln_intermediates_and_stage_exact(cuda, model, sd, og):
    imgs = [torch.rand(3, 480, 640, generator=torch.Generator().manual_seed(1)),
            torch.rand(3, 700, 500, generator=torch.Generator().manual_seed(2))]
    eng = model.engine()
    out, inter = eng.detect([i.to(cuda) for i in imgs], 1, 200, 0.5, want_intermediates=True)
    boxes, scores, labels, count, conf, gauss = out
    ref, rint = og.gln_forward(imgs, sd, detections_per_img=200, return_intermediates=True)
    assert tuple(inter['batch'].shape[1:3]) == tuple(rint['batch'].shape[-2:])
    assert rel(nchw(inter['batch'])[:, :3], rint['batch']) < 1e-2
    # (1) end to end against the literal fp32 oracle: bf16 storage through 50+ random-weight layers
    for got, want in zip(inter['features'], rint['features']):
        assert rel(nchw(got), want) < 0.04, rel(nchw(got), want)
        assert l2rel(nchw(got), want) < 0.02
    for got, want in zip(inter['cls'], rint['cls']):
        assert (got.view(2, -1).cpu() - want.view(2, -1)).abs().max() < 0.15 * want.std() + 0.05
    for got, want in zip(inter['reg'], rint['reg']):
        assert (got.view(2, -1).cpu() - want.view(2, -1)).abs().max() < 0.15 * want.std() + 0.02
    # ill-conditioned with random weights (sparse ReLU output): fp16 storage (the default) measures 0.011, bf16 0.118
    assert l2rel(gauss.cpu(), rint['gaussians']) < (0.05 if model.precision == 'fp16' else 0.25), l2rel(gauss.cpu(), rint['gaussians'])
    # (2) every stage against the CPU model of the SAME numerics, fed the GPU's own stage inputs: tight
    from oracle import bf16_model as bm
    nm = bm.FP16 if model.precision == 'fp16' else bm.BF16
    c2, c3, c4, c5 = [nchw(t) for t in inter['c']]
    feats = [nchw(t) for t in inter['features']]
    m_c = bm.body(nchw(inter['batch'])[:, :3], sd, nm=nm)
    assert l2rel(c2, m_c[0]) < 2e-3, l2rel(c2, m_c[0])     # stem + layer1 (10 convs, pool, 3 residual adds)
    for got, want in zip(feats, bm.fpn(c3, c4, c5, sd, nm=nm)):
        assert l2rel(got, want) < 2e-3, l2rel(got, want)
    assert l2rel(gauss.cpu(), bm.gaussian_branch(c2, feats[0], sd, nm=nm)) < 3e-2   # 8 layers, sparse output: see the golden test below
    m_cls, m_reg = bm.heads(feats, sd, nm=nm)
    for got, want in zip(inter['cls'], m_cls):
        assert (got.view(2, -1).cpu() - want.view(2, -1)).abs().max() < 0.03 * want.std() + 5e-3
    for got, want in zip(inter['reg'], m_reg):
        assert (got.view(2, -1).cpu() - want.view(2, -1)).abs().max() < 0.03 * want.std() + 5e-3
    # stage-isolated exactness of decode/top-k/NMS/rescale on real head outputs
    n = 2
    cls = [c.view(n, -1, 1).cpu() for c in inter['cls']]
    reg = [r.view(n, -1, 4).cpu() for r in inter['reg']]
    for i in range(n):
        b, s, l = og.postprocess_image([c[i] for c in cls], [r[i] for r in reg], rint['anchors'],
                                       rint['image_sizes'][i], 200)
        b = og.resize_boxes(b, rint['image_sizes'][i], tuple(imgs[i].shape[-2:]))
        c = int(count[i])
        assert c == len(b)
        torch.testing.assert_close(scores[i, :c].cpu(), s, rtol=0, atol=1e-6)
        torch.testing.assert_close(boxes[i, :c].cpu(), b, rtol=1e-5, atol=2e-3)
        assert int(conf[i]) == int((s > 0.5).sum())
    for i in range(n):
        c = int(count[i])
        # (fp16 measures 97.5 / 98.5 % of the oracle's boxes at IoU > 0.9 on these two images, bf16 90 %)
        _compare_detections({'boxes': boxes[i, :c], 'scores': scores[i, :c]}, ref[i], min_frac=0.95 if model.precision == 'fp16' else 0.7)


def test_pipeline_matches_per_image_api(cuda, gln_model, vgg_model):
    """BatchedPipeline (fused, device-resident) == ProposalGenerator + Classifier per image
    (the reference's production.py flow), and == the oracle matcher on the GPU's embeddings."""
    from cvpce_amd import production, synthetic
    from oracle import match as omatch, crop as ocrop, macvgg as ovgg
    det, _ = gln_model
    enc, vsd = vgg_model
    gal = synthetic.gallery_images(96, seed=100)
    ann = [f'sku_{i}' for i in range(96)]
    clf = production.Classifier(enc, synthetic.TensorGallery(gal, ann), device=cuda, emb_device=cuda, batch_size=32, k=2,
                                match_dtype=torch.float32)
    assert clf.embedding.shape == (96, 1024) and len(clf.annotations) == 96
    ref_gal = ovgg.macvgg_forward(gal[:4], vsd)
    assert F.cosine_similarity(clf.embedding[:4].cpu(), ref_gal, dim=1).min() > 0.999
    det.detections_per_img = 24
    try:
        # same-size images: like torchvision, results depend on the padded batch shape (anchor strides = padded // grid)
        imgs = [synthetic.shelf_image(7, 600, 800).to(cuda), synthetic.shelf_image(8, 600, 800).to(cuda)]
        pipe = production.BatchedPipeline(det, clf, 0.5)
        out = pipe.run(imgs)
        pg = production.ProposalGenerator(det, device=cuda)
        for i, img in enumerate(imgs):
            boxes, crops = pg.generate_proposals_and_images(img)
            c = int(out['count'][i])
            assert c == len(boxes) > 0
            assert torch.equal(out['boxes'][i, :c], boxes)
            labels, emb = clf.classify(crops, return_embedding=True)
            idx = out['indices'][i, :c].cpu()
            assert [[ann[j] for j in row] for row in idx.tolist()] == labels
            assert (out['indices'][i, c:] == -1).all()
            # crops against the oracle crop of the same boxes
            ref_crops = ocrop.crop_boxes(img.cpu(), boxes.cpu())
            torch.testing.assert_close(crops.cpu(), ref_crops, rtol=0, atol=2e-6)
            # matcher against the oracle on the same embeddings: exact on tie-free rows
            d = omatch.cosine_distance_matrix(clf.embedding.cpu(), emb.cpu())
            ref_idx = torch.sort(d, dim=-1, stable=True).indices[:, :2]
            srt = d.sort(dim=-1).values[:, :3]
            safe = (srt[:, 1:] - srt[:, :-1]).min(dim=1).values > 2e-6
            assert idx[safe].equal(ref_idx[safe])
            # embeddings against the oracle embedder on the oracle crops
            ref_emb = ovgg.macvgg_forward(ocrop.scale_to_tanh(ref_crops[:4]), vsd)
            assert F.cosine_similarity(emb[:4].cpu(), ref_emb, dim=1).min() > 0.999
    finally:
        det.detections_per_img = 200


def test_index_save_load_roundtrip(cuda, vgg_model, tmp_path):
    from cvpce_amd import production, synthetic
    enc, _ = vgg_model
    gal = synthetic.gallery_images(8, seed=3)
    clf = production.Classifier(enc, synthetic.TensorGallery(gal), device=cuda, emb_device=cuda, batch_size=4)
    p = str(tmp_path / 'idx.pkl')
    clf.save_index(p)
    saved = torch.load(p, weights_only=False)
    assert set(saved) == {'embedding', 'annotations'}          # production.py:49-56 format
    clf2 = production.Classifier(enc, None, device=cuda, emb_device=cuda, load=p)
    assert torch.equal(clf2.embedding, clf.embedding) and clf2.annotations == clf.annotations
    q = (gal[:3] + 1) / 2     # classify() takes [0,1] crops
    assert clf2.classify(q.to(cuda)) == [[a] for a in clf.annotations[:3]]
    assert clf.classify(torch.empty(0, 3, 256, 256, device=cuda)) == []


def test_classify_and_index_are_independent_of_batching(cuda, vgg_model):
    """`batch_size` (production.py:23; 8 in cli/eval.py) bounds one forward pass of the reference; here passes are coalesced
    up to production.ENGINE_BATCH images.  Per-crop results must not depend on the batch they ran in: labels and embeddings of
    classify() and the gallery index are bit-identical between 5-image passes and one coalesced pass."""
    from cvpce_amd import production, synthetic
    enc, _ = vgg_model
    gal = synthetic.gallery_images(23, seed=11)
    crops = (synthetic.gallery_images(37, seed=12) + 1) / 2          # [0,1] like resize_for_classification's output
    keep = production.ENGINE_BATCH
    try:
        production.ENGINE_BATCH = 1                                  # -> every pass is exactly batch_size images
        small = production.Classifier(enc, synthetic.TensorGallery(gal), device=cuda, emb_device=cuda, batch_size=5, num_workers=0)
        l_small, e_small = small.classify(crops.to(cuda), return_embedding=True)
        production.ENGINE_BATCH = keep
        big = production.Classifier(enc, synthetic.TensorGallery(gal), device=cuda, emb_device=cuda, batch_size=5, num_workers=0)
        l_big, e_big = big.classify(crops.to(cuda), return_embedding=True)
    finally:
        production.ENGINE_BATCH = keep
    assert torch.equal(small.embedding, big.embedding) and small.annotations == big.annotations
    assert l_small == l_big and torch.equal(e_small, e_big)
    assert len(l_big) == 37 and e_big.shape == (37, 1024)


def test_cpu_device_fails_loudly():
    from cvpce_amd import synthetic
    from cvpce_amd.models import classification as C
    m = synthetic.synthetic_gln(seed=0, calibrate=False)
    with pytest.raises(RuntimeError, match='HIP'):
        m([torch.rand(3, 64, 64)])
    with pytest.raises(RuntimeError, match='HIP'):
        C.nearest_neighbors(torch.rand(4, 8), torch.rand(2, 8))


@pytest.mark.parametrize('tanh', [False, True])
def test_gaussian_branch_vs_reference_golden(cuda, golden_dir, tanh):
    """The HIP Gaussian-branch schedule against outputs of the REFERENCE's own GaussianLayer + GaussianSubnet
    (tests/golden/gaussian_head.pt, made by importing /root/reference/cvpce/models/proposals.py:51-107).
    The fixture uses reduced widths (16/8/4/2/1 channels); channels are zero-padded to the kernels' multiple
    of 8, which leaves the mathematical result unchanged."""
    import os
    from torch import nn
    from cvpce_amd.models import proposals as P
    g = torch.load(os.path.join(golden_dir, 'gaussian_head.pt'), weights_only=False)[f'tanh_{tanh}']

    def pad8(n):
        return (n + 7) // 8 * 8

    layer = P.GaussianLayer(16, 16)
    layer.load_state_dict(g['layer_state'])
    # widen: 16->16 lateral, 16->8 block1, 8->4 (pad 8) block2, subnet 4->2->2->1->1->1 all padded to 8 (last stays 1)
    def widen_conv(conv, cin_p, cout_p):
        w = torch.zeros(cout_p, cin_p, *conv.weight.shape[2:])
        w[:conv.weight.shape[0], :conv.weight.shape[1]] = conv.weight.data
        b = torch.zeros(cout_p); b[:conv.bias.shape[0]] = conv.bias.data
        new = nn.Conv2d(cin_p, cout_p, conv.kernel_size, padding=conv.padding)
        new.weight.data, new.bias.data = w, b
        return new

    def widen_bn(bn, n):
        new = nn.BatchNorm2d(n)
        k = bn.weight.shape[0]
        new.weight.data[:k], new.bias.data[:k] = bn.weight.data, bn.bias.data
        new.running_mean[:k], new.running_var[:k] = bn.running_mean, bn.running_var
        return new.eval()

    layer.block2.conv, layer.block2.norm = widen_conv(layer.block2.conv, 8, 8), widen_bn(layer.block2.norm, 8)
    subnet = P.GaussianSubnet(4, tanh)
    subnet.load_state_dict(g['subnet_state'])
    for i, blk in enumerate(subnet.blocks):
        blk.conv = widen_conv(blk.conv, 8, 1 if i == 4 else 8)
    # both storage modes against the REFERENCE's fp32 output: bf16 (default) and the fp16 accuracy mode
    for dt, l2tol, abstol in ((torch.bfloat16, 3e-2, 0.05), (torch.float16, 4e-3, 0.006)):
        eng = P.GLNEngine.__new__(P.GLNEngine)
        eng.pack_gaussian(layer.eval(), subnet, cuda, dtype=dt)
        nhwc = lambda x: x.permute(0, 2, 3, 1).contiguous().to(dt).to(cuda)
        out = eng.gaussian_branch(nhwc(g['c2']), nhwc(g['p3'])).permute(0, 3, 1, 2).cpu()
        assert out.shape == g['gaussians'].shape
        # 16-bit storage of inputs / weights / 7 intermediate tensors vs the reference's fp32
        assert l2rel(out, g['gaussians']) < l2tol, (dt, l2rel(out, g['gaussians']))
        assert (out - g['gaussians']).abs().max() < abstol * g['gaussians'].abs().max() + (1e-2 if dt == torch.bfloat16 else 1e-3), dt


@pytest.mark.parametrize('hw', [(100, 100), (64, 88), (20, 37)])
def test_head_atlas_equals_per_level(cuda, gln_model, hw):
    """RetinaNet head on the level atlas (one masked halo launch per tower layer) against the per-level launches."""
    from cvpce_amd.models import proposals as P
    model, _ = gln_model
    eng = model.engine()
    g = torch.Generator().manual_seed(hw[0])
    shapes, (h, w) = [], hw
    for _ in range(5):
        shapes.append((h, w)); h, w = (h + 1) // 2, (w + 1) // 2
    feats = [torch.randn(2, h, w, 256, generator=g).to(eng.dtype).to(cuda) for h, w in shapes]      # (the engine's storage type: fp16 by default)
    hc, wc, offs = eng.atlas_layout(shapes)
    occ = torch.zeros(hc + 2, wc + 2, dtype=torch.int32)
    for (h, w), (oy, ox) in zip(shapes, offs):          # levels (grown by the 1-pixel halo) never overlap another level
        assert oy + h <= hc and ox + w <= wc
        occ[oy:oy + h + 2, ox:ox + w + 2] += 1
        occ[oy + 1:oy + h + 1, ox + 1:ox + w + 1] += 10
    assert int(((occ > 10) & (occ % 10 > 1)).sum()) == 0
    cls_a, reg_a = eng.heads_atlas(feats)
    cls_l, reg_l = eng.heads(feats)
    torch.cuda.synchronize()
    for a, b in zip(cls_a + reg_a, cls_l + reg_l):
        assert a.shape == b.shape
        assert (a - b).abs().max() <= 0.02 * b.abs().max() + 1e-3, ((a - b).abs().max(), b.abs().max())
    # round 6: layer i of BOTH towers as one paired launch (the towers as cout tiles of one conv; opt-in, CVPCE_PAIRED_TOWERS=1): every
    # tile is computed with the same arithmetic as in the per-tower launches, so the two forms agree bit for bit
    assert eng.tower_pairs is None
    eng.tower_pairs = eng.pack_tower_pairs(model)
    assert eng.tower_pairs is not None and eng.tower_pairs[0].cout == 512
    try:
        cls_u, reg_u = eng.heads_atlas(feats)
    finally:
        eng.tower_pairs = None
    torch.cuda.synchronize()
    for a, b in zip(cls_a + reg_a, cls_u + reg_u):
        assert torch.equal(a, b)


def test_detection_does_not_depend_on_the_batch(cuda):
    """the detector's generic convs pick their kernel (LDS-DMA, register-staged, split-K) by N * Ho * Wo inside the library,
    with different fp32 summation orders -- so "an image's result does not depend on what it is batched with" (evaluate_iter, bench.py
    --verify across world sizes) has to be CHECKED at the sizes where the thresholds sit: image 0 of BASELINE-shaped batches of 1, 3, 4
    and 8 images (2048 x 2048 -> 800 x 800) gives bit-identical boxes, scores, labels, counts and Gaussian maps."""
    from cvpce_amd import synthetic
    det = synthetic.synthetic_gln(seed=0, detections_per_img=200).to(cuda)
    eng = det.engine()
    imgs = [synthetic.shelf_image(g, 2048, 2048).to(cuda) for g in range(8)]
    ref = None
    for nb in (1, 3, 4, 8):
        out = eng.detect(imgs[:nb], 1, 200, 0.5)
        torch.cuda.synchronize()
        cur = [t[0].clone() for t in out[:4]] + [out[5][0].clone()]
        if ref is None:
            ref = cur
            assert int(cur[3]) > 0
        for a, b in zip(cur, ref):
            assert torch.equal(a, b), nb


@pytest.mark.parametrize('batch_norm,desc_layers', [(True, [2, 3]), (False, [1, 2, 4])])
def test_macresnet_parity(cuda, batch_norm, desc_layers):
    """Optional ResNet-50 MAC encoder (classification.py:53-85,111-121) against the fp32 oracle; state-dict keys follow
    the reference's Sequential nesting."""
    from cvpce_amd.models import classification as C
    from cvpce_amd import production, synthetic
    from oracle import macresnet as om
    torch.manual_seed(3)
    m = C.macresnet_encoder(pretrained=False, batch_norm=batch_norm, desc_layers=desc_layers)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.normal_(0, 0.1); mod.running_var.uniform_(0.5, 1.5)
            mod.weight.data.uniform_(0.4, 0.8); mod.bias.data.normal_(0, 0.1)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    assert m.embedding_size == sum(C.MACResNet.layer_output_sizes[l] for l in desc_layers)
    keys = set(sd)
    assert 'blocks.0.0.0.weight' in keys and 'blocks.0.1.0.conv1.weight' in keys and 'blocks.0.1.0.downsample.0.weight' in keys
    assert ('blocks.0.0.1.running_mean' in keys) == batch_norm
    if desc_layers == [2, 3]:
        assert 'blocks.1.0.5.conv3.weight' in keys and not any(k.startswith('blocks.2') for k in keys)    # layer3 has 6 blocks; layer4 unused
    x = torch.rand(5, 3, 256, 256, generator=torch.Generator().manual_seed(4)) * 2 - 1
    ref = om.macresnet_forward(x, sd, desc_layers)
    m2 = C.macresnet_encoder(pretrained=False, batch_norm=batch_norm, desc_layers=desc_layers)
    m2.load_state_dict(sd)
    got = m2.to(cuda)(x.to(cuda)).cpu()
    assert got.shape == ref.shape == (5, m.embedding_size)
    assert torch.allclose(got.norm(dim=1), torch.ones(5), atol=1e-5)
    assert F.cosine_similarity(got, ref, dim=1).min() > 0.999, F.cosine_similarity(got, ref, dim=1)
    # through the Classifier: images in [0,1] are scaled to [-1,1] and NOT normalised further for this encoder
    gal = synthetic.gallery_images(8, seed=9)
    clf = production.Classifier(m2, synthetic.TensorGallery(gal), device=cuda, emb_device=cuda, batch_size=4, match_dtype=torch.float32)
    labels, emb = clf.classify(((gal[:3] + 1) / 2).to(cuda), return_embedding=True)
    assert [l[0] for l in labels] == ['sku_00000', 'sku_00001', 'sku_00002']
    assert F.cosine_similarity(emb.cpu(), om.macresnet_forward(gal[:3], sd, desc_layers), dim=1).min() > 0.999
    with pytest.raises(RuntimeError):
        C.macresnet_encoder(pretrained=True)


def test_embedder_full_size_properties(cuda, vgg_model):
    """BASELINE-size batch (8 images x 200 proposals = 1600 crops) through the whole embedder schedule, checked by
    size-independent properties: (1) equivariance under a permutation of the batch, bit for bit -- every persistent
    kernel's tile decode and the 6 x 256 + 64 batch split; (2) duplicated crops give identical rows; (3) unit norm,
    non-negative entries (MAC of post-ReLU maps)."""
    from cvpce_amd import ops
    from cvpce_amd.models import classification as C
    enc, _ = vgg_model
    eng = enc.engine()
    g = torch.Generator().manual_seed(77)
    base = torch.rand(400, 3, 256, 256, generator=g) * 2 - 1
    x = base.repeat(4, 1, 1, 1)                       # crops i, i+400, i+800, i+1200 are identical
    perm = torch.randperm(1600, generator=g)
    packed = ops.pack_embed_input(x.to(cuda), False, C.TANH_MEAN, C.TANH_STD)
    e1 = eng.embed_packed(packed)
    e2 = eng.embed_packed(packed[perm.to(cuda)].contiguous())
    torch.cuda.synchronize()
    assert e1.shape == (1600, 1024)
    assert torch.equal(e2, e1[perm.to(cuda)])
    assert torch.equal(e1[:400], e1[400:800]) and torch.equal(e1[:400], e1[1200:])
    assert torch.allclose(e1.norm(dim=1), torch.ones(1600, device=cuda), atol=1e-5) and float(e1.min()) >= 0.0


def test_detector_full_size_properties(cuda, gln_model):
    """BASELINE-size detector batch (8 x 3 x 2048 x 2048) by size-independent properties: a second run is bit-identical
    (no atomics, no launch-order dependence -- head towers run on side streams), and reversing the batch reverses the
    per-image results bit for bit; boxes lie inside the image, scores are sorted, counts are consistent."""
    from cvpce_amd import synthetic
    det, _ = gln_model
    eng = det.engine()
    imgs = [synthetic.shelf_image(40 + i, 2048, 2048).to(cuda) for i in range(8)]
    a = eng.detect(imgs, 1, 200)
    b = eng.detect(imgs, 1, 200)
    c = eng.detect(imgs[::-1], 1, 200)
    torch.cuda.synchronize()
    for x, y, z in zip(a, b, c):
        assert torch.equal(x, y)
        assert torch.equal(x, z.flip(0))
    boxes, scores, labels, count, conf, gauss = a
    assert boxes.shape == (8, 200, 4) and gauss.shape[0] == 8
    for i in range(8):
        n = int(count[i])
        assert 0 < n <= 200 and int(conf[i]) == int((scores[i, :n] > 0.5).sum())
        s = scores[i, :n]
        assert bool((s[:-1] >= s[1:]).all())
        bx = boxes[i, :n]
        assert float(bx.min()) >= 0 and float(bx[:, 2].max()) <= 2048 and float(bx[:, 3].max()) <= 2048
        assert bool((bx[:, 2] >= bx[:, 0]).all()) and bool((bx[:, 3] >= bx[:, 1]).all())



def test_detector_graph_survives_cache_eviction(cuda, gln_model):
    """The captured detector graph keeps the engine-owned tensors it points at (level masks, tile maps, atlas buffers,
    per-geometry constants) alive: evicting every host-side cache and thrashing the allocator must not change a replay."""
    from cvpce_amd import synthetic
    det, _ = gln_model
    eng = det.engine()
    imgs = [synthetic.shelf_image(90 + i, 512, 640).to(cuda) for i in range(2)]
    a = eng.detect(imgs, 1, 200)          # eager
    b = eng.detect(imgs, 1, 200)          # capture + replay
    for name in ('_atlas_cache', '_atlas_bufs', '_pp_cache'):
        eng.__dict__.get(name, {}).clear()
    junk = [torch.full((1 << 22,), 7.0, device=cuda) for _ in range(64)]    # reuse whatever memory became free
    del junk
    torch.cuda.empty_cache()
    c = eng.detect(imgs, 1, 200)          # replay
    torch.cuda.synchronize()
    for x, y, z in zip(a, b, c):
        assert torch.equal(x, y) and torch.equal(x, z)


def test_detector_graph_cache_policy(cuda):
    """Round-2 advisor finding: first sights must not evict captured graphs, eviction is LRU, and a stream of many geometries
    backs the capture threshold off.  Results never depend on whether a call was eager, captured or replayed."""
    from cvpce_amd import synthetic
    from cvpce_amd.models import proposals as P
    det = synthetic.synthetic_gln(seed=0, detections_per_img=50).to(cuda)
    eng = det.engine()
    sizes = [(256 + 32 * i, 320) for i in range(7)]
    imgs = {s: [synthetic.shelf_image(7, *s).to(cuda)] for s in sizes}
    ref = {s: eng.detect(imgs[s], 1, 50, want_intermediates=True)[0] for s in sizes}      # eager, never captured
    hot = sizes[0]
    for _ in range(3):
        out = eng.detect(imgs[hot], 1, 50)                       # eager, capture, replay
    assert len(eng._graphs) == 1 and eng._capture_on_sight == P.CAPTURE_ON_SIGHT
    for s in sizes[1:]:                                          # six first sights: nothing is captured, nothing evicted
        out = eng.detect(imgs[s], 1, 50)
        for x, y in zip(out, ref[s]):
            assert torch.equal(x, y)
    assert len(eng._graphs) == 1 and next(iter(eng._graphs.values()))['replays'] == 1
    for s in sizes[1:]:                                          # second sights: captured one by one, LRU keeps at most 4
        eng.detect(imgs[hot], 1, 50)                             # the hot geometry keeps being replayed -> most recently used
        out = eng.detect(imgs[s], 1, 50)
        for x, y in zip(out, ref[s]):
            assert torch.equal(x, y)
        assert len(eng._graphs) <= P.MAX_DETECT_GRAPHS
    assert any(k[2] == (P.resized_hw(*hot),) for k in eng._graphs), 'the hot geometry was evicted'
    assert eng._capture_on_sight > P.CAPTURE_ON_SIGHT            # barely-used graphs were evicted: capture now needs more sights
    out = eng.detect(imgs[hot], 1, 50)
    for x, y in zip(out, ref[hot]):
        assert torch.equal(x, y)


# ---- round 3: reference-made fixtures (tests/golden/members.pt) on the HIP path --------------------------------------------
def test_macresnet_hip_matches_reference_fixture(cuda, golden_dir):
    """The HIP MACResNet loaded with the state dict the REFERENCE's MACResNet produced (key nesting pinned by load_state_dict
    strict=True) against the reference's own output (classification.py:53-85)."""
    from cvpce_amd.models import classification as C
    m = torch.load(os.path.join(golden_dir, 'members.pt'), weights_only=False)['macresnet']
    for case in m['cases']:
        src = C._ResNetSource(tuple(m['layers']), torch.nn.BatchNorm2d, stem=m['stem'], planes=tuple(m['planes']))
        model = C.MACResNet(src, list(case['descriptor_layers']))
        model.load_state_dict({k: m['source_state'][s] for k, s in case['state_key_to_source_key'].items()}, strict=True)
        assert model.embedding_size == case['embedding_size_attr']
        out = model.to(cuda)(case['input'].to(cuda)).cpu()
        assert out.shape == case['output'].shape
        cos = torch.nn.functional.cosine_similarity(out, case['output'], dim=1)
        assert cos.min() > 0.9995, cos
        assert (out - case['output']).abs().max() < 0.02


class _ToyEngine:
    """HIP-side counterpart of the fixture's toy encoder (4x4 average pool -> linear -> unit norm) behind the engine
    interface Classifier.classify drives: `embed_packed((B,S,S,8) bf16 NHWC, already scale_to_tanh'ed)`."""

    def __init__(self, weight):
        self.w = weight

    def embed_packed(self, packed):
        x = packed[..., :3].float().permute(0, 3, 1, 2)
        v = torch.nn.functional.adaptive_avg_pool2d(x, 4).flatten(1) @ self.w.t()
        return v / v.norm(dim=1, keepdim=True).clamp(min=1e-8)


class _ToyEncoder(torch.nn.Module):
    embedding_size = 64
    input_mean, input_std = (0.0, 0.0, 0.0), (1.0, 1.0, 1.0)

    def __init__(self, weight):
        super().__init__()
        self._eng = _ToyEngine(weight)

    def engine(self):
        return self._eng


@pytest.mark.parametrize('match_dtype', [torch.float32, torch.bfloat16])
def test_classifier_classify_matches_reference_fixture(cuda, golden_dir, match_dtype):
    """Classifier.classify (production.py:57-74) as executed by the REFERENCE with a toy encoder: batching (batch sizes that do
    not divide the input), k, return_embedding, empty input, label lookup -- same labels from the HIP matcher."""
    from cvpce_amd import production
    m = torch.load(os.path.join(golden_dir, 'members.pt'), weights_only=False)['classifier']
    enc = _ToyEncoder(m['encoder_state']['proj.weight'].to(cuda))
    up = lambda p: torch.nn.functional.interpolate(p, size=(256, 256), mode='nearest')
    for case in m['cases']:
        clf = production.Classifier.from_embedding(enc, m['gallery_embedding'].to(cuda), m['annotations'], device=cuda, emb_device=cuda,
                                                   batch_size=case['batch_size'], k=case['k'], match_dtype=match_dtype)
        imgs = up(m['query_patterns'][:case['n']]) if case['n'] else torch.empty(0, 3, 256, 256)
        labels = clf.classify(imgs.to(cuda))
        labels2, emb = clf.classify(imgs.to(cuda), return_embedding=True)
        assert labels == labels2 == case['labels'], (case['batch_size'], case['k'])
        assert emb.shape == case['embedding'].shape
        if case['n']:
            assert (emb.cpu() - case['embedding']).abs().max() < 5e-3      # bf16 rounding of the packed input


def test_proposal_generator_single_sync_path_equals_generic(cuda, gln_model):
    """ProposalGenerator.generate_proposals_and_images (production.py:16-20): the engine path (confidence prefix + degenerate-box
    test from ONE device-to-host copy) returns exactly what the reference-shaped generic path (boolean masks over the detector's
    result dict) returns -- including an image whose confident boxes contain degenerate ones."""
    from cvpce_amd import production, synthetic
    det, _ = gln_model

    class Plain:                       # the same detector without `.engine`: forces the generic path
        def __init__(self, d):
            self.d = d

        def __call__(self, x):
            return self.d(x)

    for seed, size in ((3, (512, 640)), (4, (96, 96))):          # 96 x 96: tiny boxes, some collapse under .to(long)
        img = synthetic.shelf_image(seed, *size).to(cuda)
        fast = production.ProposalGenerator(det, device=cuda, confidence_threshold=0.5).generate_proposals_and_images(img)
        slow = production.ProposalGenerator(Plain(det), device=cuda, confidence_threshold=0.5).generate_proposals_and_images(img)
        assert torch.equal(fast[0], slow[0]) and torch.equal(fast[1], slow[1])
        assert len(fast[0]) > 0


def test_conditioned_detector_default_mode_tight(cuda):
    """The WHOLE detector in the product default (fp16 storage) on better-conditioned weights (`residual_gain` 0.25: the damped
    seeded init, the base of the fitted-head fixture) against the fp32 oracle, with TIGHT bounds -- the random-weight bounds of
    `test_config1_plumbing_and_parity` (Gaussian map l2rel < 0.25, 80 % of boxes) are the floor of an amplifying random ResNet, not
    of the kernels (proposals.py:162-181).  Measured on MI355X: Gaussian l2rel 4e-3, FPN features l2rel <= 2e-3, 99-100 % of boxes."""
    from cvpce_amd import synthetic
    from oracle import gln as og
    m = synthetic.synthetic_gln(seed=0, detections_per_img=200, residual_gain=0.25)
    assert m.precision == 'fp16'
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.to(cuda)
    products = synthetic.product_images(64, seed=200)
    imgs = [synthetic.structured_shelf(7, 800, 800, products)[0], torch.rand(3, 640, 640, generator=torch.Generator().manual_seed(3))]
    eng = m.engine()
    for img in imgs:
        out, inter = eng.detect([img.to(cuda)], 1, 200, 0.5, want_intermediates=True)
        boxes, scores, labels, count, conf, gauss = out
        ref, rint = og.gln_forward([img], sd, detections_per_img=200, return_intermediates=True)
        for got, want in zip(inter['features'], rint['features']):
            assert l2rel(nchw(got), want) < 5e-3, l2rel(nchw(got), want)
        for got, want in zip(inter['cls'], rint['cls']):
            assert (got.view(1, -1).cpu() - want.view(1, -1)).abs().max() < 0.02 * want.std() + 5e-3
        assert l2rel(gauss.cpu(), rint['gaussians']) <= 0.03, l2rel(gauss.cpu(), rint['gaussians'])
        c = int(count[0])
        _compare_detections({'boxes': boxes[0, :c], 'scores': scores[0, :c]}, ref[0], min_frac=0.97)


def test_distance_on_device(cuda, golden_dir):
    """E1 `distance` (classification.py:87-88: 1 - cosine_similarity, eps 1e-8) on CUDA tensors: the reference's own KAT geometry and the
    reference-made distance matrices of tests/golden/nearest.pt (`distance` over the broadcast (Q,A,D) pair the reference builds)."""
    from cvpce_amd.models import classification as C
    a = torch.tensor([[1.0, 0.0], [0.0, 2.0], [-3.0, 0.0], [1.0, 1.0]], device=cuda)
    b = torch.tensor([[2.0, 0.0], [0.0, 1.0], [1.0, 0.0], [1.0, 0.0]], device=cuda)
    d = C.distance(a, b)
    torch.testing.assert_close(d.cpu(), torch.tensor([0.0, 0.0, 2.0, 1.0 - 2 ** -0.5]), rtol=0, atol=1e-6)
    assert C.distance(torch.zeros(1, 4, device=cuda), torch.ones(1, 4, device=cuda)).item() == 1.0       # eps clamp: no NaN on a zero row
    gold = torch.load(os.path.join(golden_dir, 'nearest.pt'), weights_only=False)
    for case in gold['cases'][:4]:
        A, Q = case['anchors'].to(cuda), case['queries'].to(cuda)
        q, n, dim = len(Q), len(A), A.shape[1]
        got = C.distance(A[None, :, :].expand(q, n, dim), Q[:, None, :].expand(q, n, dim), dim=-1)
        assert got.shape == (q, n) and got.dtype == torch.float32
        torch.testing.assert_close(got.cpu(), case['distances'], rtol=0, atol=2e-6)
