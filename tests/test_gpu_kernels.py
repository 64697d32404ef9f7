"""Kernel-level parity: each HIP entry point of include/cvpce_amd.h (called through
cvpce_amd.ops -> ctypes -> C ABI) against the CPU oracle on the same seeded inputs.

Tolerances: convolutions compute in bf16 x bf16 -> fp32 accumulate; the oracle is fed the SAME
bf16-rounded operands, so the only differences are fp32 summation order (+ one bf16 rounding of
the output when the output is bf16).  Index outputs (top-k, NMS, nearest neighbour) are exact.
"""
import math
import os
import zlib

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


def r16(t):
    return t.to(BF).to(torch.float32)


def nhwc(x):
    """(N,C,H,W) f32 -> NHWC bf16 with C padded to 8"""
    n, c, h, w = x.shape
    cp = (c + 7) // 8 * 8
    out = torch.zeros(n, h, w, cp, dtype=BF)
    out[..., :c] = x.permute(0, 2, 3, 1).to(BF)
    return out


def nchw(y):
    return y.float().permute(0, 3, 1, 2).cpu()


def rel_err(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp(min=1e-6)).item()


CONV_CASES = [
    # name, N, Cin, H, W, Cout, k, stride, pad, opts
    ('vgg_first', 2, 3, 40, 36, 64, 3, 1, 1, {}),
    ('stem7x7', 2, 3, 67, 45, 64, 7, 2, 3, {}),
    ('1x1', 2, 64, 25, 31, 256, 1, 1, 0, {}),
    ('3x3_s1', 1, 128, 33, 29, 128, 3, 1, 1, {}),
    ('3x3_s2', 2, 64, 26, 30, 64, 3, 2, 1, {}),
    ('1x1_s2_ds', 1, 256, 20, 20, 512, 1, 2, 0, {'act': 0}),
    ('cin32', 1, 32, 30, 30, 32, 3, 1, 1, {}),
    ('cin16_1x1', 1, 16, 30, 31, 16, 1, 1, 0, {}),
    ('residual', 2, 64, 18, 22, 256, 1, 1, 0, {'res': 'same'}),
    ('res_upsample', 2, 64, 20, 24, 256, 1, 1, 0, {'res': 'up', 'act': 0}),
    ('in_upsample', 1, 64, 12, 14, 32, 3, 1, 1, {'in_up': 1}),
    ('cls_f32', 2, 256, 13, 13, 9, 3, 1, 1, {'f32': True, 'act': 0}),
    ('reg_f32', 1, 256, 7, 7, 36, 3, 1, 1, {'f32': True, 'act': 0}),
    ('gauss_out', 1, 16, 40, 40, 1, 1, 1, 0, {'f32': True, 'act': 1}),
    ('gauss_tanh', 1, 16, 40, 40, 1, 1, 1, 0, {'f32': True, 'act': 2}),
    ('deep_k', 1, 512, 9, 9, 512, 3, 1, 1, {}),
    ('ragged_m', 3, 64, 5, 3, 192, 3, 1, 1, {}),
    # big-M shapes: routed to the LDS-DMA kernel (256x256 / 128x256 / 64x256 tiles)
    ('dma256', 2, 64, 128, 128, 256, 3, 1, 1, {}),
    ('dma512_deep', 1, 256, 128, 128, 512, 3, 1, 1, {}),
    ('dma128', 2, 64, 128, 128, 128, 3, 1, 1, {}),
    ('dma64', 2, 64, 128, 128, 64, 3, 1, 1, {}),
    ('dma_1x1_res', 2, 256, 128, 128, 256, 1, 1, 0, {'res': 'same'}),
    ('dma_s2', 2, 128, 256, 256, 256, 3, 2, 1, {}),
    ('dma_ragged', 1, 64, 181, 182, 192, 3, 1, 1, {}),
    ('dma_res_up', 2, 64, 128, 128, 256, 1, 1, 0, {'res': 'up', 'act': 0}),
    # conv + ReLU + MaxPool2d(2,2) fused
    ('pool_dma256', 2, 64, 128, 128, 256, 3, 1, 1, {'pool': True}),
    ('pool_dma64', 2, 64, 128, 128, 64, 3, 1, 1, {'pool': True}),
    ('pool_generic', 3, 32, 20, 24, 64, 3, 1, 1, {'pool': True}),
    ('pool_generic_first', 2, 3, 36, 40, 64, 3, 1, 1, {'pool': True}),
]


@pytest.mark.parametrize('case', CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv2d_parity(cuda, case):
    from cvpce_amd import ops
    name, n, cin, h, w, cout, k, stride, pad, o = case
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 1000)   # stable across processes (str hashes are salted)
    x = torch.randn(n, cin, h, w, generator=g)
    wgt = torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)
    bias = torch.randn(cout, generator=g) * 0.1
    act = o.get('act', 1)
    in_up = o.get('in_up', 0)
    pc = ops.PackedConv(wgt, bias, stride, pad, device=cuda)
    xin = r16(x)
    xl = F.interpolate(xin, scale_factor=2.0, mode='nearest') if in_up else xin
    ref = F.conv2d(xl, r16(wgt), bias, stride=stride, padding=pad)
    res_dev = None
    if o.get('res') == 'same':
        res = r16(torch.randn(ref.shape, generator=g))
        ref = ref + res
        res_dev = nhwc(res).to(cuda)
    elif o.get('res') == 'up':
        res = r16(torch.randn(n, cout, ref.shape[2] // 2, ref.shape[3] // 2, generator=g))
        ref = ref + F.interpolate(res, size=ref.shape[-2:], mode='nearest')
        res_dev = nhwc(res).to(cuda)
    if act == 1:
        ref = F.relu(ref)
    elif act == 2:
        ref = torch.tanh(ref)
    if o.get('pool'):
        ref = F.max_pool2d(ref, 2, 2)
    ring_case = name.startswith('dma') or name.startswith('pool_dma')
    ops.USE_HALO_3X3 = not ring_case          # these cases pin the LDS-ring implicit-GEMM kernels (the halo kernels have their own test)
    ops.USE_CONV1X1 = not ring_case
    try:
        y = ops.conv2d(nhwc(xin).to(cuda), pc, act=act, out_f32=o.get('f32', False), residual=res_dev, in_up_shift=in_up,
                       pool=o.get('pool', False))
    finally:
        ops.USE_HALO_3X3 = True
        ops.USE_CONV1X1 = True
    torch.cuda.synchronize()
    got = nchw(y)
    if name.startswith('dma') or name.startswith('pool_dma'):
        # the LDS-DMA kernel and the register-staged fallback accumulate every output element in the same
        # K order -> bit-identical results (an A/B of the two code paths on the same inputs)
        assert ops.ConvProfile.variant(pc, ref.shape[0] * (y.shape[1] * (2 if o.get('pool') else 1)) * (y.shape[2] * (2 if o.get('pool') else 1))).startswith('conv_dma')
        ops.FORCE_GENERIC_CONV = True
        try:
            y2 = ops.conv2d(nhwc(xin).to(cuda), pc, act=act, out_f32=o.get('f32', False), residual=res_dev,
                            in_up_shift=in_up, pool=o.get('pool', False))
        finally:
            ops.FORCE_GENERIC_CONV = False
        assert torch.equal(y2, y)
    assert got.shape == ref.shape
    tol = 2e-4 if o.get('f32') else 1e-2   # bf16 output rounding = 2^-8 relative
    assert rel_err(got, ref) < tol, (name, rel_err(got, ref))


def test_conv2d_rejects_bad_args(cuda):
    from cvpce_amd import ops
    pc = ops.PackedConv(torch.randn(64, 64, 3, 3), None, 1, 1, device=cuda)
    with pytest.raises(RuntimeError):
        ops.conv2d(torch.zeros(1, 8, 8, 64, dtype=BF), pc)  # CPU tensor: loud, no fallback
    pc9 = ops.PackedConv(torch.randn(9, 64, 3, 3), None, 1, 1, device=cuda)
    with pytest.raises(RuntimeError):
        ops.conv2d(torch.zeros(1, 8, 8, 64, dtype=BF, device=cuda), pc9, out_f32=False)  # Cout % 4 != 0 needs f32 out


def test_conv2d_empty_batch(cuda):
    from cvpce_amd import ops
    pc = ops.PackedConv(torch.randn(64, 64, 3, 3), None, 1, 1, device=cuda)
    y = ops.conv2d(torch.zeros(0, 8, 8, 64, dtype=BF, device=cuda), pc)
    assert y.shape == (0, 8, 8, 64)


@pytest.mark.parametrize('k,stride,pad,h,w', [(2, 2, 0, 16, 20), (3, 2, 1, 17, 23), (2, 2, 0, 7, 9)])
def test_maxpool_parity(cuda, k, stride, pad, h, w):
    from cvpce_amd import ops
    x = r16(torch.randn(2, 24, h, w, generator=torch.Generator().manual_seed(3)))
    ref = F.max_pool2d(x, k, stride, pad)
    got = nchw(ops.maxpool2d(nhwc(x).to(cuda), k, stride, pad))
    assert torch.equal(got, ref)


def test_relu_globalmax_l2norm(cuda):
    from cvpce_amd import ops
    g = torch.Generator().manual_seed(4)
    x = r16(torch.randn(3, 72, 9, 11, generator=g))
    assert torch.equal(nchw(ops.relu(nhwc(x).to(cuda))), F.relu(x))
    desc = torch.full((3, 100), -7.0, device=cuda)
    ops.global_max_into(nhwc(x).to(cuda), desc, 20)
    assert torch.equal(desc[:, 20:92].cpu(), x.amax(dim=(-2, -1)))
    assert (desc[:, :20] == -7).all() and (desc[:, 92:] == -7).all()
    d = torch.rand(5, 1024, generator=g)
    d[3] = 0
    out, out_bf = ops.l2_normalize(d.to(cuda), 1e-8, want_bf16=True)
    ref = d / torch.linalg.norm(d, dim=1, keepdim=True).clamp(min=1e-8)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-5, atol=1e-7)
    assert torch.equal(out_bf.cpu(), out.cpu().to(BF))


@pytest.mark.parametrize('h0,w0', [(640, 640), (2048, 2048), (300, 517), (1200, 700)])
def test_transform_parity(cuda, h0, w0):
    from cvpce_amd import ops
    from cvpce_amd.models import proposals as P
    from oracle import gln as og
    img = torch.rand(3, h0, w0, generator=torch.Generator().manual_seed(h0))
    ref = og.transform_one(img)
    h, w = P.resized_hw(h0, w0)
    assert (h, w) == tuple(ref.shape[-2:]) == og.resized_size(h0, w0)
    hp, wp = (h + 31) // 32 * 32, (w + 31) // 32 * 32
    batch = torch.full((1, hp, wp, 8), 9.0, dtype=BF, device=cuda)
    ops.gln_transform_into(img.to(cuda), batch, 0, h, w, P.IMAGE_MEAN, P.IMAGE_STD)
    got = batch[0].float().cpu()
    assert (got[..., 3:] == 0).all() and (got[h:] == 0).all() and (got[:, w:] == 0).all()
    # fp32 interpolation identical up to rounding; compare after the same bf16 rounding
    diff = (got[:h, :w, :3].permute(2, 0, 1) - ref).abs().max().item()
    assert diff < 2e-2, diff   # bf16 ulp at |x| <= 2.7 is 1.6e-2
    frac_exact = (got[:h, :w, :3].permute(2, 0, 1) == r16(ref)).float().mean().item()
    assert frac_exact > 0.99, frac_exact


def test_crop_resize_parity(cuda):
    from cvpce_amd import ops
    from cvpce_amd.models.classification import TANH_MEAN, TANH_STD
    from oracle import crop as ocrop, macvgg as ovgg
    img = torch.rand(3, 300, 400, generator=torch.Generator().manual_seed(9))
    boxes = torch.tensor([[10.7, 20.2, 110.9, 80.5],     # wide
                          [0.0, 0.0, 400.0, 300.0],      # whole image
                          [395.2, 5.0, 400.0, 299.9],    # thin, tall
                          [50.0, 60.0, 51.9, 61.2],      # 1x1 pixel
                          [100.3, 100.3, 356.3, 356.3],  # clipped by the image bottom (y2 > H)
                          [7.0, 9.0, 263.0, 265.0]])     # exactly 256x256 (identity resize)
    ref = ocrop.crop_boxes(img, boxes)
    got = ops.crop_resize(img.to(cuda), boxes.to(cuda), 256, mode=0).cpu()
    torch.testing.assert_close(got, ref, rtol=0, atol=2e-6)
    packed = ops.crop_resize(img.to(cuda), boxes.to(cuda), 256, mode=1, mean=TANH_MEAN, std=TANH_STD).float().cpu()
    refp = ovgg.normalize_tanh(ocrop.scale_to_tanh(ref)).permute(0, 2, 3, 1)
    assert (packed[..., 3:] == 0).all()
    assert (packed[..., :3] - refp).abs().max() < 2e-2
    # the API-level pack kernel must agree bit-for-bit with the fused crop path
    packed2 = ops.pack_embed_input(got.to(cuda), True, TANH_MEAN, TANH_STD).float().cpu()
    assert torch.equal(packed2, packed)
    # device-side count: rows >= count are not touched
    out = torch.full((6, 256, 256, 8), 5.0, dtype=BF, device=cuda)
    cnt = torch.tensor([2], dtype=torch.int32, device=cuda)
    ops.crop_resize(img.to(cuda), boxes.to(cuda), 256, mode=1, mean=TANH_MEAN, std=TANH_STD, count=cnt, out=out)
    assert (out[2:] == 5).all() and torch.equal(out[:2].float().cpu(), packed[:2])


def _random_head_outputs(n, grids, a, k, seed, spread=2.0, bias=-1.0, quantum=0.0):
    g = torch.Generator().manual_seed(seed)
    cls = [torch.randn(n, gh * gw * a * k, generator=g) * spread + bias for gh, gw in grids]
    if quantum:                        # logits on a coarse grid: thousands of exact ties at the top-k cut and in the NMS order
        cls = [torch.round(c / quantum) * quantum for c in cls]
    reg = [torch.randn(n, gh * gw * a, 4, generator=g) * 0.5 for gh, gw in grids]
    return cls, reg


@pytest.mark.parametrize('dpi,seed,bias,k,quantum', [(1000, 0, -1.0, 1, 0), (200, 1, 1.0, 1, 0), (300, 2, -4.5, 1, 0), (300, 3, -1.0, 3, 0),
                                                     (1000, 4, 0.5, 2, 0), (300, 5, 0.0, 1, 0.5), (1000, 6, -3.0, 1, 0.125),
                                                     (300, 7, -1.0, 3, 0.25)])
def test_detect_postprocess_parity(cuda, dpi, seed, bias, k, quantum, monkeypatch):
    """K6-K8 against the oracle on identical fp32 logits: kept sets and order identical; k > 1 classes exercise the
    per-class offsets of batched_nms and the anchor / label split of the flat candidate index (and, at k = 3, a P3 level of
    280 800 logits = 35 chunks: the one-kernel decode); quantum > 0 puts the logits on a grid, so that the top-k cut of a
    level and of its chunks falls inside a run of equal logits (lowest index first) and the score threshold on exact values.
    Every case also runs through the one-kernel decode: identical outputs."""
    from cvpce_amd import ops
    from cvpce_amd.models import proposals as P
    from oracle import gln as og
    n = 2
    padded = (800, 832)
    grids = [(100, 104), (50, 52), (25, 26), (13, 13), (7, 7)]
    cls, reg = _random_head_outputs(n, grids, 9, k, seed, bias=bias, quantum=quantum)
    resized = [(800, 810), (790, 832)]
    original = [(2048, 2073), (1000, 1053)]
    anchors = og.grid_anchors(padded, grids)
    strides = [(padded[0] // gh, padded[1] // gw) for gh, gw in grids]
    base = P._base_anchors()
    assert torch.equal(base, torch.stack([og.base_anchors(s) for s in og.ANCHOR_SIZES]))
    image_hw = torch.tensor(resized, dtype=torch.int32)
    ratios = torch.stack([torch.tensor(o, dtype=torch.float32) / torch.tensor(r, dtype=torch.float32)
                          for o, r in zip(original, resized)])
    boxes, scores, labels, count, conf = ops.detect_postprocess(
        [c.to(cuda) for c in cls], [r.to(cuda) for r in reg], grids, strides, base.to(cuda), image_hw.to(cuda),
        ratios.to(cuda), 9, k, og.TOPK_CANDIDATES, og.SCORE_THRESH, og.NMS_THRESH, og.BBOX_XFORM_CLIP, dpi, 0.5)
    torch.cuda.synchronize()
    for i in range(n):
        # oracle with the documented tie refinement: order by logit (monotone in score), then index
        b, s, l = og.postprocess_image([c[i].view(-1, k) for c in cls], [r[i] for r in reg], anchors, resized[i], dpi)
        b = og.resize_boxes(b, resized[i], original[i])
        c = int(count[i])
        assert c == len(b), (c, len(b))
        torch.testing.assert_close(scores[i, :c].cpu(), s, rtol=0, atol=1e-6)
        torch.testing.assert_close(boxes[i, :c].cpu(), b, rtol=1e-5, atol=2e-3)
        assert torch.equal(labels[i, :c].cpu(), l)
        assert k == 1 or len(set(l.tolist())) == k
        assert int(conf[i]) == int((s > 0.5).sum())
        assert (scores[i, :c - 1] >= scores[i, 1:c]).all()
    monkeypatch.setenv('CVPCE_DECODE_ONE_KERNEL', '1')
    one = ops.detect_postprocess(
        [c.to(cuda) for c in cls], [r.to(cuda) for r in reg], grids, strides, base.to(cuda), image_hw.to(cuda),
        ratios.to(cuda), 9, k, og.TOPK_CANDIDATES, og.SCORE_THRESH, og.NMS_THRESH, og.BBOX_XFORM_CLIP, dpi, 0.5)
    for x, y in zip(one, (boxes, scores, labels, count, conf)):
        assert torch.equal(x, y)


def test_detect_score_threshold_band(cuda, monkeypatch):
    """The chunked decode decides sigmoid(l) > thresh from the logit alone outside a narrow band around logit(thresh) and
    evaluates the sigmoid inside it; the one-kernel decode evaluates it for every logit.  Logits packed at 1-ulp and at
    1e-4 steps around logit(thresh), for several thresholds (incl. one outside the range where the shortcut is enabled):
    the two forms keep exactly the same candidates."""
    from cvpce_amd import ops
    from cvpce_amd.models import proposals as P
    from oracle import gln as og
    grids = [(10, 10)]
    for thresh in (0.05, 0.3, 0.5, 0.9, 0.995, 1e-7):
        t = math.log(thresh / (1 - thresh))
        t32 = torch.tensor(t, dtype=torch.float32)
        fine = t32 + torch.arange(-225, 225, dtype=torch.float32) * torch.finfo(torch.float32).eps * max(abs(t), 1e-3)
        coarse = t32 + torch.arange(-225, 225, dtype=torch.float32) * 1e-4 * (1 + abs(t))
        lg = torch.cat([fine, coarse])[torch.randperm(900, generator=torch.Generator().manual_seed(1))].view(1, 900)
        reg = torch.zeros(1, 900, 4)
        args = ([lg.to(cuda)], [reg.to(cuda)], grids, [(8, 8)], P._base_anchors()[:1].contiguous().to(cuda),
                torch.tensor([[80, 80]], dtype=torch.int32).to(cuda), torch.ones(1, 2).to(cuda), 9, 1, 1000, thresh, 2.0,
                og.BBOX_XFORM_CLIP, 1000, 0.5)          # nms_thresh 2: nothing is suppressed, every candidate comes out
        monkeypatch.delenv('CVPCE_DECODE_ONE_KERNEL', raising=False)
        got = ops.detect_postprocess(*args)
        monkeypatch.setenv('CVPCE_DECODE_ONE_KERNEL', '1')
        one = ops.detect_postprocess(*args)
        monkeypatch.delenv('CVPCE_DECODE_ONE_KERNEL')
        assert 200 < int(got[3][0]) < 700, (thresh, int(got[3][0]))
        for x, y in zip(got, one):
            assert torch.equal(x, y), thresh
        ref = int((torch.sigmoid(lg) > thresh).sum())
        assert abs(int(got[3][0]) - ref) <= 8, (thresh, int(got[3][0]), ref)   # (torch's CPU sigmoid may differ by an ulp at the cut)


def test_detect_nms_second_phase(cuda, monkeypatch):
    """NMS runs on the first 8 x detections_per_img candidates first and on all of them only for the images that did not fill
    up there.  A low IoU threshold suppresses so much that image 0 needs its remaining candidates (fewer than dpi boxes are
    kept out of > limit candidates), while image 1 -- few candidates -- is final after phase 1: both equal the oracle and the
    one-phase run."""
    from cvpce_amd import ops
    from cvpce_amd.models import proposals as P
    from oracle import gln as og
    n, dpi, nms_thresh = 2, 150, 0.02
    padded = (800, 800)
    grids = [(100, 100), (50, 50), (25, 25), (13, 13), (7, 7)]
    cls, reg = _random_head_outputs(n, grids, 9, 1, 21, bias=0.0)
    for c, r in zip(cls, reg):
        c[1] -= 9.0                                         # image 1: about a hundred candidates in all
        r[0, :, 2:] += 3.0                                  # image 0: boxes 20 x their anchors -- nearly all overlap
    resized = [(800, 800), (800, 800)]
    anchors = og.grid_anchors(padded, grids)
    strides = [(padded[0] // gh, padded[1] // gw) for gh, gw in grids]
    base = P._base_anchors()
    args = ([c.to(cuda) for c in cls], [r.to(cuda) for r in reg], grids, strides, base.to(cuda),
            torch.tensor(resized, dtype=torch.int32).to(cuda), torch.ones(n, 2).to(cuda), 9, 1, og.TOPK_CANDIDATES, og.SCORE_THRESH,
            nms_thresh, og.BBOX_XFORM_CLIP, dpi, 0.5)
    got = ops.detect_postprocess(*args)
    monkeypatch.setenv('CVPCE_NMS_ONE_PHASE', '1')
    one = ops.detect_postprocess(*args)
    for x, y in zip(got, one):
        assert torch.equal(x, y)
    boxes, scores, labels, count, conf = got
    ncand = [sum(min(1000, int((torch.sigmoid(c[i]) > og.SCORE_THRESH).sum())) for c in cls) for i in range(n)]
    assert ncand[0] > 8 * dpi and int(count[0]) < dpi, (ncand, count.tolist())       # image 0 cannot be decided in phase 1
    assert ncand[1] <= 8 * dpi and 0 < int(count[1]) < dpi, (ncand, count.tolist())
    for i in range(n):
        b, s, l = og.postprocess_image([c[i].view(-1, 1) for c in cls], [r[i] for r in reg], anchors, resized[i], dpi, nms_thresh=nms_thresh)
        c = int(count[i])
        assert c == len(b), (c, len(b))
        torch.testing.assert_close(scores[i, :c].cpu(), s, rtol=0, atol=1e-6)
        torch.testing.assert_close(boxes[i, :c].cpu(), b, rtol=1e-5, atol=2e-3)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('act', [1, 2])
def test_gauss_tail_parity(cuda, dtype, act):
    """The Gaussian subnet's two 1x1 layers in one launch (proposals.py:96-107) against (a) the same two layers in fp32 on the same
    16-bit operands with the hidden layer rounded where the two-launch form stores it, (b) the two conv launches themselves."""
    from cvpce_amd import ops
    g = torch.Generator().manual_seed(17)
    n, h, w = 2, 37, 41
    x = torch.randn(n, 16, h, w, generator=g).relu()
    w4, b4 = torch.randn(16, 16, 1, 1, generator=g) / 4, torch.randn(16, generator=g) * 0.1
    w5, b5 = torch.randn(1, 16, 1, 1, generator=g) / 4, torch.randn(1, generator=g) * 0.1
    c4 = ops.PackedConv(w4, b4, 1, 0, device=cuda, dtype=dtype)
    c5 = ops.PackedConv(w5, b5, 1, 0, device=cuda, dtype=dtype)
    rd = lambda t: t.to(dtype).to(torch.float32)
    xin = x.permute(0, 2, 3, 1).contiguous().to(dtype).to(cuda)
    assert ops.can_fuse_gauss_tail(xin, c4, c5)
    got = ops.gauss_tail(xin, c4, c5, act).cpu()
    hid = rd(F.relu(F.conv2d(rd(x), rd(w4), b4)))
    ref = F.conv2d(hid, rd(w5), b5)
    ref = torch.tanh(ref) if act == 2 else F.relu(ref)
    assert got.shape == (n, h, w, 1)
    # (a hidden value that lands on a rounding boundary of the 16-bit type may round the other way under another summation order)
    torch.testing.assert_close(got[..., 0], ref[:, 0], rtol=0, atol=2e-2 if dtype == torch.bfloat16 else 3e-3)
    assert (got[..., 0] - ref[:, 0]).abs().mean() < (2e-4 if dtype == torch.bfloat16 else 3e-5)
    two = ops.conv2d(ops.conv2d(xin, c4, act=1), c5, act=act, out_f32=True).cpu()
    torch.testing.assert_close(got, two, rtol=0, atol=2e-2 if dtype == torch.bfloat16 else 3e-3)
    assert (got - two).abs().mean() < (2e-4 if dtype == torch.bfloat16 else 3e-5)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('n,h,w,cout,act,up', [(2, 37, 45, 32, 1, 0), (1, 103, 16, 16, 1, 0), (3, 7, 70, 32, 0, 0), (1, 51, 33, 16, 0, 0),
                                                  (2, 19, 23, 32, 1, 1), (1, 52, 8, 32, 1, 1), (1, 3, 40, 32, 0, 1)])
def test_conv3x3_thin_parity(cuda, dtype, n, h, w, cout, act, up):
    """The thin 3x3 kernel (Cin 32 -> Cout 16 | 32: the Gaussian subnet's layers, proposals.py:81-107) against F.conv2d on the same
    16-bit operands and against the implicit-GEMM kernel: ragged strips (W % 16 != 0), more / fewer rows than a 50-row task, the
    image border rows and columns (zero padding through the buffer range check)."""
    from cvpce_amd import ops
    g = torch.Generator().manual_seed(100 * h + w + cout)
    rd = lambda t: t.to(dtype).to(torch.float32)
    cin = 64 if up else 32          # up: (h, w) is the STORED size, the conv runs over its nearest-2x upsample (proposals.py:79)
    x = rd(torch.randn(n, cin, h, w, generator=g))
    wgt = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin)
    bias = torch.randn(cout, generator=g) * 0.1
    pc = ops.PackedConv(wgt, bias, 1, 1, device=cuda, dtype=dtype)
    ref = F.conv2d(F.interpolate(x, scale_factor=2.0, mode='nearest') if up else x, rd(wgt), bias, padding=1)
    if act:
        ref = F.relu(ref)
    xin = x.permute(0, 2, 3, 1).contiguous().to(dtype).to(cuda)
    ops.PROFILE = ops.ConvProfile()
    try:
        y = ops.conv2d(xin, pc, act=act, in_up_shift=up)
        assert ops.PROFILE.records[-1][0] == 'thin3x3_kernel'
        ops.USE_THIN_3X3 = False
        y2 = ops.conv2d(xin, pc, act=act, in_up_shift=up)
        assert ops.PROFILE.records[-1][0] != 'thin3x3_kernel'
    finally:
        ops.USE_THIN_3X3 = True
        ops.PROFILE = None
    got = y.float().permute(0, 3, 1, 2).cpu()
    assert got.shape == ref.shape
    assert rel_err(got, ref) < (1e-2 if dtype == torch.bfloat16 else 2e-3), rel_err(got, ref)
    assert (y.float() - y2.float()).abs().max() <= (2 ** -7 if dtype == torch.bfloat16 else 2 ** -10) * y2.float().abs().max()


def _subnet_layers(g, dtype, cuda):
    from cvpce_amd import ops
    shapes = [(32, 64, 3), (32, 32, 3), (16, 32, 3), (16, 16, 1), (1, 16, 1)]
    ws = [torch.randn(co, ci, k, k, generator=g) * math.sqrt(2.0 / (k * k * ci)) for co, ci, k in shapes]
    bs = [torch.randn(co, generator=g) * 0.1 for co, _, _ in shapes]
    convs = [ops.PackedConv(w, b, 1, 1 if w.shape[-1] == 3 else 0, device=cuda, dtype=dtype) for w, b in zip(ws, bs)]
    return ws, bs, convs


def _subnet_reference(x, ws, bs, act, rd):
    """GaussianSubnet (proposals.py:81-107) over up2(x) in fp32 on the 16-bit operands, every layer rounded where the kernels store it."""
    t = F.interpolate(x, scale_factor=2.0, mode='nearest')
    for i in range(3):
        t = rd(F.relu(F.conv2d(t, rd(ws[i]), bs[i], padding=1)))
    t = rd(F.relu(F.conv2d(t, rd(ws[3]), bs[3])))
    z = F.conv2d(t, rd(ws[4]), bs[4])
    return torch.tanh(z) if act == 2 else (F.relu(z) if act == 1 else z)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('n,hs,ws_,act', [(2, 19, 23, 1), (1, 100, 14, 2), (1, 7, 29, 1), (3, 40, 40, 2), (1, 1, 1, 1), (1, 3, 57, 0), (2, 64, 15, 1)])
def test_gauss_subnet_parity(cuda, dtype, n, hs, ws_, act):
    """The whole Gaussian subnet in one launch (csrc/gauss_subnet.hip; proposals.py:81-107) against (a) the five layers in fp32 on the same
    16-bit operands with every layer rounded where the per-layer kernels store it, (b) the per-layer kernels themselves (three thin 3x3
    launches + the pointwise tail).  Shapes: one strip exactly (W = 28), ragged last strips, maps narrower than a strip, more / fewer rows
    than a task, a 2 x 2 output (every tap but the centre is zero padding), the image borders of every layer."""
    from cvpce_amd import ops
    g = torch.Generator().manual_seed(1000 * hs + ws_ + act)
    rd = lambda t: t.to(dtype).to(torch.float32)
    wts, bs, convs = _subnet_layers(g, dtype, cuda)
    x = rd(torch.randn(n, 64, hs, ws_, generator=g).relu())
    xin = x.permute(0, 2, 3, 1).contiguous().to(dtype).to(cuda)
    assert ops.can_fuse_gauss_subnet(xin, convs)
    got = ops.gauss_subnet(xin, convs, act).cpu()
    assert got.shape == (n, 2 * hs, 2 * ws_, 1) and got.dtype == torch.float32
    ref = _subnet_reference(x, wts, bs, act, rd)[:, 0]
    # (a value that lands on a rounding boundary of the 16-bit type may round the other way under another summation order, and the flip
    #  is carried through the layers behind it: max error a few ulps of the storage type, mean error far below one)
    atol, mtol = (3e-2, 1.5e-3) if dtype == torch.bfloat16 else (4e-3, 2e-4)
    scale = max(1.0, float(ref.abs().max()))
    assert (got[..., 0] - ref).abs().max() <= atol * scale, ((got[..., 0] - ref).abs().max(), scale)
    assert (got[..., 0] - ref).abs().mean() <= mtol * scale, ((got[..., 0] - ref).abs().mean(), scale)
    # (b) the per-layer launches
    t = xin
    t = ops.conv2d(t, convs[0], act=1, in_up_shift=1)
    t = ops.conv2d(t, convs[1], act=1)
    t = ops.conv2d(t, convs[2], act=1)
    per_layer = ops.gauss_tail(t, convs[3], convs[4], act).cpu()
    assert (got - per_layer).abs().max() <= atol * scale and (got - per_layer).abs().mean() <= mtol * scale
    assert torch.equal(ops.gauss_subnet(xin, convs, act).cpu(), got)                        # deterministic


def test_gauss_subnet_full_size_and_batch_independence(cuda):
    """At the detector's size (200 x 200 stored -> 400 x 400 output, BASELINE configs[1]): an image's map is bit-identical whatever batch
    it is computed in (the row split over the wave slots depends on N; the arithmetic of a pixel does not), and agrees with the per-layer
    kernels; also a portrait map (800 x 1088 input -> 400 x 544)."""
    from cvpce_amd import ops
    g = torch.Generator().manual_seed(5)
    wts, bs, convs = _subnet_layers(g, torch.float16, cuda)
    for hs, ws_ in ((200, 200), (272, 200)):
        x = torch.randn(8, hs, ws_, 64, generator=g).relu().to(torch.float16).to(cuda)
        all8 = ops.gauss_subnet(x, convs, 2)
        for nb in (1, 3, 4):
            part = ops.gauss_subnet(x[:nb].contiguous(), convs, 2)
            assert torch.equal(part, all8[:nb]), nb
        t = ops.conv2d(x[:2].contiguous(), convs[0], act=1, in_up_shift=1)
        t = ops.conv2d(ops.conv2d(t, convs[1], act=1), convs[2], act=1)
        per_layer = ops.gauss_tail(t, convs[3], convs[4], 2)
        assert (all8[:2] - per_layer).abs().max() < 4e-3 and (all8[:2] - per_layer).abs().mean() < 1e-4


def test_detect_postprocess_no_candidates(cuda):
    from cvpce_amd import ops
    from cvpce_amd.models import proposals as P
    from oracle import gln as og
    grids = [(4, 4), (2, 2)]
    cls = [torch.full((1, gh * gw * 9), -10.0) for gh, gw in grids]
    reg = [torch.zeros(1, gh * gw * 9, 4) for gh, gw in grids]
    out = ops.detect_postprocess([c.to(cuda) for c in cls], [r.to(cuda) for r in reg], grids, [(8, 8), (16, 16)],
                                 P._base_anchors()[:2].contiguous().to(cuda), torch.tensor([[32, 32]], dtype=torch.int32).to(cuda),
                                 torch.ones(1, 2).to(cuda), 9, 1, 1000, 0.05, 0.5, og.BBOX_XFORM_CLIP, 100, 0.5)
    assert int(out[3][0]) == 0 and int(out[4][0]) == 0


def test_nms_properties_full_size(cuda):
    """Full-size (800x800, 120 087 anchors x 4 images) run checked through size-independent properties."""
    from cvpce_amd import ops
    from cvpce_amd.models import proposals as P
    from oracle import gln as og
    n = 4
    grids = [(100, 100), (50, 50), (25, 25), (13, 13), (7, 7)]
    cls, reg = _random_head_outputs(n, grids, 9, 1, 11, bias=1.0)
    strides = [(800 // gh, 800 // gw) for gh, gw in grids]
    image_hw = torch.tensor([[800, 800]] * n, dtype=torch.int32)
    ratios = torch.full((n, 2), 2.56)
    boxes, scores, labels, count, conf = ops.detect_postprocess(
        [c.to(cuda) for c in cls], [r.to(cuda) for r in reg], grids, strides, P._base_anchors().to(cuda),
        image_hw.to(cuda), ratios.to(cuda), 9, 1, 1000, 0.05, 0.5, og.BBOX_XFORM_CLIP, 1000, 0.5)
    for i in range(n):
        c = int(count[i])
        assert 0 < c <= 1000
        b, s = boxes[i, :c].cpu(), scores[i, :c].cpu()
        assert (s[:-1] >= s[1:]).all() and (s > 0.05).all()
        assert (b[:, 0] >= 0).all() and (b[:, 2] <= 2048.0 + 1e-3).all() and (b[:, 2] >= b[:, 0]).all()
        area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
        lt = torch.max(b[:, None, :2], b[:, :2]); rb = torch.min(b[:, None, 2:], b[:, 2:])
        inter = (rb - lt).clamp(min=0).prod(dim=2)
        iou = torch.nan_to_num(inter / (area[:, None] + area - inter), nan=0.0)   # 0/0 for clipped-to-empty boxes
        iou.fill_diagonal_(0)
        assert iou.max() <= 0.5 + 1e-5, iou.max()   # no kept pair overlaps more than the NMS threshold
        assert int(conf[i]) == int((s > 0.5).sum())


def test_match_reference_kat_and_golden(cuda, golden_dir):
    from cvpce_amd.models import classification as C
    from oracle import match as omatch
    gold = torch.load(os.path.join(golden_dir, 'nearest.pt'), weights_only=False)
    kat = gold['kat']
    got = C.nearest_neighbors(kat['anchors'].to(cuda), kat['queries'].to(cuda))[:, 0].cpu()
    assert kat['expected'].equal(got)                       # test/models/classification_test.py:8-25
    for case in gold['cases']:
        a, q, k = case['anchors'], case['queries'], case['k']
        got = C.nearest_neighbors(a.to(cuda), q.to(cuda), k).cpu()
        assert got.dtype == torch.int64 and got.shape == (len(q), k)
        srt = case['distances'].sort(dim=-1).values[:, :k + 1]
        safe = (srt[:, 1:] - srt[:, :-1]).min(dim=1).values > 1e-5
        assert got[safe].equal(case['indices'][safe]), 'fp32 path must be index-exact on tie-free rows'
        # the remaining rows: same SET of distances within fp32 rounding
        d = case['distances']
        assert (d.gather(1, got) - d.gather(1, case['indices'])).abs().max() < 1e-5


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('qn,gn,d,k', [(200, 1000, 1024, 1), (37, 3200, 1024, 4), (130, 129, 512, 3), (5, 10000, 512, 1),
                                        (200, 10000, 512, 1), (1600, 10000, 1024, 1)])   # the last two: BASELINE configs[3] (distance-GEMM stress)
def test_match_parity(cuda, dtype, qn, gn, d, k):
    from cvpce_amd import ops
    from oracle import match as omatch
    g = torch.Generator().manual_seed(qn + gn)
    G = F.normalize(torch.rand(gn, d, generator=g), dim=1)     # non-negative like MAC descriptors
    Q = F.normalize(torch.rand(qn, d, generator=g), dim=1)
    Gd, Qd = G.to(dtype), Q.to(dtype)
    idx, dist = ops.match_topk(Qd.to(cuda), Gd.to(cuda), k, return_distance=True)
    ref_d = omatch.cosine_distance_matrix(Gd.float(), Qd.float())    # oracle on the same (rounded) operands
    ref_idx = torch.sort(ref_d, dim=-1, stable=True).indices[:, :k]
    srt = ref_d.sort(dim=-1).values[:, :k + 1]
    safe = (srt[:, 1:] - srt[:, :-1]).min(dim=1).values > 2e-6
    assert safe.float().mean() > 0.9
    assert idx.cpu()[safe].equal(ref_idx[safe])
    torch.testing.assert_close(dist.cpu(), ref_d.gather(1, idx.cpu()), rtol=0, atol=2e-6)


@pytest.mark.parametrize('qn,gn,d', [(200, 10000, 512), (1600, 10000, 1024), (333, 1357, 256), (5, 40, 64)])
def test_match_every_core_and_tile_gives_identical_results(cuda, qn, gn, d):
    """The bf16 search has two cores (the 128-row register-staged kernel; the LDS-DMA kernel of round 5 with tiles of 128 MG gallery rows x
    64 NQ queries) and, opt-in, a one-launch form of k = 1 (atomic (distance, row) keys + ticket).  Which one a launch takes is a cost
    model's choice, so every variant must return the SAME indices and the SAME distances bit for bit -- a query's result must not depend
    on how many queries it was batched with (classification.py:87-95; BASELINE configs[3] sizes first).  Includes a NaN query, equal rows
    (ties -> lower index), a ragged last tile in both dimensions, k = 1 and k = 4, and a second call on the restored state block."""
    from cvpce_amd import ops
    from cvpce_amd._lib import lib
    g = torch.Generator().manual_seed(qn * 7 + gn)
    G = F.normalize(torch.rand(gn, d, generator=g), dim=1)
    Q = F.normalize(torch.rand(qn, d, generator=g), dim=1)
    G[gn // 2] = G[3]                                    # duplicate rows: a tie for every query
    Q[1, 5] = float('nan')
    Gd, Qd = G.to(torch.bfloat16).to(cuda), Q.to(torch.bfloat16).to(cuda)
    variants = [(1, 0, 0, 0)] + [(2, nq, mg, one) for mg in (1, 2) for nq in (2, 3, 4, 5) if not (mg == 1 and nq == 5) for one in (0, 1)]
    ref = None
    try:
        for core, nq, mg, one in variants:
            assert ops.match_set_core(core, nq, mg, one) == 0
            assert ops.MATCH_ONE_LAUNCH == bool(one)
            got = [ops.match_topk(Qd, Gd, 4, return_distance=True), ops.match_topk(Qd, Gd, 1, return_distance=True),
                   ops.match_topk(Qd, Gd, 1, return_distance=True)]
            torch.cuda.synchronize()
            got = [(i.cpu(), x.cpu()) for i, x in got]
            assert int(got[0][0].min()) >= 0 and int(got[0][0].max()) < gn
            assert got[1][0].equal(got[0][0][:, :1]) and got[2][0].equal(got[1][0]) and got[2][1].equal(got[1][1])
            if ref is None:
                ref = got
            for (i, x), (ri, rx) in zip(got, ref):
                assert i.equal(ri), (core, nq, mg, one)
                assert torch.equal(x.view(torch.int32), rx.view(torch.int32)), (core, nq, mg, one)       # bit for bit (inf rows included)
    finally:
        ops.match_set_core(0, 0, 0, 0)
    assert not ops._MATCH_STATE or ops.MATCH_ONE_LAUNCH is False          # (a state block exists only because a one-launch variant ran)
    assert ref[0][0][1].tolist() == [0, 1, 2, 3]                                      # the NaN query: all distances +inf, lowest rows first
    assert lib.cvpce_match_set_core(2, 5, 1, 0) == 1 and lib.cvpce_match_set_core(3, 0, 0, 0) == 1      # no 128-row x 320-query tile; no such core


def test_match_k_clamped_and_nan_rows(cuda):
    """classification.py:95 `argsort[:, :k]` returns min(k, G) columns; a query with a non-finite embedding still gets valid
    gallery indices (NaN distances sort last), so `annotations[j]` lookups never go out of range."""
    from cvpce_amd import ops
    from cvpce_amd.models import classification as C
    g = torch.Generator().manual_seed(3)
    G = F.normalize(torch.rand(5, 64, generator=g), dim=1)
    Q = F.normalize(torch.rand(4, 64, generator=g), dim=1)
    idx = C.nearest_neighbors(G.to(cuda), Q.to(cuda), k=8).cpu()
    assert idx.shape == (4, 5)
    for row in idx.tolist():
        assert sorted(row) == [0, 1, 2, 3, 4]
    assert C.nearest_neighbors(G.to(cuda), torch.empty(0, 64).to(cuda), k=2).shape == (0, 2)
    Qn = Q.clone(); Qn[1, 3] = float('nan')
    for dt in (torch.float32, torch.bfloat16):
        idx = ops.match_topk(Qn.to(dt).to(cuda), G.to(dt).to(cuda), 3).cpu()
        assert int(idx.min()) >= 0 and int(idx.max()) < 5
        assert idx[1].tolist() == [0, 1, 2]                  # all distances of the NaN row are +inf: lowest indices first
        assert idx[[0, 2, 3]].equal(ops.match_topk(Q.to(dt).to(cuda), G.to(dt).to(cuda), 3).cpu()[[0, 2, 3]])


def test_match_ties_lowest_index(cuda):
    from cvpce_amd import ops
    G = torch.zeros(300, 64); G[:, 0] = 1.0       # all gallery rows identical -> all distances tie
    Q = torch.zeros(3, 64); Q[:, 0] = 1.0
    idx = ops.match_topk(Q.to(cuda), G.to(cuda), 5).cpu()
    assert idx.equal(torch.arange(5).expand(3, 5))


@pytest.mark.parametrize('n,h,w', [(2, 32, 48), (1, 256, 256), (3, 16, 16), (1, 16, 32), (37, 64, 64)])
def test_vgg_stem_fused_parity(cuda, n, h, w):
    """Fused conv1_1+ReLU+conv1_2+ReLU+pool kernel against the oracle ops on the same bf16-rounded operands
    (conv1_1's output is rounded to bf16 before conv1_2, exactly like the unfused schedule stores it)."""
    from cvpce_amd import ops
    g = torch.Generator().manual_seed(n * 1000 + h)
    x = r16(torch.randn(n, 3, h, w, generator=g))
    w1 = torch.randn(64, 3, 3, 3, generator=g) / math.sqrt(27)
    b1 = torch.randn(64, generator=g) * 0.1
    w2 = torch.randn(64, 64, 3, 3, generator=g) / math.sqrt(576)
    b2 = torch.randn(64, generator=g) * 0.1
    mid = r16(F.relu(F.conv2d(x, r16(w1), b1, padding=1)))
    ref = F.max_pool2d(F.relu(F.conv2d(mid, r16(w2), b2, padding=1)), 2, 2)
    ps = ops.PackedStem(w1, b1, w2, b2, device=cuda)
    for cstride in (8, 4):
        xin = torch.zeros(n, h, w, cstride, dtype=BF)
        xin[..., :3] = x.permute(0, 2, 3, 1).to(BF)
        got = nchw(ops.vgg_stem(xin.to(cuda), ps))
        assert got.shape == ref.shape
        assert rel_err(got, ref) < 1e-2, rel_err(got, ref)
    # and bit-for-bit against the unfused HIP schedule (same K order per output element is NOT guaranteed across
    # the two kernels, so compare within one bf16 ulp instead)
    pc1, pc2 = ops.PackedConv(w1, b1, 1, 1, device=cuda), ops.PackedConv(w2, b2, 1, 1, device=cuda)
    xin8 = torch.zeros(n, h, w, 8, dtype=BF); xin8[..., :3] = x.permute(0, 2, 3, 1).to(BF)
    unf = ops.conv2d(ops.conv2d(xin8.to(cuda), pc1, act=1), pc2, act=1, pool=True)
    fused = ops.vgg_stem(xin8.to(cuda), ps)
    assert (unf.float() - fused.float()).abs().max() <= 2 ** -7 * unf.float().abs().max()


def test_vgg_stem_rejects_bad_shapes(cuda):
    from cvpce_amd import ops
    ps = ops.PackedStem(torch.zeros(64, 3, 3, 3), torch.zeros(64), torch.zeros(64, 64, 3, 3), torch.zeros(64), device=cuda)
    with pytest.raises(RuntimeError):
        ops.vgg_stem(torch.zeros(1, 24, 16, 8, dtype=BF, device=cuda), ps)     # H not a multiple of 16


@pytest.mark.parametrize('n,h,w', [(2, 64, 64), (1, 800, 608), (3, 37, 51), (1, 7, 9), (2, 130, 66), (1, 1, 1)])
def test_gln_stem_fused_parity(cuda, n, h, w):
    """conv7x7/2 + FrozenBN + ReLU + maxpool3x3/2 in one launch against the oracle ops on the same bf16-rounded operands
    (the convolution output is rounded to bf16 before pooling, as the unfused schedule stores it), and against the unfused
    HIP schedule (generic conv + pool kernels) within one bf16 ulp (the two kernels accumulate K in different orders).
    Odd and tiny sizes exercise every border: conv padding, pool padding, partial tiles."""
    from cvpce_amd import ops
    g = torch.Generator().manual_seed(zlib.crc32(f'glnstem/{n}/{h}/{w}'.encode()))
    x = r16(torch.randn(n, 3, h, w, generator=g))
    wt = torch.randn(64, 3, 7, 7, generator=g) / math.sqrt(147)
    scale = torch.rand(64, generator=g) + 0.5
    shift = torch.randn(64, generator=g) * 0.2
    wf = r16(wt * scale[:, None, None, None])
    ref = F.max_pool2d(r16(F.relu(F.conv2d(x, wf, shift, stride=2, padding=3))), 3, 2, 1)
    ps = ops.PackedGlnStem(wt, scale, shift, device=cuda)
    xin = nhwc(x).to(cuda)
    got = ops.gln_stem(xin, ps)
    assert nchw(got).shape == ref.shape
    assert rel_err(nchw(got), ref) < 1e-2, rel_err(nchw(got), ref)
    pc = ops.PackedConv(wt, None, 2, 3, scale=scale, shift=shift, device=cuda)
    unf = ops.maxpool2d(ops.conv2d(xin, pc, act=1), 3, 2, 1)
    assert unf.shape == got.shape
    assert (unf.float() - got.float()).abs().max() <= 2 ** -7 * unf.float().abs().max()
    assert (unf != got).float().mean() < 0.02            # and almost everywhere bit-identical


def test_gln_stem_channel3_ignored(cuda):
    """Whatever sits in channels 3..7 of the NHWC8 input must not reach the output."""
    from cvpce_amd import ops
    g = torch.Generator().manual_seed(5)
    wt = torch.randn(64, 3, 7, 7, generator=g) / 12
    ps = ops.PackedGlnStem(wt, torch.ones(64), torch.zeros(64), device=cuda)
    x = torch.randn(1, 40, 40, 8, generator=g).to(BF)
    clean = x.clone(); clean[..., 3:] = 0
    assert torch.equal(ops.gln_stem(x.to(cuda), ps), ops.gln_stem(clean.to(cuda), ps))


@pytest.mark.parametrize('dtype,c', [(torch.bfloat16, 256), (torch.float32, 9), (torch.float32, 36), (torch.bfloat16, 8)])
def test_atlas_pack_unpack(cuda, dtype, c):
    """One-launch level atlas <-> per-level copies against plain slicing (the layout GLNEngine.atlas_layout produces)."""
    from cvpce_amd import ops
    from cvpce_amd.models.proposals import GLNEngine
    shapes = [(100, 100), (50, 50), (25, 25), (13, 13), (7, 7)]
    hc, wc, offs = GLNEngine.atlas_layout(shapes)
    g = torch.Generator().manual_seed(c)
    levels = [torch.randn(3, h, w, c, generator=g).to(dtype).to(cuda) for h, w in shapes]
    atlas = torch.zeros(3, hc, wc, c, dtype=dtype, device=cuda)
    ops.atlas_pack(levels, atlas, offs)
    want = torch.zeros_like(atlas)
    for f, (h, w), (oy, ox) in zip(levels, shapes, offs):
        want[:, oy:oy + h, ox:ox + w] = f
    assert torch.equal(atlas, want)                               # and the gaps are untouched (still zero)
    junk = torch.randn(3, hc, wc, c, generator=g).to(dtype).to(cuda)
    back = ops.atlas_unpack(junk, shapes, offs)
    for t, (h, w), (oy, ox) in zip(back, shapes, offs):
        assert t.is_contiguous() and torch.equal(t, junk[:, oy:oy + h, ox:ox + w])
    with pytest.raises(RuntimeError):
        ops.atlas_pack(levels, atlas, [(o[0] + 60, o[1]) for o in offs])      # a level would leave the canvas


HALO_CASES = [  # n, cin, h, w, cout, pool
    (4, 128, 128, 128, 128, True),     # VGG conv2_2 shape (TC = 128, pooled)
    (16, 128, 64, 64, 256, False),     # conv3_1
    (16, 256, 64, 64, 256, True),      # conv3_3 (pooled)
    (64, 256, 32, 32, 512, False),     # conv4_1: two cout tiles
    (256, 512, 16, 16, 512, False),    # conv5_x: one pixel tile per image
    (5, 64, 112, 96, 192, False),      # ragged: Cout not a multiple of the cout tile, odd tile counts
    (8, 256, 100, 100, 256, False),    # RetinaNet head on P3: H, W not multiples of the 16x16 tile
    (8, 256, 50, 50, 256, False),      # ... on P4
    (9, 128, 26, 38, 128, True),       # ragged + pooled (even sizes)
    (8, 256, 13, 13, 256, False),      # smaller than one tile
    (3, 64, 48, 80, 128, False),       # conv2_1 shape class: one 64-channel chunk, Cout = 128
    (2, 64, 30, 44, 96, True),         # ragged + pooled, Cout < 128, odd tile counts both ways
    (2, 192, 40, 70, 128, True),       # three chunks
]


@pytest.mark.parametrize('n,cin,h,w,cout,pool', HALO_CASES)
def test_conv3x3_halo_parity(cuda, n, cin, h, w, cout, pool):
    """Halo-patch kernels (16x16 tiles for Cout > 128, 16x32 tiles below) against the implicit-GEMM HIP kernel on the
    same inputs (and, for the small cases, the oracle)."""
    from cvpce_amd import ops
    g = torch.Generator().manual_seed(cin + cout + h)
    x = torch.randn(n, h, w, cin, generator=g).to(BF)
    wgt = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin)
    bias = torch.randn(cout, generator=g) * 0.1
    pc = ops.PackedConv(wgt, bias, 1, 1, device=cuda)
    xin = x.to(cuda)
    ops.HALO_RAGGED = True
    ops.PROFILE = ops.ConvProfile()
    try:
        y = ops.conv2d(xin, pc, act=1, pool=pool)
        assert [r[0] for r in ops.PROFILE.records][-1].startswith('conv3x3_halo')      # the halo path did run
        ops.USE_HALO_3X3 = False
        y2 = ops.conv2d(xin, pc, act=1, pool=pool)
    finally:
        ops.USE_HALO_3X3 = True
        ops.HALO_RAGGED = False
        ops.PROFILE = None
    torch.cuda.synchronize()
    assert y.shape == y2.shape
    assert (y.float() - y2.float()).abs().max() <= 2 ** -7 * y2.float().abs().max()
    if n * h * w * cin * cout <= 4 * 128 * 128 * 128 * 128:
        ref = F.relu(F.conv2d(x.float().permute(0, 3, 1, 2), r16(wgt), bias, padding=1))
        if pool:
            ref = F.max_pool2d(ref, 2, 2)
        assert rel_err(nchw(y), ref) < 1e-2


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('n,cin,h,w,cout', [(2, 256, 100, 151, 9), (2, 256, 100, 151, 36), (1, 64, 50, 77, 9), (3, 128, 48, 48, 36), (1, 256, 64, 64, 1)])
def test_conv3x3_thin_out_parity(cuda, dtype, n, cin, h, w, cout):
    """The RetinaNet head's output convs (cls_logits 256 -> 9, bbox_reg 256 -> 36: fp32 out, no activation; reached from
    cvpce/models/proposals.py:162-168) through the thin-output form of the wide halo kernel (round 5) against the register-staged implicit
    GEMM they ran on before and against F.conv2d on the same rounded operands: ragged tiles in both dimensions, a Cout that is not a
    multiple of 4 (scalar stores, element-wise bias), Cout = 1, both storage types of the detector."""
    from cvpce_amd import ops
    g = torch.Generator().manual_seed(cin + cout + h)
    x = torch.randn(n, h, w, cin, generator=g).to(dtype)
    wgt = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin)
    bias = torch.randn(cout, generator=g) * 0.1
    pc = ops.PackedConv(wgt, bias, 1, 1, device=cuda, dtype=dtype)
    xin = x.to(cuda)
    ops.PROFILE = ops.ConvProfile()
    try:
        y = ops.conv2d(xin, pc, out_f32=True)
        assert ops.PROFILE.layer_records[-1][0].startswith('conv3x3_halo3 thin out')          # the new path did run
        ops.USE_HALO_THIN_OUT = False
        y2 = ops.conv2d(xin, pc, out_f32=True)
        assert not ops.PROFILE.layer_records[-1][0].startswith('conv3x3_halo3 thin out')
    finally:
        ops.USE_HALO_THIN_OUT = True
        ops.PROFILE = None
    torch.cuda.synchronize()
    assert y.dtype == torch.float32 and y.shape == y2.shape == (n, h, w, cout)
    assert (y - y2).abs().max() <= 2e-5 * y2.abs().max() + 1e-6                   # the same products, another summation order
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), wgt.to(dtype).float(), bias, padding=1)
    assert rel_err(nchw(y), ref) < 1e-4
    with pytest.raises(RuntimeError):
        torch.ops.cvpce_amd.conv3x3_halo_thin_out(xin, pc.weight_halo, pc.bias, torch.empty((n, h, w, 200), dtype=torch.float32, device=cuda), 200, pc.k_pad, pc.cout_pad)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('n,cin,h,w,cout,k,stride,res,act,f32', [
    (4, 512, 25, 25, 512, 3, 1, False, 1, False),      # layer4 conv2 (72 K-steps, split 4)
    (2, 512, 50, 50, 512, 3, 2, False, 1, False),      # layer4's opening stride-2 conv
    (3, 256, 64, 64, 256, 3, 2, False, 1, False),      # a 32 x 32 output map: the largest that is split
    (4, 256, 13, 13, 256, 3, 2, False, 0, False),      # P7 from P6: 7 x 7 out, ragged pixel tile
    (1, 256, 25, 25, 256, 3, 1, True, 0, False),       # + same-size residual
    (2, 320, 9, 11, 200, 3, 1, False, 1, True),        # 45 K-steps: ksplit does not divide them; ragged cout tile; fp32 out
])
def test_conv2d_splitk_parity(cuda, dtype, n, cin, h, w, cout, k, stride, res, act, f32):
    """Split-K launches of the register-staged kernel (round 5; torchvision ResNet-50 layer3 / layer4 and the FPN's extra levels as built at
    cvpce/models/proposals.py:109-139) against the unsplit kernel (the same products, another fp32 summation order), against F.conv2d on
    the same rounded operands, and against themselves: repeated launches on one workspace are bit-identical (the partial tiles are added
    in split order)."""
    from cvpce_amd import ops
    g = torch.Generator().manual_seed(cin + cout + h + k)
    x = torch.randn(n, h, w, cin, generator=g).to(dtype)
    wgt = torch.randn(cout, cin, k, k, generator=g) / math.sqrt(k * k * cin)
    bias = torch.randn(cout, generator=g) * 0.1
    pc = ops.PackedConv(wgt, bias, stride, k // 2, device=cuda, dtype=dtype)
    xin = x.to(cuda)
    ho, wo = pc.out_hw(h, w, 0)
    r = torch.randn(n, ho, wo, cout, generator=g).to(dtype) if res else None
    rin = r.to(cuda) if res else None
    assert ops.splitk_factor(pc, ho, wo) == 4
    ops.PROFILE = ops.ConvProfile()
    try:
        ys = [ops.conv2d(xin, pc, act=act, out_f32=f32, residual=rin).clone() for _ in range(3)]
        assert 'split-K' in ops.PROFILE.layer_records[-1][0]                                   # the new path did run
        ops.CONV_SPLITK = False
        y2 = ops.conv2d(xin, pc, act=act, out_f32=f32, residual=rin)
        assert 'split-K' not in ops.PROFILE.layer_records[-1][0]
    finally:
        ops.CONV_SPLITK = True
        ops.PROFILE = None
    torch.cuda.synchronize()
    assert torch.equal(ys[0], ys[1]) and torch.equal(ys[0], ys[2])
    y = ys[0]
    assert y.dtype == (torch.float32 if f32 else dtype) and y.shape == y2.shape == (n, ho, wo, cout)
    tol = 2e-5 if f32 else 2.0 ** (-7 if dtype == torch.bfloat16 else -10)                     # one storage ulp where a rounding tips
    assert (y.float() - y2.float()).abs().max() <= tol * y2.float().abs().max() + 1e-6
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), wgt.to(dtype).float(), bias, stride=stride, padding=k // 2)
    if res:
        ref = ref + r.float().permute(0, 3, 1, 2)
    if act:
        ref = ref.relu()
    assert rel_err(nchw(y.float()), ref) < (1e-4 if f32 else (6e-3 if dtype == torch.bfloat16 else 8e-4))
    with pytest.raises(RuntimeError):                                                          # a workspace that is too small is refused
        torch.ops.cvpce_amd.conv2d_splitk(xin, pc.weight, pc.bias, rin, torch.empty_like(y), pc.cout, k, k, stride, k // 2, ho, wo, pc.k_pad, pc.cout_pad,
                                          act, int(f32), 0, 1 if res else 0, 4, torch.zeros(4096, dtype=torch.uint8, device=cuda))


@pytest.mark.parametrize('n,use_map', [(3, False), (3, True), (90, True), (50, False)])
def test_conv3x3_atlas_masked(cuda, n, use_map):
    """cvpce_conv3x3_halo_masked: two maps packed side by side with a one-pixel zero gap == the conv applied to each map
    separately; gap pixels of the output are exactly zero (so the output is again a valid atlas).  With 90 / 50 images every
    persistent workgroup walks several tiles (its tile list and their pixel masks live in LDS), with and without a tile map."""
    from cvpce_amd import ops
    g = torch.Generator().manual_seed(5)
    a = torch.randn(n, 21, 30, 128, generator=g).to(BF)
    b = torch.randn(n, 9, 14, 128, generator=g).to(BF)
    wgt = torch.randn(256, 128, 3, 3, generator=g) / math.sqrt(9 * 128)
    bias = torch.randn(256, generator=g) * 0.1
    pc = ops.PackedConv(wgt, bias, 1, 1, device=cuda)
    atlas = torch.zeros(n, 21, 61, 128, dtype=BF)      # columns 45..60: a tile column that lies wholly in the gap
    atlas[:, :, :30] = a
    atlas[:, 5:14, 31:45] = b
    mask = torch.zeros(21, 61, dtype=torch.uint8)
    mask[:, :30] = 1
    mask[5:14, 31:45] = 1
    if use_map:
        tile_map = ops.atlas_tile_map(mask.to(cuda))
        assert tile_map.numel() == 5                                   # of 2 x 4 tiles: the last column and (1, 2) are all gap
        out = torch.zeros(n, 21, 61, 256, dtype=BF, device=cuda)
        y = ops.conv3x3_atlas(atlas.to(cuda), pc, mask.to(cuda), act=1, tile_map=tile_map, out=out)
    else:
        y = ops.conv3x3_atlas(atlas.to(cuda), pc, mask.to(cuda), act=1)
    ops.HALO_RAGGED = True
    try:
        ya = ops.conv2d(a.to(cuda), pc, act=1)
        yb = ops.conv2d(b.to(cuda), pc, act=1)
    finally:
        ops.HALO_RAGGED = False
    assert torch.equal(y[:, :, :30], ya) and torch.equal(y[:, 5:14, 31:45], yb)
    gap = y.float() * (1 - mask.to(cuda).float())[None, :, :, None]
    assert float(gap.abs().max()) == 0.0
    ref = F.relu(F.conv2d(a.float().permute(0, 3, 1, 2), r16(wgt), bias, padding=1))
    assert rel_err(nchw(y[:, :, :30].contiguous()), ref) < 1e-2


C1X1_CASES = [  # n, cin, h, w, cout, stride, res ('', 'same', 'up'), act
    (2, 64, 40, 52, 256, 1, 'same', 1),      # layer1 conv3 + identity + ReLU
    (2, 256, 40, 52, 64, 1, '', 1),          # layer1 conv1
    (3, 256, 33, 47, 512, 2, '', 0),         # downsample, stride 2, odd sizes
    (1, 1024, 13, 17, 256, 1, 'up', 0),      # FPN lateral + nearest-upsampled top-down
    (2, 512, 7, 9, 2048, 1, 'same', 1),      # layer4 expand, ragged last pixel tile
    (1, 2048, 5, 5, 512, 1, '', 1),          # deep K
    (1, 128, 5, 7, 192, 1, 'same', 1),       # 3 cout tiles (not a power of two: the LDS-staged weights form), 35 pixels in all
    (2, 128, 61, 67, 512, 1, 'same', 1),     # K = 128: weights in registers, ragged last pixel tile, more tiles than persistent waves
    (1, 64, 30, 31, 128, 1, 'up', 1),        # K = 64 (one stage), upsampled residual
]


@pytest.mark.parametrize('n,cin,h,w,cout,stride,res,act', C1X1_CASES)
def test_conv1x1_parity(cuda, n, cin, h, w, cout, stride, res, act):
    """Pointwise GEMM kernel against the oracle conv and against the implicit-GEMM HIP kernel on the same operands."""
    from cvpce_amd import ops
    g = torch.Generator().manual_seed(cin + cout + h)
    x = r16(torch.randn(n, cin, h, w, generator=g))
    wgt = torch.randn(cout, cin, 1, 1, generator=g) / math.sqrt(cin)
    bias = torch.randn(cout, generator=g) * 0.1
    pc = ops.PackedConv(wgt, bias, stride, 0, device=cuda)
    ref = F.conv2d(x, r16(wgt), bias, stride=stride)
    rdev = None
    if res == 'same':
        r = r16(torch.randn(ref.shape, generator=g)); ref = ref + r; rdev = nhwc(r).to(cuda)
    elif res == 'up':
        r = r16(torch.randn(n, cout, (ref.shape[2] + 1) // 2, (ref.shape[3] + 1) // 2, generator=g))
        ref = ref + F.interpolate(r, size=ref.shape[-2:], mode='nearest'); rdev = nhwc(r).to(cuda)
    if act:
        ref = F.relu(ref)
    xin = nhwc(x).to(cuda)
    ops.PROFILE = ops.ConvProfile()
    ops.CONV1X1_ANY_SHAPE = True
    try:
        y = ops.conv2d(xin, pc, act=act, residual=rdev)
        assert ops.PROFILE.records[-1][0] == 'conv1x1_kernel'
        ops.USE_CONV1X1 = False
        y2 = ops.conv2d(xin, pc, act=act, residual=rdev)
        assert ops.PROFILE.records[-1][0] != 'conv1x1_kernel'
    finally:
        ops.USE_CONV1X1 = True
        ops.CONV1X1_ANY_SHAPE = False
        ops.PROFILE = None
    assert rel_err(nchw(y), ref) < 1e-2, rel_err(nchw(y), ref)
    assert (y.float() - y2.float()).abs().max() <= 2 ** -7 * y2.float().abs().max()


@pytest.mark.parametrize('name,n,cin,hw,cout,pool', [('conv2_1', 256, 64, 128, 128, False), ('conv2_2', 256, 128, 128, 128, True),
                                                     ('conv3_2', 256, 256, 64, 256, False), ('conv4_3', 256, 512, 32, 512, True),
                                                     ('conv5_1', 256, 512, 16, 512, False)])
def test_conv3x3_full_size_linearity(cuda, name, n, cin, hw, cout, pool):
    """VGG layer shapes at the benchmark's batch (256 crops), checked by a size-independent property: with zero bias,
    conv(2x) == 2 conv(x) and relu/pool commute with the factor -- exact in bf16 (a power of two) -- plus spot checks of
    a few output pixels against the fp32 oracle."""
    from cvpce_amd import ops
    g = torch.Generator().manual_seed(cin + hw)
    wgt = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin)
    pc = ops.PackedConv(wgt, torch.zeros(cout), 1, 1, device=cuda)
    x = torch.randn(n, hw, hw, cin, generator=g).to(BF).to(cuda)
    y1 = ops.conv2d(x, pc, act=1, pool=pool)
    y2 = ops.conv2d(x * 2, pc, act=1, pool=pool)
    torch.cuda.synchronize()
    assert torch.equal(y2, y1 * 2)
    assert float(y1.float().abs().max()) > 0.5
    # spot check: 3 images, one output row each, against F.conv2d on the same bf16-rounded operands
    for img in (0, n // 2, n - 1):
        xi = x[img:img + 1].float().permute(0, 3, 1, 2).cpu()
        ref = F.relu(F.conv2d(xi, r16(wgt), padding=1))
        if pool:
            ref = F.max_pool2d(ref, 2, 2)
        assert rel_err(nchw(y1[img:img + 1]), ref) < 1e-2


@pytest.mark.parametrize('n,cin,h,w,cout,pool,store', [(5, 128, 32, 32, 512, True, True), (3, 64, 16, 16, 512, False, False),
                                                        (2, 64, 40, 24, 256, False, True), (2, 64, 21, 19, 264, False, False)])
def test_conv_mac_fused_equals_unfused(cuda, n, cin, h, w, cout, pool, store):
    """MAC descriptor (classification.py:46-49 amax over H, W) taken in the conv epilogue -- with MaxPool2d(2,2) fused into the
    store (conv4_3 -> pool4) or no store at all (conv5_3) -- against conv -> global-max kernel -> max-pool kernel: bit for bit,
    ragged tiles and a ragged cout tile included; and against the oracle ops."""
    from cvpce_amd import ops
    g = torch.Generator().manual_seed(n * 100 + h)
    x = r16(torch.randn(n, cin, h, w, generator=g))
    wgt = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
    bias = torch.randn(cout, generator=g) * 0.1
    pc = ops.PackedConv(wgt, bias, 1, 1, device=cuda)
    xd = nhwc(x).to(cuda)
    ops.HALO_RAGGED = True
    try:
        y = ops.conv2d(xd, pc, act=1)
        want_mac = torch.zeros(n, 40 + cout, device=cuda)
        ops.global_max_into(y, want_mac, 40)
        mac = torch.zeros(n, 40 + cout, device=cuda)
        got = ops.conv2d_relu_mac(xd, pc, mac, 40, store=store, pool=pool)
    finally:
        ops.HALO_RAGGED = False
    torch.cuda.synchronize()
    assert torch.equal(mac, want_mac) and float(mac[:, :40].abs().max()) == 0.0
    if not store:
        assert got is None
    elif pool:
        assert torch.equal(got, ops.maxpool2d(y, 2, 2))
    else:
        assert torch.equal(got, y)
    ref = F.relu(F.conv2d(x, r16(wgt), bias, padding=1))
    assert rel_err(mac[:, 40:].cpu(), r16(ref).amax(dim=(-2, -1))) < 1e-2


def test_crop_resize_narrow_pixels(cuda):
    """Crop mode 2 (8-byte NHWC4 pixels for the fused stem) carries exactly the values of mode 1 (NHWC8)."""
    from cvpce_amd import ops
    from cvpce_amd.models import classification as C
    img = torch.rand(3, 300, 420, generator=torch.Generator().manual_seed(8)).to(cuda)
    boxes = torch.tensor([[10.2, 20.7, 200.1, 180.0], [0., 0., 420., 300.], [399.5, 250.2, 460.0, 330.0]], device=cuda)
    a = ops.crop_resize(img, boxes, 256, mode=1, mean=C.TANH_MEAN, std=C.TANH_STD)
    b = ops.crop_resize(img, boxes, 256, mode=2, mean=C.TANH_MEAN, std=C.TANH_STD)
    assert b.shape == (3, 256, 256, 4) and torch.equal(a[..., :4], b) and float(a[..., 3:].abs().max()) == 0.0
    # the two-pixels-per-thread embedder kernel against the one-pixel f32 kernel (mode 0, itself checked against the oracle)
    # followed by scale_to_tanh + normalisation on the host: the same fp32 expression, so the same bf16 bits
    v = ops.crop_resize(img, boxes, 256, mode=0).cpu()
    mean = torch.tensor(C.TANH_MEAN)[None, :, None, None]
    std = torch.tensor(C.TANH_STD)[None, :, None, None]
    want = ((v * 2.0 - 1.0 - mean) / std).to(torch.bfloat16).permute(0, 2, 3, 1)
    assert torch.equal(b[..., :3].cpu(), want)


BNECK_CASES = [  # n, cin, planes, h, w, separate residual tensor (the projection-shortcut case)
    (2, 256, 64, 56, 56, False),      # layer1 identity block, tiles exact (4 x 14)
    (1, 64, 64, 37, 51, True),        # layer1 first block: Cin = 64, residual = shortcut conv; ragged tiles
    (2, 512, 128, 100, 100, False),   # layer2
    (1, 512, 128, 5, 9, False),       # smaller than one tile
    (2, 1024, 256, 50, 50, False),    # layer3
    (1, 1024, 256, 15, 29, False),    # layer3, one pixel past a tile edge
]


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('n,cin,p,h,w,sep', BNECK_CASES)
def test_bottleneck_fused_parity(cuda, dtype, n, cin, p, h, w, sep):
    """cvpce_bottleneck_fused (1x1 -> 3x3 -> 1x1 + residual, intermediates in LDS) against the three-launch HIP schedule (same
    rounding points: within two ulps of the storage type, almost everywhere identical) and against the oracle ops on the same
    16-bit-rounded operands."""
    from cvpce_amd import ops
    rd = lambda t: t.to(dtype).to(torch.float32)
    g = torch.Generator().manual_seed(zlib.crc32(f'bneck/{n}/{cin}/{p}/{h}/{w}'.encode()))
    x = rd(torch.randn(n, cin, h, w, generator=g))
    mk = lambda co, ci, k: (torch.randn(co, ci, k, k, generator=g) / math.sqrt(ci * k * k), torch.randn(co, generator=g) * 0.1)
    (w1, b1), (w2, b2), (w3, b3) = mk(p, cin, 1), mk(p, p, 3), mk(4 * p, p, 1)
    res = rd(torch.randn(n, 4 * p, h, w, generator=g)) if sep else x
    assert sep or cin == 4 * p
    pcs = [ops.PackedConv(w_, b_, 1, pad, device=cuda, dtype=dtype) for (w_, b_, pad) in ((w1, b1, 0), (w2, b2, 1), (w3, b3, 0))]
    to_dev = lambda t: t.permute(0, 2, 3, 1).contiguous().to(dtype).to(cuda)
    xd = to_dev(x)
    rdv = to_dev(res) if sep else xd
    saved = ops.FUSED_BOTTLENECK_MAX_PLANES
    ops.FUSED_BOTTLENECK_MAX_PLANES = 256          # the detector only uses P = 64 (the measured win); the kernel is verified for all three widths
    try:
        assert ops.can_fuse_bottleneck(xd, *pcs, rdv)
        y = ops.bottleneck(xd, *pcs, rdv)                   # fragment-major weights (round 5, the default)
        ops.BNECK_FRAGMENT_MAJOR = False
        y_rm = ops.bottleneck(xd, *pcs, rdv)                # the same kernel reading the row-major weights: the same products in the same order
    finally:
        ops.FUSED_BOTTLENECK_MAX_PLANES = saved
        ops.BNECK_FRAGMENT_MAJOR = True
    assert torch.equal(y.view(torch.int16), y_rm.view(torch.int16))
    ops.USE_FUSED_BOTTLENECK = False
    try:
        m1 = ops.conv2d(xd, pcs[0], act=1)
        m2 = ops.conv2d(m1, pcs[1], act=1)
        y3 = ops.conv2d(m2, pcs[2], act=1, residual=rdv)
    finally:
        ops.USE_FUSED_BOTTLENECK = True
    torch.cuda.synchronize()
    assert y.shape == y3.shape == (n, h, w, 4 * p) and y.dtype == dtype
    ulp = 2.0 ** (-7 if dtype == torch.bfloat16 else -10)
    d = (y.float() - y3.float()).abs()
    assert float(d.max()) <= 2.5 * ulp * float(y3.float().abs().max()), float(d.max())
    assert float((y != y3).float().mean()) < 0.05
    t1 = rd(F.relu(F.conv2d(x, rd(w1), b1)))
    t2 = rd(F.relu(F.conv2d(t1, rd(w2), b2, padding=1)))
    ref = F.relu(F.conv2d(t2, rd(w3), b3) + res)
    assert rel_err(nchw(y), ref) < (1.5e-2 if dtype == torch.bfloat16 else 3e-3), rel_err(nchw(y), ref)


def test_bottleneck_fused_rejects_what_it_does_not_cover(cuda):
    from cvpce_amd import ops
    x = torch.zeros(1, 8, 8, 2048, dtype=BF, device=cuda)
    mk = lambda co, ci, k, s=1: ops.PackedConv(torch.zeros(co, ci, k, k), torch.zeros(co), s, k // 2, device=cuda)
    assert not ops.can_fuse_bottleneck(x, mk(512, 2048, 1), mk(512, 512, 3), mk(2048, 512, 1), x)          # layer4: P = 512 does not fit the LDS
    x2 = torch.zeros(1, 8, 8, 256, dtype=BF, device=cuda)
    assert not ops.can_fuse_bottleneck(x2, mk(128, 256, 1), mk(128, 128, 3, 2), mk(512, 128, 1), torch.zeros(1, 4, 4, 512, dtype=BF, device=cuda))  # stride 2


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_transform_batch_equals_per_image(cuda, dtype):
    """cvpce_gln_transform_batch: images of different sizes into one padded batch, bit-identical to the per-image launches."""
    from cvpce_amd import ops
    from cvpce_amd.models import proposals as P
    g = torch.Generator().manual_seed(12)
    imgs = [torch.rand(3, h0, w0, generator=g).to(cuda) for h0, w0 in ((300, 517), (640, 480), (97, 131))] * 12     # 36 images: two launches
    rs = [P.resized_hw(*i.shape[-2:]) for i in imgs]
    hp, wp = max((r[0] + 31) // 32 * 32 for r in rs), max((r[1] + 31) // 32 * 32 for r in rs)
    one = torch.full((len(imgs), hp, wp, 8), 3.0, dtype=dtype, device=cuda)
    ops.gln_transform_batch(imgs, one, rs, P.IMAGE_MEAN, P.IMAGE_STD)
    ref = torch.full((len(imgs), hp, wp, 8), 5.0, dtype=dtype, device=cuda)
    for i, (img, (h, w)) in enumerate(zip(imgs, rs)):
        ops.gln_transform_into(img, ref, i, h, w, P.IMAGE_MEAN, P.IMAGE_STD)
    assert torch.equal(one, ref)
