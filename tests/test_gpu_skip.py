"""Constant-padding tile skipping of the embedder (csrc/skiplist.hip, the *_list entry points of include/cvpce_amd.h).

`resize_for_classification` (/root/reference/cvpce/datautils.py:232-239) pads every crop to a square with 0.5; the embedder
(/root/reference/cvpce/models/classification.py:38-51) then spends its time on that constant.  The work-list kernels skip the
tiles that lie in it.  The bar: results BIT-IDENTICAL to the plain kernels, for any mix of box shapes."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

S = 256


def _boxes(n, h0, w0, seed, kinds=('wide', 'tall', 'square', 'thin', 'tiny', 'edge', 'huge')):
    g = torch.Generator().manual_seed(seed)
    out = []
    for i in range(n):
        kind = kinds[i % len(kinds)]
        r = lambda a, b: a + (b - a) * float(torch.rand(1, generator=g))
        if kind == 'wide':
            w, h = r(80, 400), r(20, 150)
        elif kind == 'tall':
            w, h = r(20, 150), r(80, 400)
        elif kind == 'square':
            w = h = r(30, 300)
        elif kind == 'thin':
            w, h = (r(200, 500), r(2, 12)) if i % 2 else (r(2, 12), r(200, 500))
        elif kind == 'tiny':
            w, h = r(1.2, 6), r(1.2, 6)
        elif kind == 'huge':
            w, h = r(0.6 * w0, 1.2 * w0), r(0.3 * h0, 0.9 * h0)
        else:
            w, h = r(50, 200), r(50, 200)
        x1, y1 = r(0, w0 - 2), r(0, h0 - 2)
        if kind == 'edge':
            x1, y1 = w0 - w / 2, h0 - h / 2           # sticks out of the image: the slice clamps
        out.append([x1, y1, x1 + w, y1 + h])
    return torch.tensor(out, dtype=torch.float32)


def _src_index(scale, dst, in_size):
    """preproc.hip src_index in fp32 (PyTorch area_pixel_compute_source_index, align_corners=False)."""
    s = torch.tensor(scale, dtype=torch.float32) * (torch.tensor(float(dst), dtype=torch.float32) + 0.5) - 0.5
    s = max(float(s), 0.0)
    return min(int(s), in_size - 1)


def _extents_model(boxes, h0, w0):
    out = []
    for b in boxes.tolist():
        x1, y1, x2, y2 = [int(v) for v in b]                     # .to(long)
        x1, x2 = min(max(x1, 0), w0), min(max(x2, 0), w0)
        y1, y2 = min(max(y1, 0), h0), min(max(y2, 0), h0)
        cw, ch = max(x2 - x1, 0), max(y2 - y1, 0)
        larger = max(cw, ch)
        if larger == 0:
            out.append([0, 0])
            continue
        sc = float(torch.tensor(float(larger), dtype=torch.float32) / torch.tensor(float(S), dtype=torch.float32))
        i0 = [_src_index(sc, o, larger) for o in range(S)]
        out.append([sum(1 for v in i0 if v < ch), sum(1 for v in i0 if v < cw)])
    return torch.tensor(out, dtype=torch.int32)


def test_crop_extents_and_exact_padding(cuda):
    from cvpce_amd import ops
    from cvpce_amd.models.classification import TANH_MEAN, TANH_STD
    h0, w0 = 600, 800
    g = torch.Generator().manual_seed(5)
    img = torch.rand(3, h0, w0, generator=g).cuda()
    boxes = _boxes(70, h0, w0, seed=11)
    ext = ops.crop_extents(boxes.cuda(), None, h0, w0, S).cpu()
    assert torch.equal(ext, _extents_model(boxes, h0, w0))
    assert (ext[:, 0] < S).any() and (ext[:, 1] < S).any() and ((ext[:, 0] == S) & (ext[:, 1] == S)).any()
    for mode, c in ((2, 4), (1, 8)):
        crops = ops.crop_resize(img, boxes.cuda(), S, mode=mode, mean=TANH_MEAN, std=TANH_STD).cpu().view(torch.int16)
        const = ops.crop_resize(torch.zeros(3, 1, 1).cuda(), torch.zeros(1, 4).cuda(), S, mode=mode, mean=TANH_MEAN, std=TANH_STD).cpu().view(torch.int16)[0]
        assert (const == const[0, 0]).all()                      # one pixel value everywhere
        for p in range(len(boxes)):
            ey, ex = ext[p].tolist()
            assert (crops[p, ey:] == const[ey:]).all() and (crops[p, :, ex:] == const[:, ex:]).all(), (p, boxes[p], ey, ex)
    # mode 0 (the reference-shaped f32 crops): padding pixels are exactly 0.5
    f = ops.crop_resize(img, boxes.cuda(), S, mode=0).cpu()
    for p in range(len(boxes)):
        ey, ex = ext[p].tolist()
        assert (f[p, :, ey:] == 0.5).all() and (f[p, :, :, ex:] == 0.5).all()
    # boxes beyond the device-side count: full extent
    cnt = torch.tensor([5], dtype=torch.int32).cuda()
    e2 = ops.crop_extents(boxes.cuda(), cnt, h0, w0, S).cpu()
    assert torch.equal(e2[:5], ext[:5]) and (e2[5:] == S).all()
    # ... and the slots of several images of one size in one launch (35 slots each, one count per image)
    e3 = ops.crop_extents(boxes.cuda(), torch.tensor([35, 6], dtype=torch.int32).cuda(), h0, w0, S, per_image=35).cpu()
    assert torch.equal(e3[:41], ext[:41]) and (e3[41:] == S).all()


def _extent(e0, pool_mask, nops, size):
    """include/cvpce_amd.h: the crop's content extent through the first `nops` ops of the pass (conv: +1, pool: halve upwards)."""
    if e0 >= S:
        return size
    e = e0
    for i in range(nops):
        e = (e + 1) // 2 if (pool_mask >> i) & 1 else e + 1
    return min(size, e)


@pytest.fixture(params=[False, True], ids=['tiles', 'tiles+rows'])
def skip_rows(request):
    """Both settings of the row-level cut (classification.SKIP_ROWS: tiles only | tiles cut at their last non-constant row)."""
    from cvpce_amd.models import classification as C
    old, C.SKIP_ROWS = C.SKIP_ROWS, request.param
    yield request.param
    C.SKIP_ROWS = old


def test_worklists_match_python_model(cuda, skip_rows):
    from cvpce_amd import ops, synthetic
    enc = synthetic.synthetic_macvgg(seed=1).cuda()
    steps, layers, pool_mask = enc.engine().skip_plan(S)
    # stem: whole tiles; conv2_x (wide-tile kernel): rows cut; conv3_1 .. conv5_3: rows cut + strip lists (the MAC layers conv4_3 /
    # conv5_3 skip too: cvpce_mac_init covers what they leave out)
    assert len(layers) == 12 and [l[-1] for l in layers] == ([1] + [2] * 2 + [3] * 9 if skip_rows else [1] * 12)
    # the op chain of VGG16 cfg 'D' up to relu5_3: 2 convs, pool, 2 convs, pool, 3 convs, pool, 3 convs, pool, 3 convs
    assert [(pool_mask >> i) & 1 for i in range(17)] == [0, 0, 1, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0]
    assert [l[4] for l in layers] == [3, 4, 6, 7, 8, 10, 11, 12, 14, 15, 16, 17] and [l[7] for l in layers] == [0, 3, 4, 6, 7, 8, 10, 11, 12, 14, 15, 16]
    g = torch.Generator().manual_seed(3)
    n = 131
    ext = torch.stack((torch.randint(1, S + 1, (n,), generator=g), torch.full((n,), S)), dim=1).to(torch.int32)
    ext[::3] = ext[::3].flip(1)                                  # tall boxes: columns
    ext[5] = torch.tensor([S, S]); ext[6] = torch.tensor([0, 0]); ext[7] = torch.tensor([1, S]); ext[8] = torch.tensor([S, 255])
    lists, counts, computed = ops.embed_worklists(ext.cuda(), n + 1, S, pool_mask, layers, 256, want_computed=True)
    lists, counts, computed = lists.cpu(), counts.cpu().tolist(), computed.cpu()
    nl = len(layers)
    assert lists.shape[0] == 2 * nl and len(counts) == 3 * nl
    allext = ext.tolist() + [[S, S]]
    for li, L in enumerate(layers):
        h, w, th, tw, out_ops, ih, iw, in_ops, skip = L
        ty_n, tx_n = -(-h // th), -(-w // tw)
        want, swant, comp = [], [], []
        for c, (ey0, ex0) in enumerate(allext):
            ny, nx = ty_n, tx_n
            if skip:
                ny = min(ty_n, -(-_extent(ey0, pool_mask, out_ops, h) // th))
                nx = min(tx_n, -(-_extent(ex0, pool_mask, out_ops, w) // tw))
            ext_in = (_extent(ey0, pool_mask, in_ops, ih) << 12) | _extent(ex0, pool_mask, in_ops, iw)
            last = 16
            for ty in range(ny):
                rows = 16                                        # conv-output rows of the tile that are not constant, rounded up to 4
                if skip >= 2 and ty == ny - 1:
                    act = min(th, _extent(ey0, pool_mask, out_ops, h) - ty * th)
                    last = rows = min(16, (-(-act * 16 // th) + 3) // 4 * 4)
                entries = [(((rows << 24) | ext_in) << 32) | (c << 16) | (ty << 8) | tx for tx in range(nx)]
                if skip >= 3 and rows == 4 and ty == ny - 1:
                    swant += entries                             # a last tile row with 4 useful rows: the strip list
                else:
                    want += entries
            comp.append([(ny - 1) * 16 + last if ny else 0, nx])
        assert counts[li] == len(want) and counts[2 * nl + li] == len(swant), (li, counts[li], len(want), counts[2 * nl + li], len(swant))
        rows_of = [(e >> 56) & 0xFF for e in want]
        assert counts[nl + li] == sum(r + 1 if r < 16 else 16 for r in rows_of) + 4 * len(swant), li       # MFMA work in sixteenths of a tile
        assert lists[li, :len(want)].tolist() == want and lists[nl + li, :len(swant)].tolist() == swant, li
        assert computed[li].tolist() == comp, li
        n_strips = locals().get('n_strips', 0) + len(swant)
    assert (n_strips > 0) == bool(skip_rows)                     # the random extents do produce strip-list tiles


@pytest.mark.parametrize('batch_norm', [False, True])
def test_embedding_is_bit_identical_with_skipping(cuda, batch_norm, skip_rows):
    from cvpce_amd import ops, synthetic
    from cvpce_amd.models import classification as C
    enc = synthetic.synthetic_macvgg(seed=1, batch_norm=batch_norm).cuda()
    eng = enc.engine()
    h0, w0 = 700, 900
    g = torch.Generator().manual_seed(9)
    img = torch.rand(3, h0, w0, generator=g).cuda()
    boxes = _boxes(83, h0, w0, seed=21).cuda()
    crops = ops.crop_resize(img, boxes, S, mode=2, mean=C.TANH_MEAN, std=C.TANH_STD)
    ext = ops.crop_extents(boxes, None, h0, w0, S)
    const = eng.const_crop(C.TANH_MEAN, C.TANH_STD, 4, S)
    plain = eng.embed_packed(crops)
    prof = ops.PROFILE = ops.ConvProfile()
    try:
        skipped = eng.embed_packed(crops, ext=ext, const_in=const)
        summ = prof.summary()
    finally:
        ops.PROFILE = None
    assert torch.equal(plain, skipped)
    done = sum(v['flops_executed'] for v in summ.values()); alg = sum(v['flops'] for v in summ.values())
    assert done < 0.9 * alg, (done, alg)                         # something was skipped
    # ... and with the switch off nothing is
    C.SKIP_PADDING = False
    try:
        assert torch.equal(eng.embed_packed(crops, ext=ext, const_in=const), plain)
    finally:
        C.SKIP_PADDING = True
    # NHWC8 input (the layout of Classifier.classify): same result
    crops8 = ops.crop_resize(img, boxes, S, mode=1, mean=C.TANH_MEAN, std=C.TANH_STD)
    assert torch.equal(eng.embed_packed(crops8, ext=ext, const_in=eng.const_crop(C.TANH_MEAN, C.TANH_STD, 8, S)), plain)
    # extents that claim less padding than there is are still exact (conservative is always valid)
    loose = torch.minimum(ext + 37, torch.full_like(ext, S))
    assert torch.equal(eng.embed_packed(crops, ext=loose, const_in=const), plain)


def test_multi_pass_and_all_padding(cuda, skip_rows):
    """More crops than one pass of the schedule takes (the constant crop rides at the end of EVERY pass), all of one shape like
    the bench's workload, plus degenerate boxes (an all-padding crop: every tile is skipped)."""
    from cvpce_amd import ops, synthetic
    from cvpce_amd.models import classification as C
    enc = synthetic.synthetic_macvgg(seed=1).cuda()
    eng = enc.engine()
    h0 = w0 = 512
    img = torch.rand(3, h0, w0, generator=torch.Generator().manual_seed(2)).cuda()
    g = torch.Generator().manual_seed(4)
    n = C.FUSED_EMBED_BATCH + 70
    x1 = torch.rand(n, generator=g) * 200; y1 = torch.rand(n, generator=g) * 300
    boxes = torch.stack((x1, y1, x1 + 260, y1 + 100), dim=1)
    boxes[3] = torch.tensor([10.0, 10.0, 10.5, 90.0])            # zero width after .to(long)
    boxes = boxes.cuda()
    crops = ops.crop_resize(img, boxes, S, mode=2, mean=C.TANH_MEAN, std=C.TANH_STD)
    ext = ops.crop_extents(boxes, None, h0, w0, S)
    assert ext[3].tolist() == [S, 0]                              # no content column: the whole crop is the constant
    const = eng.const_crop(C.TANH_MEAN, C.TANH_STD, 4, S)
    plain = eng.embed_packed(crops)
    assert torch.equal(plain, eng.embed_packed(crops, ext=ext, const_in=const))      # two early passes (768 + 70 crops), one late pass over all
    old, C.LATE_EMBED_MAX = C.LATE_EMBED_MAX, 500                                      # ... and with the late layers in two passes as well
    try:
        assert torch.equal(plain, eng.embed_packed(crops, ext=ext, const_in=const))
    finally:
        C.LATE_EMBED_MAX = old


def test_content_only_crops(cuda):
    """`ops.crop_resize(..., content_ext=ext)` (cvpce_crop_resize_content, round 5) writes the crops' content and nothing else: inside
    the extents the pixels are the full crop's bit for bit, outside them the buffer keeps what it held (NaNs here); the work-list
    embedder returns the embeddings of the full crops from it (it reads the padding from the constant crop: no NaN reaches an MFMA),
    and any other schedule refuses such crops.  `datautils.py:232-239` is the padding this rests on."""
    from cvpce_amd import ops, synthetic
    from cvpce_amd.models import classification as C
    enc = synthetic.synthetic_macvgg(seed=1).cuda()
    eng = enc.engine()
    h0, w0 = 600, 800
    img = torch.rand(3, h0, w0, generator=torch.Generator().manual_seed(3)).cuda()
    g = torch.Generator().manual_seed(9)
    n = 40
    x1 = torch.rand(n, generator=g) * 400; y1 = torch.rand(n, generator=g) * 300
    wh = 40 + torch.rand(n, 2, generator=g) * 220                                        # wide, tall and near-square boxes
    boxes = torch.stack((x1, y1, x1 + wh[:, 0], y1 + wh[:, 1]), dim=1)
    boxes[5] = torch.tensor([10.0, 10.0, 10.5, 90.0])                                     # zero width: nothing is written at all
    boxes[6] = torch.tensor([700.0, 500.0, 900.0, 700.0])                                 # clipped at the image border
    boxes = boxes.cuda()
    count = torch.tensor([n - 3], dtype=torch.int32, device='cuda')                        # the last three slots are beyond the count
    full = ops.crop_resize(img, boxes, S, mode=2, mean=C.TANH_MEAN, std=C.TANH_STD, count=count)
    ext = ops.crop_extents(boxes, count, h0, w0, S)
    nan = torch.full((n, S, S, 4), float('nan'), dtype=torch.bfloat16, device='cuda')
    part = ops.crop_resize(img, boxes, S, mode=2, mean=C.TANH_MEAN, std=C.TANH_STD, count=count, out=nan.clone(), content_ext=ext)
    torch.cuda.synchronize()
    e = ext.cpu()
    yy = torch.arange(S).view(1, S, 1); xx = torch.arange(S).view(1, 1, S)
    # (a thread writes two pixels: a content row may run one pixel past an odd column extent)
    inside = ((yy < e[:, 0].view(-1, 1, 1)) & (xx < e[:, 1].view(-1, 1, 1))).cuda()
    inside[n - 3:] = False
    outside = ((yy >= e[:, 0].view(-1, 1, 1)) | (xx >= (e[:, 1] + (e[:, 1] & 1)).view(-1, 1, 1))).cuda()
    outside[n - 3:] = True
    pv, fv = part.view(torch.int16), full.view(torch.int16)
    assert torch.equal(pv[inside], fv[inside])
    assert bool(torch.isnan(part[outside].float()).all())
    assert int(inside[5].sum()) == 0 and 0 < int(inside.sum()) < int((n - 3) * S * S * 0.8)
    const = eng.const_crop(C.TANH_MEAN, C.TANH_STD, 4, S)
    v = n - 3
    plain = eng.embed_packed(full[:v])
    assert torch.equal(eng.embed_packed(part[:v], ext=ext[:v], const_in=const, partial=True), plain)
    assert bool(torch.isfinite(plain).all())
    with pytest.raises(RuntimeError, match='work-list'):
        eng.embed_packed(part[:v], partial=True)                                          # no extents: the plain schedule would read the NaNs
    C.SKIP_PADDING = False
    try:
        assert not eng.will_skip(S)
        with pytest.raises(RuntimeError, match='work-list'):
            eng.embed_packed(part[:v], ext=ext[:v], const_in=const, partial=True)
    finally:
        C.SKIP_PADDING = True
    assert eng.will_skip(S)
    with pytest.raises(RuntimeError):                                                      # mode 0 (f32 planes) has no content-only form
        torch.ops.cvpce_amd.crop_resize_content(img, boxes, None, torch.empty(n, 3, S, S, device='cuda'), S, 0, list(C.TANH_MEAN), list(C.TANH_STD), ext)


def test_pipeline_content_only_crops_switch(cuda):
    """BatchedPipeline with the crop kernel writing content only (default) and whole crops: the same results bit for bit, for images of
    one size (extents of the batch in one launch) and of different sizes (per image)."""
    from cvpce_amd import production, synthetic
    dev = torch.device('cuda:0')
    det = synthetic.synthetic_gln(seed=0, detections_per_img=40).to(dev)
    enc = synthetic.synthetic_macvgg(seed=1).to(dev)
    clf = production.Classifier(enc, synthetic.TensorGallery(synthetic.gallery_images(32, seed=100)), device=dev, emb_device=dev, batch_size=16,
                                match_dtype=torch.bfloat16)
    pipe = production.BatchedPipeline(det, clf, 0.5)
    for sizes in (((640, 768), (640, 768)), ((640, 768), (512, 704))):
        imgs = [synthetic.shelf_image(50 + i, h, w).to(dev) for i, (h, w) in enumerate(sizes)]
        assert production.CROP_CONTENT_ONLY
        on = pipe.run(imgs)
        production.CROP_CONTENT_ONLY = False
        try:
            off = pipe.run(imgs)
        finally:
            production.CROP_CONTENT_ONLY = True
        assert int(on['count'].sum()) > 0
        for k in ('boxes', 'scores', 'indices', 'embeddings', 'count'):
            assert torch.equal(on[k], off[k]), k


def test_pipeline_results_identical_with_and_without_skipping(cuda):
    from cvpce_amd import production, synthetic
    from cvpce_amd.models import classification as C
    dev = torch.device('cuda:0')
    det = synthetic.synthetic_gln(seed=0, detections_per_img=60).to(dev)
    enc = synthetic.synthetic_macvgg(seed=1).to(dev)
    gal = synthetic.gallery_images(48, seed=100)
    clf = production.Classifier(enc, synthetic.TensorGallery(gal), device=dev, emb_device=dev, batch_size=16, match_dtype=torch.bfloat16)
    imgs = [synthetic.shelf_image(40 + i, 640, 768).to(dev) for i in range(2)]
    pipe = production.BatchedPipeline(det, clf, 0.5)
    on = pipe.run(imgs)
    C.SKIP_PADDING = False
    try:
        off = pipe.run(imgs)
    finally:
        C.SKIP_PADDING = True
    assert int(on['count'].sum()) > 0
    for k in ('boxes', 'scores', 'indices', 'embeddings', 'count'):
        assert torch.equal(on[k], off[k]), k


def test_classify_reads_extents_off_the_crops(cuda):
    """The reference-shaped API (production.py:57-74: f32 crops in, labels out): `Classifier.classify` finds the constant border
    in the crops themselves (cvpce_pad_extents) -- same labels and embeddings as without skipping, and a tensor without any
    constant border (gallery-like noise) is unaffected."""
    from cvpce_amd import ops, production, synthetic
    from cvpce_amd.models import classification as C
    enc = synthetic.synthetic_macvgg(seed=1).cuda()
    gal = synthetic.gallery_images(40, seed=3)
    clf = production.Classifier(enc, synthetic.TensorGallery(gal), device=cuda, emb_device=cuda, batch_size=16, k=2)
    h0, w0 = 600, 800
    img = torch.rand(3, h0, w0, generator=torch.Generator().manual_seed(8)).cuda()
    boxes = _boxes(45, h0, w0, seed=31).cuda()
    crops = ops.crop_resize(img, boxes, S, mode=0)
    ext = ops.pad_extents(crops).cpu()
    want = ops.crop_extents(boxes, None, h0, w0, S).cpu()
    assert bool((ext <= want).all()) and bool((ext == want).float().mean() > 0.9)      # (content that happens to equal 0.5 may shorten an extent)
    noise = torch.rand(5, 3, S, S, generator=torch.Generator().manual_seed(1)).cuda()
    assert ops.pad_extents(noise).cpu().tolist() == [[S, S]] * 5
    allc = torch.cat((crops, noise))
    on_l, on_e = clf.classify(allc, return_embedding=True)
    C.SKIP_PADDING = False
    try:
        off_l, off_e = clf.classify(allc, return_embedding=True)
    finally:
        C.SKIP_PADDING = True
    assert on_l == off_l and torch.equal(on_e, off_e)
