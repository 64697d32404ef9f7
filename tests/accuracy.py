#!/usr/bin/env python3
"""End-to-end accuracy of the HIP path against the fp32 oracle at the tolerance `north_star` states ("mAP and top-1 within
0.1 pt", "index outputs bit-exact") -- the counterpart of what the reference reports through
/root/reference/cvpce/proposals_eval.py:19-48 (AP / AR300 of the detector), cvpce/classification_eval.py:6-56 (top-k accuracy
of the matcher on ground-truth boxes) and cvpce/detection_eval.py:6-55.

TEST INFRASTRUCTURE: this is the only place besides tests/*, `__graft_entry__.smoke()` and bench.py's checker legs where
the oracle is imported; it lives under tests/ for that reason (a `tools/` script may not touch oracle/).  bench.py calls
`run()` with a bounded sample for the `parity` object of its JSON line; `python tests/accuracy.py` runs the full-size
measurement (>= 32 structured shelf images of 2048^2, galleries of 1000 and 3200 products) and writes a JSON report.

What is measured (whole bf16 HIP pipeline with the PRODUCT defaults of bench.py -- bf16 activations, bf16 distance GEMM --
against the whole fp32 oracle; nothing stage-isolated):

detection (per image: HIP detections vs the oracle's detections, cvpce_amd.metrics = the reference's metric code)
  ap50_vs_oracle / ap75_vs_oracle / ar300_vs_oracle   the oracle's detections taken as ground truth (reproduction measure;
                                                      1.0 = every oracle box found at that IoU in confidence order)
  frac_oracle_boxes_iou90                             fraction of oracle boxes with a HIP box at IoU > 0.9
  pseudo_gt.{ap50_hip, ap50_oracle, delta_pt}         ground truth := the oracle's confident boxes (score > 0.5); AP of BOTH
                                                      detectors' full outputs against it -> the "mAP delta" of north_star
  gt.{ap50_hip, ap50_oracle, delta_pt}                against the pasted products' true boxes (near zero for random weights;
                                                      reported for completeness)
matching (per gallery size G and matcher dtype)
  pairs.top1_agree / topk_agree   HIP box <-> oracle box pairs (IoU > 0.9, one to one): index the HIP pipeline matched vs the
                                  index the oracle path (its own box -> fp32 crop -> fp32 embed -> fp32 NN) matched
  gt_boxes.acc_hip / acc_oracle / delta_pt   classification_eval style: the true boxes cropped and matched by both paths,
                                  top-1 accuracy against the pasted product's id, and the difference in points
  gt_boxes.top1_agree             same crops: HIP index == oracle index
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402


FITTED_HEAD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'fitted_head.pt')


def fitted_detector(detections_per_img, precision=None):
    """The detector whose RetinaNet head was FITTED on structured shelf scenes (tests/golden/fit_head.py -> fitted_head.pt: the
    trained head tensors over the seeded base `synthetic_gln(seed=0, residual_gain=...)`): unlike the random-init detector it
    finds the products, so AP / AR against the TRUE boxes is a non-vacuous figure and its score field is bimodal like a trained
    detector's.  -> (model on the CPU, its reference-format state dict for the oracle, the fixture's recipe)."""
    from cvpce_amd import synthetic
    fx = torch.load(FITTED_HEAD, map_location='cpu')
    det = synthetic.synthetic_gln(seed=0, detections_per_img=detections_per_img, residual_gain=fx['recipe']['residual_gain'], calibrate=False,
                                  precision=precision)
    sd = det.state_dict()
    for k, v in fx['tensors'].items():
        assert k in sd and sd[k].shape == v.shape, k
        sd[k] = v.to(torch.float32)
    det.load_state_dict(sd)
    return det, {k: v.clone() for k, v in det.state_dict().items()}, fx['recipe']


def box_iou(a, b):
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = torch.max(a[:, None, :2], b[:, :2]); rb = torch.min(a[:, None, 2:], b[:, 2:])
    inter = (rb - lt).clamp(min=0).prod(dim=2)
    return inter / (area_a[:, None] + area_b - inter)


def pair_boxes(a, b, thr=0.9):
    """One-to-one pairs (i, j) with IoU(a[i], b[j]) > thr, greedy in descending IoU."""
    if not len(a) or not len(b):
        return []
    iou = box_iou(a, b)
    cand = (iou > thr).nonzero()
    order = torch.argsort(iou[cand[:, 0], cand[:, 1]], descending=True, stable=True)
    used_a, used_b, out = set(), set(), []
    for i, j in cand[order].tolist():
        if i not in used_a and j not in used_b:
            used_a.add(i); used_b.add(j); out.append((i, j))
    return out


def _ap(targets, preds, confs, thrs=(0.5, 0.75)):
    from cvpce_amd import metrics
    r = metrics.calculate_metrics(targets, preds, confs, iou_thresholds=thrs)
    return {t: {'ap': float(r[t]['ap']), 'ar_300': float(r[t]['ar_300'])} for t in thrs}


@torch.no_grad()
def oracle_embed(x_tanh, sd, device, batch=32):
    """(B,3,256,256) in [-1,1] -> (B,1024) through the oracle embedder, `device` = 'cpu' (the oracle of record) or 'cuda'
    (the SAME oracle code executed by torch's fp32 GPU kernels: used only to afford large samples, cross-checked against
    the CPU run in `run()`)."""
    from oracle import macvgg as ovgg
    sd_d = {k: v.to(device) for k, v in sd.items()}
    out = []
    for s in range(0, len(x_tanh), batch):
        out.append(ovgg.macvgg_forward(x_tanh[s:s + batch].to(device), sd_d).cpu())
    return torch.cat(out) if out else torch.empty(0, 1024)


@torch.no_grad()
def emulated_detect(img, sd, dpi):
    """The detector evaluated by oracle/bf16_model.py: the SAME graph as the fp32 oracle with the rounding points of the HIP
    schedule (BN folded, weights and inter-layer activations rounded to bf16, fp32 accumulation, fp32 head outputs), then the
    oracle's own post-processing.  Not an oracle: a CONTROL that separates "bf16 storage moves near-tie decisions" (this
    model deviates from the fp32 oracle as much as the HIP path does) from "the kernels are wrong" (the HIP path would
    deviate from this model too)."""
    from oracle import gln as og, bf16_model as bm
    x = og.transform_one(img)
    batch = og.batch_images([x])
    c2, c3, c4, c5 = bm.body(batch, sd)
    feats = bm.fpn(c3, c4, c5, sd)
    cls, reg = bm.heads(feats, sd)
    anchors = og.grid_anchors(tuple(batch.shape[-2:]), [tuple(f.shape[-2:]) for f in feats])
    b, s_, l = og.postprocess_image([c[0] for c in cls], [r[0] for r in reg], anchors, tuple(x.shape[-2:]), dpi)
    return {'boxes': og.resize_boxes(b, tuple(x.shape[-2:]), tuple(img.shape[-2:])), 'scores': s_}


@torch.no_grad()
def _ap_area(targets, preds, confs, thr=0.5):
    """All-point AP (area under the interpolated precision / recall curve) and final recall with the SAME matching code.
    The reference's 11-point AP (cvpce/metrics.py:66-73) samples precision at recall 1.0: it cannot exceed 10/11 = 0.909
    unless EVERY target is found, so it saturates as a reproduction measure; the area form does not."""
    from cvpce_amd import metrics
    r = metrics.calculate_metrics(targets, preds, confs, iou_thresholds=(thr,))[thr]['raw']
    p, rc = r['p'], r['r']
    if not len(p):
        return 0.0, 0.0
    env = torch.flip(torch.cummax(torch.flip(p, (0,)), 0).values, (0,))           # precision envelope
    prev = torch.cat((torch.zeros(1), rc[:-1]))
    return float(((rc - prev) * env).sum()), float(rc[-1])


def _detection_report(hip, orc, shelves):
    hb, hs = [h['boxes'] for h in hip], [h['scores'] for h in hip]
    ob, os_ = [o['boxes'] for o in orc], [o['scores'] for o in orc]
    vs = _ap(ob, hb, hs)
    area50, recall50 = _ap_area(ob, hb, hs, 0.5)
    area75, _ = _ap_area(ob, hb, hs, 0.75)
    pg = [o['boxes'][o['scores'] > 0.5] for o in orc]
    gt = [sh[1] for sh in shelves]
    found, total, dscore, dbox = 0, 0, [], []
    for h, o in zip(hip, orc):
        pairs = pair_boxes(h['boxes'], o['boxes'])
        found += len(pairs); total += len(o['boxes'])
        if pairs:
            i, j = torch.tensor(pairs).t()
            dscore.append((h['scores'][i] - o['scores'][j]).abs())
            dbox.append((h['boxes'][i] - o['boxes'][j]).abs().max(dim=1).values)
    dscore = torch.cat(dscore) if dscore else torch.zeros(1)
    dbox = torch.cat(dbox) if dbox else torch.zeros(1)
    a_h, a_o = _ap(pg, hb, hs), _ap(pg, ob, os_)
    g_h, g_o = _ap(gt, hb, hs), _ap(gt, ob, os_)
    pa_h, pa_o = _ap_area(pg, hb, hs, 0.5)[0], _ap_area(pg, ob, os_, 0.5)[0]
    return {
        'ap50_vs_oracle': vs[0.5]['ap'], 'ap75_vs_oracle': vs[0.75]['ap'], 'ar300_vs_oracle': vs[0.5]['ar_300'],
        'ar300_iou75_vs_oracle': vs[0.75]['ar_300'],
        'ap50_area_vs_oracle': area50, 'ap75_area_vs_oracle': area75, 'recall50_vs_oracle': recall50,
        'ap_note': 'ap50/ap75 = the reference\'s 11-point AP (cvpce/metrics.py:66-73): capped at 10/11 = 0.909 unless recall reaches 1.0; '
                   'ap*_area = all-point area under the same precision/recall curve',
        'frac_oracle_boxes_iou90': found / max(1, total), 'oracle_boxes': total,
        'paired_abs_score_diff_max': float(dscore.max()), 'paired_abs_score_diff_mean': float(dscore.mean()),
        'paired_box_diff_px_max': float(dbox.max()), 'paired_box_diff_px_mean': float(dbox.mean()),
        'count_hip': sum(len(b) for b in hb), 'count_oracle': sum(len(b) for b in ob),
        'confident_hip': sum(h['conf'] for h in hip), 'confident_oracle': sum(len(p) for p in pg),
        'pseudo_gt': {'ap50_hip': a_h[0.5]['ap'], 'ap50_oracle': a_o[0.5]['ap'], 'delta_pt': 100 * (a_h[0.5]['ap'] - a_o[0.5]['ap']),
                      'ap75_hip': a_h[0.75]['ap'], 'ap75_oracle': a_o[0.75]['ap'], 'delta75_pt': 100 * (a_h[0.75]['ap'] - a_o[0.75]['ap']),
                      'ar300_hip': a_h[0.5]['ar_300'], 'ar300_oracle': a_o[0.5]['ar_300'],
                      'delta_ar300_pt': 100 * (a_h[0.5]['ar_300'] - a_o[0.5]['ar_300']),
                      'ap50_area_hip': pa_h, 'ap50_area_oracle': pa_o, 'delta_area_pt': 100 * (pa_h - pa_o),
                      'note': 'the oracle scored against itself has recall exactly 1.0 and 11-point AP 1.0; any detector that misses one box '
                              'is capped at 10/11, so delta_pt is ~ -9 pt for every mode: delta_area_pt is the informative figure'},
        'gt': {'ap50_hip': g_h[0.5]['ap'], 'ap50_oracle': g_o[0.5]['ap'], 'delta_pt': 100 * (g_h[0.5]['ap'] - g_o[0.5]['ap']),
               'ap75_hip': g_h[0.75]['ap'], 'ap75_oracle': g_o[0.75]['ap'], 'delta75_pt': 100 * (g_h[0.75]['ap'] - g_o[0.75]['ap']),
               'ar300_hip': g_h[0.5]['ar_300'], 'ar300_oracle': g_o[0.5]['ar_300'], 'delta_ar300_pt': 100 * (g_h[0.5]['ar_300'] - g_o[0.5]['ar_300']),
               'ap50_area_hip': _ap_area(gt, hb, hs, 0.5)[0], 'ap50_area_oracle': _ap_area(gt, ob, os_, 0.5)[0],
               'true_boxes': sum(len(g_) for g_ in gt),
               'note': 'AP / AR300 of BOTH detectors against the pasted products\' TRUE boxes (cvpce/proposals_eval.py:19-48 with cvpce/metrics.py); '
                       'ap50 = the reference\'s 11-point AP (steps of 1/11 = 9.09 pt), ap50_area = area under the same precision / recall curve'}}


@torch.no_grad()
def run(n_images=32, image_size=2048, galleries=(1000, 3200), dpi=200, queries=1024, oracle_device='cpu', match_dtypes=('bf16', 'f32'),
        k=5, images_per_batch=8, seed=0, control_images=0, residual_gain=1.0, log=None, precisions=('fp16',), detector='random'):
    """precisions: detector storage modes to measure ('fp16' = the product default, 'bf16' = the opt-in); the oracle
    side is computed once.  The report's top-level `detection` / `matching` are those of precisions[0]; every mode's figures
    are under `by_precision`."""
    from cvpce_amd import ops, production, synthetic, datautils
    from oracle import gln as og, crop as ocrop, match as omatch
    log = log or (lambda *a: None)
    dev = torch.device('cuda:0')
    t_start = time.perf_counter()
    if detector == 'fitted':
        det0, det_sd, recipe = fitted_detector(dpi)
    else:
        det0 = synthetic.synthetic_gln(seed=0, detections_per_img=dpi, residual_gain=residual_gain)
        det_sd = {k_: v.clone() for k_, v in det0.state_dict().items()}
    enc = synthetic.synthetic_macvgg(seed=1)
    enc_sd = {k_: v.clone() for k_, v in enc.state_dict().items()}
    enc = enc.to(dev)
    galleries = tuple(sorted(int(g) for g in galleries))
    gmax, gmin = galleries[-1], galleries[0]
    products = synthetic.product_images(gmax, seed=200 + seed)
    gal_tanh = products * 2 - 1                                          # gallery tensors live in [-1, 1] (datautils.py:446)
    hip_gal = torch.cat([enc(gal_tanh[i:i + 128].to(dev)) for i in range(0, gmax, 128)])
    log(f'[accuracy] {gmax} products, HIP gallery embedded ({time.perf_counter() - t_start:.1f} s)')
    t = time.perf_counter()
    orc_gal = oracle_embed(gal_tanh, enc_sd, oracle_device)
    log(f'[accuracy] oracle gallery embedded on {oracle_device} ({time.perf_counter() - t:.1f} s)')
    report = {'n_images': n_images, 'image_size': image_size, 'detections_per_img': dpi, 'oracle_device_embedder': oracle_device,
              'detector_residual_gain': residual_gain, 'detector_precisions': list(precisions), 'detector': detector,
              'data': 'structured shelves (cvpce_amd.synthetic.structured_shelf), seeded random-init weights'
                      + ('; detector head fitted on such scenes (tests/golden/fit_head.py)' if detector == 'fitted' else '')}
    if detector == 'fitted':
        report['fitted_head_recipe'] = recipe
    if oracle_device != 'cpu':                                           # the GPU run of the oracle code vs its CPU run
        chk = oracle_embed(gal_tanh[:8], enc_sd, 'cpu')
        report['oracle_cuda_vs_cpu_max_abs'] = float((chk - orc_gal[:8]).abs().max())
    gal_cos = torch.nn.functional.cosine_similarity(hip_gal.cpu(), orc_gal, dim=1)
    report['gallery_embedding_cosine_min'] = float(gal_cos.min())

    shelves = [synthetic.structured_shelf(1000 * seed + i, image_size, image_size, products, pool=range(gmin)) for i in range(n_images)]
    clf = {(g, md): production.Classifier.from_embedding(enc, hip_gal[:g], list(range(g)), device=dev, emb_device=dev, k=min(k, g),
                                                         match_dtype=torch.bfloat16 if md == 'bf16' else torch.float32)
           for g in galleries for md in match_dtypes}
    first = clf[(galleries[0], match_dtypes[0])]
    t = time.perf_counter()
    orc = [og.gln_forward([sh[0]], det_sd, detections_per_img=dpi)[0] for sh in shelves]
    log(f'[accuracy] oracle detector on {n_images} images ({time.perf_counter() - t:.1f} s)')
    ob = [o['boxes'] for o in orc]

    # ---- per detector precision: the HIP pipeline (product defaults), detection agreement, paired-detection matching --------------
    report['by_precision'] = {}
    emu = None
    for prec in precisions:
        det = (fitted_detector(dpi, prec)[0] if detector == 'fitted'
               else synthetic.synthetic_gln(seed=0, detections_per_img=dpi, residual_gain=residual_gain, precision=prec)).to(dev)
        pipe = production.BatchedPipeline(det, first, 0.5)
        hip = []
        for s in range(0, n_images, images_per_batch):
            imgs = [sh[0].to(dev) for sh in shelves[s:s + images_per_batch]]
            out = pipe.run(imgs)
            emb, off = out['embeddings'], 0
            for i in range(len(imgs)):
                c, dc = int(out['count'][i]), int(out['det_count'][i])
                e = emb[off:off + c]; off += c
                idx = {key: (c_.match(e).cpu() if c else torch.empty(0, c_.k, dtype=torch.int64)) for key, c_ in clf.items()}
                hip.append({'boxes': out['boxes'][i, :dc].cpu(), 'scores': out['scores'][i, :dc].cpu(), 'conf': c, 'idx': idx})
        del pipe, det
        rp = {'detection': _detection_report(hip, orc, shelves), 'matching_pairs': {}}
        hb, hs = [h['boxes'] for h in hip], [h['scores'] for h in hip]
        if control_images and prec == 'bf16':
            t = time.perf_counter()
            m = min(control_images, n_images)
            emu = emu or [emulated_detect(sh[0], det_sd, dpi) for sh in shelves[:m]]
            eb, es = [e['boxes'] for e in emu], [e['scores'] for e in emu]
            e_vs_o, h_vs_e, h_vs_o = _ap(ob[:m], eb, es), _ap(eb, hb[:m], hs[:m]), _ap(ob[:m], hb[:m], hs[:m])
            frac = lambda a, b: sum(len(pair_boxes(x, y)) for x, y in zip(a, b)) / max(1, sum(len(y) for y in b))
            rp['detection']['control_bf16_emulation'] = {
                'images': m,
                'what': 'oracle/bf16_model.py = the fp32 oracle graph with the HIP schedule\'s bf16 rounding points, on the CPU',
                'emulation_vs_oracle': {'ap50': e_vs_o[0.5]['ap'], 'ar300': e_vs_o[0.5]['ar_300'], 'frac_boxes_iou90': frac(eb, ob[:m])},
                'hip_vs_oracle_same_images': {'ap50': h_vs_o[0.5]['ap'], 'ar300': h_vs_o[0.5]['ar_300'], 'frac_boxes_iou90': frac(hb[:m], ob[:m])},
                'hip_vs_emulation': {'ap50': h_vs_e[0.5]['ap'], 'ar300': h_vs_e[0.5]['ar_300'], 'frac_boxes_iou90': frac(hb[:m], eb)}}
            log(f'[accuracy] bf16-emulation control on {m} images ({time.perf_counter() - t:.1f} s)')
        # paired detections: HIP box <-> oracle confident box, the index each path matched
        gen = torch.Generator().manual_seed(77 + seed)
        allp = []
        for n, (h, o) in enumerate(zip(hip, orc)):
            oc = o['boxes'][o['scores'] > 0.5]
            for i, j in pair_boxes(h['boxes'][:h['conf']], oc):
                lb = oc[j].to(torch.long)
                if lb[2] > lb[0] and lb[3] > lb[1]:
                    allp.append((n, i, j))
        sel = [allp[i] for i in torch.randperm(len(allp), generator=gen)[:queries].tolist()]
        t = time.perf_counter()
        crops = []
        for n, i, j in sel:
            oc = orc[n]['boxes'][orc[n]['scores'] > 0.5]
            crops.append(ocrop.crop_boxes(shelves[n][0], oc[j:j + 1])[0])
        q_orc = oracle_embed(ocrop.scale_to_tanh(torch.stack(crops)), enc_sd, oracle_device) if crops else torch.empty(0, 1024)
        log(f'[accuracy] {prec}: oracle crop+embed of {len(sel)} paired detections ({time.perf_counter() - t:.1f} s)')
        # the true product under each paired detection (the pasted box it overlaps at IoU > 0.5), if any: end-to-end top-1 accuracy
        # of BOTH pipelines (own box -> own crop -> own embedding -> own nearest neighbour) against the pasted product's id
        true_id = []
        for n, i, j in sel:
            oc = orc[n]['boxes'][orc[n]['scores'] > 0.5]
            iou = box_iou(oc[j:j + 1], shelves[n][1])[0] if len(shelves[n][1]) else torch.zeros(0)
            true_id.append(int(shelves[n][2][int(iou.argmax())]) if len(iou) and float(iou.max()) > 0.5 else -1)
        true_id = torch.tensor(true_id, dtype=torch.int64)
        on_product = true_id >= 0
        for g in galleries:
            o_idx = omatch.nearest_neighbors(orc_gal[:g], q_orc, min(k, g)) if len(q_orc) else torch.empty(0, k, dtype=torch.int64)
            for md in match_dtypes:
                h_idx = torch.stack([hip[n]['idx'][(g, md)][i] for n, i, j in sel]) if sel else torch.empty(0, k, dtype=torch.int64)
                top1 = (h_idx[:, 0] == o_idx[:, 0]).float().mean().item() if len(sel) else None
                topk = (h_idx == o_idx[:, :1]).any(dim=1).float().mean().item() if len(sel) else None
                e = {'n': len(sel), 'top1_agree': top1, f'top{k}_contains_oracle_top1': topk}
                if int(on_product.sum()):
                    a_h = (h_idx[on_product, 0] == true_id[on_product]).float().mean().item()
                    a_o = (o_idx[on_product, 0] == true_id[on_product]).float().mean().item()
                    e['vs_true_product'] = {'n': int(on_product.sum()), 'top1_acc_hip': a_h, 'top1_acc_oracle': a_o, 'delta_pt': 100 * (a_h - a_o)}
                rp['matching_pairs'][f'G{g}_{md}'] = e
        report['by_precision'][prec] = rp
    p0 = report['by_precision'][precisions[0]]
    report['detection'] = p0['detection']

    # ---- matching: ground-truth boxes (classification_eval.py:19-42 flow; independent of the detector) -------------------
    gen = torch.Generator().manual_seed(78 + seed)
    allg = [(n, b) for n, sh in enumerate(shelves) for b in range(len(sh[1]))]
    selg = [allg[i] for i in torch.randperm(len(allg), generator=gen)[:queries].tolist()]
    t = time.perf_counter()
    gcrops = [ocrop.crop_boxes(shelves[n][0], shelves[n][1][b:b + 1])[0] for n, b in selg]
    g_orc = oracle_embed(ocrop.scale_to_tanh(torch.stack(gcrops)), enc_sd, oracle_device) if gcrops else torch.empty(0, 1024)
    log(f'[accuracy] oracle crop+embed of {len(selg)} ground-truth boxes ({time.perf_counter() - t:.1f} s)')
    g_true = torch.tensor([int(shelves[n][2][b]) for n, b in selg])
    # HIP side of the ground-truth crops: crop kernel (mode 0, like ProposalGenerator) + Classifier.classify's embed path
    g_hip_emb = []
    for n in sorted(set(n for n, _ in selg)):
        bs = [b for m, b in selg if m == n]
        cr = ops.crop_resize(shelves[n][0].to(dev).contiguous(), shelves[n][1][bs].to(dev), datautils.CLASSIFICATION_IMAGE_SIZE, mode=0)
        packed = ops.pack_embed_input(cr, True, enc.input_mean, enc.input_std)
        g_hip_emb.append((n, bs, enc.engine().embed_packed(packed)))
    order = {(n, b): None for n, b in selg}
    for n, bs, e in g_hip_emb:
        for b, row in zip(bs, e):
            order[(n, b)] = row
    g_hip = torch.stack([order[key] for key in selg]) if selg else torch.empty(0, 1024, device=dev)
    emb_cos = torch.nn.functional.cosine_similarity(g_hip.cpu(), g_orc, dim=1) if len(selg) else torch.ones(1)
    report['embedding_cosine_min_gt_crops'] = float(emb_cos.min())
    report['matching'] = {}
    for g in galleries:
        og_idx = omatch.nearest_neighbors(orc_gal[:g], g_orc, min(k, g)) if len(g_orc) else torch.empty(0, k, dtype=torch.int64)
        d = omatch.cosine_distance_matrix(orc_gal[:g], g_orc).sort(dim=1).values if len(g_orc) else torch.zeros(0, 2)
        for md in match_dtypes:
            hg_idx = clf[(g, md)].match(g_hip).cpu() if len(selg) else torch.empty(0, k, dtype=torch.int64)
            acc_h = (hg_idx[:, 0] == g_true).float().mean().item() if len(selg) else None
            acc_o = (og_idx[:, 0] == g_true).float().mean().item() if len(selg) else None
            agree = (hg_idx[:, 0] == og_idx[:, 0])
            flips = ~agree
            report['matching'][f'G{g}_{md}'] = {
                'pairs': p0['matching_pairs'][f'G{g}_{md}'],
                'gt_boxes': {'n': len(selg), 'acc_hip': acc_h, 'acc_oracle': acc_o,
                             'delta_pt': 100 * (acc_h - acc_o) if selg else None,
                             'top1_agree': agree.float().mean().item() if len(selg) else None,
                             'oracle_margin_median': float((d[:, 1] - d[:, 0]).median()) if len(selg) else None,
                             'oracle_margin_at_flips_max': float((d[:, 1] - d[:, 0])[flips].max()) if flips.any() else 0.0}}
    report['seconds'] = round(time.perf_counter() - t_start, 1)
    return report


def summary(report):
    """The handful of figures bench.py puts into its `parity` object: top level = the first detector precision of the report (bench.py
    passes the mode of its run first: fp16, the product default),
    `by_precision` = the detector agreement of every measured mode."""
    def det(d):
        return {'ap50_vs_oracle': round(d['ap50_vs_oracle'], 4), 'ap50_area_vs_oracle': round(d['ap50_area_vs_oracle'], 4),
                'ar300_vs_oracle': round(d['ar300_vs_oracle'], 4), 'frac_oracle_boxes_iou90': round(d['frac_oracle_boxes_iou90'], 4),
                'paired_box_diff_px_mean': round(d['paired_box_diff_px_mean'], 4),
                'map_delta_pt_pseudo_gt': round(d['pseudo_gt']['delta_pt'], 3), 'map_area_delta_pt_pseudo_gt': round(d['pseudo_gt']['delta_area_pt'], 3),
                'ar300_delta_pt_pseudo_gt': round(d['pseudo_gt']['delta_ar300_pt'], 3),
                'ap50_true_gt_oracle': round(d['gt']['ap50_oracle'], 4), 'ap50_true_gt_hip': round(d['gt']['ap50_hip'], 4),
                'map_delta_pt_true_gt': round(d['gt']['delta_pt'], 3),
                'map_area_delta_pt_true_gt': round(100 * (d['gt']['ap50_area_hip'] - d['gt']['ap50_area_oracle']), 3),
                'ar300_delta_pt_true_gt': round(d['gt']['delta_ar300_pt'], 3)}
    out = {'images': report['n_images'], 'detector': report.get('detector', 'random')}
    out.update(det(report['detection']))
    for key, m in report['matching'].items():
        out[key] = {'pairs': m['pairs']['n'], 'top1_agree': None if m['pairs']['top1_agree'] is None else round(m['pairs']['top1_agree'], 4),
                    'gt_crops': m['gt_boxes']['n'],
                    'gt_top1_agree': None if m['gt_boxes']['top1_agree'] is None else round(m['gt_boxes']['top1_agree'], 4),
                    'top1_acc_delta_pt': None if m['gt_boxes']['delta_pt'] is None else round(m['gt_boxes']['delta_pt'], 3)}
    out['by_precision'] = {}
    for prec, rp in report.get('by_precision', {}).items():
        e = det(rp['detection'])
        for key, m in rp['matching_pairs'].items():
            e[f'pairs_top1_agree_{key}'] = None if m['top1_agree'] is None else round(m['top1_agree'], 4)
            if 'vs_true_product' in m:
                e[f'pipeline_top1_delta_pt_{key}'] = round(m['vs_true_product']['delta_pt'], 3)
                e[f'pipeline_top1_acc_oracle_{key}'] = round(m['vs_true_product']['top1_acc_oracle'], 4)
        out['by_precision'][prec] = e
    return out


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('--images', type=int, default=32)
    ap.add_argument('--image-size', type=int, default=2048)
    ap.add_argument('--galleries', default='1000,3200')
    ap.add_argument('--queries', type=int, default=4096, help='paired detections / ground-truth crops sampled for the matching figures')
    ap.add_argument('--detections-per-img', type=int, default=200)
    ap.add_argument('--oracle-device', default='cpu', choices=['cpu', 'cuda'])
    ap.add_argument('--control-images', type=int, default=8, help='images of the bf16-emulation control (oracle/bf16_model.py)')
    ap.add_argument('--residual-gain', type=float, default=1.0, help='synthetic_gln residual_gain (conditioning of the random detector)')
    ap.add_argument('--precisions', default='fp16,bf16', help="detector storage modes to measure (first = the report's top level)")
    ap.add_argument('--detector', default='random', choices=['random', 'fitted'],
                    help='random = seeded random-init weights with the calibrated head; fitted = the head fitted on shelf scenes (tests/golden/fitted_head.pt)')
    ap.add_argument('--out', default=None)
    a = ap.parse_args()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    rep = run(a.images, a.image_size, tuple(int(g) for g in a.galleries.split(',')), a.detections_per_img, a.queries, a.oracle_device,
              control_images=a.control_images, residual_gain=a.residual_gain, log=lambda *x: print(*x, flush=True),
              precisions=tuple(a.precisions.split(',')), detector=a.detector)
    text = json.dumps(rep, indent=1)
    print(text)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        open(a.out, 'w').write(text + '\n')


if __name__ == '__main__':
    main()
