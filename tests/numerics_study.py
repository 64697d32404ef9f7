#!/usr/bin/env python3
"""CPU study of the detector's candidate numerics modes (round-2 review, "close the detector parity gap", step A).

TEST INFRASTRUCTURE (imports oracle/): for each `Numerics` policy of oracle/bf16_model.py the whole detector is evaluated on
the CPU on the control images of tests/accuracy.py (structured shelves, seeded random-init weights) and compared with the fp32
oracle (oracle/gln.py): rms error of the head logits relative to their spread, fraction of the oracle's boxes reproduced at
IoU > 0.9, AP50 / AR300 of the emulated detections scored against the oracle's.  No GPU, no HIP library: this decides
which mode is worth writing kernels for.

    python tests/numerics_study.py --images 8 --out profiles/r03_numerics_study.json
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

F16, BF, F32 = torch.float16, torch.bfloat16, torch.float32


def policies():
    from oracle.bf16_model import Numerics
    return [
        Numerics(name='bf16 (round-2 schedule)'),
        Numerics(block_out=F32, name='bf16 + fp32 carriers for the 16 block outputs / identity path'),
        Numerics(block_out=F16, name='bf16 + fp16 carriers for the 16 block outputs / identity path'),
        Numerics(fpn_sum=F32, name='bf16 + fp32 FPN top-down sums'),
        Numerics(block_out=F32, fpn_sum=F32, name='bf16 + fp32 block outputs + fp32 FPN sums'),
        Numerics(act=BF, wgt=F32, name='bf16 activations, exact weights'),
        Numerics(act=F32, wgt=BF, name='exact activations, bf16 weights'),
        Numerics(act=F16, wgt=BF, name='fp16 activations, bf16 weights (not an MFMA operand pair; for the record)'),
        Numerics(act=F16, wgt=F16, name='fp16 storage for the whole detector'),
    ]


@torch.no_grad()
def emulate(img, sd, dpi, nm):
    from oracle import gln as og, bf16_model as bm
    x = og.transform_one(img)
    batch = og.batch_images([x])
    c2, c3, c4, c5 = bm.body(batch, sd, nm)
    feats = bm.fpn(c3, c4, c5, sd, nm)
    cls, reg = bm.heads(feats, sd, nm)
    anchors = og.grid_anchors(tuple(batch.shape[-2:]), [tuple(f.shape[-2:]) for f in feats])
    b, s_, _ = og.postprocess_image([c[0] for c in cls], [r[0] for r in reg], anchors, tuple(x.shape[-2:]), dpi)
    return {'boxes': og.resize_boxes(b, tuple(x.shape[-2:]), tuple(img.shape[-2:])), 'scores': s_,
            'cls': torch.cat([c.flatten() for c in cls]), 'reg': torch.cat([r.flatten() for r in reg])}


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('--images', type=int, default=8)
    ap.add_argument('--image-size', type=int, default=1024)
    ap.add_argument('--detections-per-img', type=int, default=200)
    ap.add_argument('--out', default=None)
    a = ap.parse_args()
    torch.set_num_threads(os.cpu_count() or 1)
    import accuracy                                   # tests/accuracy.py: pair_boxes, _ap (cvpce_amd.metrics)
    from cvpce_amd import metrics                     # noqa: F401  host-only module, imports without the HIP library
    import importlib
    synthetic = importlib.import_module('cvpce_amd.synthetic')
    from oracle import gln as og
    det = synthetic.synthetic_gln(seed=0, detections_per_img=a.detections_per_img)
    sd = {k: v.clone() for k, v in det.state_dict().items()}
    products = synthetic.product_images(256, seed=200)
    shelves = [synthetic.structured_shelf(i, a.image_size, a.image_size, products) for i in range(a.images)]
    t0 = time.perf_counter()
    orc = []
    for sh in shelves:
        res, inter = og.gln_forward([sh[0]], sd, detections_per_img=a.detections_per_img, return_intermediates=True)
        orc.append({'boxes': res[0]['boxes'], 'scores': res[0]['scores'],
                    'cls': torch.cat([c.flatten() for c in inter['cls']]), 'reg': torch.cat([r.flatten() for r in inter['reg']])})
    print(f'[study] fp32 oracle on {a.images} images: {time.perf_counter() - t0:.1f} s', flush=True)
    rows = []
    for nm in policies():
        t0 = time.perf_counter()
        emu = [emulate(sh[0], sd, a.detections_per_img, nm) for sh in shelves]
        ob, eb, es = [o['boxes'] for o in orc], [e['boxes'] for e in emu], [e['scores'] for e in emu]
        apr = accuracy._ap(ob, eb, es)
        found = sum(len(accuracy.pair_boxes(x, y)) for x, y in zip(eb, ob))
        rel = lambda key: float(torch.stack([(e[key] - o[key]).pow(2).mean().sqrt() / o[key].std() for e, o in zip(emu, orc)]).mean())
        row = {'policy': nm.name, 'act': str(nm.act), 'wgt': str(nm.wgt), 'block_out': str(nm.block_out), 'fpn_sum': str(nm.fpn_sum),
               'cls_logit_rms_rel': rel('cls'), 'box_reg_rms_rel': rel('reg'),
               'frac_oracle_boxes_iou90': found / max(1, sum(len(b) for b in ob)),
               'ap50_vs_oracle': apr[0.5]['ap'], 'ap75_vs_oracle': apr[0.75]['ap'], 'ar300_vs_oracle': apr[0.5]['ar_300']}
        rows.append(row)
        print(f"[study] {nm.name}: logit rms {100 * row['cls_logit_rms_rel']:.3f} %  boxes@0.9 {100 * row['frac_oracle_boxes_iou90']:.2f} %  "
              f"AP50 {row['ap50_vs_oracle']:.4f}  ({time.perf_counter() - t0:.1f} s)", flush=True)
    rep = {'images': a.images, 'image_size': a.image_size, 'detections_per_img': a.detections_per_img,
           'data': 'structured shelves, seeded random-init weights (cvpce_amd.synthetic)', 'policies': rows}
    if a.out:
        with open(a.out, 'w') as f:
            f.write(json.dumps(rep, indent=1) + '\n')
    return rep


if __name__ == '__main__':
    main()
