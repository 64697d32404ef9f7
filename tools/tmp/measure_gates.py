"""temporary: print what the default (fp16) mode measures on the gates VERDICT r5 item 7 asks to tighten."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from cvpce_amd import synthetic, production
from oracle import gln as og
from test_gpu_models import l2rel, box_iou
cuda = torch.device('cuda:0')
m = synthetic.synthetic_gln(seed=0, detections_per_img=200)
sd = {k: v.clone() for k, v in m.state_dict().items()}
m = m.to(cuda)

def frac(gb, rb):
    iou = box_iou(gb, rb); return (iou.max(dim=1).values > 0.9).float().mean().item()

for prec in ('fp16', 'bf16'):
    m.set_precision(prec)
    img = torch.rand(3, 640, 640, generator=torch.Generator().manual_seed(0))
    r = m([img.to(cuda)])[0]
    ref = og.gln_forward([img], sd, detections_per_img=200)[0]
    print(prec, 'config1 l2rel', l2rel(r['gaussians'].cpu(), ref['gaussians']), 'frac', frac(r['boxes'].cpu(), ref['boxes']), len(r['boxes']), len(ref['boxes']), flush=True)
    imgs = [torch.rand(3, 480, 640, generator=torch.Generator().manual_seed(1)), torch.rand(3, 700, 500, generator=torch.Generator().manual_seed(2))]
    eng = m.engine()
    out, inter = eng.detect([i.to(cuda) for i in imgs], 1, 200, 0.5, want_intermediates=True)
    boxes, scores, labels, count, conf, gauss = out
    ref, rint = og.gln_forward(imgs, sd, detections_per_img=200, return_intermediates=True)
    print(prec, 'inter gauss l2rel', l2rel(gauss.cpu(), rint['gaussians']), 'fracs', [frac(boxes[i, :int(count[i])].cpu(), ref[i]['boxes']) for i in range(2)], flush=True)
m.set_precision('fp16')
# smoke
det = synthetic.synthetic_gln(seed=0, detections_per_img=16)
det_sd = {k: v.clone() for k, v in det.state_dict().items()}
det = det.to(cuda)
img = synthetic.shelf_image(3, 512, 512)
for prec in ('fp16', 'bf16'):
    det.set_precision(prec)
    eng = det.engine()
    out = eng.detect([img.to(cuda)], 1, 16, 0.5)
    c = int(out[3][0])
    ref = og.gln_forward([img], det_sd, detections_per_img=16)[0]
    print(prec, 'smoke frac', frac(out[0][0, :c].cpu(), ref['boxes']), c, flush=True)
