"""dev diagnostic (GPU box): head-logit deviation HIP vs bf16 emulation vs fp32 oracle on one structured image."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cvpce_amd import synthetic
from oracle import gln as og, bf16_model as bm
torch.set_num_threads(16)
dev = torch.device('cuda:0')
det = synthetic.synthetic_gln(seed=0, detections_per_img=200)
sd = {k: v.clone() for k, v in det.state_dict().items()}
det = det.to(dev)
products = synthetic.product_images(64, seed=200)
img = synthetic.structured_shelf(0, 1024, 1024, products)[0]
eng = det.engine()
out, inter = eng.detect([img.to(dev)], 1, 200, 0.5, want_intermediates=True)
nchw = lambda y: y.float().permute(0, 3, 1, 2).cpu()
batch = og.batch_images([og.transform_one(img)])
feats_o, _ = og.backbone_forward(batch, sd)
cls_o, reg_o = og.head(feats_o, sd)
c2, c3, c4, c5 = bm.body(batch, sd)
feats_e = bm.fpn(c3, c4, c5, sd)
cls_e, reg_e = bm.heads(feats_e, sd)
def rms(a, b): return ((a - b).pow(2).mean().sqrt() / b.std()).item()
print('transform hip vs fp32', rms(nchw(inter['batch'])[:, :3], batch))
for i, (h, e) in enumerate(zip(inter['c'], (c2, c3, c4, c5))):
    print(f'C{i+2} hip vs emu', rms(nchw(h), e))
for i in range(5):
    h = nchw(inter['features'][i])
    print(f'P{i+3}: hip-emu {rms(h, feats_e[i]):.5f}  hip-orc {rms(h, feats_o[i]):.5f}  emu-orc {rms(feats_e[i], feats_o[i]):.5f}')
for i in range(5):
    h = inter['cls'][i].view(1, -1).cpu(); e = cls_e[i].view(1, -1); o = cls_o[i].view(1, -1)
    print(f'cls{i}: hip-emu {rms(h, e):.5f}  hip-orc {rms(h, o):.5f}  emu-orc {rms(e, o):.5f}  std {o.std().item():.3f}')
