import torch, sys
sys.path.insert(0,'.')
from cvpce_amd import synthetic
from oracle import gln as og
def nchw(y): return y.float().permute(0,3,1,2).cpu()
def rel(a,b): return ((a-b).abs().max()/b.abs().max()).item(), ((a-b).norm()/b.norm()).item()
m = synthetic.synthetic_gln(seed=0, detections_per_img=200)
sd = {k:v.clone() for k,v in m.state_dict().items()}
m = m.cuda()
img = torch.rand(3,640,640,generator=torch.Generator().manual_seed(0))
eng = m.engine()
out, inter = eng.detect([img.cuda()],1,200,0.5,want_intermediates=True)
ref, rint = og.gln_forward([img], sd, 200, return_intermediates=True)
cfe = og.resnet_body(rint['batch'], sd)
for i,(g,w) in enumerate(zip(inter['c'], cfe.values())): print('C',i+2, rel(nchw(g), w))
for i,(g,w) in enumerate(zip(inter['features'], rint['features'])): print('P',i+3, rel(nchw(g), w))
for i,(g,w) in enumerate(zip(inter['cls'], rint['cls'])): print('cls',i, (g.view(1,-1).cpu()-w.view(1,-1)).abs().max().item(), w.std().item())
for i,(g,w) in enumerate(zip(inter['reg'], rint['reg'])): print('reg',i, (g.view(1,-1).cpu()-w.view(1,-1)).abs().max().item(), w.std().item())
print('gauss e2e', rel(out[5].cpu(), rint['gaussians']))
# stage isolated: oracle gaussian branch on GPU's own C2 / P3
c2 = nchw(inter['c'][0]); p3 = nchw(inter['features'][0])
gl = og.gaussian_layer(c2, p3, sd); gs = og.gaussian_subnet(gl, sd, False)
print('gauss isolated', rel(out[5].cpu(), gs))
x = og.conv_b(c2, sd, 'backbone.gaussian_layer.lateral') + torch.nn.functional.interpolate(p3, scale_factor=2.0)
print('lateral std', x.std().item(), 'block1 out std', og.gaussian_block(x, sd, 'backbone.gaussian_layer.block1').std().item(), 'final std', gs.std().item(), gs.max().item())

from oracle import bf16_model as bm
m16 = bm.gln_heads(nchw(inter['batch'])[:, :3], sd)
for i,(g,w) in enumerate(zip(inter['c'], m16['c'])): print('bf16model C',i+2, rel(nchw(g), w))
for i,(g,w) in enumerate(zip(inter['features'], m16['features'])): print('bf16model P',i+3, rel(nchw(g), w))
for i,(g,w) in enumerate(zip(inter['cls'], m16['cls'])): print('bf16model cls',i, (g.view(1,-1).cpu()-w.view(1,-1)).abs().max().item())
for i,(g,w) in enumerate(zip(inter['reg'], m16['reg'])): print('bf16model reg',i, (g.view(1,-1).cpu()-w.view(1,-1)).abs().max().item())
print('bf16model gauss', rel(out[5].cpu(), m16['gaussians']))
