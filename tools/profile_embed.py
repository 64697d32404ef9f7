"""dev tool: per-layer time of the MAC-VGG16 embed stage on real crops (8 images x 200 detections), HIP events per launch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvpce_amd import ops, production, synthetic
dev = torch.device('cuda')
det = synthetic.synthetic_gln(seed=0, detections_per_img=200).to(dev)
enc = synthetic.synthetic_macvgg(seed=1).to(dev)
gal = enc(synthetic.gallery_images(64, seed=100).to(dev))
clf = production.Classifier.from_embedding(enc, gal, list(range(64)), device=dev, emb_device=dev, match_dtype=torch.bfloat16)
pipe = production.BatchedPipeline(det, clf, 0.5)
imgs = [synthetic.shelf_image(i, 2048, 2048).to(dev) for i in range(8)]
for _ in range(2):
    pipe.run(imgs)
torch.cuda.synchronize()
recs = []
def wrap(name, fn, flops_fn, key_fn):
    def w(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); y = fn(*a, **k); e1.record()
        recs.append((name, key_fn(*a, **k), flops_fn(*a, **k), e0, e1))
        return y
    return w
oc = ops.conv2d
ops.conv2d = wrap('conv', oc, lambda x, pc, **k: 2.0 * x.shape[0] * x.shape[1] * x.shape[2] * pc.cout * 9 * pc.cin if pc.kh == 3 and pc.stride == 1 else 0.0,
                  lambda x, pc, **k: (tuple(x.shape[1:]), pc.cout, pc.kh, pc.stride, bool(k.get('pool'))))
ops.conv2d_relu_mac = wrap('conv+mac', ops.conv2d_relu_mac, lambda x, pc, *a, **k: 2.0 * x.shape[0] * x.shape[1] * x.shape[2] * pc.cout * 9 * pc.cin,
                           lambda x, pc, *a, **k: (tuple(x.shape[1:]), pc.cout, 3, 1, bool(k.get('pool'))))     # conv4_3 / conv5_3 with the fused MAC epilogue
ops.vgg_stem = wrap('stem', ops.vgg_stem, lambda x, ps: ps.flops_per_pixel * x.shape[0] * x.shape[1] * x.shape[2], lambda x, ps: tuple(x.shape[1:]))
ops.maxpool2d = wrap('maxpool', ops.maxpool2d, lambda x, *a, **k: 0.0, lambda x, *a, **k: tuple(x.shape[1:]))
ops.global_max_into = wrap('global_max', ops.global_max_into, lambda x, *a: 0.0, lambda x, *a: tuple(x.shape[1:]))
ops.l2_normalize = wrap('l2norm', ops.l2_normalize, lambda x, *a, **k: 0.0, lambda x, *a, **k: tuple(x.shape[1:]))
ops.crop_resize = wrap('crop', ops.crop_resize, lambda *a, **k: 0.0, lambda *a, **k: 'crop')
n = 3
for _ in range(n):
    det_out = det.engine().detect(imgs, 1, 200, 0.5)
    counts = det_out[4].tolist()
    del recs[:]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    crops, valid, sel = pipe._crop_embed_match(imgs, det_out, counts)
    emb = enc.engine().embed_packed(valid)
    e1.record()
    torch.cuda.synchronize()
agg = {}
for name, key, fl, a, b in recs:
    d = agg.setdefault((name, key), [0, 0.0, 0.0]); d[0] += 1; d[1] += a.elapsed_time(b); d[2] += fl
print(f'crop + embed of {valid.shape[0]} crops: {e0.elapsed_time(e1):.3f} ms (with per-launch events)')
tot = 0.0
for (name, key), (cnt, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    tot += ms
    print(f'{name:10s} {str(key):44s} x{cnt:3d} {ms:8.3f} ms  {fl / ms / 1e9 if ms else 0:7.1f} TF')
print('sum', tot)
