"""dev tool: per-layer conv timing on the REAL activations of the pipeline (random-data figures do not transfer: the chip clocks
by data).  `capture` runs the pipeline once and saves each VGG layer's input (one 256-crop pass) to /tmp; `time` loads them and
times ops.conv2d (use CVPCE_LIB=<ablation build> to time a side library)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cvpce_amd import ops, production, synthetic
dev = torch.device('cuda')
mode = sys.argv[1]
enc = synthetic.synthetic_macvgg(seed=1).to(dev)
eng = enc.engine()
names = ['conv2_1', 'conv2_2', 'conv3_1', 'conv3_2', 'conv3_3', 'conv4_1', 'conv4_2', 'conv4_3', 'conv5_1', 'conv5_2', 'conv5_3']
if mode == 'capture':
    det = synthetic.synthetic_gln(seed=0, detections_per_img=200).to(dev)
    gal = enc(synthetic.gallery_images(64, seed=100).to(dev))
    clf = production.Classifier.from_embedding(enc, gal, list(range(64)), device=dev, emb_device=dev, match_dtype=torch.bfloat16)
    pipe = production.BatchedPipeline(det, clf, 0.5)
    imgs = [synthetic.shelf_image(i, 2048, 2048).to(dev) for i in range(2)]
    det_out = det.engine().detect(imgs, 1, 200, 0.5)
    crops, valid, sel = pipe._crop_embed_match(imgs, det_out, det_out[4].tolist())
    xb = ops.vgg_stem(valid[:256].contiguous(), eng.stem)
    torch.save(valid[:256].contiguous().cpu(), '/tmp/real_stem_in.pt')
    i = 0
    for kind, pc in eng.plan:
        if kind in ('conv', 'conv_pool'):
            torch.save(xb.cpu(), f'/tmp/real_{names[i]}.pt'); i += 1
            xb = ops.conv2d(xb, pc, act=1, pool=kind == 'conv_pool')
        elif kind == 'pool':
            xb = ops.maxpool2d(xb, 2, 2)
    print('captured', i, 'layer inputs; zero fraction of conv3_2 input:', float((torch.load('/tmp/real_conv3_2.pt') == 0).float().mean()))
else:
    want = sys.argv[2].split(',') if len(sys.argv) > 2 else names
    convs = [(k, pc) for k, pc in eng.plan if k in ('conv', 'conv_pool')]
    if 'stem' in want:
        x = torch.load('/tmp/real_stem_in.pt').to(dev)
        for _ in range(3):
            ops.vgg_stem(x, eng.stem)
        torch.cuda.synchronize()
        import time
        t_end = time.perf_counter() + float(os.environ.get('REAL_LAYER_SECONDS', '0'))
        while True:                                   # (settled clocks: see the conv loop below)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.vgg_stem(x, eng.stem)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            if time.perf_counter() >= t_end:
                break
        print(f'stem {ms:.3f} ms {eng.stem.flops_per_pixel * 256 * 65536 / ms / 1e9:7.1f} TF', flush=True)
    for nm, (kind, pc) in zip(names, convs):
        if nm not in want:
            continue
        x = torch.load(f'/tmp/real_{nm}.pt').to(dev)
        for _ in range(3):
            ops.conv2d(x, pc, act=1, pool=kind == 'conv_pool')
        torch.cuda.synchronize()
        # groups of 10 launches back to back for REAL_LAYER_SECONDS (default: one group); the LAST group counts -- the card
        # needs ~1 s under load to settle at the clock its power cap allows (a cold group reads 15-20 % low)
        import time
        t_end = time.perf_counter() + float(os.environ.get('REAL_LAYER_SECONDS', '0'))
        while True:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.conv2d(x, pc, act=1, pool=kind == 'conv_pool')
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            if time.perf_counter() >= t_end:
                break
        n, h, w, c = x.shape
        print(f'{nm} {ms:.3f} ms {2.0 * n * h * w * pc.cout * 9 * pc.cin / ms / 1e9:7.1f} TF', flush=True)
