"""dev tool: one detector pass (8 x 2048^2, dpi 200 and 4 x 2048^2, dpi 1000) per storage precision, graph-replayed; ms per pass."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cvpce_amd import synthetic
dev = torch.device('cuda')
for prec in ('bf16', 'fp16'):
    for n, dpi in ((8, 200), (4, 1000)):
        det = synthetic.synthetic_gln(seed=0, detections_per_img=dpi, precision=prec).to(dev)
        eng = det.engine()
        imgs = [synthetic.shelf_image(i, 2048, 2048).to(dev) for i in range(n)]
        for _ in range(4): eng.detect(imgs, 1, dpi)
        torch.cuda.synchronize()
        ts = []
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): eng.detect(imgs, 1, dpi)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20)
        print(f'{prec} {n} images dpi {dpi}: ' + ' / '.join(f'{t:.3f}' for t in ts) + ' ms', flush=True)
