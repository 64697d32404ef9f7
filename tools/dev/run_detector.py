"""dev tool: N detector passes of a batch of `n` 2048^2 images (graph-replayed as in production) for rocprofv3 traces.
usage: run_detector.py <images> <detections_per_img> [passes] [precision]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cvpce_amd import synthetic
n, dpi = int(sys.argv[1]), int(sys.argv[2])
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 6
prec = sys.argv[4] if len(sys.argv) > 4 else 'bf16'
dev = torch.device('cuda')
det = synthetic.synthetic_gln(seed=0, detections_per_img=dpi, precision=prec).to(dev)
imgs = [synthetic.shelf_image(i, 2048, 2048).to(dev) for i in range(n)]
eng = det.engine()
for _ in range(3):
    eng.detect(imgs, 1, dpi)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(passes):
    eng.detect(imgs, 1, dpi)
e1.record(); torch.cuda.synchronize()
print(f'{n} images, dpi {dpi}, {prec}: {e0.elapsed_time(e1) / passes:.3f} ms per pass', flush=True)
