"""dev tool: wall time of one detector pass (8 x 2048^2 images) under the engine's switches."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvpce_amd import ops, synthetic
import cvpce_amd.models.proposals as P
dev = torch.device('cuda')
det = synthetic.synthetic_gln(seed=0, detections_per_img=200).to(dev)
imgs = [synthetic.shelf_image(i, 2048, 2048).to(dev) for i in range(8)]
for side in (4, 0):
    for atlas in (False, True):
        P.N_SIDE_STREAMS = side
        P.USE_HEAD_ATLAS = atlas
        det._engine = None if hasattr(det, '_engine') else None
        eng = P.GLNEngine(det, dev)
        for _ in range(3): eng.detect(imgs, 1, 200)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): eng.detect(imgs, 1, 200)
        e1.record(); torch.cuda.synchronize()
        print(f'side_streams={side} atlas={atlas}: {e0.elapsed_time(e1) / 10:.3f} ms per detector pass')
