"""dev tool: run under `rocprofv3 --kernel-trace --output-format csv`; one warm pipeline step is isolated by a 1 s gap."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvpce_amd import production, synthetic
dev = torch.device('cuda')
det = synthetic.synthetic_gln(seed=0, detections_per_img=200).to(dev)
enc = synthetic.synthetic_macvgg(seed=1).to(dev)
gal = torch.nn.functional.normalize(torch.rand(3200, 1024), dim=1).to(dev)
clf = production.Classifier.from_embedding(enc, gal, [str(i) for i in range(3200)], device=dev, emb_device=dev)
pipe = production.BatchedPipeline(det, clf, 0.5)
imgs = [synthetic.shelf_image(i, 2048, 2048).to(dev) for i in range(8)]
pipe.run(imgs); pipe.run(imgs)
torch.cuda.synchronize(); time.sleep(1.0)
pipe.run(imgs)
torch.cuda.synchronize()
