"""dev tool: Classifier.build_index throughput against the number of gallery images staged per pass (host-bound: stack + pinned copy +
upload of f32 images), all settings in one process so that the host is the same."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cvpce_amd import production, synthetic
dev = torch.device('cuda')
enc = synthetic.synthetic_macvgg(seed=1).to(dev)
G = 1000
gal = synthetic.TensorGallery(synthetic.gallery_images(G, seed=100))
for rep in range(2):
    for ib in (32, 64, 256):
      production.INDEX_BATCH = ib
      for pinned in (False, True):
        production.PINNED_STAGING = pinned
        for workers in (0, 8):
            torch.cuda.synchronize(); t = time.time()
            clf = production.Classifier(enc, gal, device=dev, emb_device=dev, batch_size=8, num_workers=workers)
            torch.cuda.synchronize(); dt = time.time() - t
            print(f'INDEX_BATCH {ib:4d} pinned-staging {pinned} workers {workers}: {G / dt:7.0f} gallery images/s', flush=True)
