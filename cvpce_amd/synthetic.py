"""Seeded synthetic weights / inputs for benchmarks and parity tests.

There is no network and no dataset here, so the benchmark uses random-init weights of the
reference architecture.  torchvision's cls-logit bias init (-log(99)) makes every score ~0.01 <
score_thresh, i.e. zero detections; `calibrate_head` therefore rescales the final cls layer so the
logits have O(1) spread and sets its bias so that every level saturates its top-k and more than
`detections_per_img` boxes survive NMS with score > 0.5.  The arithmetic per image is the same as
for trained weights; only the data-dependent counts are pinned.
"""
import torch

from .models import proposals, classification


def calibrate_head(model, logit_gain=8.0, logit_bias=1.0):
    ch = model.head.classification_head.cls_logits
    with torch.no_grad():
        ch.weight.mul_(logit_gain)
        ch.bias.fill_(logit_bias)
    model._engine = None
    return model


def synthetic_gln(seed=0, detections_per_img=200, tanh=False, calibrate=True):
    """Seeded GLN (CPU tensors; call .cuda() to run)."""
    state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    try:
        model = proposals.gln(pretrained_backbone=False, tanh=tanh, detections_per_img=detections_per_img)
        # give the Gaussian-branch BatchNorms non-trivial running statistics
        for m in model.backbone.gaussian_layer.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.5)
                m.running_var.uniform_(0.5, 2.0)
                m.weight.data.uniform_(0.5, 1.5)
                m.bias.data.normal_(0, 0.2)
    finally:
        torch.random.set_rng_state(state)
    return calibrate_head(model) if calibrate else model


def synthetic_macvgg(seed=1):
    state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    try:
        model = classification.macvgg_embedder('vgg16', pretrained=False)
        with torch.no_grad():
            for m in model.modules():
                if isinstance(m, torch.nn.Conv2d):
                    m.bias.normal_(0, 0.05)
    finally:
        torch.random.set_rng_state(state)
    return model


def shelf_image(seed, h=2048, w=2048, device='cpu'):
    """`torch.rand(3,H,W)` f32 -- the range of ttf.to_tensor output (datautils.py:185)."""
    g = torch.Generator().manual_seed(seed)
    return torch.rand(3, h, w, generator=g).to(device)


def gallery_images(n, seed=100, device='cpu'):
    """(n,3,256,256) in [-1,1] like the GP gallery tensors (datautils.py:446)."""
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(n, 3, 256, 256, generator=g) * 2 - 1).to(device)


class TensorGallery:
    """Minimal sample_set for Classifier.build_index: items are (image, annotation)."""

    def __init__(self, images, annotations=None):
        self.images = images
        self.annotations = annotations if annotations is not None else [f'sku_{i:05d}' for i in range(len(images))]

    def __len__(self):
        return len(self.images)

    def __getitem__(self, i):
        return self.images[i], self.annotations[i]
