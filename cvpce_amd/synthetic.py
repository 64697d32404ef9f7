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


def synthetic_gln(seed=0, detections_per_img=200, tanh=False, calibrate=True, residual_gain=1.0, precision=None):
    """Seeded GLN (CPU tensors; call .cuda() to run).

    residual_gain < 1 scales the last FrozenBN of every bottleneck (the residual branch's output gain).  Plain random init
    (gain 1) makes the ResNet body a strongly amplifying map: rounding differences grow ~10x from C2 to C5, which is not how a
    trained detector behaves; a damped init (e.g. 0.25, in the spirit of zero-gamma / Fixup initialisation) is the better-
    conditioned stand-in used by tests/accuracy.py to show how the agreement figures depend on the weights' conditioning."""
    state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    try:
        model = proposals.gln(pretrained_backbone=False, tanh=tanh, detections_per_img=detections_per_img, precision=precision)
        # give the Gaussian-branch BatchNorms non-trivial running statistics
        for m in model.backbone.gaussian_layer.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.5)
                m.running_var.uniform_(0.5, 2.0)
                m.weight.data.uniform_(0.5, 1.5)
                m.bias.data.normal_(0, 0.2)
        if residual_gain != 1.0:
            for name, m in model.backbone.body.named_modules():
                if name.endswith('bn3'):
                    m.weight.mul_(residual_gain)
    finally:
        torch.random.set_rng_state(state)
    return calibrate_head(model) if calibrate else model


def synthetic_macvgg(seed=1, batch_norm=False):
    """Seeded MAC-VGG16; batch_norm=True is `macvgg_embedder('vgg16_bn')` (the reference default, classification.py:97) with
    non-trivial eval-mode BatchNorm statistics."""
    state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    try:
        model = classification.macvgg_embedder('vgg16_bn' if batch_norm else 'vgg16', pretrained=False)
        with torch.no_grad():
            for m in model.modules():
                if isinstance(m, torch.nn.Conv2d):
                    m.bias.normal_(0, 0.05)
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.running_mean.normal_(0, 0.1)
                    m.running_var.uniform_(0.6, 1.6)
                    m.weight.uniform_(0.7, 1.3)
                    m.bias.normal_(0, 0.1)
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


def gallery_shard(start, end, seed=100, device='cpu'):
    """Gallery images [start, end) of an unbounded seeded gallery, each from its own generator: a rank materialises only the
    block it embeds (bench.py shards the gallery build), and image i is the same tensor whatever the world size."""
    out = torch.empty(max(0, end - start), 3, 256, 256)
    for j, i in enumerate(range(start, end)):
        g = torch.Generator().manual_seed(seed * 1000003 + i)
        out[j] = torch.rand(3, 256, 256, generator=g) * 2 - 1
    return out.to(device)


class TensorGallery:
    """Minimal sample_set for Classifier.build_index: items are (image, annotation)."""

    def __init__(self, images, annotations=None):
        self.images = images
        self.annotations = annotations if annotations is not None else [f'sku_{i:05d}' for i in range(len(images))]

    def __len__(self):
        return len(self.images)

    def __getitem__(self, i):
        return self.images[i], self.annotations[i]


# ---------------------------------------------------------------------------------------------------------------------
# Structured data for the accuracy measurements (tests/accuracy.py): products with distinct low-frequency appearance pasted
# on shelf rows.  Pure `rand` noise is the worst case for agreement statistics -- every gallery embedding is nearly the same
# vector, so nearest-neighbour margins are at rounding level; real product photos are not like that.
# ---------------------------------------------------------------------------------------------------------------------
def product_images(n, seed=200, size=256):
    """(n,3,size,size) in [0,1]: per product a base colour, a coarse random colour layout (bilinear-upsampled 4x4 and 16x16
    grids), a few hard-edged rectangles ("labels") and a stripe pattern of random period / orientation."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(seed)
    out = torch.empty(n, 3, size, size)
    yy, xx = torch.meshgrid(torch.arange(size, dtype=torch.float32), torch.arange(size, dtype=torch.float32), indexing='ij')
    for s0 in range(0, n, 64):                      # chunks bound the temporaries
        m = min(64, n - s0)
        base = torch.rand(m, 3, 1, 1, generator=g)
        coarse = F.interpolate(torch.rand(m, 3, 4, 4, generator=g), size=(size, size), mode='bilinear', align_corners=False)
        fine = F.interpolate(torch.rand(m, 3, 16, 16, generator=g), size=(size, size), mode='bilinear', align_corners=False)
        img = 0.45 * base + 0.35 * coarse + 0.2 * fine
        nrect = torch.randint(2, 6, (m,), generator=g).tolist()
        geo = torch.randint(0, 1 << 30, (m, 5, 4), generator=g)
        col = torch.rand(m, 5, 3, generator=g)
        for i in range(m):
            for r in range(nrect[i]):
                x0, y0 = int(geo[i, r, 0]) % (size - 32), int(geo[i, r, 1]) % (size - 32)
                w, h = 24 + int(geo[i, r, 2]) % (size // 2 - 24), 24 + int(geo[i, r, 3]) % (size // 2 - 24)
                img[i, :, y0:y0 + h, x0:x0 + w] = col[i, r][:, None, None] * 0.8 + 0.1 * fine[i, :, y0:y0 + h, x0:x0 + w]
        period = torch.rand(m, 1, 1, generator=g) * 24 + 6
        ang = torch.rand(m, 1, 1, generator=g) * 3.14159
        amp = torch.rand(m, 1, 1, generator=g) * 0.15
        stripes = torch.sin((xx[None] * torch.cos(ang) + yy[None] * torch.sin(ang)) * (6.28318 / period))
        out[s0:s0 + m] = (img + (amp * stripes)[:, None]).clamp(0, 1)
    return out


def structured_shelf(seed, h, w, products, pool=None, scale=(0.45, 0.95), background=0.32):
    """One shelf image: rows of products (rescaled copies of `products[i]`, i drawn from `pool`) standing on shelf boards.
    -> (image (3,h,w) in [0,1], gt boxes (P,4) xyxy f32, gt product ids (P,) int64)."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(seed)
    pool = torch.arange(len(products)) if pool is None else torch.as_tensor(pool)
    img = torch.full((3, h, w), background) + 0.04 * torch.rand(3, h, w, generator=g)
    boxes, ids = [], []
    size = products.shape[-1]
    y = int(torch.randint(8, 40, (1,), generator=g))
    while True:
        row_h = int(size * (scale[0] + (scale[1] - scale[0]) * float(torch.rand(1, generator=g))))
        if y + row_h + 14 > h:
            break
        x = int(torch.randint(4, 30, (1,), generator=g))
        while True:
            s = row_h - int(torch.randint(0, max(1, row_h // 5), (1,), generator=g))      # product height (stands on the board)
            ws = max(16, int(s * (0.6 + 0.5 * float(torch.rand(1, generator=g)))))         # product width
            if x + ws + 2 > w:
                break
            pid = int(pool[int(torch.randint(0, len(pool), (1,), generator=g))])
            patch = F.interpolate(products[pid][None], size=(s, ws), mode='bilinear', align_corners=False)[0]
            y0 = y + row_h - s
            img[:, y0:y0 + s, x:x + ws] = patch
            boxes.append([float(x), float(y0), float(x + ws), float(y0 + s)])
            ids.append(pid)
            x += ws + int(torch.randint(2, 12, (1,), generator=g))
        img[:, y + row_h:y + row_h + 10, :] = 0.62                                         # the shelf board
        y += row_h + 10 + int(torch.randint(6, 30, (1,), generator=g))
    return img.clamp(0, 1), torch.tensor(boxes, dtype=torch.float32).reshape(-1, 4), torch.tensor(ids, dtype=torch.int64)
