"""Oracle: MACVGG embedder forward (VGG16 without BN), functional over a state dict.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows
/root/reference/cvpce/models/classification.py:20-51; the VGG cfg 'D' feature
stack itself is torchvision 0.9 ("parity unpinned").

State-dict keys keep torchvision's `features` indices after slicing
(classification.py:36-37): block1.{0,2,5,7,10,12,14,17,19,21}, block2.{24,26,28}.
"""
import torch
import torch.nn.functional as F

VGG_CFG_D = (64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M')
CUTOFF_1 = 23  # features[:23]  = conv1_1 .. relu4_3
CUTOFF_2 = 30  # features[23:30] = pool4, conv5_1 .. relu5_3 (final pool dropped)
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
EMBEDDING_SIZE = 1024


def feature_plan():
    """[(features_index, 'conv'|'pool', cout)] for indices < CUTOFF_2 (ReLU follows every conv)."""
    plan, idx = [], 0
    for v in VGG_CFG_D:
        if v == 'M':
            plan.append((idx, 'pool', None)); idx += 1
        else:
            plan.append((idx, 'conv', v)); idx += 2
    return [p for p in plan if p[0] < CUTOFF_2]


def normalize_tanh(x):
    """ttf.normalize with mean 2m-1, std 2s (classification.py:41-44); x in [-1,1]."""
    mean = torch.tensor([m * 2 - 1 for m in IMAGENET_MEAN], dtype=x.dtype)[None, :, None, None]
    std = torch.tensor([s * 2 for s in IMAGENET_STD], dtype=x.dtype)[None, :, None, None]
    return (x - mean) / std


@torch.no_grad()
def macvgg_forward(x, sd, eps=1e-8, return_descs=False):
    """(B,3,256,256) f32 in [-1,1] -> (B,1024) unit-norm MAC descriptors."""
    x = normalize_tanh(x)
    desc_1 = None
    for idx, kind, _ in feature_plan():
        if idx == CUTOFF_1:
            desc_1 = x.amax(dim=(-2, -1))
        if kind == 'pool':
            x = F.max_pool2d(x, kernel_size=2, stride=2)
        else:
            blk = 'block1' if idx < CUTOFF_1 else 'block2'
            x = F.relu(F.conv2d(x, sd[f'{blk}.{idx}.weight'], sd[f'{blk}.{idx}.bias'], padding=1))
    desc_2 = x.amax(dim=(-2, -1))
    desc = torch.cat((desc_1, desc_2), dim=1)
    out = desc / torch.linalg.norm(desc, dim=1, keepdim=True).clamp(min=eps)
    return (out, desc) if return_descs else out
