"""Oracle: MACVGG embedder forward (VGG16 cfg 'D', with or without BatchNorm), functional over a state dict.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows
/root/reference/cvpce/models/classification.py:20-51; the VGG cfg 'D' feature
stack itself is torchvision 0.9 ("parity unpinned").

State-dict keys keep torchvision's `features` indices after slicing
(classification.py:36-37): without BN block1.{0,2,5,7,10,12,14,17,19,21}, block2.{24,26,28};
with BN (`macvgg_embedder('vgg16_bn')`, the reference default, classification.py:97) every conv at index i is
followed by its BatchNorm2d at i+1 and ReLU at i+2 (44 `features` modules), cut-offs 33 / 43
(classification.py:29-37: layers_per_conv = 3).  The variant is read off the state dict (running_mean keys).
"""
import torch
import torch.nn.functional as F

VGG_CFG_D = (64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M')
CONVS_PER_BLOCK = (2, 2, 3, 3, 3)
BN_EPS = 1e-5


def cutoffs(batch_norm):
    """classification.py:29-33 -> (cutoff_1, cutoff_2): (23, 30) without BN, (33, 43) with."""
    per_conv = 3 if batch_norm else 2
    per_block = [c * per_conv + 1 for c in CONVS_PER_BLOCK]
    return sum(per_block[:-1]) - 1, sum(per_block) - 1


CUTOFF_1, CUTOFF_2 = cutoffs(False)  # features[:23] = conv1_1 .. relu4_3; features[23:30] = pool4, conv5_1 .. relu5_3
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
EMBEDDING_SIZE = 1024


def feature_plan(batch_norm=False):
    """[(features_index, 'conv'|'pool', cout)] for indices < cutoff_2 ([BatchNorm and] ReLU follow every conv)."""
    plan, idx = [], 0
    for v in VGG_CFG_D:
        if v == 'M':
            plan.append((idx, 'pool', None)); idx += 1
        else:
            plan.append((idx, 'conv', v)); idx += 3 if batch_norm else 2
    return [p for p in plan if p[0] < cutoffs(batch_norm)[1]]


def has_batch_norm(sd):
    return any(k.endswith('running_mean') for k in sd)


def normalize_tanh(x):
    """ttf.normalize with mean 2m-1, std 2s (classification.py:41-44); x in [-1,1]."""
    mean = torch.tensor([m * 2 - 1 for m in IMAGENET_MEAN], dtype=x.dtype, device=x.device)[None, :, None, None]
    std = torch.tensor([s * 2 for s in IMAGENET_STD], dtype=x.dtype, device=x.device)[None, :, None, None]
    return (x - mean) / std


@torch.no_grad()
def macvgg_forward(x, sd, eps=1e-8, return_descs=False):
    """(B,3,256,256) f32 in [-1,1] -> (B,1024) unit-norm MAC descriptors."""
    x = normalize_tanh(x)
    desc_1 = None
    bn = has_batch_norm(sd)
    cut1, _ = cutoffs(bn)
    for idx, kind, _ in feature_plan(bn):
        if idx == cut1:
            desc_1 = x.amax(dim=(-2, -1))
        if kind == 'pool':
            x = F.max_pool2d(x, kernel_size=2, stride=2)
        else:
            blk = 'block1' if idx < cut1 else 'block2'
            x = F.conv2d(x, sd[f'{blk}.{idx}.weight'], sd[f'{blk}.{idx}.bias'], padding=1)
            if bn:    # nn.BatchNorm2d in eval mode (torchvision vgg.make_layers(batch_norm=True))
                q = f'{blk}.{idx + 1}'
                x = F.batch_norm(x, sd[q + '.running_mean'], sd[q + '.running_var'], sd[q + '.weight'], sd[q + '.bias'],
                                 training=False, eps=BN_EPS)
            x = F.relu(x)
    desc_2 = x.amax(dim=(-2, -1))
    desc = torch.cat((desc_1, desc_2), dim=1)
    out = desc / torch.linalg.norm(desc, dim=1, keepdim=True).clamp(min=eps)
    return (out, desc) if return_descs else out
