"""Oracle: MACResNet encoder forward (ResNet-50 v1.5 + eval BatchNorm), functional over a state dict.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows /root/reference/cvpce/models/classification.py:53-85,111-121; the
ResNet-50 itself is torchvision 0.9 ("parity unpinned"): Bottleneck with the stride on the 3x3 conv,
downsample = 1x1 conv (stride) + norm on the first block of every layer, BatchNorm2d eps 1e-5.

State-dict keys (descriptor layers [2, 3]): blocks.0.0.{0,1} = conv1 / bn1, blocks.0.1.* = layer1, blocks.0.2.* = layer2,
blocks.1.0.* = layer3 (Sequential nesting of classification.py:72-76)."""
import torch
import torch.nn.functional as F

LAYERS = (3, 4, 6, 3)
EPS = 1e-5


def _bn(x, sd, prefix):
    if prefix + '.weight' not in sd:          # norm_layer = nn.Identity
        return x
    return F.batch_norm(x, sd[prefix + '.running_mean'], sd[prefix + '.running_var'], sd[prefix + '.weight'], sd[prefix + '.bias'],
                        training=False, eps=EPS)


def _bottleneck(x, sd, p, stride, has_ds):
    idt = x
    y = F.relu(_bn(F.conv2d(x, sd[p + '.conv1.weight']), sd, p + '.bn1'))
    y = F.relu(_bn(F.conv2d(y, sd[p + '.conv2.weight'], stride=stride, padding=1), sd, p + '.bn2'))
    y = _bn(F.conv2d(y, sd[p + '.conv3.weight']), sd, p + '.bn3')
    if has_ds:
        idt = _bn(F.conv2d(x, sd[p + '.downsample.0.weight'], stride=stride), sd, p + '.downsample.1')
    return F.relu(y + idt)


@torch.no_grad()
def macresnet_forward(x, sd, desc_layers=(2, 3), eps=1e-8, layers=LAYERS):
    """(B,3,H,W) f32 -> (B, sum of descriptor-layer channels) unit-norm MAC descriptors.  `layers`: bottlenecks per stage
    (ResNet-50: 3, 4, 6, 3; the reference-made fixture tests/golden/members.pt uses a smaller hand-built source); channel widths
    come from the state dict.  Pinned by that fixture: key nesting, per-block amax, concatenation order, L2 normalisation."""
    descs, prev = [], 0
    for bi, l in enumerate(desc_layers):
        for pos, layer in enumerate(range(prev, l + 1)):
            p = f'blocks.{bi}.{pos}'
            if layer == 0:
                x = F.relu(_bn(F.conv2d(x, sd[p + '.0.weight'], stride=2, padding=3), sd, p + '.1'))
                x = F.max_pool2d(x, 3, 2, 1)
            else:
                for b in range(layers[layer - 1]):
                    x = _bottleneck(x, sd, f'{p}.{b}', 2 if (b == 0 and layer > 1) else 1, b == 0)
        descs.append(x.amax(dim=(-2, -1)))
        prev = l + 1
    desc = torch.cat(descs, dim=1)
    return desc / torch.linalg.norm(desc, dim=1, keepdim=True).clamp(min=eps)
