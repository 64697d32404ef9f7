"""Oracle: GLN detector forward (eval mode), functional over a state dict.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows
  /root/reference/cvpce/models/proposals.py:51-139,162-203
and, for everything inherited from torchvision 0.9 (not in the reference tree:
"parity unpinned"), SURVEY.md Appendix A.  Of those inherited parts the ResNet-50
body (`resnet_body`, `bottleneck`, `frozen_bn`) is pinned to an independent third
implementation of the architecture -- transformers.ResNetModel, fixture
tests/golden/resnet_body_hf.pt made by tests/golden/make_thirdparty.py; FPN, head,
anchors, box coder, NMS and the transform have no vector to be checked against.

`sd` is a reference-format state dict (keys `backbone.body.*`, `backbone.fpn.*`,
`backbone.gaussian_layer.*`, `backbone.gaussian_subnet.*`, `head.*`).
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

# ---- torchvision 0.9 RetinaNet defaults (Appendix A) -----------------------
MIN_SIZE = 800
MAX_SIZE = 1333
IMAGE_MEAN = (0.485, 0.456, 0.406)
IMAGE_STD = (0.229, 0.224, 0.225)
SIZE_DIVISIBLE = 32
SCORE_THRESH = 0.05
NMS_THRESH = 0.5
TOPK_CANDIDATES = 1000
BBOX_XFORM_CLIP = math.log(1000.0 / 16)
FROZEN_BN_EPS = 1e-5
BN_EPS = 1e-5
ANCHOR_SIZES = tuple((x, int(x * 2 ** (1.0 / 3)), int(x * 2 ** (2.0 / 3))) for x in (32, 64, 128, 256, 512))
ASPECT_RATIOS = (0.5, 1.0, 2.0)
RESNET50_LAYERS = (3, 4, 6, 3)


# ---- transform (GeneralizedRCNNTransform, eval) ---------------------------
def transform_one(img):
    """(3,H,W) f32 in [0,1] -> normalised + resized (3,h,w)."""
    mean = torch.tensor(IMAGE_MEAN, dtype=img.dtype)[:, None, None]
    std = torch.tensor(IMAGE_STD, dtype=img.dtype)[:, None, None]
    x = (img - mean) / std
    h, w = img.shape[-2:]
    scale = float(MIN_SIZE) / float(min(h, w))
    if float(max(h, w)) * scale > MAX_SIZE:
        scale = float(MAX_SIZE) / float(max(h, w))
    x = F.interpolate(x[None], scale_factor=scale, mode='bilinear',
                      recompute_scale_factor=True, align_corners=False)[0]
    return x


def resized_size(h, w):
    """Output size of transform_one (floor(h*scale), floor(w*scale)) computed like torch does (double)."""
    scale = float(MIN_SIZE) / float(min(h, w))
    if float(max(h, w)) * scale > MAX_SIZE:
        scale = float(MAX_SIZE) / float(max(h, w))
    return int(math.floor(float(h) * scale)), int(math.floor(float(w) * scale))


def batch_images(images):
    max_h = max(i.shape[1] for i in images)
    max_w = max(i.shape[2] for i in images)
    max_h = int(math.ceil(max_h / SIZE_DIVISIBLE) * SIZE_DIVISIBLE)
    max_w = int(math.ceil(max_w / SIZE_DIVISIBLE) * SIZE_DIVISIBLE)
    out = images[0].new_zeros((len(images), 3, max_h, max_w))
    for i, img in enumerate(images):
        out[i, :, :img.shape[1], :img.shape[2]] = img
    return out


# ---- ResNet-50 body with FrozenBatchNorm2d -------------------------------
def frozen_bn(x, sd, p):
    w, b = sd[p + '.weight'], sd[p + '.bias']
    rm, rv = sd[p + '.running_mean'], sd[p + '.running_var']
    scale = w * (rv + FROZEN_BN_EPS).rsqrt()
    bias = b - rm * scale
    return x * scale[None, :, None, None] + bias[None, :, None, None]


def bottleneck(x, sd, p, stride):
    out = F.conv2d(x, sd[p + '.conv1.weight'])
    out = F.relu(frozen_bn(out, sd, p + '.bn1'))
    out = F.conv2d(out, sd[p + '.conv2.weight'], stride=stride, padding=1)  # v1.5: stride on the 3x3
    out = F.relu(frozen_bn(out, sd, p + '.bn2'))
    out = F.conv2d(out, sd[p + '.conv3.weight'])
    out = frozen_bn(out, sd, p + '.bn3')
    if (p + '.downsample.0.weight') in sd:
        identity = F.conv2d(x, sd[p + '.downsample.0.weight'], stride=stride)
        identity = frozen_bn(identity, sd, p + '.downsample.1')
    else:
        identity = x
    return F.relu(out + identity)


def resnet_body(x, sd, prefix='backbone.body'):
    x = F.conv2d(x, sd[prefix + '.conv1.weight'], stride=2, padding=3)
    x = F.relu(frozen_bn(x, sd, prefix + '.bn1'))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    feats = OrderedDict()
    for li, nblocks in enumerate(RESNET50_LAYERS):
        for bi in range(nblocks):
            stride = 2 if (bi == 0 and li > 0) else 1
            x = bottleneck(x, sd, f'{prefix}.layer{li + 1}.{bi}', stride)
        feats[str(li)] = x
    return feats


# ---- FPN + LastLevelP6P7(256,256) ----------------------------------------
def conv_b(x, sd, p, stride=1, padding=0):
    return F.conv2d(x, sd[p + '.weight'], sd[p + '.bias'], stride=stride, padding=padding)


def fpn(c_feats, sd, prefix='backbone.fpn'):
    """c_feats: [C3, C4, C5] -> OrderedDict '1','2','3','p6','p7'."""
    last_inner = conv_b(c_feats[-1], sd, f'{prefix}.inner_blocks.{len(c_feats) - 1}')
    results = [conv_b(last_inner, sd, f'{prefix}.layer_blocks.{len(c_feats) - 1}', padding=1)]
    for idx in range(len(c_feats) - 2, -1, -1):
        lateral = conv_b(c_feats[idx], sd, f'{prefix}.inner_blocks.{idx}')
        top_down = F.interpolate(last_inner, size=lateral.shape[-2:], mode='nearest')
        last_inner = lateral + top_down
        results.insert(0, conv_b(last_inner, sd, f'{prefix}.layer_blocks.{idx}', padding=1))
    p5 = results[-1]  # use_P5: LastLevelP6P7(256, 256) (proposals.py:118)
    p6 = conv_b(p5, sd, f'{prefix}.extra_blocks.p6', stride=2, padding=1)
    p7 = conv_b(F.relu(p6), sd, f'{prefix}.extra_blocks.p7', stride=2, padding=1)
    return OrderedDict(zip(['1', '2', '3', 'p6', 'p7'], results + [p6, p7]))


# ---- Gaussian branch (proposals.py:51-107) -- pinned by tests/golden/gaussian_head.pt
def gaussian_block(x, sd, p):
    x = F.conv2d(x, sd[p + '.conv.weight'], sd[p + '.conv.bias'], padding=1)
    x = F.batch_norm(x, sd[p + '.norm.running_mean'], sd[p + '.norm.running_var'],
                     sd[p + '.norm.weight'], sd[p + '.norm.bias'], training=False, eps=BN_EPS)
    return F.relu(x)


def gaussian_layer(c2, p3, sd, prefix='backbone.gaussian_layer'):
    up = lambda t: F.interpolate(t, scale_factor=2.0, mode='nearest')  # nn.Upsample(scale_factor=2)
    x = conv_b(c2, sd, prefix + '.lateral') + up(p3)
    x = gaussian_block(x, sd, prefix + '.block1')
    x = gaussian_block(x, sd, prefix + '.block2')
    return up(x)


def gaussian_subnet(x, sd, tanh=False, prefix='backbone.gaussian_subnet'):
    for i in range(5):
        w = sd[f'{prefix}.blocks.{i}.conv.weight']
        pad = 1 if w.shape[-1] > 1 else 0
        x = F.conv2d(x, w, sd[f'{prefix}.blocks.{i}.conv.bias'], padding=pad)
        x = torch.tanh(x) if (tanh and i == 4) else F.relu(x)
    return x


# ---- RetinaNet head ---------------------------------------------------------
def head_tower(x, sd, prefix, final):
    for i in (0, 2, 4, 6):
        x = F.relu(conv_b(x, sd, f'{prefix}.conv.{i}', padding=1))
    return conv_b(x, sd, f'{prefix}.{final}', padding=1)


def head(features, sd, num_classes=1):
    """-> per-level lists of (N, HWA, K) logits and (N, HWA, 4) regressions."""
    cls, reg = [], []
    for f in features:
        n, _, h, w = f.shape
        c = head_tower(f, sd, 'head.classification_head', 'cls_logits')
        c = c.view(n, -1, num_classes, h, w).permute(0, 3, 4, 1, 2).reshape(n, -1, num_classes)
        r = head_tower(f, sd, 'head.regression_head', 'bbox_reg')
        r = r.view(n, -1, 4, h, w).permute(0, 3, 4, 1, 2).reshape(n, -1, 4)
        cls.append(c)
        reg.append(r)
    return cls, reg


# ---- anchors ----------------------------------------------------------------
def base_anchors(scales, ratios=ASPECT_RATIOS):
    scales = torch.as_tensor(scales, dtype=torch.float32)
    ratios = torch.as_tensor(ratios, dtype=torch.float32)
    h_ratios = torch.sqrt(ratios)
    w_ratios = 1 / h_ratios
    ws = (w_ratios[:, None] * scales[None, :]).view(-1)
    hs = (h_ratios[:, None] * scales[None, :]).view(-1)
    return (torch.stack([-ws, -hs, ws, hs], dim=1) / 2).round()


def grid_anchors(padded_hw, grid_sizes):
    out = []
    for (gh, gw), sizes in zip(grid_sizes, ANCHOR_SIZES):
        stride_h, stride_w = padded_hw[0] // gh, padded_hw[1] // gw
        sx = torch.arange(0, gw, dtype=torch.float32) * stride_w
        sy = torch.arange(0, gh, dtype=torch.float32) * stride_h
        yy, xx = torch.meshgrid(sy, sx, indexing='ij')
        xx, yy = xx.reshape(-1), yy.reshape(-1)
        shifts = torch.stack((xx, yy, xx, yy), dim=1)
        out.append((shifts.view(-1, 1, 4) + base_anchors(sizes).view(1, -1, 4)).reshape(-1, 4))
    return out


# ---- post-processing ---------------------------------------------------------
def decode_single(rel, anchors):
    w = anchors[:, 2] - anchors[:, 0]
    h = anchors[:, 3] - anchors[:, 1]
    cx = anchors[:, 0] + 0.5 * w
    cy = anchors[:, 1] + 0.5 * h
    dx, dy = rel[:, 0], rel[:, 1]
    dw = torch.clamp(rel[:, 2], max=BBOX_XFORM_CLIP)
    dh = torch.clamp(rel[:, 3], max=BBOX_XFORM_CLIP)
    pcx = dx * w + cx
    pcy = dy * h + cy
    pw = torch.exp(dw) * w
    ph = torch.exp(dh) * h
    return torch.stack((pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw, pcy + 0.5 * ph), dim=1)


def clip_boxes(boxes, hw):
    h, w = hw
    x = boxes[:, 0::2].clamp(min=0, max=w)
    y = boxes[:, 1::2].clamp(min=0, max=h)
    return torch.stack((x[:, 0], y[:, 0], x[:, 1], y[:, 1]), dim=1)


def nms(boxes, scores, thresh, order_key=None):
    """torchvision nms: greedy, score-descending, suppress IoU > thresh (strict).

    Ties in `scores` are undefined in the reference (unstable sort).  The oracle
    and the HIP path both refine the order deterministically: by `order_key`
    (the fp32 logit, a monotone refinement of the sigmoid score -- distinct
    logits may round to one score) and then lowest-index-first (stable sort).
    Any such order is one the reference itself could produce.
    """
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    order = torch.sort(scores if order_key is None else order_key, descending=True, stable=True).indices
    b = boxes[order]
    areas = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    n = b.shape[0]
    suppressed = torch.zeros(n, dtype=torch.bool)
    keep = []
    for i in range(n):
        if suppressed[i]:
            continue
        keep.append(i)
        if i + 1 >= n:
            break
        xx1 = torch.maximum(b[i, 0], b[i + 1:, 0])
        yy1 = torch.maximum(b[i, 1], b[i + 1:, 1])
        xx2 = torch.minimum(b[i, 2], b[i + 1:, 2])
        yy2 = torch.minimum(b[i, 3], b[i + 1:, 3])
        inter = (xx2 - xx1).clamp(min=0) * (yy2 - yy1).clamp(min=0)
        iou = inter / (areas[i] + areas[i + 1:] - inter)
        suppressed[i + 1:] |= iou > thresh
    return order[torch.tensor(keep, dtype=torch.int64)]


def topk_stable(keys, k):
    """torch.topk with the tie rule fixed: by key (logit) descending, then lowest index."""
    return torch.sort(keys, descending=True, stable=True).indices[:k]


def postprocess_image(cls_levels, reg_levels, anchor_levels, image_hw, detections_per_img,
                      score_thresh=SCORE_THRESH, nms_thresh=NMS_THRESH, topk=TOPK_CANDIDATES):
    """One image: per-level lists of (HWA,K) logits / (HWA,4) regs / (HWA,4) anchors."""
    boxes, scores, labels, keys = [], [], [], []
    for logits, reg, anchors in zip(cls_levels, reg_levels, anchor_levels):
        num_classes = logits.shape[-1]
        lg = logits.flatten()
        s = torch.sigmoid(lg)
        keep = s > score_thresh
        cand = torch.where(keep)[0]
        idx = topk_stable(lg[keep], min(topk, cand.numel()))
        cand = cand[idx]
        a_idx = torch.div(cand, num_classes, rounding_mode='floor')
        labels.append(cand % num_classes)
        b = decode_single(reg[a_idx], anchors[a_idx])
        boxes.append(clip_boxes(b, image_hw))
        scores.append(s[cand])
        keys.append(lg[cand])
    boxes, scores, labels, keys = torch.cat(boxes), torch.cat(scores), torch.cat(labels), torch.cat(keys)
    # batched_nms: offsets = label * (max_coord + 1); num_classes == 1 -> all zero
    if boxes.numel():
        offs = labels.to(boxes) * (boxes.max() + 1)
        keep = nms(boxes + offs[:, None], scores, nms_thresh, order_key=keys)
    else:
        keep = torch.empty((0,), dtype=torch.int64)
    keep = keep[:detections_per_img]
    return boxes[keep], scores[keep], labels[keep]


def resize_boxes(boxes, from_hw, to_hw):
    rh = torch.tensor(to_hw[0], dtype=torch.float32) / torch.tensor(from_hw[0], dtype=torch.float32)
    rw = torch.tensor(to_hw[1], dtype=torch.float32) / torch.tensor(from_hw[1], dtype=torch.float32)
    return torch.stack((boxes[:, 0] * rw, boxes[:, 1] * rh, boxes[:, 2] * rw, boxes[:, 3] * rh), dim=1)


# ---- whole forward ----------------------------------------------------------
def backbone_forward(x, sd, tanh=False):
    """(N,3,H,W) transformed batch -> (list of 5 feature maps, gaussians (N,1,H/2,W/2))."""
    c = resnet_body(x, sd)
    c2 = c.pop('0')  # proposals.py:133
    p = fpn(list(c.values()), sd)
    feats = list(p.values())
    gl = gaussian_layer(c2, feats[0], sd)
    g = gaussian_subnet(gl, sd, tanh)
    return feats, g


@torch.no_grad()
def gln_forward(images, sd, detections_per_img=1000, tanh=False, num_classes=1, return_intermediates=False):
    """images: list of (3,H,W) f32 in [0,1] -> list of dict(boxes, scores, labels, gaussians).

    GaussianLayerNetwork.forward in eval mode (proposals.py:176-181 over RetinaNet.forward).
    """
    orig_sizes = [tuple(i.shape[-2:]) for i in images]
    resized = [transform_one(i) for i in images]
    image_sizes = [tuple(i.shape[-2:]) for i in resized]
    batch = batch_images(resized)
    feats, gauss = backbone_forward(batch, sd, tanh)
    cls, reg = head(feats, sd, num_classes)
    anchors = grid_anchors(tuple(batch.shape[-2:]), [tuple(f.shape[-2:]) for f in feats])
    results = []
    for i in range(len(images)):
        b, s, l = postprocess_image([c[i] for c in cls], [r[i] for r in reg], anchors,
                                    image_sizes[i], detections_per_img)
        b = resize_boxes(b, image_sizes[i], orig_sizes[i])
        results.append({'boxes': b, 'scores': s, 'labels': l, 'gaussians': gauss[i]})
    if return_intermediates:
        return results, {'batch': batch, 'features': feats, 'cls': cls, 'reg': reg, 'anchors': anchors,
                         'image_sizes': image_sizes, 'gaussians': gauss}
    return results
