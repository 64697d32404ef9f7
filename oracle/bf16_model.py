"""Oracle companion: a CPU model of the GPU path's *numerics* (TEST INFRASTRUCTURE).

oracle/gln.py and oracle/macvgg.py are the literal fp32 restatement of the reference and remain
the parity oracle.  This module evaluates the SAME graphs with the rounding points of the HIP
schedule -- BN folded into the weights, weights and inter-layer activations rounded to a 16-bit
storage type, fp32 accumulation, fp32 head/gaussian outputs -- so that a GPU-vs-CPU comparison
can be made at ~1e-3 instead of the ~1e-2 that bf16 storage costs against pure fp32.  It
separates "the kernel schedule has a bug" from "16-bit storage rounds differently", nothing more.

Round 3: the rounding points are a `Numerics` policy instead of hard-wired bf16, so that the
candidate accuracy modes of the detector can be compared on the CPU before any kernel is written
(tests/numerics_study.py): storage type of activations and weights (bf16 | fp16 | fp32), a
separate carrier type for the 16 bottleneck block outputs (the identity path) and for the FPN
top-down sums.  `BF16` is the schedule the HIP path runs by default, `FP16` its opt-in
accuracy mode (`gln(..., precision='fp16')`).
"""
from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn.functional as F

from . import gln as og
from . import macvgg as ovgg

BF = torch.bfloat16
F16_MAX = 65504.0


def _round(x, dt):
    """x rounded (RNE) to storage type dt and widened back to f32.  fp16 saturates at +-65504 like the kernels' epilogue."""
    if dt is None or dt == torch.float32:
        return x
    if dt == torch.float16:
        x = x.clamp(-F16_MAX, F16_MAX)
    return x.to(dt).to(torch.float32)


@dataclass(frozen=True)
class Numerics:
    act: Optional[torch.dtype] = BF            # inter-layer activations as stored in HBM
    wgt: Optional[torch.dtype] = BF            # packed conv weights (after BN folding)
    block_out: Optional[torch.dtype] = None    # bottleneck block outputs = the identity path; None: same as `act`
    fpn_sum: Optional[torch.dtype] = None      # FPN lateral + top-down sums (i4, i3); None: same as `act`
    name: str = 'bf16'

    def a(self, x):
        return _round(x, self.act)

    def w(self, x):
        return _round(x, self.wgt)


BF16 = Numerics()
FP16 = Numerics(act=torch.float16, wgt=torch.float16, name='fp16')
FP32 = Numerics(act=None, wgt=None, name='fp32')


def q(x):
    return x.to(BF).to(torch.float32)


def fconv(x, w, b=None, stride=1, pad=0, scale=None, shift=None, residual=None, act=0, f32_out=False, nm=BF16, out_dt='act'):
    """act(conv(x, round(w*scale)) + (b*scale + shift) + residual), output rounded to the storage type unless f32_out.
    A conv reads its input in the activation type: an input carried in a wider type (block_out / fpn_sum carriers) is
    rounded to `nm.act` at the read, as an MFMA operand would be."""
    if scale is not None:
        w = w * scale[:, None, None, None]
        b = b * scale if b is not None else None
    if shift is not None:
        b = shift if b is None else b + shift
    y = F.conv2d(nm.a(x), nm.w(w), b, stride=stride, padding=pad)
    if residual is not None:
        y = y + residual
    if act == 1:
        y = F.relu(y)
    elif act == 2:
        y = torch.tanh(y)
    if f32_out:
        return y
    return _round(y, nm.act if out_dt == 'act' else out_dt)


def _fbn(sd, p, eps=og.FROZEN_BN_EPS):
    scale = sd[p + '.weight'] * (sd[p + '.running_var'] + eps).rsqrt()
    return scale, sd[p + '.bias'] - sd[p + '.running_mean'] * scale


def _cb(sd, nm):
    return lambda x, name, stride=1, pad=0, **kw: fconv(x, sd[name + '.weight'], sd[name + '.bias'], stride, pad, nm=nm, **kw)


def _up(t, ref):
    return F.interpolate(t, size=ref.shape[-2:], mode='nearest')


@torch.no_grad()
def body(batch, sd, nm=BF16):
    """(N,3,H,W) transformed batch -> [C2, C3, C4, C5] with the GPU schedule's rounding points."""
    x = nm.a(batch)
    p = 'backbone.body'
    s, b = _fbn(sd, p + '.bn1')
    x = fconv(x, sd[p + '.conv1.weight'], None, 2, 3, s, b, act=1, nm=nm)
    x = F.max_pool2d(x, 3, 2, 1)
    carrier = nm.block_out if nm.block_out is not None else 'act'
    cs = []
    for li, nblocks in enumerate(og.RESNET50_LAYERS):
        for bi in range(nblocks):
            bp = f'{p}.layer{li + 1}.{bi}'
            stride = 2 if (bi == 0 and li > 0) else 1
            if (bp + '.downsample.0.weight') in sd:
                s, b = _fbn(sd, bp + '.downsample.1')
                identity = fconv(x, sd[bp + '.downsample.0.weight'], None, stride, 0, s, b, nm=nm, out_dt=carrier)
            else:
                identity = x
            s, b = _fbn(sd, bp + '.bn1')
            y = fconv(x, sd[bp + '.conv1.weight'], None, 1, 0, s, b, act=1, nm=nm)
            s, b = _fbn(sd, bp + '.bn2')
            y = fconv(y, sd[bp + '.conv2.weight'], None, stride, 1, s, b, act=1, nm=nm)
            s, b = _fbn(sd, bp + '.bn3')
            x = fconv(y, sd[bp + '.conv3.weight'], None, 1, 0, s, b, residual=identity, act=1, nm=nm, out_dt=carrier)
        cs.append(x)
    return cs


@torch.no_grad()
def fpn(c3, c4, c5, sd, nm=BF16):
    f = 'backbone.fpn'
    cb = _cb(sd, nm)
    sums = nm.fpn_sum if nm.fpn_sum is not None else 'act'
    i5 = cb(c5, f + '.inner_blocks.2', out_dt=sums)
    p5 = cb(i5, f + '.layer_blocks.2', 1, 1)
    i4 = cb(c4, f + '.inner_blocks.1', residual=_up(i5, c4), out_dt=sums)
    p4 = cb(i4, f + '.layer_blocks.1', 1, 1)
    i3 = cb(c3, f + '.inner_blocks.0', residual=_up(i4, c3), out_dt=sums)
    p3 = cb(i3, f + '.layer_blocks.0', 1, 1)
    p6 = cb(p5, f + '.extra_blocks.p6', 2, 1)
    p7 = cb(F.relu(p6), f + '.extra_blocks.p7', 2, 1)
    return [p3, p4, p5, p6, p7]


@torch.no_grad()
def gaussian_branch(c2, p3, sd, tanh=False, nm=BF16):
    g = 'backbone.gaussian_layer'
    x = fconv(c2, sd[g + '.lateral.weight'], sd[g + '.lateral.bias'], residual=_up(p3, c2), nm=nm)
    for blk in ('block1', 'block2'):
        bn = f'{g}.{blk}.norm'
        s = sd[bn + '.weight'] * (sd[bn + '.running_var'] + og.BN_EPS).rsqrt()
        sh = sd[bn + '.bias'] - sd[bn + '.running_mean'] * s
        x = fconv(x, sd[f'{g}.{blk}.conv.weight'], sd[f'{g}.{blk}.conv.bias'], 1, 1, s, sh, act=1, nm=nm)
    x = F.interpolate(x, scale_factor=2.0, mode='nearest')
    gs = 'backbone.gaussian_subnet.blocks'
    for i in range(5):
        w = sd[f'{gs}.{i}.conv.weight']
        last = i == 4
        x = fconv(x, w, sd[f'{gs}.{i}.conv.bias'], 1, 1 if w.shape[-1] > 1 else 0,
                  act=2 if (last and tanh) else 1, f32_out=last, nm=nm)
    return x


@torch.no_grad()
def heads(feats, sd, nm=BF16):
    cb = _cb(sd, nm)
    cls, reg = [], []
    for ft in feats:
        n, _, h, w_ = ft.shape
        for prefix, final, store, k in (('head.classification_head', 'cls_logits', cls, 1),
                                        ('head.regression_head', 'bbox_reg', reg, 4)):
            t = ft
            for i in (0, 2, 4, 6):
                t = cb(t, f'{prefix}.conv.{i}', 1, 1, act=1)
            o = cb(t, f'{prefix}.{final}', 1, 1, f32_out=True)
            store.append(o.view(n, -1, k, h, w_).permute(0, 3, 4, 1, 2).reshape(n, -1, k))
    return cls, reg


@torch.no_grad()
def macvgg_desc(x_normalised, sd):
    """x: (B,3,256,256) already scale_to_tanh'ed + normalised -> raw (B,1024) MAC descriptor (before L2 norm)."""
    x = q(x_normalised)
    d1 = None
    for idx, kind, _ in ovgg.feature_plan():
        if idx == ovgg.CUTOFF_1:
            d1 = x.amax(dim=(-2, -1))
        if kind == 'pool':
            x = F.max_pool2d(x, 2, 2)
        else:
            blk = 'block1' if idx < ovgg.CUTOFF_1 else 'block2'
            x = fconv(x, sd[f'{blk}.{idx}.weight'], sd[f'{blk}.{idx}.bias'], 1, 1, act=1)
    return torch.cat((d1, x.amax(dim=(-2, -1))), dim=1)
