"""MAC-VGG16 product embedder + cosine matcher on MI355X -- drop-in for
/root/reference/cvpce/models/classification.py (`MACVGG`, `macvgg_embedder`,
`distance`, `nearest_neighbors`, and the optional `MACResNet` / `macresnet_encoder`
at the end of the file; inference only -- the GAN wrappers are training code).

nn.Modules hold parameters under the reference's state-dict keys
(`block1.{0,2,5,...}`, `block2.{24,26,28}`: torchvision `features` indices
survive the slicing at classification.py:36-37); the forward pass is a schedule
of HIP kernels: 13 MFMA implicit-GEMM 3x3 convolutions with fused bias+ReLU,
4 max-pools, two global-max (MAC) reductions and an L2 normalisation.
"""
from collections import OrderedDict

import torch
from torch import nn

from .. import ops

VGG_CFGS = {'D': (64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M')}
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
# classification.py:41-44: ImageNet statistics rescaled to the [-1, 1] input range
TANH_MEAN = tuple(m * 2 - 1 for m in IMAGENET_MEAN)
TANH_STD = tuple(s * 2 for s in IMAGENET_STD)
INPUT_SIZE = 256
USE_FUSED_STEM = True   # A/B switch (tests compare the fused stem against the unfused conv kernels)
import os as _os
MAX_EMBED_BATCH = int(_os.environ.get('CVPCE_EMBED_BATCH', 256))  # crops per kernel schedule pass (keeps every NHWC tensor < 2^31 elements)
# ... with the fused stem the largest tensor of a pass is conv2_x's output (128 x 128 x 128 per crop), so a pass can take 3x as
# many crops under the 4 GB buffer-offset limit.  Whole multiples of 256 keep every layer's workgroup count a multiple of the CU
# count (conv5_x: two workgroups per crop); measured on the pipeline bench, same call: 256 -> 156.4, 512 -> 158.1, 768 -> 158.6,
# 800 -> 157.6, 400 -> 153.5, 200 -> 150.6 images/s.
FUSED_EMBED_BATCH = int(_os.environ.get('CVPCE_EMBED_BATCH', 768))
FUSED_EMBED_MAX = int(_os.environ.get('CVPCE_EMBED_MAX', 960))     # largest pass the 32-bit tensor limits allow with the fused stem (conv2_x output: 960 * 128 * 128 * 128 < 2^31)
FUSED_EMBED_BATCH = min(FUSED_EMBED_BATCH, FUSED_EMBED_MAX)        # (an over-large CVPCE_EMBED_BATCH would hit the kernels' 32-bit guards instead of splitting the pass)
# Constant-padding tile skipping (csrc/skiplist.hip): `resize_for_classification` (datautils.py:232-239) pads every crop to a square
# with 0.5, so the part of the 256 x 256 embedder input below / right of the box content is the same constant in every crop; conv
# tiles whose receptive field lies inside it are not computed (bit-identical results: tests/test_gpu_skip.py).  A/B switch.
SKIP_PADDING = _os.environ.get('CVPCE_SKIP_PADDING', '1') != '0'
LATE_EMBED_MAX = int(_os.environ.get('CVPCE_LATE_EMBED_MAX', 2000))   # crops per pass of the late layers (conv3_1 on) of the work-list schedule
# ... and the LISTED tiles are cut at their last non-constant row (list entries carry `rows` in {4, 8, 12, 16}; the halo kernels stop
# streaming patch rows there).  Pays since the work-list kernels take CONTIGUOUS blocks of the list (with the old strided
# assignment a workgroup got nothing but cut tiles or nothing but full ones): same call, 29.15 / 29.41 -> 27.98 ms per 1 600
# bench-shaped crops.  A/B switch; tests/test_gpu_skip.py runs both settings.
SKIP_ROWS = _os.environ.get('CVPCE_SKIP_ROWS', '1') != '0'
# ... and the two layers that carry the MAC maximum (conv4_3, conv5_3; classification.py:46-49) skip like the others: what their lists
# leave out is the constant crop's own map there, whose row / column suffix maxima are tabulated once per engine and start the
# descriptor (`cvpce_mac_init`) before the kernels take their atomic maxima over what they do compute.  A/B switch.
SKIP_MAC = _os.environ.get('CVPCE_SKIP_MAC', '1') != '0'
# ... and tiles of which only the first 4 rows are not constant (a crop's content ending just below a tile boundary: conv4_2 / conv4_3
# on the bench's boxes) are computed three at a time (`cvpce_conv3x3_halo_strips`): alone such a tile issues a third of the MFMAs of a
# full one against the same weight stream, which bounds it.  A/B switch.
SKIP_STRIPS = _os.environ.get('CVPCE_SKIP_STRIPS', '1') != '0'


def _passes(n, step, longest):
    """[(start, end), ...] covering n crops in passes of `step`; a short tail is folded into the pass before it when the sum stays
    within `longest` (1600 crops: 768 + 832 instead of 768 + 768 + 64 -- the tail pass costs 13 launches for 4 % of the work)."""
    sizes = [step] * (n // step) + ([n % step] if n % step else [])
    if len(sizes) >= 2 and sizes[-1] + sizes[-2] <= longest:
        sizes[-2:] = [sizes[-1] + sizes[-2]]
    out, s = [], 0
    for z in sizes:
        out.append((s, s + z))
        s += z
    return out


def _vgg_features(cfg, batch_norm):
    layers, cin = [], 3
    for v in cfg:
        if v == 'M':
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            conv = nn.Conv2d(cin, v, kernel_size=3, padding=1)
            nn.init.kaiming_normal_(conv.weight, mode='fan_out', nonlinearity='relu')
            nn.init.constant_(conv.bias, 0)
            layers += [conv, nn.BatchNorm2d(v), nn.ReLU(inplace=True)] if batch_norm else [conv, nn.ReLU(inplace=True)]
            cin = v
    return layers


class MACVGGEngine:
    def __init__(self, model, device):
        self.plan = []
        mods = list(model.block1) + ['desc1'] + list(model.block2)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, str):
                self.plan.append(('desc', None))
            elif isinstance(m, nn.MaxPool2d):
                if self.plan and self.plan[-1][0] == 'conv':
                    self.plan[-1] = ('conv_pool', self.plan[-1][1])   # MaxPool2d(2,2) fused into the conv epilogue
                else:
                    self.plan.append(('pool', None))
            elif isinstance(m, nn.Conv2d):
                scale = shift = None
                if i + 1 < len(mods) and isinstance(mods[i + 1], nn.BatchNorm2d):
                    bn = mods[i + 1]
                    scale = bn.weight * (bn.running_var + bn.eps).rsqrt()
                    shift = bn.bias - bn.running_mean * scale
                self.plan.append(('conv', ops.PackedConv(m.weight, m.bias, 1, 1, scale=scale, shift=shift, device=device)))
            i += 1
        # conv1_1 + conv1_2 + pool1 fused into one persistent kernel when the plan starts with exactly that
        self.stem = None
        if (USE_FUSED_STEM and len(self.plan) >= 2 and self.plan[0][0] == 'conv' and self.plan[1][0] == 'conv_pool'
                and self.plan[0][1].cin == 3 and self.plan[0][1].cout == 64 and self.plan[1][1].cout == 64):
            packed = []
            for i, m in enumerate(mods):
                if isinstance(m, nn.Conv2d) and len(packed) < 4:
                    w, b = m.weight.detach().float().cpu(), m.bias.detach().float().cpu()
                    if i + 1 < len(mods) and isinstance(mods[i + 1], nn.BatchNorm2d):   # eval-mode BN folded in
                        bn = mods[i + 1]
                        sc = (bn.weight * (bn.running_var + bn.eps).rsqrt()).detach().float().cpu()
                        w = w * sc[:, None, None, None]
                        b = b * sc + (bn.bias.detach().float().cpu() - bn.running_mean.detach().float().cpu() * sc)
                    packed += [w, b]
            self.stem = ops.PackedStem(*packed, device=device)
            self.plan = self.plan[2:]
        self.device = device
        self.embedding_size = model.embedding_size
        self._skip_plans = {}      # input size -> the work-list schedule of `_embed_skip` (None: this plan has no such schedule)
        self._mac_tables = {}      # id of a constant crop -> per MAC layer, the row / column suffix maxima of the constant crop's map
        self._const_crops = {}     # (mean, std, channels, size) -> the all-padding crop

    # ---- constant-padding tile skipping --------------------------------------------------------------------------------------
    def skip_plan(self, size):
        """The work-list schedule for size x size inputs: (steps [(kind, PackedConv | None, pool, mac, store)], one `ops.skip_layer`
        per step, pool_mask), or None when this engine's plan is not stem + 3x3 halo convolutions with the MAC descriptors fused
        (then nothing is skipped).  Extent bookkeeping (include/cvpce_amd.h `cvpce_skip_layer`): the pass is a chain of ops on
        the crop -- conv (+1) and pool (halve upwards) -- and every tensor is named by the number of ops before it."""
        key = (size, SKIP_ROWS, SKIP_MAC, SKIP_STRIPS)
        if key in self._skip_plans:
            return self._skip_plans[key]
        steps, layers, chain = None, [], []
        if self.stem is not None and size % 16 == 0 and size <= 1024:
            steps = [('stem', None, True, False)]
            chain += [0, 0, 1]                               # conv1_1, conv1_2, pool1
            h = size // 2
            layers.append(ops.skip_layer(h, h, 8, 8, len(chain), size, size, 0, 1))
            plan = self.plan + [('desc', None)]
            i = 0
            while steps is not None and i < len(plan):
                kind, pc = plan[i]
                if kind not in ('conv', 'conv_pool') or not (pc.cin_pad % 64 == 0 and (pc.kh, pc.kw, pc.stride, pc.pad) == (3, 3, 1, 1)
                                                             and pc.cout % 8 == 0 and pc.cout > 64 and pc.dtype == ops.BF16):
                    steps = None
                    break
                mac = kind == 'conv' and i + 1 < len(plan) and plan[i + 1][0] == 'desc' and pc.cout > 128 and ops.USE_FUSED_MAC
                after = plan[i + 2][0] if i + 2 < len(plan) else None
                pool = kind == 'conv_pool' or (mac and after == 'pool' and h % 2 == 0)
                store = not mac or after is not None
                th, tw = (16, 32) if pc.cout <= 128 else (16, 16)
                in_ops, oh = len(chain), h
                chain.append(0)
                if pool:
                    chain.append(1)
                    th, tw, oh = th // 2, tw // 2, h // 2
                if not store:
                    oh, th, tw = h, 16, 16
                mode = 0 if (mac and not SKIP_MAC) else (1 if not SKIP_ROWS else (3 if (SKIP_STRIPS and pc.cout > 128) else 2))
                layers.append(ops.skip_layer(oh, oh, th, tw, len(chain), h, h, in_ops, mode))
                steps.append(('conv', pc, pool, mac, store))
                i += (3 if pool else 2) if mac else 1
                if not mac and i < len(plan) and plan[i][0] != 'conv' and plan[i][0] != 'conv_pool':
                    steps = None                              # a stand-alone pool / descriptor: not this schedule
                    break
                h = oh
                if h % 16 != 0 and i < len(plan):
                    steps = None
            if steps is not None and len(chain) > 32:
                steps = None
        pool_mask = sum(1 << i for i, p in enumerate(chain) if p)
        self._skip_plans[key] = None if steps is None else (steps, layers, pool_mask)
        return self._skip_plans[key]

    def const_crop(self, mean, std, channels, size):
        """The all-padding crop in the layout the crop kernel writes: made BY the crop kernel from a zero-area box, so its
        pixels are the very constant the kernel writes to every padding pixel."""
        key = (tuple(float(v) for v in mean), tuple(float(v) for v in std), int(channels), int(size))
        if key not in self._const_crops:
            img = torch.zeros((3, 1, 1), dtype=torch.float32, device=self.device)
            box = torch.zeros((1, 4), dtype=torch.float32, device=self.device)
            self._const_crops[key] = ops.crop_resize(img, box, size, mode=2 if channels == 4 else 1, mean=mean, std=std)[0].contiguous()
        return self._const_crops[key]

    def mac_tables(self, const_in, sched):
        """Per MAC layer of the schedule: (row suffix maxima (H + 1, C), column suffix maxima (W + 1, C)) f32 of the CONSTANT crop's
        post-ReLU map at that layer -- entry r = the maximum over rows >= r (all columns), the last entry 0.  The constant crop goes
        through the plain kernels once (same arithmetic per pixel as the work-list instances); a crop's descriptor starts from the
        entry at the first row / column its list leaves out."""
        key = (const_in.data_ptr(), id(sched))
        if key in self._mac_tables:
            return self._mac_tables[key]
        steps = sched[0]
        t = ops.vgg_stem(const_in[None].contiguous(), self.stem)
        tables = {}
        for li, (kind, pc, pool, mac, store) in enumerate(steps[1:], start=1):
            if not mac:
                t = ops.conv2d(t, pc, act=1, pool=pool)
                continue
            full = ops.conv2d(t, pc, act=1)                               # the map the MAC epilogue takes its maximum over
            m = full[0].to(torch.float32)                                 # (H, W, C), values >= 0
            zero = torch.zeros((1, m.shape[2]), dtype=torch.float32, device=m.device)
            rows = torch.cat((torch.cummax(m.amax(dim=1).flip(0), 0).values.flip(0), zero)).contiguous()
            cols = torch.cat((torch.cummax(m.amax(dim=0).flip(0), 0).values.flip(0), zero)).contiguous()
            tables[li] = (rows, cols)
            if store:
                t = ops.maxpool2d(full, 2, 2) if pool else full
        self._mac_tables[key] = tables
        self._keep_const = getattr(self, '_keep_const', []) + [const_in]  # (the key holds a data pointer: keep the tensor alive)
        return tables

    def _embed_skip(self, x, ext, const_in, sched):
        """The schedule over work lists: x (N,S,S,c) crops -> MAC descriptor (N,1024).  Two phases:
          * EARLY layers (the stem and the Cout <= 128 convs, whose per-crop tensors are 2-4 MB) in passes of <= FUSED_EMBED_MAX images,
            each with the constant crop as its last image; the last early layer of every pass writes straight into ONE tensor over
            all crops, pass after pass, so that the constant crop of a pass is overwritten by the next pass's first crop and the
            final tensor is [crop 0 .. crop N-1, constant crop];
          * LATE layers (conv3_1 on: <= 2 MB per crop) in ONE pass over all N + 1 images: half as many launches, each with twice the
            tiles per workgroup -- the uneven tail of a launch (workgroups run out of tiles a tile apart) costs half as much."""
        steps, layers, pool_mask = sched
        N, S, nl = x.shape[0], x.shape[1], len(steps)
        early = 1 + sum(1 for st in steps[1:] if st[1].cout <= 128 and not st[3])
        if any(st[1].cout <= 128 for st in steps[early:]):
            early = nl                                     # (not the VGG16 shape: wide-tile layers after halo2 ones -- everything in passes)
        dev = x.device
        mac_skips = any(st[3] and layers[i][-1] for i, st in enumerate(steps) if i)      # a MAC layer whose list leaves tiles / rows out
        tables = self.mac_tables(const_in, sched) if mac_skips else None

        def run(t, ext_, first, last, n_img, out_last=None):
            """steps[first:last] over n_img images (the constant crop last); the last of them writes into out_last if given."""
            sub = layers[first:last]
            k = last - first
            res = ops.embed_worklists(ext_, n_img, S, pool_mask, sub, (S // 16) ** 2 if first == 0 else max(1, (sub[0][5] // 16)) ** 2,
                                      want_computed=mac_skips)
            lists, counts = res[0], res[1]
            if mac_skips:
                o = sum(st[1].cout for st in steps[1:first] if st[3])
                for j in range(k):
                    if steps[first + j][3] if first + j else False:
                        ops.mac_init(desc, o, tables[first + j][0], tables[first + j][1], res[2][j])
                        o += steps[first + j][1].cout
            off = sum(st[1].cout for st in steps[1:first] if st[3])
            for j in range(k):
                li = first + j
                if li == 0:
                    t = ops.vgg_stem_list(t, const_in, self.stem, lists[0], counts[0:1])
                    continue
                kind, pc, pool, mac, store = steps[li]
                t = ops.conv2d_list(t, pc, lists[j], counts[j:j + 1], act=1, pool=pool, mac=desc if mac else None, mac_off=off, store=store,
                                    units=counts[k + j:k + j + 1], out=out_last if (j == k - 1 and store) else None,
                                    strips=(lists[k + j], counts[2 * k + j:2 * k + j + 1]) if layers[li][-1] == 3 else None)
                if mac:
                    off += pc.cout
            return t

        desc = (torch.empty if mac_skips else torch.zeros)((N + 1, self.embedding_size), dtype=torch.float32, device=dev)
        if early >= nl:                                     # everything in passes (each pass has its own descriptor rows)
            outs = []
            for s_, e_ in _passes(N, FUSED_EMBED_BATCH, FUSED_EMBED_MAX - 1):
                desc = (torch.empty if mac_skips else torch.zeros)((e_ - s_ + 1, self.embedding_size), dtype=torch.float32, device=dev)
                run(x[s_:e_], ext[s_:e_], 0, nl, e_ - s_ + 1)
                outs.append(desc[:e_ - s_])
            return torch.cat(outs)
        eh, ec = layers[early - 1][0], steps[early - 1][1].cout if early > 1 else 64         # the last early layer's output: (eh, eh, ec) per image
        big = torch.empty((N + 1, eh, eh, ec), dtype=torch.bfloat16, device=dev)
        for s_, e_ in _passes(N, FUSED_EMBED_BATCH, FUSED_EMBED_MAX - 1):
            if early == 1:                                  # (a plan whose first conv after the stem is already a late layer)
                big[s_:e_ + 1].copy_(run(x[s_:e_], ext[s_:e_], 0, 1, e_ - s_ + 1))
            else:
                run(x[s_:e_], ext[s_:e_], 0, early, e_ - s_ + 1, out_last=big[s_:e_ + 1])
        run(big, ext, early, nl, N + 1)
        return desc[:N]

    def will_skip(self, size):
        """Whether `embed_packed(x, ext=..., const_in=...)` on (B, size, size, c) crops runs the work-list schedule (the condition
        `embed_packed` evaluates: the module switch, the fused stem, a schedule for this size)."""
        return bool(SKIP_PADDING and self.stem is not None and size * size <= INPUT_SIZE * INPUT_SIZE and self.skip_plan(size) is not None)

    def embed_packed(self, x, eps=1e-8, want_bf16=False, batch=None, ext=None, const_in=None, partial=False):
        """x: (B,256,256,8) bf16, already normalised -> (B,1024) f32 unit-norm [, bf16 copy].
        batch: crops per pass of the kernel schedule (default MAX_EMBED_BATCH; a host that runs on fewer CUs passes that
        CU count so that the persistent kernels' tile counts stay whole multiples of their grid).
        ext (B,2) int32 + const_in (S,S,c): the crops' content extents (`ops.crop_extents`) and the all-padding crop
        (`const_crop`) -- tiles that lie in a crop's constant padding are then skipped (bit-identical results)."""
        outs, outs_bf = [], []
        fused = batch is None and self.stem is not None and x.shape[1] * x.shape[2] <= INPUT_SIZE * INPUT_SIZE
        step = batch or (FUSED_EMBED_BATCH if fused else MAX_EMBED_BATCH)
        plan = self.plan + [('desc', None)]            # the second descriptor: amax of the last map (classification.py:48-49)
        sched = self.skip_plan(x.shape[1]) if (SKIP_PADDING and ext is not None and const_in is not None and fused and x.shape[1] == x.shape[2]) else None
        if partial and sched is None:
            raise RuntimeError('embed_packed: crops that hold their content only (ops.crop_resize content_ext) need the work-list schedule')
        if sched is not None and x.shape[0]:
            # (the late layers' tensors: 64 x 64 x 256 per crop -> at most LATE_EMBED_MAX crops per call under the 32-bit tensor limits)
            for s, e in _passes(x.shape[0], LATE_EMBED_MAX, LATE_EMBED_MAX):
                r = ops.l2_normalize(self._embed_skip(x[s:e], ext[s:e], const_in, sched).contiguous(), eps, want_bf16)
                if want_bf16:
                    outs.append(r[0]); outs_bf.append(r[1])
                else:
                    outs.append(r)
            if want_bf16:
                return torch.cat(outs), torch.cat(outs_bf)
            return torch.cat(outs)
        for s, e in _passes(x.shape[0], step, FUSED_EMBED_MAX if fused else step):
            xb = x[s:e]
            desc = torch.zeros((xb.shape[0], self.embedding_size), dtype=torch.float32, device=x.device)   # (zeros: the fused MAC epilogue takes atomic maxima of values >= 0)
            off = 0
            if self.stem is not None:
                xb = ops.vgg_stem(xb, self.stem)
            i = 0
            while i < len(plan):
                kind, pc = plan[i]
                nxt = plan[i + 1][0] if i + 1 < len(plan) else None
                if kind == 'conv' and nxt == 'desc' and ops.can_fuse_mac(xb, pc):
                    # conv + ReLU with the MAC descriptor taken in its epilogue; the map itself is stored only if something
                    # reads it: pooled when MaxPool2d(2,2) follows (conv4_3 -> pool4), not at all at the end (conv5_3)
                    after = plan[i + 2][0] if i + 2 < len(plan) else None
                    fuse_pool = after == 'pool' and xb.shape[1] % 2 == 0 and xb.shape[2] % 2 == 0
                    xb = ops.conv2d_relu_mac(xb, pc, desc, off, store=after is not None, pool=fuse_pool)
                    off += pc.cout
                    i += 3 if fuse_pool else 2
                    continue
                if kind == 'conv':
                    xb = ops.conv2d(xb, pc, act=1)
                elif kind == 'conv_pool':
                    xb = ops.conv2d(xb, pc, act=1, pool=True)
                elif kind == 'pool':
                    xb = ops.maxpool2d(xb, 2, 2)
                else:
                    ops.global_max_into(xb, desc, off)
                    off += xb.shape[3]
                i += 1
            r = ops.l2_normalize(desc, eps, want_bf16)
            if want_bf16:
                outs.append(r[0]); outs_bf.append(r[1])
            else:
                outs.append(r)
        if not outs:
            e = torch.empty((0, self.embedding_size), dtype=torch.float32, device=x.device)
            return (e, e.to(torch.bfloat16)) if want_bf16 else e
        if want_bf16:
            return torch.cat(outs), torch.cat(outs_bf)
        return torch.cat(outs)


class MACVGG(nn.Module):
    """classification.py:20-51.  forward((B,3,256,256) in [-1,1]) -> (B,1024) unit-norm MAC descriptor."""

    embedding_size = 512 * 2
    input_mean, input_std = TANH_MEAN, TANH_STD

    def __init__(self, config='D', convs_per_block=[2, 2, 3, 3, 3], batch_norm=True, vgg_state_dict=None):
        super().__init__()
        feats = _vgg_features(VGG_CFGS[config], batch_norm)
        if vgg_state_dict is not None:
            holder = nn.Module()
            holder.features = nn.Sequential(*feats)
            holder.load_state_dict({k: v for k, v in vgg_state_dict.items() if k.startswith('features.')}, strict=True)
        per_conv = 3 if batch_norm else 2
        per_block = [c * per_conv + 1 for c in convs_per_block]
        cut1, cut2 = sum(per_block[:-1]) - 1, sum(per_block) - 1
        named = [(str(i), m) for i, m in enumerate(feats)]
        self.block1 = nn.Sequential(OrderedDict(named[:cut1]))
        self.block2 = nn.Sequential(OrderedDict(named[cut1:cut2]))
        self._engine = None
        self.eval()

    def load_state_dict(self, *a, **k):
        self._engine = None
        return super().load_state_dict(*a, **k)

    def _apply(self, fn, *a, **k):
        self._engine = None
        return super()._apply(fn, *a, **k)

    def engine(self):
        dev = next(self.parameters()).device
        if dev.type != 'cuda':
            raise RuntimeError('MACVGG runs on an MI355X (HIP) device only: call .cuda() first (no CPU fallback).')
        if self._engine is None:
            self._engine = MACVGGEngine(self, dev)
        return self._engine

    @torch.no_grad()
    def forward(self, x, eps=1e-8):
        eng = self.engine()
        x = x.to(device=eng.device, dtype=torch.float32)
        packed = ops.pack_embed_input(x, False, TANH_MEAN, TANH_STD)
        return eng.embed_packed(packed, eps)


def distance(emb1, emb2, dim=1):
    """classification.py:87-88 (elementwise form; plumbing-sized, plain torch on the caller's device)."""
    return 1 - torch.nn.functional.cosine_similarity(emb1, emb2, dim=dim)


@torch.no_grad()
def nearest_neighbors(anchors, queries, k=1, dtype=torch.float32):
    """classification.py:90-95 -> (Q,k) int64 indices into `anchors`, ascending cosine distance.

    One MFMA GEMM with a fused top-k epilogue (csrc/match.hip).  dtype=float32 uses the exact-f32
    matrix pipe (index-exact vs the fp32 oracle on tie-free data); bfloat16 is the fast path.
    Ties resolve to the lower index (the reference's unstable argsort leaves them unspecified).
    """
    if not anchors.is_cuda or not queries.is_cuda:
        raise RuntimeError('nearest_neighbors runs on an MI355X (HIP) device only (no CPU fallback)')
    if len(queries) == 0:
        return torch.empty((0, k), dtype=torch.int64, device=queries.device)
    a = ops.pad_features(anchors.to(dtype))
    q = ops.pad_features(queries.to(dtype))
    return ops.match_topk(q, a, k)


def macvgg_embedder(model='vgg16_bn', pretrained=True, progress=True):
    """classification.py:97-109.  No network here: `pretrained=True` is rejected."""
    model_to_config = {'vgg16': ('D', False), 'vgg16_bn': ('D', True)}
    if model not in model_to_config:
        raise NotImplementedError(f'MACVGG not implemented for {model}')
    if pretrained:
        raise RuntimeError('pretrained VGG weights cannot be downloaded here; use pretrained=False and load_state_dict')
    config, batchnorm = model_to_config[model]
    return MACVGG(config, batch_norm=batchnorm)


# ---------------------------------------------------------------------------------------------------------------------
# MACResNet: the optional ResNet-50 MAC encoder (classification.py:53-85,111-121), selectable in
# `cvpce dihe eval --model resnet50`.  Same kernels as the detector's ResNet body, eval-mode BatchNorm folded in.
# ---------------------------------------------------------------------------------------------------------------------
class _ResBottleneck(nn.Module):
    """torchvision.models.resnet.Bottleneck (v1.5: stride on the 3x3) as a parameter container."""
    expansion = 4

    def __init__(self, inplanes, planes, stride, downsample, norm_layer):
        super().__init__()
        self.conv1, self.bn1 = nn.Conv2d(inplanes, planes, 1, bias=False), norm_layer(planes)
        self.conv2, self.bn2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False), norm_layer(planes)
        self.conv3, self.bn3 = nn.Conv2d(planes, planes * 4, 1, bias=False), norm_layer(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = (nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride, bias=False), norm_layer(planes * 4))
                           if downsample else None)
        self.stride = stride


class _ResNetSource(nn.Module):
    """The attributes of torchvision's ResNet that MACResNet reads (conv1, bn1, relu, maxpool, layer1..4)."""

    def __init__(self, layers=(3, 4, 6, 3), norm_layer=nn.BatchNorm2d, stem=64, planes=(64, 128, 256, 512)):
        super().__init__()
        self.conv1, self.bn1 = nn.Conv2d(3, stem, 7, 2, 3, bias=False), norm_layer(stem)
        self.relu, self.maxpool = nn.ReLU(inplace=True), nn.MaxPool2d(3, 2, 1)
        inpl = stem
        for li, (planes, blocks) in enumerate(zip(planes, layers)):
            seq = []
            for bi in range(blocks):
                seq.append(_ResBottleneck(inpl, planes, 2 if (bi == 0 and li > 0) else 1, bi == 0, norm_layer))
                inpl = planes * 4
            setattr(self, f'layer{li + 1}', nn.Sequential(*seq))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')


class MACResNetEngine:
    def __init__(self, model, device):
        def fold(conv, bn):
            scale = shift = None
            if isinstance(bn, nn.BatchNorm2d):
                scale = bn.weight * (bn.running_var + bn.eps).rsqrt()
                shift = bn.bias - bn.running_mean * scale
            return ops.PackedConv(conv.weight, conv.bias, conv.stride[0], conv.padding[0], scale=scale, shift=shift, device=device)

        self.blocks = []          # per descriptor block: list of ('stem', pc) | ('bottleneck', (c1, c2, c3, ds))
        for block in model.blocks:
            stages = []
            for layer in block:
                if isinstance(layer[0], nn.Conv2d):           # Sequential(conv1, bn1, relu, maxpool)
                    stages.append(('stem', fold(layer[0], layer[1])))
                else:
                    for b in layer:
                        stages.append(('bottleneck', (fold(b.conv1, b.bn1), fold(b.conv2, b.bn2), fold(b.conv3, b.bn3),
                                                      fold(b.downsample[0], b.downsample[1]) if b.downsample is not None else None)))
            self.blocks.append(stages)
        # width of the descriptor actually produced = channels of the last stage of every block (the reference concatenates
        # whatever the blocks emit, classification.py:77-84; its `embedding_size` attribute assumes ResNet-50 widths)
        self.device = device
        self.embedding_size = sum((st[-1][1].cout if st[-1][0] == 'stem' else st[-1][1][2].cout) for st in self.blocks)

    def embed_packed(self, x, eps=1e-8, batch=None):
        """x: (B,S,S,8) bf16 -> (B, embedding_size) f32 unit-norm."""
        outs = []
        step = batch or MAX_EMBED_BATCH
        for s in range(0, x.shape[0], step):
            xb = x[s:s + step]
            desc = torch.empty((xb.shape[0], self.embedding_size), dtype=torch.float32, device=x.device)
            off = 0
            for stages in self.blocks:
                for kind, p in stages:
                    if kind == 'stem':
                        xb = ops.maxpool2d(ops.conv2d(xb, p, act=1), 3, 2, 1)
                    else:
                        c1, c2, c3, ds = p
                        identity = ops.conv2d(xb, ds) if ds is not None else xb
                        y = ops.conv2d(ops.conv2d(xb, c1, act=1), c2, act=1)
                        xb = ops.conv2d(y, c3, act=1, residual=identity)
                ops.global_max_into(xb, desc, off)
                off += xb.shape[3]
            outs.append(ops.l2_normalize(desc, eps))
        return torch.cat(outs) if outs else torch.empty((0, self.embedding_size), dtype=torch.float32, device=x.device)


class MACResNet(nn.Module):
    """classification.py:53-85.  forward((B,3,H,W)) -> (B, sum of the descriptor layers' channels) unit-norm MAC descriptor;
    `blocks[i]` = the ResNet stages between descriptor layer i-1 (exclusive) and i (inclusive), 0 = stem."""
    layer_output_sizes = [64, 256, 512, 1024, 2048]
    input_mean, input_std = (0.0, 0.0, 0.0), (1.0, 1.0, 1.0)      # no normalisation inside the module (classification.py:77-84)

    def __init__(self, source_resnet, descriptor_layers=[2, 3]):
        super().__init__()
        layers = [nn.Sequential(source_resnet.conv1, source_resnet.bn1, source_resnet.relu, source_resnet.maxpool),
                  source_resnet.layer1, source_resnet.layer2, source_resnet.layer3, source_resnet.layer4]
        prev = 0
        self.blocks = nn.ModuleList()
        for l in descriptor_layers:
            self.blocks.append(nn.Sequential(*layers[prev:l + 1]))
            prev = l + 1
        self.embedding_size = sum(self.layer_output_sizes[l] for l in descriptor_layers)
        self._engine = None
        self.eval()

    def load_state_dict(self, *a, **k):
        self._engine = None
        return super().load_state_dict(*a, **k)

    def _apply(self, fn, *a, **k):
        self._engine = None
        return super()._apply(fn, *a, **k)

    def engine(self):
        dev = next(self.parameters()).device
        if dev.type != 'cuda':
            raise RuntimeError('MACResNet runs on an MI355X (HIP) device only: call .cuda() first (no CPU fallback).')
        if self._engine is None:
            self._engine = MACResNetEngine(self, dev)
        return self._engine

    @torch.no_grad()
    def forward(self, x, eps=1e-8):
        eng = self.engine()
        x = x.to(device=eng.device, dtype=torch.float32)
        assert x.shape[2] == x.shape[3], 'square inputs (the reference feeds 256x256 crops)'
        return eng.embed_packed(ops.pack_embed_input(x, False, self.input_mean, self.input_std), eps)


def macresnet_encoder(model='resnet50', pretrained=True, progress=True, batch_norm=True, desc_layers=[2, 3]):
    """classification.py:111-121.  No network here: `pretrained=True` is rejected."""
    if model != 'resnet50':
        raise NotImplementedError(f'MACResNet not implemented for {model}')
    if pretrained:
        raise RuntimeError('pretrained ResNet weights cannot be downloaded here; use pretrained=False and load_state_dict')
    return MACResNet(_ResNetSource((3, 4, 6, 3), nn.BatchNorm2d if batch_norm else nn.Identity), desc_layers)
