"""Tensor-level wrappers of the hot-path kernels.

Layering: this module (shapes, output allocation, kernel choice) -> `torch.ops.cvpce_amd.*` (cvpce_amd/torch_ops.py: the
PyTorch-ROCm custom ops, CUDA/HIP dispatch key only) -> ctypes -> the C ABI of include/cvpce_amd.h -> hand-written HIP
kernels in libcvpce_hip.so.  torch is plumbing: device memory (tensors), the current HIP stream, the dispatcher.  There is
no CPU kernel anywhere: CPU tensors are rejected loudly.
"""
import ctypes
import math

import os as _os

import torch

from . import _lib
from ._lib import lib, check
from .torch_ops import T

BF16 = torch.bfloat16
F16 = torch.float16
STORAGE_TYPES = {'bf16': BF16, 'fp16': F16}    # element types the detector's kernels exist for (csrc/common.h ElemBF16 / ElemF16)
# the detector's default storage type: fp16 is the mode that meets north_star's 0.1 pt tolerance on AP50 / AP75 / AR300 / top-1
# (profiles/r04_accuracy.json: bf16 misses it on AP75 and end-to-end top-1) at the same MFMA rate; bf16 is the opt-in
DEFAULT_DETECTOR_PRECISION = 'fp16'
F16_MAX = 65504.0


def to_storage(t, dtype):
    """f32 host tensor -> the 16-bit storage type, RNE; fp16 saturates at +-65504 like the kernels' epilogues."""
    return t.clamp(-F16_MAX, F16_MAX).to(F16) if dtype == F16 else t.to(dtype)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError('cvpce_amd ops run on the GPU only (HIP kernels); got a CPU tensor. '
                               'There is no CPU fallback in the product path.')


# ---------------------------------------------------------------------------
# weights
# ---------------------------------------------------------------------------
class PackedConv:
    """bf16 [Cout_pad][K_pad] weight + fp32 bias in the K order of include/cvpce_amd.h."""

    def __init__(self, weight, bias=None, stride=1, pad=0, scale=None, shift=None, device='cuda', dtype=BF16):
        """weight (Cout,Cin,KH,KW) f32; optional per-Cout affine folded in:
        y = conv(x, w) * scale + shift (+ bias * scale)  -- FrozenBN / BN-eval folding.
        dtype: storage type of the packed weight = the activations' type this layer runs on (bf16 | fp16)."""
        assert dtype in (BF16, F16)
        self.dtype = dtype
        w = weight.detach().to(torch.float32).cpu()
        cout, cin, kh, kw = w.shape
        b = bias.detach().to(torch.float32).cpu() if bias is not None else None
        if scale is not None:
            scale = scale.detach().to(torch.float32).cpu()
            w = w * scale[:, None, None, None]
            b = b * scale if b is not None else None
        if shift is not None:
            shift = shift.detach().to(torch.float32).cpu()
            b = shift if b is None else b + shift
        self.cin = cin
        self.cin_pad = (cin + 7) // 8 * 8
        self.cout, self.kh, self.kw, self.stride, self.pad = cout, kh, kw, stride, pad
        k = kh * kw * self.cin_pad
        kq = 64 if self.cin_pad % 64 == 0 else 32   # K-step of the kernel that will run this layer
        self.k_pad = (k + kq - 1) // kq * kq
        self.cout_pad = (cout + 255) // 256 * 256
        wp = torch.zeros(self.cout_pad, kh, kw, self.cin_pad, dtype=torch.float32)
        wp[:cout, :, :, :cin] = w.permute(0, 2, 3, 1)
        if self.cin_pad % 64 == 0:
            # chunk-major K: (64-channel chunk, kh, kw, channel-in-chunk) -- the layout the kernels walk for Cin % 64 == 0
            wp = wp.reshape(self.cout_pad, kh, kw, self.cin_pad // 64, 64).permute(0, 3, 1, 2, 4)
        packed = torch.zeros(self.cout_pad, self.k_pad, dtype=torch.float32)
        packed[:, :k] = wp.reshape(self.cout_pad, k)
        w16 = to_storage(packed, dtype)
        self.weight = w16.to(device)
        self.bias = b.to(device) if b is not None else None
        # the 3x3 halo kernels (csrc/conv3x3_halo2.hip, conv3x3_halo3.hip) read the same values fragment-major
        self._weight_halo = (self._to_halo_layout(w16, self.cout_pad, self.cin_pad).to(device)
                             if (kh, kw) == (3, 3) and self.cin_pad % 64 == 0 else None)

    @property
    def weight_halo(self):
        """The weights in the fragment-major "halo weight layout" of include/cvpce_amd.h (3x3, Cin % 64 == 0; built with the layer)."""
        if _os.environ.get('CVPCE_WEIGHT_ROWMAJOR') == '1':     # dev A/B against a library built before the layout change
            return self.weight
        return self._weight_halo

    @staticmethod
    def _to_halo_layout(weight, cout_pad, cin_pad):
        """[64-channel chunk][32-cout group][kw][32-channel half][kh][16-cout block mt][lane = 16 q + m][8 channels], where lane (m, q)
        of block mt holds cout 32 g + 8 (m >> 2) + 4 mt + (m & 3), channels 64 c + 32 half + 8 q .. + 7 of tap (kh, kw) -- one
        contiguous KiB per MFMA weight fragment of the 3x3 halo kernels."""
        nch = cin_pad // 64
        # row-major packed: [cout_pad][chunk][kh][kw][64] -> (g, mq 4, mt 2, mj 4, c, kh, kw, half 2, q 4, e 8)
        v = weight.view(torch.int16).reshape(cout_pad // 32, 4, 2, 4, nch, 3, 3, 2, 4, 8)
        #                 -> (c, g, kw, half, kh, mt, q, mq, mj, e): lane = 16 q + 4 mq + mj
        return v.permute(4, 0, 6, 7, 5, 2, 8, 1, 3, 9).contiguous().view(weight.dtype).reshape(-1)

    def out_hw(self, h, w, in_up_shift=0):
        h, w = h << in_up_shift, w << in_up_shift
        return ((h + 2 * self.pad - self.kh) // self.stride + 1, (w + 2 * self.pad - self.kw) // self.stride + 1)


class ConvProfile:
    """Opt-in per-launch timing of the conv kernel with HIP events on the launch stream (bench.py roofline leg).

    Kernel variant names mirror the dispatch in csrc/conv_igemm.hip (tile TC x TP, K-step BK)."""

    class _Records(list):
        """The launch records; while `stage == 'detect'` the 3x3 halo kernels' records are filed under '<kernel>[detector]': the detector's
        launches (fp16 / non-list instances on small maps and the head atlas, 0.2-0.5 of the rate) are other template instances than the
        embedder's and are listed apart by rocprofv3 too -- merged, they hid what the dominant kernel (VGG conv3_1 ... conv5_3) reaches."""
        SPLIT = ('conv3x3_halo2_kernel', 'conv3x3_halo3_kernel')
        stage = None

        def append(self, rec):
            if self.stage == 'detect' and rec[0] in self.SPLIT:
                rec = (rec[0] + '[detector]',) + tuple(rec[1:])
            list.append(self, rec)

    def __init__(self):
        self.records = ConvProfile._Records()   # (variant, flops, start_event, end_event)
        self.byte_records = []   # HBM-bound launches: (kernel, algorithmic bytes, start_event, end_event)
        self.layer_records = []  # detector launch classes: (kernel + layer shape, algorithmic FLOPs, algorithmic bytes, start_event, end_event)

    def layer(self, key, flops, nbytes, e0, e1):
        self.layer_records.append((key, float(flops), float(nbytes), e0, e1))

    def summary_layers(self):
        """per launch class (kernel, layer shape): launches, algorithmic FLOPs and bytes (inputs + outputs + weights once), milliseconds --
        bench.py `workloads.detector_configs1.layers`."""
        torch.cuda.synchronize()
        out = {}
        for key, flops, nbytes, e0, e1 in self.layer_records:
            d = out.setdefault(key, {'launches': 0, 'flops': 0.0, 'bytes': 0.0, 'ms': 0.0})
            d['launches'] += 1
            d['flops'] += flops
            d['bytes'] += nbytes
            d['ms'] += e0.elapsed_time(e1)
        return out

    @staticmethod
    def variant(pc, m):
        bk64 = pc.cin_pad % 64 == 0
        tiles256 = (m + 255) // 256
        if bk64 and not FORCE_GENERIC_CONV:
            if pc.cout >= 192 and tiles256 * ((pc.cout + 255) // 256) >= 128:
                return 'conv_dma16_kernel<256,256,2,4,2,4>'
            if 64 < pc.cout <= 128 and tiles256 >= 128:
                return 'conv_dma_kernel<128,128,2,2,2>'
            if 32 < pc.cout <= 64 and tiles256 >= 128:
                return 'conv_dma_kernel<64,128,2,2,2>'
        tc = 128 if pc.cout > 64 else (32 if (pc.cout <= 32 and bk64) else 64)
        return f'conv_igemm_kernel<{tc},128,{64 if bk64 else 32},{1 if tc == 32 else 2},{4 if tc == 32 else 2}>'

    @staticmethod
    def is_igemm128(pc, m):
        """whether `variant(pc, m)` is the 128-cout register-staged kernel (the one with a split-K form), without formatting a name on the
        launch path."""
        if pc.cout <= 64:
            return False
        if pc.cin_pad % 64 == 0 and not FORCE_GENERIC_CONV:
            tiles256 = (m + 255) // 256
            if (pc.cout >= 192 and tiles256 * ((pc.cout + 255) // 256) >= 128) or (pc.cout <= 128 and tiles256 >= 128):
                return False
        return True

    def summary(self):
        """per kernel: launches, algorithmic FLOPs (`flops`), EXECUTED FLOPs (`flops_executed`: work-list launches of the embedder
        compute only the tiles on their list -- records carry (device count, FLOPs per listed tile) for those), milliseconds."""
        torch.cuda.synchronize()
        out = {}
        for rec in self.records:
            name, flops, e0, e1 = rec[:4]
            d = out.setdefault(name, {'launches': 0, 'flops': 0.0, 'flops_executed': 0.0, 'ms': 0.0})
            d['launches'] += 1
            d['flops'] += flops
            d['flops_executed'] += (float(rec[4].item()) * rec[5] if rec[4] is not None else 0.0) if len(rec) > 4 else flops
            d['ms'] += e0.elapsed_time(e1)
        return out


    def summary_bytes(self):
        """per HBM-bound kernel: launches, ALGORITHMIC bytes (every input and output tensor of the launch once, weights included) and
        milliseconds -- bench.py `roofline.hbm_stages`."""
        torch.cuda.synchronize()
        out = {}
        for name, nbytes, e0, e1 in self.byte_records:
            d = out.setdefault(name, {'launches': 0, 'bytes': 0.0, 'ms': 0.0})
            d['launches'] += 1
            d['bytes'] += float(nbytes)
            d['ms'] += e0.elapsed_time(e1)
        return out


PROFILE = None   # set to a ConvProfile() to record


def _nbytes(*ts):
    return sum(t.numel() * t.element_size() for t in ts if t is not None)


CONV1X1_ANY_SHAPE = bool(int(_os.environ.get('CVPCE_CONV1X1_ANY', '0')))   # test switch: every eligible 1x1 conv through the pointwise kernel
CONV1X1_STRIDE1 = _os.environ.get('CVPCE_CONV1X1_STRIDE1', '1') != '0'   # every stride-1 1x1 conv through the pointwise library entry (its streaming kernel, round 4; A/B switch)
CONV1X1_MAX_CIN = int(_os.environ.get('CVPCE_CONV1X1_MAX_CIN', '256'))   # expansion convs up to this Cin (512 -- the 25 x 25 stage of ResNet-50 -- measured slower: 3.32 -> 3.36 ms detector only)
ONE_OUTPUT_BLOCK = _os.environ.get('CVPCE_ONE_OUTPUT_BLOCK', '1') != '0'   # detect_postprocess outputs as views of one block (A/B switch)
USE_CONV1X1 = True           # 1x1 convs with Cin, Cout % 64 == 0 through the LDS-free pointwise GEMM kernel (A/B switch)
HALO_RAGGED = bool(int(_os.environ.get('CVPCE_HALO_RAGGED', '0')))          # test switch: also send small maps and maps that 16x16 tiles do not cover exactly through the halo kernel
USE_HALO_3X3 = True          # A/B switch: 3x3 s1 layers with Cin % 64 == 0 through the halo-patch kernel
FORCE_GENERIC_CONV = False   # A/B switch: route every conv through the register-staged fallback kernel


def conv2d(x, pc, act=0, out_f32=False, residual=None, res_mode=0, in_up_shift=0, out=None, pool=False):
    """x: NHWC bf16 | fp16 (N,H,W,Cin_pad), the type `pc` was packed for -> NHWC (N,Ho,Wo,Cout) of the same type | f32."""
    _need_cuda(x, residual)
    assert x.dtype == pc.dtype and x.is_contiguous() and x.dim() == 4, (x.dtype, pc.dtype)
    n, h, w, cin = x.shape
    assert cin == pc.cin_pad, (cin, pc.cin_pad)
    ho, wo = pc.out_hw(h, w, in_up_shift)
    if out is None:
        oshape = (n, ho // 2, wo // 2, pc.cout) if pool else (n, ho, wo, pc.cout)
        out = torch.empty(oshape, dtype=torch.float32 if out_f32 else x.dtype, device=x.device)
    hr = wr = 0
    if residual is not None:
        assert residual.dtype == x.dtype and residual.is_contiguous() and residual.shape[0] == n and residual.shape[3] == pc.cout
        hr, wr = residual.shape[1], residual.shape[2]
        if res_mode == 0:
            res_mode = 1 if (hr, wr) == (ho, wo) else 2
    halo = (USE_HALO_3X3 and not FORCE_GENERIC_CONV and pc.cin_pad % 64 == 0 and pc.kh == 3 and pc.kw == 3
            and pc.stride == 1 and pc.pad == 1 and not out_f32 and residual is None and not in_up_shift
            and act in (0, 1) and pc.cout % 8 == 0 and pc.cout > 64 and (HALO_RAGGED or (h % 16 == 0 and w % 16 == 0) or min(h, w) >= 48)   # ragged (masked) tiles pay from about 50x50 up (detector: 200x200 -1/3, 100x100 -1/3, 50x50 -5 %, 25x25 +20 %); the choice never depends on the batch size: a crop's embedding must not change with the crops it is batched with
            and n * h * w * pc.cin_pad * 2 < 2 ** 32)
    prof = PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    # the RetinaNet head's output convs (256 -> 9 | 36, fp32 out, no activation): the 16 x 32-pixel halo-patch kernel with only the waves
    # that hold a real cout computing (csrc/conv3x3_halo3.hip THIN) -- 72 + 60 us per 4 images as register-staged implicit GEMMs
    if (USE_HALO_THIN_OUT and not FORCE_GENERIC_CONV and out_f32 and act == 0 and residual is None and not in_up_shift and not pool
            and (pc.kh, pc.kw, pc.stride, pc.pad) == (3, 3, 1, 1) and pc.cin_pad % 64 == 0 and cin == pc.cin_pad and pc.cout <= 64
            and min(h, w) >= 48 and n * h * w * pc.cin_pad * 2 < 2 ** 32):
        T.conv3x3_halo_thin_out(x, pc.weight_halo, pc.bias, out, pc.cout, pc.k_pad, pc.cout_pad)
        if prof is not None:
            e1.record()
            prof.records.append(('conv3x3_halo3_kernel', 2.0 * n * ho * wo * pc.cout * 9 * pc.cin, e0, e1))
            prof.layer(f'conv3x3_halo3 thin out {ho}x{wo} {pc.cin}->{pc.cout} f32 out', 2.0 * n * ho * wo * pc.cout * 9 * pc.cin, _nbytes(x, out, pc.weight), e0, e1)
        return out
    if halo:
        # (Cout <= 128 is forwarded to the wide-tile kernel, conv3x3_halo3.hip, inside the library)
        T.conv3x3_halo(x, pc.weight_halo, pc.bias, out, pc.cout, pc.k_pad, pc.cout_pad, int(act), int(pool))
        if prof is not None:
            e1.record()
            prof.records.append(('conv3x3_halo3_kernel' if pc.cout <= 128 else 'conv3x3_halo2_kernel',
                                 2.0 * n * ho * wo * pc.cout * 9 * pc.cin, e0, e1))
            prof.layer(f"{'conv3x3_halo3' if pc.cout <= 128 else 'conv3x3_halo2'} {h}x{w} {pc.cin}->{pc.cout}" + (' +pool' if pool else ''),
                       2.0 * n * ho * wo * pc.cout * 9 * pc.cin, _nbytes(x, out, pc.weight), e0, e1)
        return out
    # thin 3x3 layers (the Gaussian subnet's 64 -> 32 over an upsampled input, 32 -> 32, 32 -> 16, all on the 400 x 400 map): weights in
    # registers, input rows streamed through LDS (csrc/thin3x3.hip)
    thin_shape = ((cin, pc.cin, pc.k_pad) == (32, 32, 288) and pc.cout in (16, 32) and not in_up_shift) or \
                 ((cin, pc.cin, pc.k_pad) == (64, 64, 576) and pc.cout == 32 and in_up_shift == 1)
    if (USE_THIN_3X3 and not FORCE_GENERIC_CONV and (pc.kh, pc.kw, pc.stride, pc.pad) == (3, 3, 1, 1) and thin_shape
            and not out_f32 and residual is None and not pool and act in (0, 1) and n * ho * wo * 64 < 2 ** 32 - 65536):
        T.conv3x3_thin(x, pc.weight, pc.bias, out, pc.cout, pc.k_pad, pc.cout_pad, int(act), int(in_up_shift))
        if prof is not None:
            e1.record()
            prof.records.append(('thin3x3_kernel', 2.0 * n * ho * wo * pc.cout * 9 * pc.cin, e0, e1))
            prof.byte_records.append(('thin3x3_kernel', _nbytes(x, out, pc.weight), e0, e1))
            prof.layer(f'thin3x3 {ho}x{wo} {pc.cin}->{pc.cout}' + (' (2x upsampled input)' if in_up_shift else ''), 2.0 * n * ho * wo * pc.cout * 9 * pc.cin,
                       _nbytes(x, out, pc.weight), e0, e1)
        return out
    # pointwise GEMM kernel: wins on the expansion convs (short K, 4x wider output: HBM-bound, 1.3-1.8x), loses on long K
    if (USE_CONV1X1 and not FORCE_GENERIC_CONV and pc.kh == 1 and pc.kw == 1 and pc.pad == 0 and cin % 64 == 0 and pc.k_pad == cin
            and (CONV1X1_ANY_SHAPE or (pc.stride == 1 and CONV1X1_STRIDE1) or (cin <= CONV1X1_MAX_CIN and pc.cout >= 4 * cin)) and pc.cout % 64 == 0 and not out_f32 and not in_up_shift and not pool and act in (0, 1) and n * h * w * cin * 2 < 2 ** 32):
        T.conv1x1_nhwc(x, pc.weight, pc.bias, residual, out, pc.cout, pc.stride, ho, wo, pc.k_pad, pc.cout_pad, int(act), int(res_mode))
        if prof is not None:
            e1.record()
            prof.records.append(('conv1x1_kernel', 2.0 * n * ho * wo * pc.cout * pc.cin, e0, e1))
            # a strided 1x1 reads only the pixels it keeps; an upsampled residual (res_mode 2) is read at its stored size
            prof.byte_records.append(('conv1x1_kernel', n * ho * wo * cin * x.element_size() + _nbytes(out, residual, pc.weight), e0, e1))
            prof.layer(f'conv1x1 {ho}x{wo} {pc.cin}->{pc.cout} s{pc.stride}' + (' +res' if residual is not None else ''), 2.0 * n * ho * wo * pc.cout * pc.cin,
                       n * ho * wo * cin * x.element_size() + _nbytes(out, residual, pc.weight), e0, e1)
        return out
    # small maps with a long K (layer4's 3x3 convs on 25 x 25 maps, the stride-2 3x3s that open layer3 / layer4, P5 / P6 / P7): the
    # register-staged kernel with K split over `ks` workgroups per output tile (csrc/conv_igemm.hip SPLIT)
    ks = splitk_factor(pc, ho, wo) if (not pool and not FORCE_GENERIC_CONV and ConvProfile.is_igemm128(pc, n * ho * wo)) else 0
    ws = None
    if ks:
        # One fp32 workspace per (launch shape, device), kept with the layer: a layer of an engine runs on ONE stream at a time (the
        # side streams of a schedule run DIFFERENT layers), so its launches are ordered and may share the block.  The key must not
        # depend on the stream: a hipGraph is captured on another stream than the eager first call of its geometry, and a capture that
        # found no workspace would have to take the unsplit kernel -- another fp32 summation order, i.e. replayed results that differ
        # from eager ones (tests/test_gpu_configs1.py).  Captured graphs hold a workspace's address, and nothing here knows which graphs
        # are still alive: an entry is NEVER freed (tens of MB per layer and batch geometry; an engine's graph cache bounds the geometries).
        cache = pc.__dict__.setdefault('_splitk_ws', {})
        key = (n, ho, wo, ks, x.device)
        ws = cache.get(key)
        if ws is None:
            ws = cache[key] = torch.empty(int(lib.cvpce_conv2d_splitk_workspace_bytes(n * ho * wo, pc.cout, ks)), dtype=torch.uint8, device=x.device)
    if ks:
        T.conv2d_splitk(x, pc.weight, pc.bias, residual, out, pc.cout, pc.kh, pc.kw, pc.stride, pc.pad, ho, wo, pc.k_pad, pc.cout_pad,
                        int(act), int(out_f32), int(in_up_shift), int(res_mode), ks, ws)
    else:
        T.conv2d_nhwc(x, pc.weight, pc.bias, residual, out, pc.cout, pc.kh, pc.kw, pc.stride, pc.pad, ho, wo, pc.k_pad, pc.cout_pad,
                      int(act), int(out_f32), int(in_up_shift), int(res_mode), int(pool), int(FORCE_GENERIC_CONV))   # FORCE_GENERIC_CONV: False/True or 2, 3 = A/B variants
    if prof is not None:
        e1.record()
        # algorithmic FLOPs: real (unpadded) channels, 2 FLOP per MAC
        prof.records.append((prof.variant(pc, n * ho * wo), 2.0 * n * ho * wo * pc.cout * pc.kh * pc.kw * pc.cin, e0, e1))
        prof.layer(f"{prof.variant(pc, n * ho * wo).split('<')[0]} {ho}x{wo} {pc.cin}->{pc.cout} k{pc.kh} s{pc.stride}" + (' f32 out' if out_f32 else '') + (f' split-K {ks}' if ks else ''),
                   2.0 * n * ho * wo * pc.cout * pc.kh * pc.kw * pc.cin, _nbytes(x, out, residual, pc.weight), e0, e1)
    return out


CONV_SPLITK = _os.environ.get('CVPCE_CONV_SPLITK', '1') != '0'   # A/B switch: split-K launches of the register-staged kernel


def splitk_factor(pc, ho, wo):
    """Workgroups per output tile for a conv that the 128-cout register-staged kernel would run: a function of the LAYER shape only
    (K-steps, map size), 0 = unsplit.  36+ K-steps on an output map of at most 32 x 32: 4 (tools/dev/prof_splitk.sh, kernel durations on 4 images:
    layer4's 3x3 62.6 -> 28.8 + 6.4 us, P6 / P7 34 -> 11 + 4 us; a 50 x 50 output map has workgroups enough: 37.8 -> 29.0 + 7-10 us, not split)."""
    if not CONV_SPLITK or pc.cin_pad % 64 != 0 or pc.cout <= 64 or pc.k_pad // 64 < 32:
        return 0
    return 4 if ho * wo <= 1024 else 0


USE_HALO_THIN_OUT = _os.environ.get('CVPCE_HALO_THIN_OUT', '1') != '0'   # A/B switch: the head's output convs through the thin-output form of the wide halo kernel
USE_THIN_3X3 = _os.environ.get('CVPCE_THIN_3X3', '1') != '0'   # A/B switch: the thin 3x3 layers through csrc/thin3x3.hip
USE_FUSED_BOTTLENECK = _os.environ.get('CVPCE_FUSED_BOTTLENECK', '1') != '0'   # A/B switch: stride-1 ResNet bottlenecks (P <= 256) in one launch (csrc/bneck.hip)


FUSED_BOTTLENECK_MAX_PLANES = int(_os.environ.get('CVPCE_FUSED_BOTTLENECK_MAXP', '128'))   # measured (tools/dev/bench_bneck.py): with fragment-major weights (round 5) layer1 (P = 64: 66 vs 88 us per 4-image block) AND layer2 (P = 128: 49 vs 70 us) win over three launches; P = 256 still loses (120 vs 67 us: a 50 x 50 map has too few tiles for a 110-us tile chain).  Round 3-4 (row-major weights): only P = 64 won


def can_fuse_bottleneck(x, c1, c2, c3, residual):
    """cvpce_bottleneck_fused covers: 1x1 (s1) -> 3x3 (s1, p1) -> 1x1 (s1), planes P in {64, 128, 256}, 4P outputs, a same-size residual.
    `residual`: the tensor, or just its shape (a caller deciding BEFORE it computes the projection shortcut)."""
    p = c1.cout
    n, h, w, cin = x.shape
    if residual is not None and not torch.is_tensor(residual):
        residual = torch.empty(tuple(residual), device='meta')
    return (USE_FUSED_BOTTLENECK and not FORCE_GENERIC_CONV and p in (64, 128, 256) and p <= FUSED_BOTTLENECK_MAX_PLANES and cin % 64 == 0 and c1.cin_pad == cin
            and (c1.kh, c1.stride, c1.pad) == (1, 1, 0) and (c2.kh, c2.kw, c2.stride, c2.pad) == (3, 3, 1, 1) and c2.cin == p and c2.cout == p
            and (c3.kh, c3.stride, c3.pad) == (1, 1, 0) and c3.cin == p and c3.cout == 4 * p and c2.k_pad == 9 * p
            and all(c.bias is not None for c in (c1, c2, c3)) and residual is not None and tuple(residual.shape) == (n, h, w, 4 * p)
            and n * h * w * max(cin, 4 * p) * 2 < 2 ** 32)


BNECK_FRAGMENT_MAJOR = _os.environ.get('CVPCE_BNECK_FM', '1') != '0'   # A/B switch: the fused bottleneck reads fragment-major weights (round 5)


def pack_bottleneck_weights(c1, c2, c3):
    """The three convs of a bottleneck in the FRAGMENT-MAJOR layouts of cvpce_bottleneck_fused_fm (include/cvpce_amd.h): every MFMA weight
    fragment csrc/bneck.hip loads becomes one contiguous KiB, lane L = 16 lq + l16 at byte 16 L, 8 consecutive k per lane.
      w1f [Cin/32 K-steps][P/16 cout blocks b][lane][8]:  cout = 32 (b >> 1) + 8 (l16 >> 2) + (l16 & 3) + 4 (b & 1),  k = 32 ks + 8 lq + e
      w2f [P/CW cout groups][6 P/64 steps s = (c64, kw, half)][kh][CW/16 blocks h][lane][8]:  cout = CW cg + (CW == 32 ? 8 (l16 >> 2) + (l16 & 3) + 4 h : l16),
           k = ((3 c64 + kh) 3 + kw) 64 + 32 half + 8 lq + e   (CW = 16 at P = 64, else 32; K chunk-major as in the packed conv weight)
      w3f [4P/32 cout groups g][P/32 K-steps][2 blocks h][lane][8]:  cout = 32 g + 8 (l16 >> 2) + (l16 & 3) + 4 h,  k = 32 ks + 8 lq + e
    -> (w1f, w2f, w3f) flat tensors of the convs' storage type on their device; cached on c1."""
    cached = c1.__dict__.get('_bneck_fm')
    if cached is not None and cached[0] is c2 and cached[1] is c3:
        return cached[2]
    p, cin = c1.cout, c1.cin_pad
    dev = c1.weight.device
    l16, lq, e = torch.arange(16), torch.arange(4), torch.arange(8)
    pair = 8 * (l16 >> 2) + (l16 & 3)                                     # cout within a 32-cout pair of blocks, + 4 h

    def gather(w, cout, k):                                               # w [rows][k_pad] -> w[cout, k] broadcast, int16 view
        return w.view(torch.int16)[cout.to(dev), k.to(dev)].contiguous().view(w.dtype).reshape(-1)

    # w1f [ks][b][lq][l16][e]
    ks, b = torch.arange(cin // 32), torch.arange(p // 16)
    cout1 = (32 * (b >> 1) + 4 * (b & 1))[None, :, None, None, None] + pair[None, None, None, :, None]
    k1 = (32 * ks)[:, None, None, None, None] + (8 * lq)[None, None, :, None, None] + e[None, None, None, None, :]
    w1f = gather(c1.weight, cout1.expand(len(ks), len(b), 4, 16, 8), k1.expand(len(ks), len(b), 4, 16, 8))
    # w2f [cg][s][kh][h][lq][l16][e]
    cw = 16 if p == 64 else 32
    ncb, ncg, ns = cw // 16, p // cw, (p // 64) * 6
    cg, st, kh, hh = torch.arange(ncg), torch.arange(ns), torch.arange(3), torch.arange(ncb)
    c64, kw, hf = st // 6, (st % 6) >> 1, st & 1
    row = (pair[None, :] + 4 * hh[:, None]) if ncb == 2 else l16[None, :].expand(1, 16)                  # [h][l16]
    cout2 = (cw * cg)[:, None, None, None, None, None, None] + row[None, None, None, :, None, :, None]
    k2 = (((3 * c64)[:, None] + kh[None, :]) * 3 + kw[:, None]) * 64 + (32 * hf)[:, None]                 # [s][kh]
    k2 = k2[None, :, :, None, None, None, None] + (8 * lq)[None, None, None, None, :, None, None] + e[None, None, None, None, None, None, :]
    shape2 = (ncg, ns, 3, ncb, 4, 16, 8)
    w2f = gather(c2.weight, cout2.expand(shape2), k2.expand(shape2))
    # w3f [g][ks][h][lq][l16][e]
    g32, ks3, h2 = torch.arange(4 * p // 32), torch.arange(p // 32), torch.arange(2)
    cout3 = (32 * g32)[:, None, None, None, None, None] + (4 * h2)[None, None, :, None, None, None] + pair[None, None, None, None, :, None]
    k3 = (32 * ks3)[None, :, None, None, None, None] + (8 * lq)[None, None, None, :, None, None] + e[None, None, None, None, None, :]
    shape3 = (len(g32), len(ks3), 2, 4, 16, 8)
    w3f = gather(c3.weight, cout3.expand(shape3), k3.expand(shape3))
    c1.__dict__['_bneck_fm'] = (c2, c3, (w1f, w2f, w3f))
    return w1f, w2f, w3f


def bottleneck(x, c1, c2, c3, residual):
    """relu(c3(relu(c2(relu(c1(x))))) + residual) -- one ResNet bottleneck block (stride 1) in ONE launch."""
    _need_cuda(x, residual)
    assert can_fuse_bottleneck(x, c1, c2, c3, residual) and x.dtype == c1.dtype == c2.dtype == c3.dtype == residual.dtype
    assert x.is_contiguous() and residual.is_contiguous()
    n, h, w, cin = x.shape
    p = c1.cout
    out = torch.empty((n, h, w, 4 * p), dtype=x.dtype, device=x.device)
    fm = pack_bottleneck_weights(c1, c2, c3) if BNECK_FRAGMENT_MAJOR else None      # (built once per block: cached on c1)
    prof = PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if fm is not None:
        T.bottleneck_fused_fm(x, residual, fm[0], c1.bias, fm[1], c2.bias, fm[2], c3.bias, out, p)
    else:
        T.bottleneck_fused(x, residual, c1.weight, c1.bias, c2.weight, c2.bias, c3.weight, c3.bias, out, p, c1.k_pad, c2.k_pad, c3.k_pad,
                           c1.cout_pad, c2.cout_pad, c3.cout_pad)
    if prof is not None:
        e1.record()
        prof.records.append(('bneck_kernel', 2.0 * n * h * w * (cin * p + 9 * p * p + 4 * p * p), e0, e1))
        prof.layer(f'bneck {h}x{w} {cin}->{p}->{4 * p}', 2.0 * n * h * w * (cin * p + 9 * p * p + 4 * p * p),
                   _nbytes(x, out, c1.weight, c2.weight, c3.weight) + (0 if residual.data_ptr() == x.data_ptr() else _nbytes(residual)), e0, e1)
    return out


USE_FUSED_MAC = True   # A/B switch: MAC descriptor (+ pool4, + no store of conv5_3) fused into the conv epilogue


def can_fuse_mac(x, pc):
    """The halo kernel's fused-MAC epilogue covers 3x3 / s1 / p1 layers with Cin % 64 == 0 and Cout > 128 (VGG16 conv4_3, conv5_3)."""
    n, h, w, _ = x.shape
    return (USE_FUSED_MAC and USE_HALO_3X3 and not FORCE_GENERIC_CONV and pc.cin_pad % 64 == 0 and pc.kh == 3 and pc.kw == 3 and pc.stride == 1
            and pc.pad == 1 and pc.cout % 8 == 0 and pc.cout > 128 and n * h * w * pc.cin_pad * 2 < 2 ** 32)


def conv2d_relu_mac(x, pc, mac, mac_off, store=True, pool=False):
    """relu(conv3x3(x)) with `mac[:, mac_off:mac_off+Cout] = amax over H, W` taken in the epilogue (classification.py:46-49).
    `mac` (N, D) f32 must be zero-filled.  store=False: the map is not written at all (returns None); pool=True: the
    returned map is MaxPool2d(2,2) of it (the descriptor is still over the unpooled map)."""
    _need_cuda(x, mac)
    assert x.dtype == BF16 and pc.dtype == BF16 and x.is_contiguous() and x.dim() == 4 and mac.dtype == torch.float32 and mac.is_contiguous()
    assert can_fuse_mac(x, pc) and mac.shape[0] == x.shape[0] and mac_off + pc.cout <= mac.shape[1]
    n, h, w, cin = x.shape
    out = None
    if store:
        out = torch.empty((n, h // 2, w // 2, pc.cout) if pool else (n, h, w, pc.cout), dtype=BF16, device=x.device)
    prof = PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    T.conv3x3_halo_mac(x, pc.weight_halo, pc.bias, out, mac, int(mac_off), pc.cout, pc.k_pad, pc.cout_pad, int(pool and store))
    if prof is not None:
        e1.record()
        prof.records.append(('conv3x3_halo2_kernel', 2.0 * n * h * w * pc.cout * 9 * pc.cin, e0, e1))
    return out


class PackedStem:
    """Weights of conv3x3(3->64) + conv3x3(64->64) in the layout of cvpce_vgg_stem_fused (include/cvpce_amd.h)."""

    def __init__(self, w1, b1, w2, b2, device='cuda'):
        w1 = w1.detach().to(torch.float32).cpu()
        w2 = w2.detach().to(torch.float32).cpu()
        assert tuple(w1.shape) == (64, 3, 3, 3) and tuple(w2.shape) == (64, 64, 3, 3)
        p1 = torch.zeros(64, 3, 4, 4)                       # (cout, kh, kw 0..3, c 0..3)
        p1[:, :, :3, :3] = w1.permute(0, 2, 3, 1)           # (cout, kh, kw, c)
        self.w1 = p1.reshape(64, 48).to(BF16).to(device)
        self.w2 = w2.permute(2, 3, 0, 1).reshape(9, 64, 64).contiguous().to(BF16).to(device)   # (tap, cout, cin)
        self.b1 = b1.detach().to(torch.float32).to(device)
        self.b2 = b2.detach().to(torch.float32).to(device)
        self.flops_per_pixel = 2.0 * 64 * (27 + 576)


class PackedGlnStem:
    """conv 7x7/2 (3->64) with its FrozenBatchNorm folded in, in the MFMA fragment order of cvpce_gln_stem_fused."""

    def __init__(self, weight, scale, shift, device='cuda', dtype=BF16):
        self.dtype = dtype
        w = weight.detach().to(torch.float32).cpu() * scale.detach().to(torch.float32).cpu()[:, None, None, None]
        assert tuple(w.shape) == (64, 3, 7, 7)
        slots = torch.zeros(64, 7, 8, 4)                       # (cout, kh, kw slot, channel slot)
        slots[:, :, :7, :3] = w.permute(0, 2, 3, 1)
        # [ct][kh][h][lh][r][kw_local 2][c 4]: kw = 4h + 2lh + kw_local
        f = slots.reshape(2, 32, 7, 2, 2, 2, 4).permute(0, 2, 3, 4, 1, 5, 6)
        self.w_frag = to_storage(f.reshape(2, 14, 64, 8).contiguous(), dtype).to(device)
        self.bias = shift.detach().to(torch.float32).to(device)


def gln_stem(x, ps):
    """x: (N,H,W,8) bf16 | fp16 transformed batch -> (N,Hp,Wp,64) of the same type = maxpool3x3/2(relu(bn(conv7x7/2(x))))."""
    _need_cuda(x)
    assert x.dtype == ps.dtype and x.is_contiguous() and x.shape[3] == 8
    n, h, w, _ = x.shape
    hc, wc = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    out = torch.empty((n, (hc - 1) // 2 + 1, (wc - 1) // 2 + 1, 64), dtype=x.dtype, device=x.device)
    prof = PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    T.gln_stem_fused(x, ps.w_frag, ps.bias, out)
    if prof is not None:
        e1.record()
        prof.records.append(('gln_stem_kernel', 2.0 * n * hc * wc * 64 * 147, e0, e1))
        prof.layer(f'gln_stem {h}x{w} 3->64 k7 s2 +pool', 2.0 * n * hc * wc * 64 * 147, _nbytes(x, out), e0, e1)
    return out


def atlas_tile_map(mask, tile=16):
    """(H,W) uint8 level mask (host or device) -> int32 device tensor of the 16x16 tiles that contain a level pixel, (ty << 16) | tx."""
    m = mask.cpu().bool()
    h, w = m.shape
    tiles = [(ty << 16) | tx for ty in range((h + tile - 1) // tile) for tx in range((w + tile - 1) // tile)
             if bool(m[ty * tile:(ty + 1) * tile, tx * tile:(tx + 1) * tile].any())]
    return torch.tensor(tiles, dtype=torch.int32).to(mask.device)


def conv3x3_atlas(x, pc, mask, act=1, tile_map=None, out=None, mask_pixels=None):
    """3x3 / stride 1 / pad 1 conv over a LEVEL ATLAS x (N,H,W,Cin) bf16: several maps sharing `pc`, packed with zero
    gaps; mask (H,W) uint8 is 1 on level pixels.  Output pixels on the gaps are written as zeros.  tile_map (atlas_tile_map):
    only those tiles are computed -- `out` must then be given and hold zeros on the skipped (all-gap) tiles."""
    _need_cuda(x, mask, tile_map, out)
    assert x.dtype == pc.dtype and x.is_contiguous() and x.dim() == 4 and mask.dtype == torch.uint8 and mask.is_contiguous()
    n, h, w, cin = x.shape
    assert tuple(mask.shape) == (h, w) and cin == pc.cin_pad and pc.kh == 3 and pc.kw == 3 and pc.stride == 1 and pc.pad == 1
    assert pc.cin_pad % 64 == 0 and pc.cout % 8 == 0 and pc.cout > 128 and act in (0, 1)
    if tile_map is not None:
        assert out is not None and tile_map.dtype == torch.int32 and tile_map.is_contiguous()
    if out is None:
        out = torch.empty((n, h, w, pc.cout), dtype=x.dtype, device=x.device)
    assert tuple(out.shape) == (n, h, w, pc.cout) and out.dtype == x.dtype and out.is_contiguous()
    prof = PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    T.conv3x3_halo_masked(x, pc.weight_halo, pc.bias, mask, tile_map, out, pc.cout, pc.k_pad, pc.cout_pad, int(act))
    if prof is not None:
        e1.record()
        npix = float(mask.sum().item()) if mask_pixels is None else float(mask_pixels)
        prof.records.append(('conv3x3_halo2_kernel', 2.0 * npix * n * pc.cout * 9 * pc.cin, e0, e1))
        prof.layer(f'conv3x3_halo2 atlas {pc.cin}->{pc.cout} (head tower)', 2.0 * npix * n * pc.cout * 9 * pc.cin,
                   2 * npix * n * (pc.cin + pc.cout) * x.element_size() + _nbytes(pc.weight), e0, e1)
    return out


def conv3x3_atlas_paired(x, pc, mask, act=1, tile_map=None, out=None, mask_pixels=None, in_paired=False):
    """Layer i of the head's TWO towers (torchvision RetinaNetHead) as ONE masked launch: `pc` packs the towers' layer-i weights
    concatenated along Cout (256 x towers); x is one atlas (N,H,W,Cin) every tower reads (in_paired=False: the first layer) or the
    tower-major output of the previous paired launch (towers,N,H,W,Cin); -> (towers,N,H,W,256), tower t's output in part t."""
    _need_cuda(x, mask, tile_map, out)
    towers = pc.cout // 256
    assert pc.cout == 256 * towers and towers >= 2 and x.dtype == pc.dtype and x.is_contiguous() and mask.dtype == torch.uint8 and mask.is_contiguous()
    n, h, w, cin = x.shape[-4:]
    assert tuple(x.shape) == ((towers, n, h, w, cin) if in_paired else (n, h, w, cin))
    assert tuple(mask.shape) == (h, w) and cin == pc.cin_pad and (pc.kh, pc.kw, pc.stride, pc.pad) == (3, 3, 1, 1) and pc.cin_pad % 64 == 0 and act in (0, 1)
    if tile_map is not None:
        assert out is not None and tile_map.dtype == torch.int32 and tile_map.is_contiguous()
    if out is None:
        out = torch.empty((towers, n, h, w, 256), dtype=x.dtype, device=x.device)
    assert tuple(out.shape) == (towers, n, h, w, 256) and out.dtype == x.dtype and out.is_contiguous()
    prof = PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    T.conv3x3_halo_masked_paired(x, pc.weight_halo, pc.bias, mask, tile_map, out, pc.cout, pc.k_pad, pc.cout_pad, int(act), int(bool(in_paired)))
    if prof is not None:
        e1.record()
        npix = float(mask.sum().item()) if mask_pixels is None else float(mask_pixels)
        prof.records.append(('conv3x3_halo2_kernel', 2.0 * npix * n * pc.cout * 9 * pc.cin, e0, e1))
        prof.layer(f'conv3x3_halo2 atlas {pc.cin}->{towers}x256 (both head towers)', 2.0 * npix * n * pc.cout * 9 * pc.cin,
                   2 * npix * n * ((towers if in_paired else 1) * pc.cin + pc.cout) * x.element_size() + _nbytes(pc.weight), e0, e1)
    return out


def atlas_pack(levels, atlas, offs):
    """levels[l] (N,h,w,C) -> atlas[:, oy:oy+h, ox:ox+w] for every level, one launch."""
    _need_cuda(atlas, *levels)
    assert atlas.is_contiguous()
    T.atlas_pack(list(levels), atlas, [int(o[0]) for o in offs], [int(o[1]) for o in offs])


def atlas_unpack(atlas, shapes, offs):
    """-> [atlas[:, oy:oy+h, ox:ox+w].contiguous() for every level], one launch."""
    _need_cuda(atlas)
    assert atlas.is_contiguous()
    n, _, _, c = atlas.shape
    levels = [torch.empty((n, h, w, c), dtype=atlas.dtype, device=atlas.device) for (h, w) in shapes]
    T.atlas_unpack(atlas, levels, [int(o[0]) for o in offs], [int(o[1]) for o in offs])
    return levels


def vgg_stem(x, ps):
    """x: (N,H,W,4|8) bf16 normalised input -> (N,H/2,W/2,64) bf16 = pool(relu(conv(relu(conv(x)))))."""
    _need_cuda(x)
    assert x.dtype == BF16 and x.is_contiguous() and x.shape[3] in (4, 8)
    n, h, w, c = x.shape
    out = torch.empty((n, h // 2, w // 2, 64), dtype=BF16, device=x.device)
    prof = PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    T.vgg_stem_fused(x, ps.w1, ps.b1, ps.w2, ps.b2, out)
    if prof is not None:
        e1.record()
        prof.records.append(('vgg_stem2_kernel', ps.flops_per_pixel * n * h * w, e0, e1))
    return out


# ---------------------------------------------------------------------------
# constant-padding tile skipping of the embedder (csrc/skiplist.hip; include/cvpce_amd.h)
# ---------------------------------------------------------------------------
def crop_extents(boxes, count, h0, w0, size=256, out=None, per_image=0):
    """boxes (P,4) f32 device [+ count (1,) int32 device] -> (P,2) int32: content rows / columns of every crop that
    `crop_resize` makes of them; pixels beyond are exactly the pad constant.  per_image > 0: the slots of several images of one
    size (per_image each) with one count per image, in one launch."""
    _need_cuda(boxes, count)
    boxes = boxes.to(torch.float32).contiguous()
    if out is None:
        out = torch.empty((boxes.shape[0], 2), dtype=torch.int32, device=boxes.device)
    T.crop_extents(boxes, count, int(per_image), int(h0), int(w0), int(size), out)
    return out


def pad_extents(images, pad=0.5):
    """(B,3,S,S) f32 crops -> (B,2) int32: rows / columns beyond which every pixel equals `pad` in all channels (read off the data)."""
    _need_cuda(images)
    ext = torch.empty((images.shape[0], 2), dtype=torch.int32, device=images.device)
    T.pad_extents(images, float(pad), ext)
    return ext


def skip_layer(h, w, tile_h, tile_w, out_ops, in_h, in_w, in_ops, skip):
    """One `cvpce_skip_layer` as the flat int list `embed_worklists` takes."""
    return [int(v) for v in (h, w, tile_h, tile_w, out_ops, in_h, in_w, in_ops, skip)]


def embed_worklists(ext0, n_images, size, pool_mask, layers, max_tiles, want_computed=False):
    """ext0 (n_images - 1, 2) int32 (the last image is the implied constant crop); pool_mask: the pass's op chain (bit i: op i is
    a 2x2 pool, else a 3x3 conv); layers: list of `skip_layer` -> (lists (2 L, n_images * max_tiles) int64: the layers' lists, then
    their strip lists; counts (3 L,) int32: tiles per layer, the layers' MFMA work in sixteenths of a tile, strip entries [, computed (L, n_images, 2) int32: conv rows computed /
    tile columns listed per crop]), all on the device, no synchronisation."""
    _need_cuda(ext0)
    dev = ext0.device if ext0 is not None else torch.device('cuda', torch.cuda.current_device())
    lists = torch.empty((2 * len(layers), n_images * max_tiles), dtype=torch.int64, device=dev)       # rows L ..: the layers' strip lists
    counts = torch.empty((3 * len(layers),), dtype=torch.int32, device=dev)
    computed = torch.empty((len(layers), n_images, 2), dtype=torch.int32, device=dev) if want_computed else None
    T.embed_worklists(ext0, int(n_images), int(size), int(pool_mask), [v for l in layers for v in l], lists, counts, computed)
    return (lists, counts, computed) if want_computed else (lists, counts)


def mac_init(desc, off, row_suffix_max, col_suffix_max, computed):
    """desc[:, off:off+C] = what the work list of a MAC layer leaves to the constant crop (see include/cvpce_amd.h cvpce_mac_init)."""
    _need_cuda(desc, row_suffix_max, col_suffix_max, computed)
    T.mac_init(desc, int(off), row_suffix_max, col_suffix_max, computed)


def vgg_stem_list(x, const_in, ps, work, count):
    """`vgg_stem` over a work list: x (N-1,H,W,4|8) + the constant crop const_in (H,W,c) -> (N,H/2,W/2,64)."""
    _need_cuda(x, const_in, work, count)
    assert x.dtype == BF16 and x.is_contiguous() and x.shape[3] in (4, 8) and const_in.is_contiguous()
    n1, h, w, c = x.shape
    out = torch.empty((n1 + 1, h // 2, w // 2, 64), dtype=BF16, device=x.device)
    prof = PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    T.vgg_stem_fused_list(x, const_in, ps.w1, ps.b1, ps.w2, ps.b2, out, work, count)
    if prof is not None:
        e1.record()
        prof.records.append(('vgg_stem2_kernel', ps.flops_per_pixel * n1 * h * w, e0, e1, count, ps.flops_per_pixel * 256, f'stem@{h}'))   # (the stem computes whole tiles)
    return out


def conv2d_list(x, pc, work, count, act=1, pool=False, mac=None, mac_off=0, store=True, units=None, strips=None, out=None):
    """3x3 / s1 / p1 conv (+ReLU, + fused MaxPool2d(2,2), + fused MAC descriptor) over a work list; x (N,H,W,Cin) with the
    constant crop as image N - 1.  strips = (strip list, its count): a second launch computes the layer's strip list (tiles with
    only 4 useful rows, three at a time) into the same output.  Returns the output tensor (None with store=False)."""
    _need_cuda(x, work, count, mac)
    assert x.dtype == BF16 and pc.dtype == BF16 and x.is_contiguous() and x.dim() == 4
    assert pc.cin_pad % 64 == 0 and (pc.kh, pc.kw, pc.stride, pc.pad) == (3, 3, 1, 1) and pc.cout % 8 == 0 and pc.cout > 64
    n, h, w, cin = x.shape
    oshape = (n, h // 2, w // 2, pc.cout) if pool else (n, h, w, pc.cout)
    if not store:
        out = None
    elif out is None:
        out = torch.empty(oshape, dtype=BF16, device=x.device)
    else:
        assert tuple(out.shape) == oshape and out.dtype == BF16 and out.is_contiguous()
    prof = PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    T.conv3x3_halo_list(x, pc.weight_halo, pc.bias, out, mac, int(mac_off), pc.cout, pc.k_pad, pc.cout_pad, int(act), int(pool and store), work, count)
    if prof is not None:
        e1.record()
        wide = pc.cout <= 128
        tile_flops = 2.0 * 16 * (32 if wide else 16) * pc.cout * 9 * pc.cin
        # (`units` counts the layer's whole MFMA work in sixteenths of a tile, its strip launch included: a strip-list entry is 4 of
        # them (csrc/skiplist.hip) and is filed with the STRIP launch below, a different template instance with its own duration)
        lunits = (units - 4 * strips[1]) if (units is not None and strips is not None) else units
        prof.records.append(('conv3x3_halo3_kernel' if wide else 'conv3x3_halo2_kernel', 2.0 * (n - 1) * h * w * pc.cout * 9 * pc.cin, e0, e1)
                            + ((lunits, tile_flops / 16) if units is not None else (count, tile_flops)) + (f'{pc.cin}->{pc.cout}@{h}' + ('mac' if mac is not None else ''),))
    if strips is not None:
        assert pc.cout > 128
        T.conv3x3_halo_strips(x, pc.weight_halo, pc.bias, out, mac, int(mac_off), pc.cout, pc.k_pad, pc.cout_pad, int(act), int(pool and store), strips[0], strips[1])
        if prof is not None:
            e2 = torch.cuda.Event(enable_timing=True)
            e2.record()
            prof.records.append(('conv3x3_halo2_kernel[strips]', 0.0, e1, e2, strips[1], tile_flops / 4, f'{pc.cin}->{pc.cout}@{h}' + ('mac' if mac is not None else '')))
    return out


def maxpool2d(x, k, stride, pad=0):
    _need_cuda(x)
    n, h, w, c = x.shape
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    out = torch.empty((n, ho, wo, c), dtype=x.dtype, device=x.device)
    T.maxpool2d_nhwc(x, out, k, stride, pad)
    return out


def relu(x):
    _need_cuda(x)
    out = torch.empty_like(x)
    T.relu(x, out)
    return out


USE_GAUSS_TAIL = _os.environ.get('CVPCE_GAUSS_TAIL', '1') != '0'   # A/B switch: the Gaussian subnet's two 1x1 layers in one launch


def can_fuse_gauss_tail(x, c4, c5):
    """cvpce_gauss_tail covers conv1x1(16 -> 16) + ReLU followed by conv1x1(16 -> 1) (proposals.py:96-107)."""
    return (USE_GAUSS_TAIL and not FORCE_GENERIC_CONV and x.shape[-1] == 16 and c4.dtype == c5.dtype == x.dtype
            and (c4.cin, c4.cin_pad, c4.cout, c4.kh, c4.kw, c4.stride, c4.pad) == (16, 16, 16, 1, 1, 1, 0)
            and (c5.cin, c5.cin_pad, c5.cout, c5.kh, c5.kw, c5.stride, c5.pad) == (16, 16, 1, 1, 1, 1, 0))


def gauss_tail(x, c4, c5, act):
    """act(conv1x1_c5(relu(conv1x1_c4(x)))) -> (N,H,W,1) f32; x (N,H,W,16) bf16 | fp16; act 1 = ReLU, 2 = Tanh."""
    _need_cuda(x)
    assert can_fuse_gauss_tail(x, c4, c5) and x.is_contiguous() and act in (0, 1, 2)
    out = torch.empty(x.shape[:3] + (1,), dtype=torch.float32, device=x.device)
    prof = PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    T.gauss_tail(x, c4.weight, c4.bias, c5.weight, c5.bias, out, c4.k_pad, int(act))
    if prof is not None:
        e1.record()
        prof.records.append(('gauss_tail_kernel', 2.0 * x.shape[0] * x.shape[1] * x.shape[2] * (16 * 16 + 16), e0, e1))
        prof.byte_records.append(('gauss_tail_kernel', _nbytes(x, out), e0, e1))
        prof.layer(f'gauss_tail {x.shape[1]}x{x.shape[2]} 16->16->1', 2.0 * x.shape[0] * x.shape[1] * x.shape[2] * (16 * 16 + 16), _nbytes(x, out), e0, e1)
    return out


USE_GAUSS_SUBNET = _os.environ.get('CVPCE_GAUSS_SUBNET', '1') != '0'   # A/B switch: the whole Gaussian subnet in one launch (csrc/gauss_subnet.hip)


def can_fuse_gauss_subnet(x, convs):
    """cvpce_gauss_subnet covers GaussianSubnet as the reference builds it (proposals.py:81-107): 3x3 64 -> 32 over the 2x-upsampled input,
    3x3 32 -> 32, 3x3 32 -> 16, 1x1 16 -> 16, 1x1 16 -> 1."""
    if not (USE_GAUSS_SUBNET and not FORCE_GENERIC_CONV and len(convs) == 5 and x.dim() == 4 and x.shape[-1] == 64):
        return False
    c1, c2, c3, c4, c5 = convs
    shape = lambda c: (c.cin, c.cin_pad, c.cout, c.kh, c.kw, c.stride, c.pad, c.k_pad)
    return (all(c.dtype == x.dtype for c in convs) and shape(c1) == (64, 64, 32, 3, 3, 1, 1, 576) and shape(c2) == (32, 32, 32, 3, 3, 1, 1, 288)
            and shape(c3) == (32, 32, 16, 3, 3, 1, 1, 288) and shape(c4)[:7] == (16, 16, 16, 1, 1, 1, 0) and shape(c5)[:7] == (16, 16, 1, 1, 1, 1, 0)
            and c4.k_pad % 4 == 0 and x.numel() * 2 < 2 ** 32 - 65536)


def gauss_subnet(x, convs, act):
    """GaussianSubnet.forward on x (N,H/2,W/2,64) bf16 | fp16, GaussianLayer's output BEFORE its 2x upsample -> (N,H,W,1) f32;
    act 1 = ReLU, 2 = Tanh (proposals.py:85).  One launch; no intermediate layer reaches memory."""
    _need_cuda(x)
    assert can_fuse_gauss_subnet(x, convs) and x.is_contiguous() and act in (0, 1, 2)
    c1, c2, c3, c4, c5 = convs
    n, hs, ws, _ = x.shape
    out = torch.empty((n, 2 * hs, 2 * ws, 1), dtype=torch.float32, device=x.device)
    prof = PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    T.gauss_subnet(x, c1.weight, c1.bias, c2.weight, c2.bias, c3.weight, c3.bias, c4.weight, c4.bias, c4.k_pad, c5.weight, c5.bias, c5.k_pad, out, int(act))
    if prof is not None:
        e1.record()
        flops = 2.0 * n * 4 * hs * ws * (9 * 64 * 32 + 9 * 32 * 32 + 9 * 32 * 16 + 16 * 16 + 16)
        prof.records.append(('gauss_subnet_kernel', flops, e0, e1))
        # (not an HBM-bound stage any more: 46 MB of compulsory traffic for 83 GFLOP at 8 images -- it is filed with the conv kernels and, by
        #  its arithmetic intensity, against the MFMA roof in the detector's per-layer table)
        prof.layer(f'gauss_subnet {2 * hs}x{2 * ws} 64->32->32->16->16->1 (2x upsampled input)', flops, _nbytes(x, out), e0, e1)
    return out


def global_max_into(x, out, out_off):
    """x NHWC bf16 -> out[:, out_off:out_off+C] (f32) = amax over H,W."""
    _need_cuda(x, out)
    n, h, w, c = x.shape
    T.global_max_nhwc(x, out, out_off)


def l2_normalize(desc, eps=1e-8, want_bf16=False):
    _need_cuda(desc)
    out = torch.empty_like(desc)
    out_bf = torch.empty(desc.shape, dtype=BF16, device=desc.device) if want_bf16 else None
    T.l2_normalize(desc, out, out_bf, float(eps))
    return (out, out_bf) if want_bf16 else out


# ---------------------------------------------------------------------------
# input side
# ---------------------------------------------------------------------------
def gln_transform_into(img, batch, index, h, w, mean, std):
    """img (3,H0,W0) f32 cuda -> batch[index] (Hp,Wp,8) bf16 | fp16 (the batch tensor's type)."""
    _need_cuda(img, batch)
    assert img.dtype == torch.float32 and img.is_contiguous()
    _, hp, wp, c8 = batch.shape
    assert c8 == 8
    T.gln_transform(img, batch, index, h, w, [float(v) for v in mean], [float(v) for v in std])


def gln_transform_batch(images, batch, sizes, mean, std):
    """images[i] (3,H0,W0) f32 cuda -> batch[i] (Hp,Wp,8), resized to sizes[i] = (h, w): the whole batch in one launch."""
    _need_cuda(batch, *images)
    assert batch.shape[0] == len(images) == len(sizes) and batch.shape[3] == 8 and batch.is_contiguous()
    prof = PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    T.gln_transform_batch([i.contiguous() for i in images], batch, [int(s[0]) for s in sizes], [int(s[1]) for s in sizes],
                          [float(v) for v in mean], [float(v) for v in std])
    if prof is not None:
        e1.record()
        prof.byte_records.append(('gln_transform_batch_kernel', _nbytes(batch, *images), e0, e1))
        prof.layer('gln_transform_batch', 0.0, _nbytes(batch, *images), e0, e1)


MAX_CROPS_PER_LAUNCH = 65528   # (the grid z limit, a multiple of 8: the crop kernel deals crops to XCDs in groups of 8)


def crop_resize(img, boxes, size=256, mode=0, mean=None, std=None, count=None, out=None, content_ext=None):
    """img (3,H0,W0) f32, boxes (P,4) f32 xyxy (device) -> (P,3,S,S) f32 [mode 0] | (P,S,S,8) bf16 [mode 1] | (P,S,S,4) bf16
    [mode 2: the 8-byte pixels the fused VGG stem reads; half the bytes of mode 1].
    content_ext (P,2) int32 (`crop_extents` of the same boxes; modes 1 / 2): only the crops' CONTENT is written -- the constant padding
    beyond the extents stays unwritten, which only the work-list embedder may be given (it reads it from the constant crop)."""
    _need_cuda(img, boxes)
    assert img.dtype == torch.float32 and img.is_contiguous()
    boxes = boxes.to(torch.float32).contiguous()
    p = boxes.shape[0]
    if out is None:
        out = (torch.empty((p, 3, size, size), dtype=torch.float32, device=img.device) if mode == 0
               else torch.empty((p, size, size, 8 if mode == 1 else 4), dtype=BF16, device=img.device))
    assert out.shape[-1] == {0: size, 1: 8, 2: 4}[mode]
    m = [float(v) for v in mean] if mean is not None else None
    s = [float(v) for v in std] if std is not None else None
    prof = PROFILE
    for start in range(0, p, MAX_CROPS_PER_LAUNCH):
        nb = min(MAX_CROPS_PER_LAUNCH, p - start)
        cnt = None
        if count is not None:
            assert start == 0 and p <= MAX_CROPS_PER_LAUNCH
            cnt = count
        if prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if content_ext is not None:
            assert mode in (1, 2) and size % 2 == 0
            T.crop_resize_content(img, boxes[start:start + nb], cnt, out[start:start + nb], size, mode, m, s, content_ext[start:start + nb])
        else:
            T.crop_resize(img, boxes[start:start + nb], cnt, out[start:start + nb], size, mode, m, s)
        if prof is not None:
            e1.record()
            # algorithmic bytes: every source pixel of every (valid) box once (3 planes of f32) + the crop pixels written
            v = nb if cnt is None else int(cnt.reshape(-1)[0])
            b = boxes[start:start + v].to(torch.long)
            area = int(((b[:, 2] - b[:, 0]).clamp(min=0) * (b[:, 3] - b[:, 1]).clamp(min=0)).sum())
            if content_ext is not None:
                e = content_ext[start:start + v].to(torch.long)
                written = int((e[:, 0] * e[:, 1]).sum()) * out.shape[-1] * out.element_size()
            else:
                written = v * out[0].numel() * out.element_size()
            prof.byte_records.append(('crop_resize_kernel', area * 3 * 4 + written, e0, e1))
    return out


def pack_embed_input(images, to_tanh, mean, std):
    """(B,3,S,S) f32 -> (B,S,S,8) bf16 normalised."""
    _need_cuda(images)
    images = images.to(torch.float32).contiguous()
    b, _, s, s2 = images.shape
    assert s == s2
    out = torch.empty((b, s, s, 8), dtype=BF16, device=images.device)
    T.pack_embed_input(images, out, int(to_tanh), [float(v) for v in mean], [float(v) for v in std])
    return out


# ---------------------------------------------------------------------------
# detector post-processing
# ---------------------------------------------------------------------------
def detect_postprocess(logits, regs, grids, strides, base_anchors, image_hw, ratios, num_anchors, num_classes,
                       topk, score_thresh, nms_thresh, xform_clip, detections_per_img, conf_thresh):
    """logits[l] (N, gh*gw*A*K) f32, regs[l] (N, gh*gw*A, 4) f32 (views of the NHWC conv outputs).

    Returns boxes (N,dpi,4), scores (N,dpi), labels (N,dpi) i64, count (N,) i32, conf_count (N,) i32.
    """
    _need_cuda(*logits, *regs, base_anchors, image_hw, ratios)
    L, n = len(logits), logits[0].shape[0]
    dev = logits[0].device
    ws_bytes = lib.cvpce_detect_workspace_bytes(n, L, topk)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    nd = n * detections_per_img
    if ONE_OUTPUT_BLOCK:
        # the five zero-padded outputs are views of ONE zero-filled block: one fill launch on the chain instead of five, and a
        # caller that has to copy the results out of a captured graph's memory copies one block (clone_views)
        zero = torch.zeros(nd * 28 + n * 8, dtype=torch.uint8, device=dev)
        labels = zero[:nd * 8].view(torch.int64).view(n, detections_per_img)
        boxes = zero[nd * 8:nd * 24].view(torch.float32).view(n, detections_per_img, 4)
        scores = zero[nd * 24:nd * 28].view(torch.float32).view(n, detections_per_img)
        count = zero[nd * 28:nd * 28 + n * 4].view(torch.int32)
        conf = zero[nd * 28 + n * 4:].view(torch.int32)
    else:
        boxes = torch.zeros((n, detections_per_img, 4), dtype=torch.float32, device=dev)
        scores = torch.zeros((n, detections_per_img), dtype=torch.float32, device=dev)
        labels = torch.zeros((n, detections_per_img), dtype=torch.int64, device=dev)
        count = torch.zeros((n,), dtype=torch.int32, device=dev)
        conf = torch.zeros((n,), dtype=torch.int32, device=dev)
    for t in list(logits) + list(regs):
        assert t.dtype == torch.float32 and t.is_contiguous()
    prof = PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    T.detect_postprocess(list(logits), list(regs), [int(g[0]) for g in grids], [int(g[1]) for g in grids], [int(s[0]) for s in strides],
                         [int(s[1]) for s in strides], base_anchors, image_hw, ratios, num_anchors, num_classes, topk, float(score_thresh),
                         float(nms_thresh), float(xform_clip), detections_per_img, float(conf_thresh), ws, boxes, scores, labels, count, conf)
    if prof is not None:
        e1.record()
        prof.layer(f'detect_postprocess (top-k {topk}, NMS, dpi {detections_per_img}: 5 launches)', 0.0, _nbytes(*logits, *regs, boxes, scores, labels), e0, e1)
    return boxes, scores, labels, count, conf


def clone_views(tensors):
    """tuple of clones of `tensors`; tensors that share one storage (views of one block) are cloned with ONE copy of it.
    (Grouped by storage, not by `_base`: a dtype-changing `.view()` starts a new base chain on the same storage.)"""
    blocks, out = {}, []
    for t in tensors:
        st = t.untyped_storage()
        key = st.data_ptr()
        if key not in blocks:
            whole = torch.empty(0, dtype=torch.uint8, device=t.device).set_(st, 0, (st.nbytes(),), (1,))
            blocks[key] = whole.clone().untyped_storage()
        out.append(torch.empty(0, dtype=t.dtype, device=t.device).set_(blocks[key], t.storage_offset(), t.size(), t.stride()))
    return tuple(out)


# ---------------------------------------------------------------------------
# matcher
# ---------------------------------------------------------------------------
def row_norms(x, eps=1e-8):
    _need_cuda(x)
    assert x.is_contiguous() and x.dtype in (BF16, torch.float32)
    out = torch.empty((x.shape[0],), dtype=torch.float32, device=x.device)
    T.row_norms(x, out, float(eps))
    return out


def match_topk(queries, gallery, k=1, q_norms=None, g_norms=None, return_distance=False):
    """(Q,D), (G,D) same dtype (bf16 | f32), D % 64 == 0 -> (Q,k) int64 [+ (Q,k) f32 distances]."""
    _need_cuda(queries, gallery)
    assert queries.dtype == gallery.dtype and queries.dtype in (BF16, torch.float32)
    assert queries.is_contiguous() and gallery.is_contiguous() and queries.shape[1] == gallery.shape[1]
    qn, d = queries.shape
    gn = gallery.shape[0]
    k = min(int(k), gn)          # the reference's argsort[:, :k] returns min(k, G) columns (classification.py:95)
    if k == 0 or qn == 0:
        idx = torch.empty((qn, k), dtype=torch.int64, device=queries.device)
        return (idx, torch.empty((qn, k), dtype=torch.float32, device=queries.device)) if return_distance else idx
    if q_norms is None:
        q_norms = row_norms(queries)
    if g_norms is None:
        g_norms = row_norms(gallery)
    idx = torch.empty((qn, k), dtype=torch.int64, device=queries.device)
    dist = torch.empty((qn, k), dtype=torch.float32, device=queries.device) if return_distance else None
    ws_bytes = lib.cvpce_match_workspace_bytes(qn, gn, k)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=queries.device)
    if MATCH_ONE_LAUNCH and k == 1 and queries.dtype == BF16 and qn <= MATCH_STATE_QUERIES and (_match_state_key(queries.device) in _MATCH_STATE or not torch.cuda.is_current_stream_capturing()):
        # the one-launch form: a state block per (device, stream) -- launches of one stream are ordered, so they can share a block
        T.match_topk_state(queries, gallery, q_norms, g_norms, k, ws, match_state(queries.device), idx, dist)
    else:
        T.match_topk(queries, gallery, q_norms, g_norms, k, ws, idx, dist)
    return (idx, dist) if return_distance else idx


MATCH_STATE_QUERIES = 16384      # queries per launch the one-launch top-1 search holds keys for (128 KiB per state block)
_MATCH_STATE = {}
# The one-launch form of the k = 1 search is an opt-in (measured slower than two launches, profiles/r05_rejected_experiments.md): only
# when it is on does a launch get a state block (128 KiB per device and stream) and go through cvpce_match_topk_state.  A hipGraph
# captured with it on must be replayed on its capture stream: the block belongs to that stream's launch order.
MATCH_ONE_LAUNCH = _os.environ.get('CVPCE_MATCH_FUSED', '0') not in ('', '0')


def match_set_core(core=0, nq=0, mg=0, one_launch=0):
    """`cvpce_match_set_core` (include/cvpce_amd.h): pin the bf16 search's core / tile (0 = the cost model's choice) and switch the
    one-launch form of k = 1 on or off -- here AND in this module's dispatch.  -> the library's status (0 = ok)."""
    global MATCH_ONE_LAUNCH
    rc = int(lib.cvpce_match_set_core(int(core), int(nq), int(mg), int(one_launch)))
    if rc == 0:
        MATCH_ONE_LAUNCH = bool(one_launch)
    return rc


def _match_state_key(device):
    return (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(device).cuda_stream)


def match_state(device):
    """The persistent state block of the one-launch top-1 search for the CURRENT stream of `device` (include/cvpce_amd.h
    cvpce_match_topk_state): initialised once, restored by every launch that uses it.  (Never created inside a stream capture -- its
    memory would belong to that graph's pool: a capture on a stream without a block takes the plain two-launch entry point.)"""
    key = _match_state_key(device)
    st = _MATCH_STATE.get(key)
    if st is None:
        st = torch.empty(lib.cvpce_match_state_bytes(MATCH_STATE_QUERIES) // 8, dtype=torch.int64, device=device)
        T.match_state_init(st)
        _MATCH_STATE[key] = st
    return st


def pad_features(x, multiple=64):
    d = x.shape[1]
    dp = (d + multiple - 1) // multiple * multiple
    if dp == d:
        return x.contiguous()
    out = torch.zeros((x.shape[0], dp), dtype=x.dtype, device=x.device)
    out[:, :d] = x
    return out


# ---------------------------------------------------------------------------
# calibration
# ---------------------------------------------------------------------------
def probe_mfma_bf16(shape, seconds=2.0, iters=20000, workgroups=256):
    """Sustained bare-MFMA rate of this device (TFLOP/s): `shape` 0 = 32x32x16, 1 = 16x16x32; register operands, random
    data, one wave per SIMD, run back to back for at least `seconds` (the clock settles under the power limit)."""
    import time
    dev = torch.device('cuda', torch.cuda.current_device())
    operands = torch.randn(1 << 16, device=dev).to(BF16)
    sink = torch.empty(workgroups * 256, dtype=torch.float32, device=dev)
    flop = workgroups * 4 * iters * 16 * (32768 if shape == 0 else 16384)
    launch = lambda: T.probe_mfma_bf16(int(shape), int(iters), operands, sink, int(workgroups))
    launch()
    torch.cuda.synchronize()
    t0, last = time.perf_counter(), 0.0
    while True:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            launch()
        e1.record()
        torch.cuda.synchronize()
        last = 4 * flop / (e0.elapsed_time(e1) * 1e-3) / 1e12
        if time.perf_counter() - t0 >= seconds:
            break
    return last        # the rate of the LAST group of launches: the settled clock, not the cold-start burst


def probe_l2_stream(mib=2, seconds=1.0, workgroups=512, iters=64):
    """TB/s at which all CUs together stream one L2-resident buffer of `mib` MiB into registers (csrc/probe.hip)."""
    import time
    dev = torch.device('cuda', torch.cuda.current_device())
    buf = torch.randint(0, 1 << 30, (mib << 18,), dtype=torch.int32, device=dev)
    sink = torch.empty(workgroups * 512, dtype=torch.float32, device=dev)
    T.probe_l2_stream(buf, 2, sink, workgroups)
    torch.cuda.synchronize()
    t0, last = time.perf_counter(), 0.0
    while True:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        T.probe_l2_stream(buf, iters, sink, workgroups)
        e1.record()
        torch.cuda.synchronize()
        last = workgroups * iters * (mib << 20) / (e0.elapsed_time(e1) * 1e-3) / 1e12
        if time.perf_counter() - t0 >= seconds:
            break
    return last
