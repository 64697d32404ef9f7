"""PyTorch-ROCm custom ops over the C ABI: `torch.ops.cvpce_amd.*` (north_star: "hand-written CDNA4 HIP kernels exposed
to Python through PyTorch-ROCm custom ops").

One dispatcher entry per C-ABI entry point of include/cvpce_amd.h, registered with `torch.library` for the CUDA (= HIP on
ROCm) dispatch key ONLY: there is no CPU kernel, so a CPU tensor fails in the dispatcher ("no CPU fallback" holds at the
op level too).  All ops are out-variants -- the caller (cvpce_amd/ops.py) allocates, the op launches one hand-written
kernel schedule on the current HIP stream through ctypes -> libcvpce_hip.so, no allocation, no synchronisation -- so they
are safe under HIP stream capture (cvpce_amd.models.proposals captures the detector schedule in a hipGraph).
"""
import ctypes

import torch

from . import _lib
from ._lib import lib, check

_L = torch.library.Library('cvpce_amd', 'DEF')
NAMES = []


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _by_dtype(x, bf16_name, f16_name, *same):
    """The C entry point for x's element type: bf16 (default storage) or fp16 (the detector's accuracy mode).  Every other 16-bit
    operand of the call must have the same type -- the kernels cannot mix them."""
    if x.dtype not in (torch.bfloat16, torch.float16):
        raise RuntimeError(f'{bf16_name}: activations must be bfloat16 or float16, got {x.dtype}')
    for t in same:
        if t is not None and t.dtype != x.dtype:
            raise RuntimeError(f'{bf16_name}: operand of type {t.dtype} with {x.dtype} activations')
    return getattr(lib, f16_name if x.dtype == torch.float16 else bf16_name)


def _op(schema):
    def deco(fn):
        name = schema.split('(')[0]
        _L.define(schema)
        _L.impl(name, fn, 'CUDA')
        NAMES.append(name)
        return fn
    return deco


@_op('conv2d_nhwc(Tensor x, Tensor weight, Tensor? bias, Tensor? residual, Tensor(a!) out, int cout, int kh, int kw, int stride, '
     'int pad, int ho, int wo, int k_pad, int cout_pad, int act, int out_f32, int in_up_shift, int res_mode, int pool, '
     'int force_generic) -> ()')
def _conv2d_nhwc(x, weight, bias, residual, out, cout, kh, kw, stride, pad, ho, wo, k_pad, cout_pad, act, out_f32, in_up_shift,
                 res_mode, pool, force_generic):
    n, h, w, cin = x.shape
    hr, wr = (residual.shape[1], residual.shape[2]) if residual is not None else (0, 0)
    fn = _by_dtype(x, 'cvpce_conv2d_nhwc_bf16', 'cvpce_conv2d_nhwc_f16', weight, residual, None if out_f32 else out)
    check(fn(_p(x), _p(weight), _p(bias), _p(residual), _p(out), n, h, w, cin, cout, kh, kw, stride, pad, ho, wo,
             k_pad, cout_pad, act, out_f32, in_up_shift, res_mode if residual is not None else 0, hr, wr, pool,
             force_generic, _stream()), 'cvpce_conv2d_nhwc')


@_op('conv2d_splitk(Tensor x, Tensor weight, Tensor? bias, Tensor? residual, Tensor(a!) out, int cout, int kh, int kw, int stride, '
     'int pad, int ho, int wo, int k_pad, int cout_pad, int act, int out_f32, int in_up_shift, int res_mode, int ksplit, '
     'Tensor(b!) workspace) -> ()')
def _conv2d_splitk(x, weight, bias, residual, out, cout, kh, kw, stride, pad, ho, wo, k_pad, cout_pad, act, out_f32, in_up_shift,
                   res_mode, ksplit, workspace):
    n, h, w, cin = x.shape
    hr, wr = (residual.shape[1], residual.shape[2]) if residual is not None else (0, 0)
    fn = _by_dtype(x, 'cvpce_conv2d_splitk_bf16', 'cvpce_conv2d_splitk_f16', weight, residual, None if out_f32 else out)
    check(fn(_p(x), _p(weight), _p(bias), _p(residual), _p(out), n, h, w, cin, cout, kh, kw, stride, pad, ho, wo,
             k_pad, cout_pad, act, out_f32, in_up_shift, res_mode if residual is not None else 0, hr, wr, ksplit,
             _p(workspace), workspace.numel() * workspace.element_size(), _stream()), 'cvpce_conv2d_splitk')


@_op('conv1x1_nhwc(Tensor x, Tensor weight, Tensor? bias, Tensor? residual, Tensor(a!) out, int cout, int stride, int ho, int wo, '
     'int k_pad, int cout_pad, int relu, int res_mode) -> ()')
def _conv1x1_nhwc(x, weight, bias, residual, out, cout, stride, ho, wo, k_pad, cout_pad, relu, res_mode):
    n, h, w, cin = x.shape
    hr, wr = (residual.shape[1], residual.shape[2]) if residual is not None else (0, 0)
    fn = _by_dtype(x, 'cvpce_conv1x1_nhwc_bf16', 'cvpce_conv1x1_nhwc_f16', weight, residual, out)
    check(fn(_p(x), _p(weight), _p(bias), _p(residual), _p(out), n, h, w, cin, cout, stride, ho, wo, k_pad,
             cout_pad, relu, res_mode if residual is not None else 0, hr, wr, _stream()), 'cvpce_conv1x1_nhwc')


@_op('conv3x3_halo(Tensor x, Tensor weight, Tensor? bias, Tensor(a!) out, int cout, int k_pad, int cout_pad, int relu, int pool) -> ()')
def _conv3x3_halo(x, weight, bias, out, cout, k_pad, cout_pad, relu, pool):
    n, h, w, cin = x.shape
    fn = _by_dtype(x, 'cvpce_conv3x3_halo', 'cvpce_conv3x3_halo_f16', weight, out)
    check(fn(_p(x), _p(weight), _p(bias), _p(out), n, h, w, cin, cout, k_pad, cout_pad, relu, pool, _stream()), 'cvpce_conv3x3_halo')


@_op('conv3x3_halo_thin_out(Tensor x, Tensor weight, Tensor? bias, Tensor(a!) out, int cout, int k_pad, int cout_pad) -> ()')
def _conv3x3_halo_thin_out(x, weight, bias, out, cout, k_pad, cout_pad):
    n, h, w, cin = x.shape
    if out.dtype != torch.float32:
        raise RuntimeError('cvpce_conv3x3_halo_thin_out: the output is float32')
    fn = _by_dtype(x, 'cvpce_conv3x3_halo_thin_out', 'cvpce_conv3x3_halo_thin_out_f16', weight)
    check(fn(_p(x), _p(weight), _p(bias), _p(out), n, h, w, cin, cout, k_pad, cout_pad, _stream()), 'cvpce_conv3x3_halo_thin_out')


@_op('conv3x3_halo_mac(Tensor x, Tensor weight, Tensor? bias, Tensor(a!)? out, Tensor(b!) mac, int mac_off, int cout, int k_pad, '
     'int cout_pad, int pool) -> ()')
def _conv3x3_halo_mac(x, weight, bias, out, mac, mac_off, cout, k_pad, cout_pad, pool):
    n, h, w, cin = x.shape
    if x.dtype != torch.bfloat16 or weight.dtype != torch.bfloat16:
        raise RuntimeError('cvpce_conv3x3_halo_mac: bf16 only (the embedder has no fp16 mode)')
    check(lib.cvpce_conv3x3_halo_mac(_p(x), _p(weight), _p(bias), _p(out), _p(mac), mac.shape[1], mac_off, n, h, w, cin, cout, k_pad,
                                     cout_pad, pool, _stream()), 'cvpce_conv3x3_halo_mac')


@_op('conv3x3_halo_masked_paired(Tensor x, Tensor weight, Tensor? bias, Tensor mask, Tensor? tile_map, Tensor(a!) out, int cout, int k_pad, '
     'int cout_pad, int relu, int in_paired) -> ()')
def _conv3x3_halo_masked_paired(x, weight, bias, mask, tile_map, out, cout, k_pad, cout_pad, relu, in_paired):
    towers = cout // 256
    n, h, w, cin = x.shape[-4:]
    assert cout == 256 * towers and towers >= 2 and tuple(out.shape) == (towers, n, h, w, 256)
    assert tuple(x.shape) == ((towers, n, h, w, cin) if in_paired else (n, h, w, cin))
    fn = _by_dtype(x, 'cvpce_conv3x3_halo_masked_paired', 'cvpce_conv3x3_halo_masked_paired_f16', weight, out)
    check(fn(_p(x), _p(weight), _p(bias), _p(mask), _p(tile_map), tile_map.numel() if tile_map is not None else 0,
             _p(out), n, h, w, cin, cout, k_pad, cout_pad, relu, in_paired, _stream()), 'cvpce_conv3x3_halo_masked_paired')


@_op('conv3x3_halo_masked(Tensor x, Tensor weight, Tensor? bias, Tensor mask, Tensor? tile_map, Tensor(a!) out, int cout, int k_pad, '
     'int cout_pad, int relu) -> ()')
def _conv3x3_halo_masked(x, weight, bias, mask, tile_map, out, cout, k_pad, cout_pad, relu):
    n, h, w, cin = x.shape
    fn = _by_dtype(x, 'cvpce_conv3x3_halo_masked', 'cvpce_conv3x3_halo_masked_f16', weight, out)
    check(fn(_p(x), _p(weight), _p(bias), _p(mask), _p(tile_map), tile_map.numel() if tile_map is not None else 0,
             _p(out), n, h, w, cin, cout, k_pad, cout_pad, relu, _stream()), 'cvpce_conv3x3_halo_masked')


@_op('bottleneck_fused_fm(Tensor x, Tensor res, Tensor w1f, Tensor b1, Tensor w2f, Tensor b2, Tensor w3f, Tensor b3, Tensor(a!) out, int planes) -> ()')
def _bottleneck_fused_fm(x, res, w1f, b1, w2f, b2, w3f, b3, out, planes):
    n, h, w, cin = x.shape
    if w1f.numel() != planes * cin or w2f.numel() != 9 * planes * planes or w3f.numel() != 4 * planes * planes:
        raise RuntimeError('cvpce_bottleneck_fused_fm: fragment-major weights of the wrong size')
    fn = _by_dtype(x, 'cvpce_bottleneck_fused_fm', 'cvpce_bottleneck_fused_fm_f16', res, w1f, w2f, w3f, out)
    check(fn(_p(x), _p(res), _p(w1f), _p(b1), _p(w2f), _p(b2), _p(w3f), _p(b3), _p(out), n, h, w, cin, planes, _stream()), 'cvpce_bottleneck_fused_fm')


@_op('bottleneck_fused(Tensor x, Tensor res, Tensor w1, Tensor b1, Tensor w2, Tensor b2, Tensor w3, Tensor b3, Tensor(a!) out, int planes, '
     'int k1_pad, int k2_pad, int k3_pad, int c1_pad, int c2_pad, int c3_pad) -> ()')
def _bottleneck_fused(x, res, w1, b1, w2, b2, w3, b3, out, planes, k1_pad, k2_pad, k3_pad, c1_pad, c2_pad, c3_pad):
    n, h, w, cin = x.shape
    fn = _by_dtype(x, 'cvpce_bottleneck_fused', 'cvpce_bottleneck_fused_f16', res, w1, w2, w3, out)
    check(fn(_p(x), _p(res), _p(w1), _p(b1), _p(w2), _p(b2), _p(w3), _p(b3), _p(out), n, h, w, cin, planes, k1_pad, k2_pad, k3_pad,
             c1_pad, c2_pad, c3_pad, _stream()), 'cvpce_bottleneck_fused')


@_op('vgg_stem_fused(Tensor x, Tensor w1, Tensor b1, Tensor w2, Tensor b2, Tensor(a!) out) -> ()')
def _vgg_stem_fused(x, w1, b1, w2, b2, out):
    n, h, w, c = x.shape
    if x.dtype != torch.bfloat16 or w1.dtype != torch.bfloat16 or w2.dtype != torch.bfloat16:
        raise RuntimeError('cvpce_vgg_stem_fused: bf16 only (the embedder has no fp16 mode)')
    check(lib.cvpce_vgg_stem_fused(_p(x), c, _p(w1), _p(b1), _p(w2), _p(b2), _p(out), n, h, w, _stream()), 'cvpce_vgg_stem_fused')


@_op('gln_stem_fused(Tensor x, Tensor w_frag, Tensor bias, Tensor(a!) out) -> ()')
def _gln_stem_fused(x, w_frag, bias, out):
    n, h, w, c = x.shape
    fn = _by_dtype(x, 'cvpce_gln_stem_fused', 'cvpce_gln_stem_fused_f16', w_frag, out)
    check(fn(_p(x), _p(w_frag), _p(bias), _p(out), n, h, w, _stream()), 'cvpce_gln_stem_fused')


@_op('maxpool2d_nhwc(Tensor x, Tensor(a!) out, int k, int stride, int pad) -> ()')
def _maxpool2d_nhwc(x, out, k, stride, pad):
    n, h, w, c = x.shape
    fn = _by_dtype(x, 'cvpce_maxpool2d_nhwc_bf16', 'cvpce_maxpool2d_nhwc_f16', out)
    check(fn(_p(x), _p(out), n, h, w, c, k, stride, pad, out.shape[1], out.shape[2], _stream()), 'maxpool')


@_op('relu(Tensor x, Tensor(a!) out) -> ()')
def _relu(x, out):
    check(_by_dtype(x, 'cvpce_relu_bf16', 'cvpce_relu_f16', out)(_p(x), _p(out), x.numel(), _stream()), 'relu')


@_op('conv3x3_thin(Tensor x, Tensor weight, Tensor? bias, Tensor(a!) out, int cout, int k_pad, int cout_pad, int relu, int in_up_shift) -> ()')
def _conv3x3_thin(x, weight, bias, out, cout, k_pad, cout_pad, relu, in_up_shift):
    n, h, w, cin = x.shape
    fn = _by_dtype(x, 'cvpce_conv3x3_thin_bf16', 'cvpce_conv3x3_thin_f16', weight, out)
    check(fn(_p(x), _p(weight), _p(bias), _p(out), n, h << in_up_shift, w << in_up_shift, cin, cout, k_pad, cout_pad, relu, in_up_shift, _stream()),
          'cvpce_conv3x3_thin')


@_op('gauss_tail(Tensor x, Tensor w2, Tensor? b2, Tensor w3, Tensor? b3, Tensor(a!) out, int k2_pad, int act) -> ()')
def _gauss_tail(x, w2, b2, w3, b3, out, k2_pad, act):
    assert x.shape[-1] == 16 and out.dtype == torch.float32 and out.numel() == x.numel() // 16
    check(_by_dtype(x, 'cvpce_gauss_tail_bf16', 'cvpce_gauss_tail_f16', w2, w3)(_p(x), _p(w2), _p(b2), _p(w3), _p(b3), _p(out), x.numel() // 16,
                                                                               k2_pad, act, _stream()), 'gauss_tail')


@_op('gauss_subnet(Tensor x, Tensor w1, Tensor? b1, Tensor w2, Tensor? b2, Tensor w3, Tensor? b3, Tensor w4, Tensor? b4, int k4_pad, '
     'Tensor w5, Tensor? b5, int k5_pad, Tensor(a!) out, int act) -> ()')
def _gauss_subnet(x, w1, b1, w2, b2, w3, b3, w4, b4, k4_pad, w5, b5, k5_pad, out, act):
    n, hs, ws, c = x.shape
    assert c == 64 and out.dtype == torch.float32 and tuple(out.shape[:3]) == (n, 2 * hs, 2 * ws) and out.numel() == n * 4 * hs * ws
    fn = _by_dtype(x, 'cvpce_gauss_subnet_bf16', 'cvpce_gauss_subnet_f16', w1, w2, w3, w4, w5)
    check(fn(_p(x), _p(w1), _p(b1), _p(w2), _p(b2), _p(w3), _p(b3), _p(w4), _p(b4), k4_pad, _p(w5), _p(b5), k5_pad, _p(out), n, 2 * hs, 2 * ws, act,
             _stream()), 'gauss_subnet')


@_op('global_max_nhwc(Tensor x, Tensor(a!) out, int out_off) -> ()')
def _global_max_nhwc(x, out, out_off):
    n, h, w, c = x.shape
    check(lib.cvpce_global_max_nhwc_bf16(_p(x), _p(out), n, h * w, c, out.shape[1], out_off, _stream()), 'global_max')


@_op('l2_normalize(Tensor x, Tensor(a!) out, Tensor(b!)? out_bf16, float eps) -> ()')
def _l2_normalize(x, out, out_bf16, eps):
    check(lib.cvpce_l2_normalize_f32(_p(x), _p(out), _p(out_bf16), x.shape[0], x.shape[1], eps, _stream()), 'l2norm')


@_op('gln_transform(Tensor img, Tensor(a!) batch, int index, int h, int w, float[] mean, float[] std) -> ()')
def _gln_transform(img, batch, index, h, w, mean, std):
    _, hp, wp, _ = batch.shape
    fn = _by_dtype(batch, 'cvpce_gln_transform', 'cvpce_gln_transform_f16')
    check(fn(_p(img), ctypes.c_void_p(batch[index].data_ptr()), img.shape[1], img.shape[2], h, w, hp, wp,
             _lib.float3(mean), _lib.float3(std), _stream()), 'gln_transform')


@_op('gln_transform_batch(Tensor[] imgs, Tensor(a!) batch, int[] h, int[] w, float[] mean, float[] std) -> ()')
def _gln_transform_batch(imgs, batch, h, w, mean, std):
    n, hp, wp, _ = batch.shape
    if len(imgs) != n or len(h) != n or len(w) != n:
        raise RuntimeError('gln_transform_batch: one image and one resized size per batch slot')
    for t in imgs:
        if t.dtype != torch.float32 or not t.is_contiguous() or t.dim() != 3 or t.shape[0] != 3:
            raise RuntimeError('gln_transform_batch: images must be contiguous (3,H,W) float32 tensors')
    ci = lambda v: (ctypes.c_int * n)(*[int(x) for x in v])
    ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in imgs])
    fn = _by_dtype(batch, 'cvpce_gln_transform_batch', 'cvpce_gln_transform_batch_f16')
    check(fn(ptrs, ci([t.shape[1] for t in imgs]), ci([t.shape[2] for t in imgs]), ci(h), ci(w), n, _p(batch), hp, wp,
             _lib.float3(mean), _lib.float3(std), _stream()), 'gln_transform_batch')


@_op('crop_resize(Tensor img, Tensor boxes, Tensor? count, Tensor(a!) out, int size, int mode, float[]? mean, float[]? std) -> ()')
def _crop_resize(img, boxes, count, out, size, mode, mean, std):
    check(lib.cvpce_crop_resize(_p(img), _p(boxes), _p(count), boxes.shape[0], _p(out), img.shape[1], img.shape[2], size, mode,
                                _lib.float3(mean) if mean is not None else None, _lib.float3(std) if std is not None else None,
                                _stream()), 'crop_resize')


@_op('crop_resize_content(Tensor img, Tensor boxes, Tensor? count, Tensor(a!) out, int size, int mode, float[] mean, float[] std, Tensor ext) -> ()')
def _crop_resize_content(img, boxes, count, out, size, mode, mean, std, ext):
    if ext.dtype != torch.int32 or not ext.is_contiguous() or ext.shape != (boxes.shape[0], 2):
        raise RuntimeError('crop_resize_content: ext must be a contiguous (P,2) int32 tensor')
    check(lib.cvpce_crop_resize_content(_p(img), _p(boxes), _p(count), boxes.shape[0], _p(out), img.shape[1], img.shape[2], size, mode,
                                        _lib.float3(mean), _lib.float3(std), _p(ext), _stream()), 'crop_resize_content')


@_op('crop_extents(Tensor boxes, Tensor? count, int per_image, int h0, int w0, int size, Tensor(a!) ext) -> ()')
def _crop_extents(boxes, count, per_image, h0, w0, size, ext):
    if boxes.dtype != torch.float32 or not boxes.is_contiguous() or ext.dtype != torch.int32 or not ext.is_contiguous() or ext.numel() < 2 * boxes.shape[0] \
            or (per_image > 0 and (count is None or count.numel() * per_image < boxes.shape[0])):
        raise RuntimeError('crop_extents: boxes (P,4) float32, ext (P,2) int32, both contiguous; per_image > 0 needs one count per image')
    check(lib.cvpce_crop_extents(_p(boxes), _p(count), boxes.shape[0], per_image, h0, w0, size, _p(ext), _stream()), 'cvpce_crop_extents')


@_op('pad_extents(Tensor images, float pad, Tensor(a!) ext) -> ()')
def _pad_extents(images, pad, ext):
    if images.dtype != torch.float32 or not images.is_contiguous() or images.dim() != 4 or images.shape[1] != 3 or images.shape[2] != images.shape[3] \
            or ext.dtype != torch.int32 or not ext.is_contiguous() or ext.numel() < 2 * images.shape[0]:
        raise RuntimeError('pad_extents: images (B,3,S,S) float32 contiguous, ext (B,2) int32')
    check(lib.cvpce_pad_extents(_p(images), images.shape[0], images.shape[2], pad, _p(ext), _stream()), 'cvpce_pad_extents')


@_op('mac_init(Tensor(a!) desc, int off, Tensor row_suffix_max, Tensor col_suffix_max, Tensor computed) -> ()')
def _mac_init(desc, off, row_suffix_max, col_suffix_max, computed):
    """desc (N, D) f32; tables (H + 1, C) / (W + 1, C) f32; computed (N, 2) int32 = one layer's slice of embed_worklists' `computed`."""
    c = row_suffix_max.shape[1]
    if desc.dtype != torch.float32 or not desc.is_contiguous() or row_suffix_max.dtype != torch.float32 or col_suffix_max.dtype != torch.float32 \
            or not row_suffix_max.is_contiguous() or not col_suffix_max.is_contiguous() or col_suffix_max.shape[1] != c \
            or computed.dtype != torch.int32 or not computed.is_contiguous() or computed.numel() < 2 * desc.shape[0]:
        raise RuntimeError('mac_init: bad argument shapes / types')
    check(lib.cvpce_mac_init(_p(desc), desc.shape[0], desc.shape[1], off, c, _p(row_suffix_max), _p(col_suffix_max), row_suffix_max.shape[0] - 1,
                             col_suffix_max.shape[0] - 1, _p(computed), _stream()), 'cvpce_mac_init')


@_op('embed_worklists(Tensor? ext0, int n_images, int size, int pool_mask, int[] layers, Tensor(a!) lists, Tensor(b!) counts, '
     'Tensor(c!)? computed) -> ()')
def _embed_worklists(ext0, n_images, size, pool_mask, layers, lists, counts, computed):
    """layers: 9 ints per layer in the field order of `cvpce_skip_layer`; lists (2 * n_layers, stride) int64 (lists, then strip lists); counts (3 * n_layers,) int32 (tiles, row units, strip entries)."""
    nf = len(_lib.SkipLayer._fields_)
    nl = len(layers) // nf
    if len(layers) != nl * nf or lists.dtype != torch.int64 or not lists.is_contiguous() or lists.shape[0] != 2 * nl or counts.dtype != torch.int32 \
            or counts.numel() != 3 * nl or (ext0 is not None and (ext0.dtype != torch.int32 or not ext0.is_contiguous() or ext0.numel() < 2 * (n_images - 1))):
        raise RuntimeError('embed_worklists: bad argument shapes / types')
    arr = (_lib.SkipLayer * nl)(*[_lib.SkipLayer(*[int(v) for v in layers[i * nf:(i + 1) * nf]]) for i in range(nl)])
    if computed is not None and (computed.dtype != torch.int32 or not computed.is_contiguous() or computed.numel() < 2 * nl * n_images):
        raise RuntimeError('embed_worklists: computed must be a contiguous int32 tensor of (n_layers, n_images, 2)')
    check(lib.cvpce_embed_worklists(_p(ext0), n_images, size, pool_mask, ctypes.cast(arr, ctypes.c_void_p), nl, _p(lists), lists.shape[1], _p(counts),
                                    _p(computed), _stream()), 'cvpce_embed_worklists')


@_op('vgg_stem_fused_list(Tensor x, Tensor const_in, Tensor w1, Tensor b1, Tensor w2, Tensor b2, Tensor(a!) out, Tensor work, Tensor count) -> ()')
def _vgg_stem_fused_list(x, const_in, w1, b1, w2, b2, out, work, count):
    """x (N-1,H,W,c) + const_in (H,W,c) -> out (N,H/2,W/2,64): image N-1 of the pass is the constant crop."""
    n1, h, w, c = x.shape
    if x.dtype != torch.bfloat16 or const_in.dtype != torch.bfloat16 or tuple(const_in.shape[-3:]) != (h, w, c) or out.shape[0] != n1 + 1 \
            or work.dtype != torch.int64 or count.dtype != torch.int32:
        raise RuntimeError('cvpce_vgg_stem_fused_list: bad argument shapes / types')
    check(lib.cvpce_vgg_stem_fused_list(_p(x), c, _p(const_in), _p(w1), _p(b1), _p(w2), _p(b2), _p(out), n1 + 1, h, w, _p(work), _p(count),
                                        _stream()), 'cvpce_vgg_stem_fused_list')


def _halo_list(fn, what, x, weight, bias, out, mac, mac_off, cout, k_pad, cout_pad, relu, pool, work, count):
    n, h, w, cin = x.shape
    if x.dtype != torch.bfloat16 or weight.dtype != torch.bfloat16 or work.dtype != torch.int64 or count.dtype != torch.int32:
        raise RuntimeError(f'{what}: bf16 activations, int64 work list, int32 count')
    check(fn(_p(x), _p(weight), _p(bias), _p(out), _p(mac), mac.shape[1] if mac is not None else 0, mac_off, n, h, w,
             cin, cout, k_pad, cout_pad, relu, pool, _p(work), _p(count), _stream()), what)


@_op('conv3x3_halo_list(Tensor x, Tensor weight, Tensor? bias, Tensor(a!)? out, Tensor(b!)? mac, int mac_off, int cout, int k_pad, '
     'int cout_pad, int relu, int pool, Tensor work, Tensor count) -> ()')
def _conv3x3_halo_list(x, weight, bias, out, mac, mac_off, cout, k_pad, cout_pad, relu, pool, work, count):
    _halo_list(lib.cvpce_conv3x3_halo_list, 'cvpce_conv3x3_halo_list', x, weight, bias, out, mac, mac_off, cout, k_pad, cout_pad, relu, pool, work, count)


@_op('conv3x3_halo_strips(Tensor x, Tensor weight, Tensor? bias, Tensor(a!)? out, Tensor(b!)? mac, int mac_off, int cout, int k_pad, '
     'int cout_pad, int relu, int pool, Tensor work, Tensor count) -> ()')
def _conv3x3_halo_strips(x, weight, bias, out, mac, mac_off, cout, k_pad, cout_pad, relu, pool, work, count):
    _halo_list(lib.cvpce_conv3x3_halo_strips, 'cvpce_conv3x3_halo_strips', x, weight, bias, out, mac, mac_off, cout, k_pad, cout_pad, relu, pool, work, count)


@_op('pack_embed_input(Tensor images, Tensor(a!) out, int to_tanh, float[] mean, float[] std) -> ()')
def _pack_embed_input(images, out, to_tanh, mean, std):
    check(lib.cvpce_pack_embed_input(_p(images), _p(out), images.shape[0], images.shape[2], to_tanh, _lib.float3(mean), _lib.float3(std),
                                     _stream()), 'pack_embed_input')


@_op('detect_postprocess(Tensor[] logits, Tensor[] regs, int[] gh, int[] gw, int[] sh, int[] sw, Tensor base_anchors, Tensor image_hw, '
     'Tensor ratios, int num_anchors, int num_classes, int topk, float score_thresh, float nms_thresh, float xform_clip, '
     'int detections_per_img, float conf_thresh, Tensor(a!) workspace, Tensor(b!) boxes, Tensor(c!) scores, Tensor(d!) labels, '
     'Tensor(e!) count, Tensor(f!) conf) -> ()')
def _detect_postprocess(logits, regs, gh, gw, sh, sw, base_anchors, image_hw, ratios, num_anchors, num_classes, topk, score_thresh,
                        nms_thresh, xform_clip, detections_per_img, conf_thresh, workspace, boxes, scores, labels, count, conf):
    L, n = len(logits), logits[0].shape[0]
    lp = (ctypes.c_void_p * L)(*[t.data_ptr() for t in logits])
    rp = (ctypes.c_void_p * L)(*[t.data_ptr() for t in regs])
    ci = lambda v: (ctypes.c_int * L)(*[int(x) for x in v])
    check(lib.cvpce_detect_postprocess(lp, rp, ci(gh), ci(gw), ci(sh), ci(sw), _p(base_anchors), _p(image_hw), _p(ratios), L, n,
                                       num_anchors, num_classes, topk, score_thresh, nms_thresh, xform_clip, detections_per_img,
                                       conf_thresh, _p(workspace), workspace.numel(), _p(boxes), _p(scores), _p(labels), _p(count),
                                       _p(conf), _stream()), 'cvpce_detect_postprocess')


def _atlas_copy(levels, atlas, oy, ox, to_atlas):
    L = len(levels)
    n, hc, wc, c = atlas.shape
    lp = (ctypes.c_void_p * L)(*[t.data_ptr() for t in levels])
    ci = lambda v: (ctypes.c_int * L)(*[int(x) for x in v])
    for t in levels:
        if t.dtype != atlas.dtype or t.shape[0] != n or t.shape[3] != c or not t.is_contiguous():
            raise RuntimeError('atlas_copy: levels must be contiguous (N,h,w,C) tensors of the atlas dtype / batch / channels')
    check(lib.cvpce_atlas_copy(lp, ci([t.shape[1] for t in levels]), ci([t.shape[2] for t in levels]), ci(oy), ci(ox), L, n, _p(atlas),
                               hc, wc, c * atlas.element_size(), int(to_atlas), _stream()), 'cvpce_atlas_copy')


@_op('atlas_pack(Tensor[] levels, Tensor(a!) atlas, int[] oy, int[] ox) -> ()')
def _atlas_pack(levels, atlas, oy, ox):
    _atlas_copy(levels, atlas, oy, ox, True)


@_op('atlas_unpack(Tensor atlas, Tensor(a!)[] levels, int[] oy, int[] ox) -> ()')
def _atlas_unpack(atlas, levels, oy, ox):
    _atlas_copy(levels, atlas, oy, ox, False)


@_op('row_norms(Tensor x, Tensor(a!) out, float eps) -> ()')
def _row_norms(x, out, eps):
    check(lib.cvpce_row_norms(_p(x), _p(out), x.shape[0], x.shape[1], int(x.dtype == torch.float32), eps, _stream()), 'row_norms')


@_op('match_topk(Tensor queries, Tensor gallery, Tensor q_norms, Tensor g_norms, int k, Tensor(a!) workspace, Tensor(b!) idx, '
     'Tensor(c!)? dist) -> ()')
def _match_topk(queries, gallery, q_norms, g_norms, k, workspace, idx, dist):
    check(lib.cvpce_match_topk(_p(queries), _p(gallery), _p(q_norms), _p(g_norms), queries.shape[0], gallery.shape[0], queries.shape[1], k,
                               int(queries.dtype == torch.float32), _p(workspace), workspace.numel(), _p(idx), _p(dist), _stream()),
          'cvpce_match_topk')


@_op('match_topk_state(Tensor queries, Tensor gallery, Tensor q_norms, Tensor g_norms, int k, Tensor(a!) workspace, Tensor(b!) state, '
     'Tensor(c!) idx, Tensor(d!)? dist) -> ()')
def _match_topk_state(queries, gallery, q_norms, g_norms, k, workspace, state, idx, dist):
    check(lib.cvpce_match_topk_state(_p(queries), _p(gallery), _p(q_norms), _p(g_norms), queries.shape[0], gallery.shape[0], queries.shape[1], k,
                                     int(queries.dtype == torch.float32), _p(workspace), workspace.numel(), _p(state), state.numel() * state.element_size(),
                                     _p(idx), _p(dist), _stream()),
          'cvpce_match_topk_state')


@_op('match_state_init(Tensor(a!) state) -> ()')
def _match_state_init(state):
    check(lib.cvpce_match_state_init(_p(state), state.numel() * state.element_size(), _stream()), 'cvpce_match_state_init')


@_op('probe_mfma_bf16(int shape, int iters, Tensor operands, Tensor(a!) sink, int workgroups) -> ()')
def _probe_mfma_bf16(shape, iters, operands, sink, workgroups):
    check(lib.cvpce_probe_mfma_bf16(shape, iters, _p(operands), _p(sink), workgroups, _stream()), 'probe')


@_op('probe_l2_stream(Tensor buf, int iters, Tensor(a!) sink, int workgroups) -> ()')
def _probe_l2_stream(buf, iters, sink, workgroups):
    check(lib.cvpce_probe_l2_stream(_p(buf), buf.numel() * buf.element_size(), iters, _p(sink), workgroups, _stream()), 'probe_l2_stream')


T = torch.ops.cvpce_amd
