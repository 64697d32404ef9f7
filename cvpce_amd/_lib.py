"""ctypes binding of the C ABI declared in include/cvpce_amd.h.

The HIP library is the product: there is NO CPU fallback.  Importing this
module without a built `libcvpce_hip.so` raises immediately (build it with
`python -c "import __graft_entry__ as g; g.build()"` or `make -C cvpce_amd/csrc`).
"""
import ctypes
import os
from ctypes import c_int, c_float, c_void_p, c_size_t, c_longlong, POINTER

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('CVPCE_LIB') or os.path.join(_HERE, 'libcvpce_hip.so')   # CVPCE_LIB: dev-only A/B builds


class HipLibraryMissing(ImportError):
    pass


if not os.path.exists(LIB_PATH):
    raise HipLibraryMissing(
        f'{LIB_PATH} not found: the cvpce_amd hot path has no CPU fallback. '
        'Build the HIP library first (python -c "import __graft_entry__ as g; g.build()").')

lib = ctypes.CDLL(LIB_PATH)

_vp = c_void_p
_fp = c_void_p   # float* passed as raw device address
_ip = c_void_p

SIGNATURES = {
    'cvpce_conv2d_nhwc_bf16': (c_int, [_vp, _vp, _fp, _vp, _vp] + [c_int] * 21 + [_vp]),
    'cvpce_conv2d_splitk_bf16': (c_int, [_vp, _vp, _fp, _vp, _vp] + [c_int] * 20 + [_vp, c_size_t, _vp]),
    'cvpce_conv2d_splitk_workspace_bytes': (c_size_t, [c_longlong, c_int, c_int]),
    'cvpce_set_persistent_workgroups': (c_int, [c_int]),
    'cvpce_conv1x1_nhwc_bf16': (c_int, [_vp, _vp, _fp, _vp, _vp] + [c_int] * 14 + [_vp]),
    'cvpce_vgg_stem_fused': (c_int, [_vp, c_int, _vp, _fp, _vp, _fp, _vp, c_int, c_int, c_int, _vp]),
    'cvpce_gln_stem_fused': (c_int, [_vp, _vp, _fp, _vp, c_int, c_int, c_int, _vp]),
    'cvpce_atlas_copy': (c_int, [POINTER(_vp), POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int), c_int, c_int, _vp, c_int, c_int, c_int, c_int, _vp]),
    'cvpce_conv3x3_halo': (c_int, [_vp, _vp, _fp, _vp] + [c_int] * 9 + [_vp]),
    'cvpce_conv3x3_halo_mac': (c_int, [_vp, _vp, _fp, _vp, _fp, c_int, c_int] + [c_int] * 8 + [_vp]),
    'cvpce_conv3x3_halo_wide': (c_int, [_vp, _vp, _fp, _vp] + [c_int] * 9 + [_vp]),
    'cvpce_bottleneck_fused': (c_int, [_vp, _vp, _vp, _fp, _vp, _fp, _vp, _fp, _vp] + [c_int] * 11 + [_vp]),
    'cvpce_conv3x3_halo_masked': (c_int, [_vp, _vp, _fp, _vp, _vp, c_int, _vp] + [c_int] * 8 + [_vp]),
    'cvpce_conv3x3_halo_masked_paired': (c_int, [_vp, _vp, _fp, _vp, _vp, c_int, _vp] + [c_int] * 9 + [_vp]),
    'cvpce_maxpool2d_nhwc_bf16': (c_int, [_vp, _vp] + [c_int] * 9 + [_vp]),
    'cvpce_relu_bf16': (c_int, [_vp, _vp, c_longlong, _vp]),
    'cvpce_pack_halo_weights': (c_int, [_vp, _vp, c_int, c_int]),
    'cvpce_conv3x3_thin_bf16': (c_int, [_vp, _vp, _fp, _vp] + [c_int] * 9 + [_vp]),
    'cvpce_gauss_tail_bf16': (c_int, [_vp, _vp, _fp, _vp, _fp, _fp, c_longlong, c_int, c_int, _vp]),
    'cvpce_gauss_subnet_bf16': (c_int, [_vp, _vp, _fp, _vp, _fp, _vp, _fp, _vp, _fp, c_int, _vp, _fp, c_int, _fp, c_int, c_int, c_int, c_int, _vp]),
    'cvpce_global_max_nhwc_bf16': (c_int, [_vp, _fp, c_int, c_int, c_int, c_int, c_int, _vp]),
    'cvpce_l2_normalize_f32': (c_int, [_fp, _fp, _vp, c_int, c_int, c_float, _vp]),
    'cvpce_gln_transform': (c_int, [_fp, _vp] + [c_int] * 6 + [POINTER(c_float), POINTER(c_float), _vp]),
    'cvpce_gln_transform_batch': (c_int, [POINTER(c_void_p), POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int), c_int, _vp, c_int, c_int,
                                          POINTER(c_float), POINTER(c_float), _vp]),
    'cvpce_crop_resize': (c_int, [_fp, _fp, _ip, c_int, _vp, c_int, c_int, c_int, c_int,
                                  POINTER(c_float), POINTER(c_float), _vp]),
    'cvpce_crop_resize_content': (c_int, [_fp, _fp, _ip, c_int, _vp, c_int, c_int, c_int, c_int,
                                          POINTER(c_float), POINTER(c_float), _ip, _vp]),
    'cvpce_pack_embed_input': (c_int, [_fp, _vp, c_int, c_int, c_int, POINTER(c_float), POINTER(c_float), _vp]),
    'cvpce_crop_extents': (c_int, [_fp, _ip, c_int, c_int, c_int, c_int, c_int, _ip, _vp]),
    'cvpce_pad_extents': (c_int, [_fp, c_int, c_int, c_float, _ip, _vp]),
    'cvpce_embed_worklists': (c_int, [_ip, c_int, c_int, ctypes.c_uint, _vp, c_int, _vp, c_longlong, _ip, _ip, _vp]),
    'cvpce_mac_init': (c_int, [_fp, c_int, c_int, c_int, c_int, _fp, _fp, c_int, c_int, _ip, _vp]),
    'cvpce_vgg_stem_fused_list': (c_int, [_vp, c_int, _vp, _vp, _fp, _vp, _fp, _vp, c_int, c_int, c_int, _vp, _ip, _vp]),
    'cvpce_bottleneck_fused_fm': (c_int, [_vp, _vp, _vp, _fp, _vp, _fp, _vp, _fp, _vp] + [c_int] * 5 + [_vp]),
    'cvpce_conv3x3_halo_thin_out': (c_int, [_vp, _vp, _fp, _fp] + [c_int] * 7 + [_vp]),
    'cvpce_conv3x3_halo_list': (c_int, [_vp, _vp, _fp, _vp, _fp, c_int, c_int] + [c_int] * 9 + [_vp, _ip, _vp]),
    'cvpce_conv3x3_halo_strips': (c_int, [_vp, _vp, _fp, _vp, _fp, c_int, c_int] + [c_int] * 9 + [_vp, _ip, _vp]),
    'cvpce_detect_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'cvpce_detect_postprocess': (c_int, [POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int), POINTER(c_int),
                                         POINTER(c_int), POINTER(c_int), _fp, _ip, _fp,
                                         c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_int, c_float,
                                         _vp, c_size_t, _fp, _fp, _vp, _ip, _ip, _vp]),
    'cvpce_row_norms': (c_int, [_vp, _fp, c_int, c_int, c_int, c_float, _vp]),
    'cvpce_match_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'cvpce_match_set_core': (c_int, [c_int, c_int, c_int, c_int]),
    'cvpce_match_state_bytes': (c_size_t, [c_int]),
    'cvpce_match_state_init': (c_int, [_vp, c_size_t, _vp]),
    'cvpce_match_topk_state': (c_int, [_vp, _vp, _fp, _fp, c_int, c_int, c_int, c_int, c_int, _vp, c_size_t, _vp, c_size_t, _vp, _fp, _vp]),
    'cvpce_probe_l2_stream': (c_int, [_vp, c_longlong, c_int, _fp, c_int, _vp]),
    'cvpce_probe_mfma_bf16': (c_int, [c_int, c_int, _vp, _fp, c_int, _vp]),
    'cvpce_match_topk': (c_int, [_vp, _vp, _fp, _fp, c_int, c_int, c_int, c_int, c_int, _vp, c_size_t, _vp, _fp, _vp]),
}
# fp16 twins of the detector's kernels (the opt-in accuracy mode): same argument lists as the functions they are named after
for _base, _twin in (('cvpce_conv2d_nhwc_bf16', 'cvpce_conv2d_nhwc_f16'), ('cvpce_conv2d_splitk_bf16', 'cvpce_conv2d_splitk_f16'), ('cvpce_conv1x1_nhwc_bf16', 'cvpce_conv1x1_nhwc_f16'),
                     ('cvpce_gln_stem_fused', 'cvpce_gln_stem_fused_f16'), ('cvpce_conv3x3_halo', 'cvpce_conv3x3_halo_f16'),
                     ('cvpce_conv3x3_halo_wide', 'cvpce_conv3x3_halo_wide_f16'), ('cvpce_conv3x3_halo_thin_out', 'cvpce_conv3x3_halo_thin_out_f16'), ('cvpce_conv3x3_halo_masked', 'cvpce_conv3x3_halo_masked_f16'), ('cvpce_conv3x3_halo_masked_paired', 'cvpce_conv3x3_halo_masked_paired_f16'),
                     ('cvpce_bottleneck_fused', 'cvpce_bottleneck_fused_f16'), ('cvpce_bottleneck_fused_fm', 'cvpce_bottleneck_fused_fm_f16'),
                     ('cvpce_maxpool2d_nhwc_bf16', 'cvpce_maxpool2d_nhwc_f16'), ('cvpce_relu_bf16', 'cvpce_relu_f16'), ('cvpce_gauss_tail_bf16', 'cvpce_gauss_tail_f16'), ('cvpce_gauss_subnet_bf16', 'cvpce_gauss_subnet_f16'), ('cvpce_conv3x3_thin_bf16', 'cvpce_conv3x3_thin_f16'),
                     ('cvpce_gln_transform', 'cvpce_gln_transform_f16'), ('cvpce_gln_transform_batch', 'cvpce_gln_transform_batch_f16')):
    SIGNATURES[_twin] = SIGNATURES[_base]

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)   # AttributeError here = header/library mismatch: fail loudly
    _fn.restype = _res
    _fn.argtypes = _args

_ERR = {1: 'bad argument', 2: 'kernel launch failure'}


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f'{what} failed: {_ERR.get(rc, rc)}')


class SkipLayer(ctypes.Structure):
    """`cvpce_skip_layer` of include/cvpce_amd.h."""
    _fields_ = [(k, c_int) for k in ('H', 'W', 'tile_h', 'tile_w', 'out_ops', 'in_H', 'in_W', 'in_ops', 'skip')]


def float3(vals):
    return (c_float * 3)(*[float(v) for v in vals])
