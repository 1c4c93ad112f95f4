"""ctypes binding of libvkas.so (the C ABI declared in include/vkas.h).

The library is built in-tree (``vkit_ocr_model_adaptive_scaling_amd/libvkas.so``) by
``__graft_entry__.build()`` / ``csrc/Makefile``.  There is no fallback: if the library is missing
or a symbol cannot be resolved, importing this module raises.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int64, c_long, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# VKAS_LIB_PATH: profiling aid only (timing-only ablation builds from profiles/build_*_variants.sh)
LIB_PATH = os.environ.get('VKAS_LIB_PATH') or os.path.join(_HERE, 'libvkas.so')

F32, BF16, F16 = 0, 1, 2
EPI_NONE, EPI_GELU, EPI_SCALE_RES, EPI_DGELU, EPI_ADD, EPI_PATCH, EPI_HEAD = range(7)
LOSS_FOCAL, LOSS_DICE, LOSS_L1, LOSS_SMOOTH_L1, LOSS_L2 = range(5)


class ConvGeom(Structure):
    _fields_ = [('B', c_int), ('Hin', c_int), ('Win', c_int), ('Hout', c_int), ('Wout', c_int), ('Cp', c_int),
                ('ldx', c_int), ('KH', c_int), ('KW', c_int), ('stride', c_int), ('pad', c_int)]


class HeadDesc(Structure):
    _fields_ = [('n_heads', c_int), ('pw', c_int), ('n0', c_int * 4), ('np', c_int * 4), ('c', c_int * 4),
                ('oc', c_int * 4), ('params', c_void_p), ('stats', c_void_p), ('proj', c_void_p)]


class Epilogue(Structure):
    _fields_ = [('mode', c_int), ('bias', c_void_p), ('out', c_void_p), ('ldo', c_long), ('out2', c_void_p),
                ('ldo2', c_long), ('aux', c_void_p), ('ldaux', c_long), ('colscale', c_void_p),
                ('rowscale', c_void_p), ('rows_per_image', c_int), ('patch', c_int), ('patch_Hs', c_int),
                ('patch_Ws', c_int), ('patch_Cp', c_int), ('head', HeadDesc)]


class PackDesc(Structure):
    _fields_ = [('w', c_void_p), ('out', c_void_p), ('total', c_long), ('kind', c_int), ('N', c_int), ('C', c_int),
                ('KH', c_int), ('KW', c_int), ('Np', c_int), ('Cp', c_int), ('mode', c_int), ('n_off', c_int),
                ('Nt', c_int), ('dtype', c_int)]


class RoughLossCfg(Structure):
    _fields_ = [('focal_factor', c_float), ('dice_factor', c_float), ('l1_factor', c_float), ('score_min', c_float),
                ('height_min', c_float), ('focal_alpha', c_float), ('focal_gamma', c_float), ('out_scale', c_float)]


class PreciseLossCfg(Structure):
    _fields_ = [('pos_l2', c_float), ('neg_l2', c_float), ('offset_l1', c_float), ('reg_l1', c_float),
                ('angle_ce', c_float), ('dist_l1', c_float), ('loss_factor', c_float), ('smooth_beta', c_float),
                ('out_scale', c_float)]


class VkasError(RuntimeError):
    pass


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f'{LIB_PATH} not found: build the HIP kernels first (python -c "import __graft_entry__ as g; g.build()" '
        f'or make -C {os.path.join(_HERE, "csrc")}).  There is no CPU fallback.')

lib = ctypes.CDLL(LIB_PATH)

_P = c_void_p
_SIGS = {
    'vkas_last_error': (c_char_p, []),
    'vkas_abi_version': (c_int, []),
    'vkas_pack_conv_weight': (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    'vkas_pack_conv_weight_slice': (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    'vkas_unpack_conv_wgrad': (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    'vkas_accumulate_many': (c_int, [c_int, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int), _P]),
    'vkas_finalize_many': (c_int, [c_int, POINTER(c_void_p), POINTER(c_long), POINTER(c_int), POINTER(c_int), POINTER(c_void_p),
                                   POINTER(c_int), _P]),
    'vkas_layernorm_bwd_parts': (c_long, [c_long, c_int]),
    'vkas_scale_res_bwd_parts': (c_long, [c_long, c_int]),
    'vkas_dwconv7x7_wgrad_parts': (c_long, [c_int, c_int, c_int, c_int, c_int]),
    'vkas_pad_vector': (c_int, [_P, _P, c_int, c_int, _P]),
    'vkas_dw_weight_elems': (c_size_t, [c_int]),
    'vkas_pack_dw_weight': (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    'vkas_unpack_dw_wgrad': (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    'vkas_image_nchw_to_nhwc8': (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    'vkas_nhwc_to_nchw_f32': (c_int, [_P, c_long, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    'vkas_nchw_f32_to_nhwc': (c_int, [_P, _P, c_long, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    'vkas_conv_gemm_fwd': (c_int, [_P, POINTER(ConvGeom), _P, c_int, POINTER(Epilogue), c_int, _P]),
    'vkas_conv_gemm_wgrad': (c_int, [_P, POINTER(ConvGeom), _P, c_long, c_int, _P, _P, c_int, _P]),
    'vkas_conv_gemm_wgrad_ordered': (c_int, [_P, POINTER(ConvGeom), _P, c_long, c_int, _P, c_int, _P]),
    'vkas_conv_gemm_wgrad_gelu': (c_int, [_P, POINTER(ConvGeom), _P, c_long, c_int, _P, _P, c_int, _P]),
    'vkas_mlp_chain_image_elems': (c_size_t, [c_int]),
    'vkas_mlp_chain_pack': (c_int, [_P, _P, _P, c_int, c_int, _P, c_int, _P]),
    'vkas_mlp_chain_fwd': (c_int, [_P, c_long, _P, _P, _P, c_long, _P, _P, c_int, _P, c_long, _P, c_long, _P, c_long,
                                   c_long, c_int, c_int, _P]),
    'vkas_mlp_chain_ln_fwd': (c_int, [_P, c_long, _P, _P, _P, c_long, _P, _P, _P, _P, c_long, _P, _P, c_int, _P, c_long, _P,
                                      c_long, _P, c_long, c_long, c_int, c_int, _P]),
    'vkas_mlp_chain_bwd': (c_int, [_P, c_long, _P, _P, c_long, _P, c_long, _P, c_long, c_long, c_int, c_int, _P]),
    'vkas_conv_gemm_tile': (c_int, [c_int, c_long, c_int, c_int]),
    'vkas_conv_gemm_kernel_id': (c_int, [c_int, POINTER(ConvGeom), c_int, c_long, c_int]),
    'vkas_colsum': (c_int, [_P, c_long, c_long, c_int, _P, c_int, _P, c_size_t, c_int, _P]),
    'vkas_colsum_ws_bytes': (c_size_t, [c_long, c_int]),
    'vkas_dwconv7x7_fwd': (c_int, [_P, c_long, _P, _P, _P, c_long, _P, c_long, c_int, c_int, c_int, c_int, c_int, _P]),
    'vkas_dwconv7x7_wgrad': (c_int, [_P, c_long, _P, c_long, _P, _P, _P, c_size_t, c_int, c_int, c_int, c_int, c_int, _P]),
    'vkas_dwconv7x7_wgrad_ws_bytes': (c_size_t, [c_int, c_int, c_int, c_int]),
    'vkas_layernorm_fwd': (c_int, [_P, c_long, _P, _P, _P, c_long, _P, c_long, c_int, c_int, c_int, c_int, _P]),
    'vkas_layernorm_bwd': (c_int, [_P, c_long, _P, _P, _P, _P, c_long, _P, c_long, _P, _P, _P, c_size_t, c_long, c_int,
                                   c_int, c_int, c_int, _P]),
    'vkas_layernorm_bwd_ws_bytes': (c_size_t, [c_long, c_int]),
    'vkas_pack_head_params': (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P, _P]),
    'vkas_head_tail_fwd': (c_int, [_P, c_long, _P, c_long, c_int, _P]),
    'vkas_head_tail_bwd': (c_int, [_P, c_long, POINTER(HeadDesc), POINTER(c_void_p), _P, c_long, _P, _P, c_size_t, c_long, c_int, _P]),
    'vkas_head_tail_bwd_ws_bytes': (c_size_t, [c_long, c_int]),
    'vkas_scale_res_bwd': (c_int, [_P, c_long, _P, c_long, _P, _P, c_int, _P, c_long, _P, _P, _P, c_size_t, c_long,
                                   c_int, c_int, _P]),
    'vkas_scale_res_bwd_ws_bytes': (c_size_t, [c_long, c_int]),
    'vkas_resize_fwd': (c_int, [_P, c_long, _P, c_long, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    'vkas_resize_bwd': (c_int, [_P, c_long, _P, c_long, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    'vkas_resize_bwd_ws_bytes': (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    'vkas_resize_bwd_ws': (c_int, [_P, c_long, _P, c_long, _P, c_size_t, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                   c_int, _P]),
    'vkas_adaptive_avgpool_fwd': (c_int, [_P, c_long, _P, c_long, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    'vkas_adaptive_avgpool_bwd': (c_int, [_P, c_long, _P, c_long, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    'vkas_copy_channels': (c_int, [_P, c_long, _P, c_long, c_long, c_int, c_int, c_int, _P]),
    'vkas_copy_channel_range': (c_int, [_P, c_long, c_int, _P, c_long, c_int, c_long, c_int, c_int, c_int, _P]),
    'vkas_softplus_fwd': (c_int, [_P, _P, c_long, _P]),
    'vkas_softplus_bwd': (c_int, [_P, _P, _P, c_long, _P]),
    'vkas_rough_loss_fwd': (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                    POINTER(RoughLossCfg), _P, _P, _P]),
    'vkas_rough_loss_bwd': (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                    POINTER(RoughLossCfg), _P, _P, _P, _P, _P]),
    'vkas_points_margin': (c_int, [_P, _P, c_long, c_int, c_int, _P, _P]),
    'vkas_precise_loss_fwd': (c_int, [_P] * 11 + [c_int] * 8 + [POINTER(PreciseLossCfg), _P, _P, _P]),
    'vkas_precise_loss_bwd': (c_int, [_P] * 11 + [c_int] * 8 + [POINTER(PreciseLossCfg), _P, _P, _P, _P, _P, _P, _P]),
    'vkas_elementwise_loss_fwd': (c_int, [c_int, _P, _P, _P, c_long, c_float, c_float, c_float, _P, _P, _P]),
    'vkas_elementwise_loss_bwd': (c_int, [c_int, _P, _P, _P, c_long, c_float, c_float, c_float, _P, _P, _P, _P]),
    'vkas_cross_entropy_fwd': (c_int, [_P, _P, c_int, c_long, c_int, _P, _P, _P]),
    'vkas_cross_entropy_bwd': (c_int, [_P, _P, c_int, c_long, c_int, _P, _P, _P]),
    'vkas_points_prepare': (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P, c_long, _P]),
    'vkas_points_gather_rows': (c_int, [_P, c_long, c_int, c_int, _P, _P, c_int, c_long, _P, c_long, _P, _P, _P, c_int, _P]),
    'vkas_points_gather_patches': (c_int, [_P, c_long, c_int, c_int, c_int, c_int, _P, c_long, _P, c_int, _P]),
    'vkas_points_scatter3x3': (c_int, [_P, _P, _P, c_long, c_int, c_int, c_int, c_int, _P, c_long, c_int, _P]),
    'vkas_points_scatter_vec8': (c_int, [_P, _P, c_long, _P, _P]),
    'vkas_points_gather_vec8': (c_int, [_P, _P, c_long, _P, _P]),
    'vkas_pack_many': (c_int, [_P, _P, c_int, c_int, _P]),
    'vkas_pack_many_blocks': (c_int, [_P]),
    'vkas_rough_postprocess': (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, c_float, c_float, _P, _P, _P]),
    'vkas_precise_postprocess': (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P]),
    'vkas_l2norm_sq': (c_int, [_P, c_long, _P, _P]),
    'vkas_adamw_step': (c_int, [_P, _P, _P, _P, c_long, _P, c_float, c_float, c_float, c_float, c_float, c_float,
                                c_float, c_int, _P]),
}

EXPORTS = tuple(_SIGS)

for _name, (_res, _args) in _SIGS.items():
    _fn = getattr(lib, _name)  # AttributeError here = the library does not export a declared symbol
    _fn.restype = _res
    _fn.argtypes = _args


def check(rc: int, what: str = ''):
    if rc != 0:
        msg = lib.vkas_last_error()
        raise VkasError(f'{what}: vkas error {rc}: {msg.decode() if msg else "?"}')
