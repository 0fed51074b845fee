"""ctypes binding of libseg2eye_hip.so (include/seg2eye_hip.h).

The library is the product path: if it is missing this module raises on import
of any op -- there is NO CPU or stock-torch fallback for the ops it exports.
"""
import ctypes as C
import os

import torch  # noqa: F401  -- MUST load before the library: both link libamdhip64 and the process must
#                       end up with torch's HIP runtime, or launches from the .so see "no ROCm-capable device"

_HERE = os.path.dirname(os.path.abspath(__file__))
# S2E_LIB_PATH: an alternative build of the SAME library (same-box A/B runs of experiment builds); never a fallback
LIB_PATH = os.environ.get('S2E_LIB_PATH') or os.path.join(_HERE, 'lib', 'libseg2eye_hip.so')

S2E_F32, S2E_BF16 = 0, 1
UNI_REPLICAS = 16           # S2E_UNI_REPLICAS of include/seg2eye_hip.h
ACT_NONE, ACT_LRELU, ACT_TANH = 0, 1, 2
AUX_NONE, AUX_RELU_MASK, AUX_LRELU_GRAD = 0, 1, 2
NORM_SPADE_STYLE, NORM_PLAIN_IN, NORM_SPADE_STYLE_BATCH = 0, 1, 2
NORM_ACCUMULATE_DX = 0x100
LOSS_NEG_MEAN, LOSS_HINGE_REAL, LOSS_HINGE_FAKE, LOSS_L1 = 0, 1, 2, 3


class ConvDesc(C.Structure):
    """s2e_conv_desc"""
    _fields_ = [(n, C.c_int) for n in (
        'N', 'Hi', 'Wi', 'Cin', 'Ho', 'Wo', 'Cout', 'KH', 'KW', 'stride', 'pad',
        'transposed', 'in_act', 'out_act', 'aux_mode')]


class WgradC8Job(C.Structure):
    _fields_ = [('x', C.c_void_p), ('gy', C.c_void_p), ('dw_oihw', C.c_void_p), ('dbias', C.c_void_p),
                ('H', C.c_int), ('W', C.c_int), ('ncls', C.c_int), ('rect_list', C.c_void_p), ('rect_count', C.c_void_p)]


class WgradBatchJob(C.Structure):
    """s2e_wgrad_batch_job"""
    _fields_ = [('x', C.c_void_p), ('gy', C.c_void_p), ('dw', C.c_void_p), ('dbias', C.c_void_p),
                ('rect_list', C.c_void_p), ('rect_count', C.c_void_p),
                ('N', C.c_int), ('H', C.c_int), ('W', C.c_int), ('Cin', C.c_int), ('Cout', C.c_int), ('flags', C.c_int)]


class WgradMultiJob(C.Structure):
    """s2e_wgrad_multi_job"""
    _fields_ = [('x', C.c_void_p), ('gy', C.c_void_p), ('dw', C.c_void_p), ('dbias', C.c_void_p), ('d', ConvDesc)]


class LabelConvJob(C.Structure):
    _fields_ = [('weight', C.c_void_p), ('bias', C.c_void_p), ('out_off', C.c_long), ('h', C.c_int), ('w', C.c_int),
                ('cout', C.c_int), ('relu', C.c_int)]


class ClassTableJob(C.Structure):
    _fields_ = [('w_sh', C.c_void_p), ('b_sh', C.c_void_p), ('w_packed', C.c_void_p), ('bias', C.c_void_p), ('table_off', C.c_long),
                ('nh', C.c_int), ('C', C.c_int)]


class SnLayer(C.Structure):
    """s2e_sn_layer"""
    _fields_ = [('w', C.c_void_p), ('u', C.c_void_p), ('v', C.c_void_p), ('t', C.c_void_p), ('s', C.c_void_p),
                ('rows', C.c_int), ('cols', C.c_int), ('t2', C.c_void_p), ('s2', C.c_void_p), ('cin', C.c_int), ('taps', C.c_int)]


class PackJob(C.Structure):
    """s2e_pack_job"""
    _fields_ = [('w', C.c_void_p), ('out', C.c_void_p), ('sigma_index', C.c_int), ('cout', C.c_int), ('cin', C.c_int),
                ('taps', C.c_int), ('cin_pad', C.c_int), ('transposed', C.c_int), ('out_fwd', C.c_void_p)]


class GradJob(C.Structure):
    """s2e_grad_job"""
    _fields_ = [('gw_packed', C.c_void_p), ('out', C.c_void_p), ('w_orig', C.c_void_p), ('u', C.c_void_p), ('v', C.c_void_p),
                ('sigma', C.c_void_p), ('cout', C.c_int), ('cin', C.c_int), ('taps', C.c_int), ('cin_pad', C.c_int),
                ('dot_index', C.c_int), ('reserved', C.c_int)]


class SpadeUniJob(C.Structure):
    """s2e_spade_uni_job"""
    _fields_ = [('R', C.c_void_p), ('A', C.c_void_p), ('w_gb', C.c_void_p), ('w_sc', C.c_long), ('w_sk', C.c_long), ('w_st', C.c_long),
                ('w_sh', C.c_void_p), ('b_sh', C.c_void_p), ('dw_sh', C.c_void_p), ('db_sh', C.c_void_p), ('dw_gb', C.c_void_p),
                ('db_gb', C.c_void_p), ('C2', C.c_int), ('nh', C.c_int), ('ncls', C.c_int), ('act_bf16', C.c_int)]


class SnGradJob(C.Structure):
    """s2e_sngrad_job"""
    _fields_ = [('g', C.c_void_p), ('w', C.c_void_p), ('u', C.c_void_p), ('v', C.c_void_p), ('sigma', C.c_void_p),
                ('rows', C.c_int), ('cin', C.c_int), ('taps', C.c_int), ('part0', C.c_int), ('nparts', C.c_int), ('vmem0', C.c_int)]


_vp, _i, _l, _f = C.c_void_p, C.c_int, C.c_long, C.c_float
# name -> argtypes; every entry returns int except the two noted below.  Must list EVERY symbol
# declared in include/seg2eye_hip.h (tests/test_abi.py checks header <-> table <-> .so).
SIGNATURES = {
    's2e_version': [],
    's2e_last_error': [],
    's2e_conv_cout_pad': [_i],
    's2e_conv_k_pad': [_i, _i],
    's2e_pack_conv_weight': [_i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    's2e_sn_block_shape': [_i, _vp, _vp],
    's2e_sn_chain_max_cols': [],
    's2e_sn_power_iteration': [_vp, _i, _vp, _i, _vp, _i, _vp, C.c_size_t, _vp, _i, _i, _f, _i, _vp],
    's2e_sn_weight_grad': [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    's2e_grad_block_map': [_vp, _i, _vp],
    's2e_sngrad_block_map': [_vp, _i, _vp],
    's2e_sngrad_scratch_floats': [_vp, _i],
    's2e_sn_grads_inplace': [_vp, _vp, _i, _vp, _vp],
    's2e_weight_grads_batched': [_vp, _vp, _i, _i, _i, _vp, _vp],
    's2e_unpack_weight_grad': [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    's2e_pack_block_map': [_i, _vp, _i, _vp],
    's2e_pack_conv_weights': [_i, _vp, _vp, _i, _i, _vp, _vp],
    's2e_conv2d_workspace_bytes': [_i, C.POINTER(ConvDesc)],
    's2e_conv2d_kernel_kind': [_i, C.POINTER(ConvDesc)],
    's2e_conv2d_wgrad_kernel_kind': [_i, C.POINTER(ConvDesc)],
    's2e_conv2d': [_i, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(ConvDesc), _vp, C.c_size_t, _vp],
    's2e_conv2d_plane_supported': [_i, C.POINTER(ConvDesc)],
    's2e_conv_plane_weight_elems': [C.POINTER(ConvDesc)],
    's2e_conv2d_plane': [_i, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(ConvDesc), _vp],
    's2e_conv2d_stats_slots': [_i, C.POINTER(ConvDesc)],
    's2e_conv2d_stats': [_i, _vp, _vp, _vp, _vp, _vp, C.POINTER(ConvDesc), _vp, _vp],
    's2e_in_stats_from_partials': [_vp, _i, _i, _i, _i, _f, _vp, _vp, _vp],
    's2e_label_rect_lists_bwd': [_vp, _i, _i, _i, _vp, _vp, _vp, _vp],
    's2e_conv2d_rects_supported': [_i, C.POINTER(ConvDesc)],
    's2e_conv2d_rects': [_i, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(ConvDesc), _vp, _vp, _vp],
    's2e_conv2d_wgrad_rects_workspace_bytes': [_i, C.POINTER(ConvDesc)],
    's2e_conv2d_wgrad_rects': [_i, _vp, _vp, _vp, _vp, C.POINTER(ConvDesc), _vp, _vp, _vp, C.c_size_t, _vp],
    's2e_spade_uniform_sums': [_i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    's2e_spade_uniform_grads': [_vp, _i, _vp],
    's2e_conv2d_wgrad_workspace_bytes': [_i, C.POINTER(ConvDesc)],
    's2e_conv2d_wgrad': [_i, _vp, _vp, _vp, _vp, C.POINTER(ConvDesc), _vp, C.c_size_t, _vp],
    's2e_conv2d_wgrad_multi_supported': [_i, C.POINTER(ConvDesc)],
    's2e_conv2d_wgrad_multi_kind': [_i, C.POINTER(ConvDesc)],
    's2e_conv2d_wgrad_multi_workspace_bytes': [_i, _vp, _i],
    's2e_conv2d_wgrad_multi': [_i, _vp, _i, _vp, C.c_size_t, _vp],
    's2e_in_stats_workspace_bytes': [_i, _i, _i, _i],
    's2e_modulate_bwd_workspace_bytes': [_i, _i, _i, _i],
    's2e_instance_norm_fwd': [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _vp],
    's2e_instance_norm_bwd': [_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    's2e_in_stats_counters': [_i, _i, _i, _i],
    's2e_in_stats': [_i, _vp, _i, _i, _i, _f, _vp, _vp, _vp, _vp],
    's2e_modulate_fwd': [_i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    's2e_modulate_bwd': [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    's2e_modulate_bwd_gamma': [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    's2e_modulate_bwd_staged': [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, C.c_double, _i, _i, _vp],
    's2e_modulate_bwd_relay': [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, C.c_double, _i, _i, _vp],
    's2e_spade_conv_modulate_rect': [_i, _i, _i, _i, _i, _i, _i, _vp, _vp],
    's2e_label_rect_classify': [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    's2e_spade_conv_modulate_sparse': [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp],
    's2e_spade_class_table': [_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    's2e_spade_modulate_uniform': [_i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    's2e_spade_conv_modulate_supported': [_i, _i, _i, _i, _i, _i, _i],
    's2e_spade_conv_modulate': [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    's2e_colsum': [_i, _vp, _l, _i, _vp, _vp],
    's2e_wgrad_c8_batch_supported': [_i, _i, _i, _i],
    's2e_wgrad_c8_batch_workspace_bytes': [_i, _vp, _i],
    's2e_wgrad_c8_batch': [_i, _i, _vp, _i, _vp, C.c_size_t, _vp],
    's2e_wgrad_batch_supported': [_i, _i, _i, _i, _i, _i],
    's2e_wgrad_batch_workspace_bytes': [],
    's2e_wgrad_batch': [_i, _vp, _i, _vp, C.c_size_t, _vp],
    's2e_label_conv_block_map': [_i, _vp, _i, _i, _vp],
    's2e_label_conv3x3_batch': [_i, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp],
    's2e_class_table_block_map': [_vp, _i, _vp],
    's2e_spade_class_table_batch': [_i, _vp, _vp, _i, _vp, _i, _vp],
    's2e_label_conv3x3': [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    's2e_onehot_nhwc': [_i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    's2e_openeds_error': [_i, _vp, _vp, _i, _i, _i, _vp, _vp],
    's2e_openeds_error_u8': [_vp, _vp, _i, _i, _i, _vp, _vp],
    's2e_resize_to255': [_i, _vp, _i, _i, _i, _vp, _i, _i, _vp],
    's2e_bilinear_resize_fwd': [_i, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    's2e_bilinear_resize_bwd': [_i, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    's2e_upsample2x_fwd': [_i, _vp, _vp, _i, _i, _i, _i, _vp],
    's2e_upsample2x_bwd': [_i, _vp, _vp, _i, _i, _i, _i, _vp],
    's2e_avgpool3x3s2_fwd': [_i, _vp, _vp, _i, _i, _i, _i, _vp],
    's2e_avgpool3x3s2_bwd': [_i, _vp, _vp, _i, _i, _i, _i, _vp],
    's2e_tanh_bwd': [_i, _vp, _vp, _vp, _l, _vp],
    's2e_lrelu_bwd': [_i, _vp, _vp, _vp, _l, _vp],
    's2e_loss_reduce': [_i, _i, _vp, _vp, _l, _f, _vp, _vp],
    's2e_loss_grad': [_i, _i, _vp, _vp, _l, _f, _vp, _vp, _i, _vp],
    's2e_style_fc_supported': [_i, _i],
    's2e_style_fc_bwd_workspace_bytes': [_i, _i, _i],
    's2e_style_fc_fwd': [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp],
    's2e_style_fc_bwd': [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_size_t, _i, _i, _i, _f, _vp],
    's2e_fc_head_supported': [_i, _i],
    's2e_fc_head_fwd_workspace_bytes': [_i, _i, _i, _i],
    's2e_fc_head_fwd': [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp, C.c_size_t, _vp],
    's2e_fc_head_bwd': [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp],
    's2e_adam_flat': [_vp, _vp, _vp, _vp, _l, _vp, _vp],
    's2e_shard_sum': [_i, _vp, _vp, _i, _l, _vp],
}

_lib = None


class Seg2EyeHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the CDLL.  Raises if the extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Seg2EyeHipError(
                'libseg2eye_hip.so not found at %s -- build it with `python -m seg2eye_amd.build` '
                '(or __graft_entry__.build()).  There is no fallback path.' % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = argtypes
            fn.restype = (C.c_char_p if name == 's2e_last_error' else
                          C.c_size_t if (name.endswith('_workspace_bytes') or name == 's2e_conv_plane_weight_elems') else
                          C.c_long if name in ('s2e_pack_block_map', 's2e_grad_block_map', 's2e_sngrad_block_map', 's2e_sngrad_scratch_floats', 's2e_label_conv_block_map', 's2e_class_table_block_map') else C.c_int)
        _lib = L
    return _lib


def check(rc, what=''):
    if rc != 0:
        msg = lib().s2e_last_error()
        raise Seg2EyeHipError('%s failed (%d): %s' % (what, rc, msg.decode() if msg else '?'))
