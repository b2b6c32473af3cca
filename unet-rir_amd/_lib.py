"""ctypes binding of libunetrir.so (the C ABI declared in include/unetrir.h).

There is no CPU fallback: if the library is missing or a call fails, this raises.
"""
import ctypes as C
import os

from . import build as _build

_LIB = None

c_f32p = C.c_void_p   # device pointers travel as integers
c_stream = C.c_void_p


class CastDesc(C.Structure):
    """unetrir_cast_desc (include/unetrir.h)."""
    _fields_ = [("w", C.c_void_p), ("same", C.c_void_p), ("transposed", C.c_void_p), ("N", C.c_int), ("T", C.c_int), ("C", C.c_int),
                ("Cp", C.c_int), ("Np", C.c_int), ("reserved", C.c_int), ("packed_s2", C.c_void_p)]


class ReduceDesc(C.Structure):
    """unetrir_reduce_desc (include/unetrir.h): one deferred split-K reduction."""
    _fields_ = [("part", C.c_void_p), ("nsplit", C.c_int), ("n", C.c_size_t), ("out", C.c_void_p), ("reg", C.c_float), ("w", C.c_void_p)]


class Config(C.Structure):
    """unetrir_config: kernel-selection switches (include/unetrir.h)."""
    _fields_ = [(n, C.c_int) for n in ("conv3x3", "conv3x3g", "conv3x3g_pair", "conv3x3h", "conv3x3s", "conv3x3r", "stem",
                                       "upconv3x3g", "wgrad3x3g", "wgrad3x3r", "head_mfma", "wgrad3x3d", "conv3x3d", "conv3x3p", "upconv3x3q", "dyn_tiles", "pw1x1", "igemm2")]


class ConvGeom(C.Structure):
    """unetrir_conv_geom"""
    _fields_ = [("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("Cin", C.c_int), ("Cout", C.c_int),
                ("k", C.c_int), ("stride", C.c_int)]


_SIGS = {
    "unetrir_abi_version": (C.c_int, []),
    "unetrir_get_config": (C.c_int, [C.POINTER(Config)]),
    "unetrir_set_config": (C.c_int, [C.POINTER(Config)]),
    "unetrir_conv2d_fwd_f32": (C.c_int, [C.POINTER(ConvGeom), c_f32p, C.c_int, c_f32p, c_f32p, c_f32p, C.c_int,
                                         c_f32p, C.c_int, c_stream]),
    "unetrir_conv2d_dgrad_f32": (C.c_int, [C.POINTER(ConvGeom), c_f32p, C.c_int, c_f32p, c_f32p, C.c_int, c_f32p,
                                           C.c_int, c_stream]),
    "unetrir_conv2d_wgrad_ws_bytes": (C.c_size_t, [C.POINTER(ConvGeom)]),
    "unetrir_conv2d_wgrad_f32": (C.c_int, [C.POINTER(ConvGeom), c_f32p, C.c_int, c_f32p, C.c_int, c_f32p, C.c_float,
                                           c_f32p, C.c_void_p, C.c_size_t, c_stream]),
    "unetrir_conv2d_transpose_fwd_f32": (C.c_int, [C.POINTER(ConvGeom), c_f32p, C.c_int, c_f32p, c_f32p, c_f32p,
                                                   C.c_int, c_stream]),
    "unetrir_conv2d_transpose_dgrad_f32": (C.c_int, [C.POINTER(ConvGeom), c_f32p, C.c_int, c_f32p, c_f32p, C.c_int,
                                                     c_f32p, C.c_int, c_stream]),
    "unetrir_conv2d_transpose_wgrad_ws_bytes": (C.c_size_t, [C.POINTER(ConvGeom)]),
    "unetrir_conv2d_transpose_wgrad_f32": (C.c_int, [C.POINTER(ConvGeom), c_f32p, C.c_int, c_f32p, C.c_int, c_f32p,
                                                     C.c_float, c_f32p, C.c_void_p, C.c_size_t, c_stream]),
    "unetrir_dense_fwd_ws_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "unetrir_dense_fwd_f32": (C.c_int, [c_f32p, C.c_int, c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p, C.c_size_t, c_stream]),
    "unetrir_transpose_weight_f32": (C.c_int, [c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, c_stream]),
    "unetrir_bn_ws_bytes": (C.c_size_t, [C.c_longlong, C.c_int]),
    "unetrir_bn_stats_f32": (C.c_int, [c_f32p, C.c_int, C.c_longlong, C.c_int, c_f32p, c_f32p, C.c_float, C.c_float,
                                       c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p, C.c_size_t, c_stream]),
    "unetrir_bn_apply_f32": (C.c_int, [c_f32p, C.c_int, C.c_longlong, C.c_int, c_f32p, C.c_int, c_f32p, C.c_int,
                                       c_stream]),
    "unetrir_bn_bwd_f32": (C.c_int, [c_f32p, C.c_int, c_f32p, C.c_int, C.c_longlong, C.c_int, c_f32p, c_f32p, c_f32p,
                                     C.c_int, c_f32p, C.c_int, c_f32p, c_f32p, C.c_void_p, C.c_size_t, c_stream]),
    "unetrir_colsum_f32": (C.c_int, [c_f32p, C.c_int, C.c_longlong, C.c_int, c_f32p, C.c_void_p, C.c_size_t,
                                     c_stream]),
    "unetrir_relu_fwd_f32": (C.c_int, [c_f32p, C.c_int, C.c_longlong, C.c_int, c_f32p, C.c_int, c_stream]),
    "unetrir_relu_bwd_f32": (C.c_int, [c_f32p, C.c_int, c_f32p, C.c_int, C.c_longlong, C.c_int, c_f32p, C.c_int,
                                       c_stream]),
    "unetrir_bn_act_add_f32": (C.c_int, [c_f32p, C.c_int, C.c_longlong, C.c_int, c_f32p, C.c_int, c_f32p, C.c_int, c_f32p,
                                         C.c_int, c_stream]),
    "unetrir_act_bwd_f32": (C.c_int, [c_f32p, C.c_int, c_f32p, C.c_int, C.c_longlong, C.c_int, C.c_int, c_f32p, C.c_int,
                                      c_stream]),
    "unetrir_add_f32": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_longlong, c_stream]),
    "unetrir_nchw_to_nhwc_pad_f32": (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, c_f32p, C.c_int,
                                               c_stream]),
    "unetrir_head6x6_supported": (C.c_int, [C.c_int]),
    "unetrir_head6x6_fwd_f32": (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_f32p, c_f32p, c_f32p,
                                          C.c_int, c_stream]),
    "unetrir_head6x6_wgrad_ws_bytes": (C.c_size_t, [C.c_int]),
    "unetrir_head6x6_wgrad_f32": (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_f32p, C.c_int, c_f32p,
                                            C.c_void_p, C.c_size_t, c_stream]),
    "unetrir_loss_ws_bytes": (C.c_size_t, [C.c_longlong]),
    "unetrir_sigmoid_loss_f32": (C.c_int, [c_f32p, C.c_int, c_f32p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                           c_f32p, c_f32p, c_f32p, C.c_void_p, C.c_size_t, c_stream]),
    "unetrir_sigmoid_nchw_f32": (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, c_f32p, c_stream]),
    "unetrir_sigmoid_bwd_f32": (C.c_int, [c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, c_f32p, c_stream]),
    "unetrir_embedding_fwd_f32": (C.c_int, [C.c_void_p, C.c_int, c_f32p, C.c_int, C.c_int, c_f32p, c_stream]),
    "unetrir_embedding_bwd_f32": (C.c_int, [C.c_void_p, C.c_int, c_f32p, C.c_int, C.c_int, c_f32p, c_stream]),
    "unetrir_mul_f32": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_longlong, c_stream]),
    "unetrir_sumsq_f32": (C.c_int, [c_f32p, C.c_longlong, C.c_float, c_f32p, C.c_int, C.c_void_p, C.c_size_t,
                                    c_stream]),
    "unetrir_adam_f32": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, C.c_longlong, C.c_float, C.c_float, C.c_float,
                                   C.c_float, C.c_float, c_stream]),
    # ---- bf16-storage variants (same argument lists; pointers are void*) ----
    "unetrir_conv2d_fwd_bf16": (C.c_int, [C.POINTER(ConvGeom), c_f32p, C.c_int, c_f32p, c_f32p, c_f32p, C.c_int,
                                          c_f32p, C.c_int, c_stream]),
    "unetrir_conv3x3s2_packed_elems": (C.c_size_t, [C.c_int, C.c_int]),
    "unetrir_conv2d_fwd_packed_bf16": (C.c_int, [C.POINTER(ConvGeom), c_f32p, C.c_int, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int,
                                                 c_f32p, C.c_int, c_stream]),
    "unetrir_conv2d_transpose_dgrad_packed_bf16": (C.c_int, [C.POINTER(ConvGeom), c_f32p, C.c_int, c_f32p, c_f32p, c_f32p, C.c_int,
                                                             c_f32p, C.c_int, c_stream]),
    "unetrir_conv2d_dgrad_bf16": (C.c_int, [C.POINTER(ConvGeom), c_f32p, C.c_int, c_f32p, c_f32p, C.c_int, c_f32p,
                                            C.c_int, c_stream]),
    "unetrir_conv2d_wgrad_bf16": (C.c_int, [C.POINTER(ConvGeom), c_f32p, C.c_int, c_f32p, C.c_int, c_f32p, C.c_float,
                                            c_f32p, C.c_void_p, C.c_size_t, c_stream]),
    "unetrir_conv2d_transpose_fwd_bf16": (C.c_int, [C.POINTER(ConvGeom), c_f32p, C.c_int, c_f32p, c_f32p, c_f32p,
                                                    C.c_int, c_stream]),
    "unetrir_conv2d_transpose_dgrad_bf16": (C.c_int, [C.POINTER(ConvGeom), c_f32p, C.c_int, c_f32p, c_f32p, C.c_int,
                                                      c_f32p, C.c_int, c_stream]),
    "unetrir_conv2d_wgrad_partials_bf16": (C.c_int, [C.POINTER(ConvGeom), c_f32p, C.c_int, c_f32p, C.c_int, c_f32p, C.c_float, c_f32p,
                                                     C.c_void_p, C.c_size_t, C.POINTER(ReduceDesc), c_stream]),
    "unetrir_conv2d_transpose_wgrad_partials_bf16": (C.c_int, [C.POINTER(ConvGeom), c_f32p, C.c_int, c_f32p, C.c_int, c_f32p, C.c_float,
                                                               c_f32p, C.c_void_p, C.c_size_t, C.POINTER(ReduceDesc), c_stream]),
    "unetrir_splitk_reduce_batched": (C.c_int, [C.POINTER(ReduceDesc), C.c_int, c_stream]),
    "unetrir_conv2d_transpose_wgrad_bf16": (C.c_int, [C.POINTER(ConvGeom), c_f32p, C.c_int, c_f32p, C.c_int, c_f32p,
                                                      C.c_float, c_f32p, C.c_void_p, C.c_size_t, c_stream]),
    "unetrir_cast_weight_bf16": (C.c_int, [c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, c_stream]),
    "unetrir_transpose_cast_weight_bf16": (C.c_int, [c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, c_stream]),
    "unetrir_bn_stats_bf16": (C.c_int, [c_f32p, C.c_int, C.c_longlong, C.c_int, c_f32p, c_f32p, C.c_float, C.c_float,
                                        c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p, C.c_size_t, c_stream]),
    "unetrir_bn_apply_bf16": (C.c_int, [c_f32p, C.c_int, C.c_longlong, C.c_int, c_f32p, C.c_int, c_f32p, C.c_int,
                                        c_stream]),
    "unetrir_bn_bwd_bf16": (C.c_int, [c_f32p, C.c_int, c_f32p, C.c_int, C.c_longlong, C.c_int, c_f32p, c_f32p,
                                      C.c_int, c_f32p, C.c_int, c_f32p, c_f32p, C.c_void_p, C.c_size_t, c_stream]),
    "unetrir_colsum_bf16": (C.c_int, [c_f32p, C.c_int, C.c_longlong, C.c_int, c_f32p, C.c_void_p, C.c_size_t,
                                      c_stream]),
    "unetrir_relu_bwd_bf16": (C.c_int, [c_f32p, C.c_int, c_f32p, C.c_int, C.c_longlong, C.c_int, c_f32p, C.c_int,
                                        c_stream]),
    "unetrir_nchw_to_nhwc_pad_bf16": (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, c_f32p, C.c_int,
                                                c_stream]),
    "unetrir_head6x6_fwd_bf16": (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_f32p, c_f32p, c_f32p,
                                           C.c_int, c_stream]),
    "unetrir_head6x6_wgrad_bf16": (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_f32p, C.c_int,
                                             c_f32p, C.c_void_p, C.c_size_t, c_stream]),
    "unetrir_stage_h2d": (C.c_int, [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), c_stream]),
    "unetrir_conv2d_colstat_rows_bf16": (C.c_longlong, [C.POINTER(ConvGeom), C.c_int, C.c_int]),
    "unetrir_conv3x3_kernel_id_bf16": (C.c_int, [C.POINTER(ConvGeom), C.c_int, C.c_int]),
    "unetrir_conv2d_fwd_colstat_bf16": (C.c_int, [C.POINTER(ConvGeom), c_f32p, C.c_int, c_f32p, c_f32p, c_f32p, C.c_int, c_f32p, C.c_int,
                                                  c_f32p, c_stream]),
    "unetrir_conv2d_dgrad_colstat_bf16": (C.c_int, [C.POINTER(ConvGeom), c_f32p, C.c_int, c_f32p, c_f32p, C.c_int, c_f32p, C.c_int,
                                                    c_f32p, c_stream]),
    "unetrir_conv2d_transpose_colstat_rows_bf16": (C.c_longlong, [C.POINTER(ConvGeom), C.c_int]),
    "unetrir_conv2d_transpose_fwd_colstat_bf16": (C.c_int, [C.POINTER(ConvGeom), c_f32p, C.c_int, c_f32p, c_f32p, c_f32p, C.c_int, c_f32p,
                                                            c_stream]),
    "unetrir_bn_stats_colstat": (C.c_int, [c_f32p, C.c_longlong, C.c_longlong, C.c_int, c_f32p, c_f32p, C.c_float, C.c_float,
                                           c_f32p, c_f32p, c_f32p, c_f32p, c_stream]),
    "unetrir_bn_colstat_act_add_f32": (C.c_int, [c_f32p, C.c_longlong, c_f32p, C.c_int, C.c_longlong, C.c_int, c_f32p, c_f32p, C.c_float, C.c_float, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int, c_f32p, C.c_int, c_f32p, C.c_int, c_stream]),
    "unetrir_bn_colstat_act_add_bf16": (C.c_int, [c_f32p, C.c_longlong, c_f32p, C.c_int, C.c_longlong, C.c_int, c_f32p, c_f32p, C.c_float, C.c_float, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int, c_f32p, C.c_int, c_f32p, C.c_int, c_stream]),
    "unetrir_colsum_colstat": (C.c_int, [c_f32p, C.c_longlong, C.c_int, C.c_int, C.c_int, c_f32p, c_stream]),
    "unetrir_cast_weights_batched_bf16": (C.c_int, [C.c_void_p, C.c_int, c_stream]),
    "unetrir_dense_dgrad_supported": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "unetrir_dense_dgrad_ws_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "unetrir_dense_dgrad_f32": (C.c_int, [c_f32p, C.c_int, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t,
                                          c_stream]),
    "unetrir_stft_frames": (C.c_int, [C.c_int, C.c_int]),
    "unetrir_stft_features_f32": (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                            c_f32p, C.c_int, C.c_int, c_stream]),
    "unetrir_istft_features_f32": (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.c_int, c_f32p, c_stream]),
    "unetrir_head6x6_dgrad_supported": (C.c_int, [C.c_int, C.c_int]),
    "unetrir_head6x6_dgrad_bf16": (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, c_f32p, C.c_int, c_f32p, C.c_int,
                                             c_stream]),
    "unetrir_sigmoid_loss_bf16": (C.c_int, [c_f32p, C.c_int, c_f32p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                            c_f32p, c_f32p, c_f32p, C.c_void_p, C.c_size_t, c_stream]),
    "unetrir_sigmoid_bwd_bf16": (C.c_int, [c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, c_f32p, c_stream]),
    "unetrir_add_f32_to_bf16": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_longlong, c_stream]),
    "unetrir_cast_bf16_to_f32": (C.c_int, [c_f32p, c_f32p, C.c_longlong, c_stream]),
    "unetrir_cast_f32_to_bf16": (C.c_int, [c_f32p, c_f32p, C.c_longlong, c_stream]),
    "unetrir_bn_act_add_bf16": (C.c_int, [c_f32p, C.c_int, C.c_longlong, C.c_int, c_f32p, C.c_int, c_f32p, C.c_int, c_f32p,
                                          C.c_int, c_stream]),
    "unetrir_act_bwd_bf16": (C.c_int, [c_f32p, C.c_int, c_f32p, C.c_int, C.c_longlong, C.c_int, C.c_int, c_f32p, C.c_int,
                                       c_stream]),
    "unetrir_bn_inference_affine_f32": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, C.c_float, C.c_int, c_f32p, c_stream]),
    "unetrir_dropout_mask_f32": (C.c_int, [c_f32p, C.c_longlong, C.c_float, C.c_ulonglong, C.c_ulonglong, c_stream]),
    "unetrir_index_to_i32": (C.c_int, [C.c_void_p, C.c_int, C.c_longlong, C.c_void_p, c_stream]),
    "unetrir_reset_tile_tickets": (C.c_int, []),
    "unetrir_bn_bwd_junction_f32": (C.c_int, [c_f32p, C.c_int, c_f32p, C.c_int, c_f32p, C.c_int, C.c_longlong, C.c_int, c_f32p, c_f32p, C.c_int, c_f32p, C.c_int, c_f32p, C.c_int, c_f32p, C.c_int, c_f32p, c_f32p, C.c_void_p, C.c_size_t, c_stream]),
    "unetrir_bn_bwd_junction_bf16": (C.c_int, [c_f32p, C.c_int, c_f32p, C.c_int, c_f32p, C.c_int, C.c_longlong, C.c_int, c_f32p, c_f32p, C.c_int, c_f32p, C.c_int, c_f32p, C.c_int, c_f32p, C.c_int, c_f32p, c_f32p, C.c_void_p, C.c_size_t, c_stream]),
    "unetrir_sigmoid_loss_ex_f32": (C.c_int, [c_f32p, C.c_int, c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                              c_f32p, c_f32p, c_f32p, C.c_void_p, C.c_size_t, c_stream]),
    "unetrir_sigmoid_loss_ex_bf16": (C.c_int, [c_f32p, C.c_int, c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                               c_f32p, c_f32p, c_f32p, C.c_void_p, C.c_size_t, c_stream]),
    "unetrir_sgd_f32": (C.c_int, [c_f32p, c_f32p, C.c_longlong, C.c_float, C.c_float, c_stream]),
    "unetrir_nadam_f32": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, C.c_longlong, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                    C.c_float, C.c_float, C.c_float, c_stream]),
    "unetrir_step_advance": (C.c_int, [C.c_void_p, c_f32p, c_f32p, C.c_int, C.c_int, c_stream]),
    "unetrir_adam_dev_f32": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, C.c_longlong, c_f32p, c_stream]),
    "unetrir_dropout_mask_dev_f32": (C.c_int, [c_f32p, C.c_longlong, C.c_float, C.c_ulonglong, C.c_void_p, C.c_ulonglong, c_stream]),
    "unetrir_prof_enable": (C.c_int, [C.c_int]),
    "unetrir_prof_collect": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
}

EXPORTS = tuple(_SIGS)
PROF_FAMILIES = 8


class UnetrirError(RuntimeError):
    pass


_LIB_PATH = None


def use_library(path):
    """Load another build of the library instead of libunetrir.so (scripts/: the timing-ablation build).  Before first use."""
    global _LIB_PATH
    if _LIB is not None:
        raise UnetrirError("the library is already loaded")
    _LIB_PATH = path


def lib():
    """Load (building first if the sources are newer) libunetrir.so.  Raises if it cannot."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = _LIB_PATH or _build.LIB
    if _LIB_PATH is None and _build.needs_build():
        path = _build.build()
    if not os.path.exists(path):
        raise UnetrirError(f"HIP extension missing: {path}")
    L = C.CDLL(path)
    for name, (res, args) in _SIGS.items():
        fn = getattr(L, name)      # AttributeError if the export is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    if L.unetrir_abi_version() != 1:
        raise UnetrirError("libunetrir.so ABI version mismatch")
    _LIB = L
    return L


def check(err, what):
    if err != 0:
        raise UnetrirError(f"{what} failed with code {err}")
