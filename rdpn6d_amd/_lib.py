"""ctypes binding of librdpn6d_hip.so (the C ABI declared in include/rdpn6d.h).

There is deliberately NO fallback: if the library is missing, or a call fails, a RuntimeError is
raised.  Nothing in this package computes the hot path on the CPU.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RDPN6D_LIB") or os.path.join(HERE, "librdpn6d_hip.so")  # (RDPN6D_LIB: another build of the same ABI, for A/B runs)

_ll = ctypes.c_longlong
c_float_p = ctypes.c_void_p
c_int_p = ctypes.c_void_p
_vp, _i, _f, _u = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_uint


class ConvDesc(ctypes.Structure):
    """Mirror of rdpn6d_conv_desc (include/rdpn6d.h)."""

    _fields_ = [
        ("x", _vp), ("w", _vp), ("scale", _vp), ("shift", _vp), ("res", _vp), ("y", _vp),
        ("B", _i), ("H", _i), ("W", _i), ("Cin", _i), ("in_cs", _i), ("in_co", _i), ("Ho", _i), ("Wo", _i),
        ("stride", _i), ("ntaps", _i), ("dy", _i * 9), ("dx", _i * 9), ("N", _i), ("Npad", _i), ("OH", _i), ("OW", _i),
        ("osy", _i), ("osx", _i), ("ooy", _i), ("oox", _i), ("out_cs", _i), ("out_co", _i), ("res_cs", _i),
        ("res_co", _i), ("act", _i), ("slope", _f),
    ]


# name -> (restype, argtypes); every symbol include/rdpn6d.h declares
SIGNATURES = {
    "rdpn6d_last_error": (ctypes.c_char_p, []),
    "rdpn6d_version": (_i, []),
    "rdpn6d_device_count": (_i, []),
    "farthest_point_sampling": (None, [_vp, _vp, _i, _i]),
    "farthest_point_sampling_init_center": (None, [_vp, _vp, _i, _i]),
    "rdpn6d_fps_host": (_i, [_vp, _vp, _i, _i, _i]),
    "rdpn6d_fps_device": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "rdpn6d_fps_workspace_bytes": (_ll, [_i]),
    "rdpn6d_fps_device_ws": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _ll, _vp]),
    "rdpn6d_conv2d_f32": (_i, [ctypes.POINTER(ConvDesc), _vp]),
    "rdpn6d_conv_tile_for": (_i, [ctypes.POINTER(ConvDesc), _vp, _vp]),
    "rdpn6d_conv_force_tile": (None, [_i, _i]),
    "rdpn6d_conv_set_tap_inner": (None, [_i]),
    "rdpn6d_conv_splitk_ws_floats": (ctypes.c_longlong, [ctypes.POINTER(ConvDesc), _i]),
    "rdpn6d_conv2d_splitk_f32": (_i, [ctypes.POINTER(ConvDesc), _i, _vp, _vp]),
    "rdpn6d_conv2d_bf16": (_i, [ctypes.POINTER(ConvDesc), _i, _vp]),
    "rdpn6d_conv2d_splitk_bf16": (_i, [ctypes.POINTER(ConvDesc), _i, _i, _vp, _vp]),
    "rdpn6d_conv_bf16_force_tile": (None, [_i, _i]),
    "rdpn6d_conv_bf16_uses_pingpong": (_i, [ctypes.POINTER(ConvDesc), _i]),
    "rdpn6d_conv2d_bf16_bnstats": (_i, [ctypes.POINTER(ConvDesc), _vp, _i, _vp, _vp]),
    "rdpn6d_conv2d_bf16_bnbwd": (_i, [ctypes.POINTER(ConvDesc), _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rdpn6d_bn_relu_backward_apply_bf16": (_i, [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _ll, _i, _vp, _i, _vp]),
    "rdpn6d_conv2d_bf16_bnbwd_y": (_i, [ctypes.POINTER(ConvDesc), _vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "rdpn6d_bn_backward_apply_bf16": (_i, [_vp, _i, _i, _vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _ll, _i,
                                           _vp, _i, _vp]),
    "rdpn6d_bn_stats_finalize": (_i, [_vp, _i, _i, _ll, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    "rdpn6d_conv_bf16_tile_for": (_i, [ctypes.POINTER(ConvDesc), _vp, _vp]),
    "rdpn6d_split_bf16x3": (_i, [_vp, _ll, _vp, _ll, _vp]),
    "rdpn6d_conv_bf16x3_eligible": (_i, [ctypes.POINTER(ConvDesc)]),
    "rdpn6d_conv2d_bf16x3": (_i, [ctypes.POINTER(ConvDesc), _ll, _ll, _vp, _ll, _vp]),
    "rdpn6d_conv_bf16x3_kernel_for": (_i, [ctypes.POINTER(ConvDesc)]),
    "rdpn6d_conv2d_bf16x3_ex": (_i, [ctypes.POINTER(ConvDesc), _ll, _ll, _vp, _ll, _vp, _ll, _vp]),
    "rdpn6d_conv_bf16_force_chunk": (None, [_i]),
    "rdpn6d_conv_bf16_force_stages": (None, [_i]),
    "rdpn6d_stem_conv7x7_bf16": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "rdpn6d_maxpool3x3s2_bf16": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "rdpn6d_upsample_bilinear_bf16": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "rdpn6d_xyz_subsample_bf16": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _i, _vp]),
    "rdpn6d_global_max_concat_bf16": (_i, [_vp, _i, _i, _i, _i, _vp]),
    "rdpn6d_cast_f32_bf16": (_i, [_vp, _i, _i, _i, _vp, _i, ctypes.c_longlong, _vp]),
    "rdpn6d_stem_conv7x7_f32": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "rdpn6d_maxpool3x3s2_f32": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "rdpn6d_upsample_bilinear_f32": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "rdpn6d_xyz_subsample_f32": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _i, _vp]),
    "rdpn6d_global_max_concat_f32": (_i, [_vp, _i, _i, _i, _i, _vp]),
    "rdpn6d_groupnorm_relu_f32": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "rdpn6d_groupnorm_relu_h2": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "rdpn6d_dense_glue_f32": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp]),
    "rdpn6d_dense_glue_h2": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "rdpn6d_dense_glue_mt_f32": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp]),
    "rdpn6d_dense_glue_mt_h2": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "rdpn6d_pose_decode_f32": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "rdpn6d_ransac_kabsch_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _i, _f, _u, _vp, _vp, _vp, _vp]),
    "rdpn6d_ransac_kabsch_ex": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _i, _f, _u, _vp, _vp, _vp, _vp, _vp]),
    "rdpn6d_ransac_kabsch_net_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _i, _f, _u, _i, _f, _vp, _vp, _vp,
                                          _vp, _vp]),
    "rdpn6d_ransac_workspace_bytes": (ctypes.c_longlong, [_i]),
    "rdpn6d_ransac_kabsch_ws": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _i, _f, _u, _i, _f, _vp, _vp, _vp, _vp, _vp, _ll, _vp]),
    "rdpn6d_ransac_kabsch_ws_mt": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _f, _i, _f, _u, _i, _f, _vp, _vp, _vp, _vp, _vp, _ll, _vp]),
    "rdpn6d_stem_conv7x7_raw_f32": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp]),
    "rdpn6d_bn_train_stats_f32": (_i, [_vp, _ll, _i, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rdpn6d_bn_apply_f32": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _ll, _i, _i, _vp]),
    "rdpn6d_bn_backward_f32": (_i, [_vp, _i, _i, _vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _i,
                                    _ll, _i, _i, _vp, _vp]),
    "rdpn6d_bn_relu_backward_f32": (_i, [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _ll, _i, _vp, _vp]),
    "rdpn6d_channel_sum_f32": (_i, [_vp, _ll, _i, _i, _i, _vp, _i, _vp, _vp]),
    "rdpn6d_groupnorm_relu_train_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "rdpn6d_groupnorm_relu_backward_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "rdpn6d_wgrad_scratch_floats": (_ll, [_i, _i, _i, _i, _i, _i]),
    "rdpn6d_wgrad_f32": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "rdpn6d_wgrad_bf16": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "rdpn6d_wgrad_f32_strided": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _ll, _ll, _ll, _i, _i,
                                      _vp, _vp]),
    "rdpn6d_wgrad_bf16_strided": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _ll, _ll, _ll,
                                       _i, _i, _vp, _vp]),
    "rdpn6d_wgrad_group_scratch_floats": (_ll, [_i, _i, _i, _i, _i, _i, _i]),
    "rdpn6d_wgrad_bf16_group": (_i, [_i, _vp, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _ll, _ll, _ll,
                                     _i, _i, _vp, _ll, _vp]),
    "rdpn6d_wgrad_bf16x3_strided": (_i, [_vp, _ll, _i, _i, _i, _i, _vp, _ll, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp,
                                         _ll, _ll, _ll, _i, _i, _vp, _vp]),
    "rdpn6d_maxpool3x3s2_backward_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "rdpn6d_upsample_bilinear_backward_f32": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "rdpn6d_global_max_concat_backward_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "rdpn6d_dense_losses_f32": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _f, _vp, _vp, _vp, _vp]),
    "rdpn6d_dense_glue_backward_f32": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "rdpn6d_dense_losses_mt_f32": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _f, _i, _vp, _vp, _vp, _vp]),
    "rdpn6d_dense_glue_backward_mt_f32": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "rdpn6d_pose_train_f32": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _f, _f, _vp, _vp, _vp,
                                   _vp, _vp, _vp]),
    "rdpn6d_pose_train_sym_f32": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _f, _f, _vp, _vp, _i,
                                       _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rdpn6d_ranger_step_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _f, _f, _f, _f, _f, _i, _i, _f, _vp]),
    "rdpn6d_ranger_step_scaled_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _f, _f, _f, _f, _f, _i, _i, _f, _f, _vp, _vp]),
    "rdpn6d_grad_nonfinite_f32": (_i, [_vp, _ll, _vp, _vp]),
    "rdpn6d_region_targets_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "rdpn6d_pose_errors_f64": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "rdpn6d_crop_builder_f32": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "rdpn6d_act_backward_f32": (_i, [_vp, _vp, _ll, _f, _vp]),
    "rdpn6d_transpose_rc_f32": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "rdpn6d_train_vis_scalars_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp]),
    "rdpn6d_rgb_to_nhwc4_f32": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "rdpn6d_stem_im2col_f32": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "rdpn6d_stem_rowpatch_f32": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "rdpn6d_stem_conv7x7_raw_bf16": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp]),
    "rdpn6d_bn_train_stats_bf16": (_i, [_vp, _ll, _i, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rdpn6d_bn_apply_bf16": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _ll, _i, _i, _vp]),
    "rdpn6d_bn_backward_bf16": (_i, [_vp, _i, _i, _vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _i,
                                     _ll, _i, _i, _vp, _vp]),
    "rdpn6d_bn_relu_backward_bf16": (_i, [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _ll, _i, _vp, _vp]),
    "rdpn6d_channel_sum_bf16": (_i, [_vp, _ll, _i, _i, _i, _vp, _i, _vp, _vp]),
    "rdpn6d_maxpool3x3s2_backward_bf16": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "rdpn6d_upsample_bilinear_backward_bf16": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "rdpn6d_global_max_concat_backward_bf16": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "rdpn6d_stem_im2col_bf16": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "rdpn6d_stem_rowpatch_bf16": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "rdpn6d_repack_f32": (_i, [_vp, _vp, _vp, _i, _vp]),
    "rdpn6d_split_h2": (_i, [_vp, _i, _i, _i, _vp, _ll, _vp, _vp]),
    "rdpn6d_conv_h2_kernel_for": (_i, [_vp]),
    "rdpn6d_conv2d_h2": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "rdpn6d_upsample_bilinear_h2": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "rdpn6d_xyz_subsample_h2": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _i, _vp, _vp]),
    "rdpn6d_stem_pool_h2": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rdpn6d_stem_pool_h2_ex": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "rdpn6d_stem_pack_h2": (_i, [_vp, _vp, _vp, _vp]),
    "rdpn6d_global_max_concat_h2": (_i, [_vp, _i, _i, _i, _i, _vp]),
    "rdpn6d_upsample_bilinear_h2_ex": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _vp, _vp]),
    "rdpn6d_global_max_h2": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "rdpn6d_convt3x3s2_const_bias_f32": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "rdpn6d_conv2d_h2_cb": (_i, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "rdpn6d_conv2d_h2_ws": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_longlong, _vp]),
    "rdpn6d_conv_h2_fuse1x1_ok": (_i, [ctypes.POINTER(ConvDesc)]),
    "rdpn6d_conv2d_h2_fuse1x1": (_i, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "rdpn6d_conv_h2_workspace_bytes": (ctypes.c_longlong, [_vp]),
    "rdpn6d_conv_h2_colmax_ok": (_i, [ctypes.POINTER(ConvDesc), _i]),
    "rdpn6d_conv2d_h2_colmax": (_i, [ctypes.POINTER(ConvDesc), _vp, _i, _vp, _vp]),
    "rdpn6d_h2_colmax_decode": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "rdpn6d_conv_h2_set_wfrag": (None, [_i]),
    "rdpn6d_conv_h2_set_clock_probe": (None, [_vp]),
    "rdpn6d_conv_h2_wfrag_wanted": (_i, [ctypes.POINTER(ConvDesc)]),
    "rdpn6d_h2_weight_frag": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "rdpn6d_conv2d_h2_wf": (_i, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp]),
    "rdpn6d_ransac_pnp_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _f, _i, _f, ctypes.c_uint, _i, _f, _vp, _vp, _vp, _vp, _vp]),
    "rdpn6d_ransac_pnp_ex": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _f, _i, _f, ctypes.c_uint, _i, _f, _i, _vp, _vp, _vp, _vp, _vp]),
    "rdpn6d_ransac_pnp_workspace_bytes": (_ll, [_i]),
    "rdpn6d_ransac_pnp_ws": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _f, _i, _f, ctypes.c_uint, _i, _f, _i, _vp, _vp, _vp, _vp, _vp, _ll, _vp]),
    "rdpn6d_select_correspondences_f32": (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rdpn6d_select_correspondences_mt_f32": (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
}

for _k in [k for k in list(SIGNATURES) if "_bf16" in k and "bf16x3" not in k]:  # the fp16 twins (include/rdpn6d.h, last section)
    SIGNATURES[_k.replace("_bf16", "_fp16")] = SIGNATURES[_k]
SIGNATURES["rdpn6d_repack_fp16"] = SIGNATURES["rdpn6d_repack_f32"]

_lib = None


def load():
    """Load (once) and return the ctypes handle; raises if the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build the HIP extension first (python -m rdpn6d_amd.build). "
                "rdpn6d_amd has no CPU fallback."
            )
        # torch first: it brings its own libamdhip64 and this library must bind to THAT copy (one HIP runtime per process).  Loaded
        # the other way round - this .so first, e.g. build() followed by smoke() in one process - the system runtime gets in,
        # torch then initialises a second one and one of the two sees no device.
        import torch  # noqa: F401

        if torch.cuda.device_count() > 0:
            torch.cuda.init()
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = load().rdpn6d_last_error()
        raise RuntimeError(f"rdpn6d HIP call failed ({what}, rc={rc}): {msg.decode() if msg else '?'}")
