"""ctypes binding of ``libspacap_hip.so`` (C ABI: ``include/spacap_hip.h``).

The library is built in-tree by ``spacap3d_amd/csrc/Makefile`` (``__graft_entry__.build()``).
There is no fallback: if the shared object is missing or a symbol does not resolve, importing
this module raises, and every operator raises ``RuntimeError`` on a non-zero return code.
"""
import ctypes
import os

# torch must be imported BEFORE the library is dlopen'ed: torch ships its own libamdhip64.so and loads it
# RTLD_GLOBAL; loaded after it, libspacap_hip.so binds its hip* symbols to that same runtime, so the
# hipStream_t handles torch hands us are valid.  Loaded first, it would bind to /opt/rocm's runtime and the
# process would hold two HIP runtimes ("no ROCm-capable device is detected" on the first launch).
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libspacap_hip.so")
ABI_VERSION = 4

_i = ctypes.c_int
_l = ctypes.c_long
_f = ctypes.c_float
_p = ctypes.c_void_p
_u64 = ctypes.c_uint64

# name -> (restype, argtypes); one row per declaration in include/spacap_hip.h
SIGNATURES = {
    "spacap_abi_version": (_i, []),
    "spacap_last_error": (ctypes.c_char_p, []),
    "spacap_device_count": (_i, []),
    "spacap_opt_n_threads": (_i, [_i]),
    "spacap_fps_workspace_bytes": (ctypes.c_size_t, [_i, _i]),
    "spacap_fps_f32": (_i, [_p, _i, _i, _i, _p, _p, _p]),
    "spacap_gather_points_f32": (_i, [_p, _p, _i, _i, _i, _i, _p, _p]),
    "spacap_gather_points_grad_f32": (_i, [_p, _p, _i, _i, _i, _i, _p, _p]),
    "spacap_ball_query_f32": (_i, [_p, _p, _i, _i, _i, _f, _i, _p, _p]),
    "spacap_ball_query_grid_workspace_bytes": (ctypes.c_size_t, [_i, _i]),
    "spacap_ball_query_grid_f32": (_i, [_p, _p, _i, _i, _i, _f, _i, _p, _p, ctypes.c_size_t, _p]),
    "spacap_group_points_f32": (_i, [_p, _p, _i, _i, _i, _i, _i, _p, _p]),
    "spacap_group_points_grad_workspace_bytes": (ctypes.c_size_t, [_i, _i, _i, _i, _i]),
    "spacap_group_points_grad_f32": (_i, [_p, _p, _i, _i, _i, _i, _i, _p, _p, _p]),
    "spacap_group_max_f32": (_i, [_p, _l, _i, _p, _p, _p]),
    "spacap_group_max_grad_f32": (_i, [_p, _p, _l, _i, _p, _p]),
    "spacap_three_nn_f32": (_i, [_p, _p, _i, _i, _i, _p, _p, _p]),
    "spacap_three_nn_weights_f32": (_i, [_p, _p, _i, _i, _i, _p, _p, _p]),
    "spacap_gather_xyz_f32": (_i, [_p, _p, _i, _i, _i, _p, _p]),
    "spacap_three_interpolate_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _p, _p]),
    "spacap_three_interpolate_grad_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _p, _p]),
    "spacap_three_interpolate_grad_pm_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _p, _p]),
    "spacap_bn_workspace_bytes": (ctypes.c_size_t, [_i]),
    "spacap_bn_set_single_launch": (_i, [_i]),
    "spacap_bn_stats_f32": (_i, [_p, _i, _i, _l, _f, _f, _p, _p, _p, _p, _p]),
    "spacap_bn_relu_apply_f32": (_i, [_p, _p, _p, _p, _i, _i, _l, _p, _p]),
    "spacap_bn_relu_train_f32": (_i, [_p, _i, _i, _l, _f, _f, _p, _p, _p, _p, _p, _p, _p, _p]),
    "spacap_bn_relu_max_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p]),
    "spacap_bn_relu_bwd_f32": (_i, [_p, _p, _p, _p, _p, _i, _i, _l, _p, _p, _p, _p, _p]),
    "spacap_bn_relu_max_bwd_f32": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p]),
    "spacap_relation_feature_fwd_f32": (_i, [_p, _p, _l, _l, _l, _i, _i, _i, _i, _p, _p]),
    "spacap_relation_feature_bwd_f32": (_i, [_p, _p, _p, _l, _l, _l, _i, _i, _i, _i, _p, _p, _p]),
    "spacap_relation_l1_isplit": (_i, []),
    "spacap_relation_l1_supported": (_i, [_i, _i, _i]),
    "spacap_relation_l1_blocks": (_i, [_i, _i, _i]),
    "spacap_relation_l1_fwd_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _p, _p]),
    "spacap_relation_l1_bwd_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p]),
    "spacap_relation_fused_supported": (_i, [_i, _i, _i, _i]),
    "spacap_relation_fused_leave_cus": (_i, [_i]),
    "spacap_relation_fused_zsplit": (_i, [_i, _i, _i]),
    "spacap_relation_fused_nparts": (_i, [_i, _i]),
    "spacap_relation_fused_part_floats": (_i, []),
    "spacap_relation_fused_fwd_f32": (_i, [_p] * 7 + [_i, _i, _p, _p, _p]),
    "spacap_relation_fused_bwd_f32": (_i, [_p] * 7 + [_i, _i, _i, _i, _p, _p, _p, _p]),
    "spacap_caption_prep_fwd_f32": (_i, [_p] * 7 + [_i] * 5 + [_f, _u64, _p] + [_p] * 7 + [_p]),
    "spacap_caption_prep_bwd_f32": (_i, [_p] * 3 + [_i] * 5 + [_f, _u64, _p, _p, _p, _p]),
    "spacap_conv1x1_cm_supported": (_i, [_i, _i, _l]),
    "spacap_conv1x1_cm_f32": (_i, [_i, _p, _p, _p, _i, _i, _i, _l, _p, _p]),
    "spacap_sa_l1_stats_f32": (_i, [_p, _p, _p, _p, _p, _i, _f, _i, _i, _i, _i, _i, _p, _p, _p]),
    "spacap_sa_l1_moments_f32": (_i, [_p, _p, _p, _p, _f, _i, _i, _i, _i, _p, _p, _p]),
    "spacap_sa_l1_moments_finalize_f32": (_i, [_p, _p, _i, _i, _i, _l, _f, _f, _p, _p, _p, _p, _p, _p]),
    "spacap_sa_mid_fwd_l1in_f32": (_i, [_p, _p, _i, _i, _p, _p, _l, _p, _p, _p]),
    "spacap_sa_wgrad_l1in_f32": (_i, [_p, _p, _p, _p, _p, _i, _i, _p, _l, _p, _p]),
    "spacap_sa_dgrad_l1in_f32": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _p, _i, _i, _i, _p, _p, _p]),
    "spacap_sa_dgrad_wgrad_l1in_slabs": (_i, [_l]),
    "spacap_sa_dgrad_wgrad_l1in_f32": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _p, _i, _i, _i, _p, _p, _p, _p]),
    "spacap_sa_l1_dw_f32": (_i, [_p, _i, _p, _i, _i, _p, _p]),
    "spacap_vote_assemble_fwd_f32": (_i, [_p, _p, _p, _i, _i, _i, _p, _p, _p]),
    "spacap_vote_assemble_bwd_f32": (_i, [_p, _p, _i, _i, _i, _p, _p, _p]),
    "spacap_copy_batched": (_i, [_p, _p, _p, _i, _p]),
    "spacap_lab_stamp": (_i, [_p, _p]),
    "spacap_stream_delay": (_i, [_i, _p]),
    "spacap_stream_wait_ge": (_i, [_p, _l, _i, _p, _p]),
    "spacap_stream_signal": (_i, [_p, _p, _p]),
    "spacap_sa_nparts": (_i, []),
    "spacap_sa_wgrad_slabs": (_i, [_l, _i, _i, _i]),
    "spacap_sa_mlp_supported": (_i, [_i, _i, _i]),
    "spacap_sa_l1_fwd_f32": (_i, [_p, _p, _p, _p, _p, _p, _i, _f, _i, _i, _i, _i, _i, _p, _p, _p]),
    "spacap_sa_bn_finalize_f32": (_i, [_p, _i, _l, _f, _f, _p, _p, _p, _p, _p, _p]),
    "spacap_sa_mid_fwd_f32": (_i, [_p, _p, _p, _l, _i, _i, _p, _p, _p]),
    "spacap_sa_pool_fwd_f32": (_i, [_p, _p, _l, _i, _i, _p, _p, _p]),
    "spacap_gemm_rows_supported": (_i, [_i, _i]),
    "spacap_gemm_rows_f32": (_i, [_p, _p, _l, _i, _i, _p, _p]),
    "spacap_sa_mid_fwd_pool_supported": (_i, [_i, _i, _i]),
    "spacap_sa_reserve_cus": (_i, [_i]),
    "spacap_sa_mid_fwd_pool_f32": (_i, [_p, _p, _p, _p, _l, _i, _i, _i, _p, _p, _p, _p, _p]),
    "spacap_sa_pool_finalize_f32": (_i, [_p, _p, _p, _p, _l, _i, _i, _p, _p, _p, _p]),
    "spacap_sa_pool_bwd_f32": (_i, [_p, _p, _p, _p, _p, _p, _l, _i, _i, _p, _p, _p]),
    "spacap_fp_concat_fwd_f32": (_i, [_p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p]),
    "spacap_fp_concat_bwd_f32": (_i, [_p, _i, _i, _i, _i, _p, _p, _p]),
    "spacap_dense_rows_slices": (_i, [_l, _i, _i]),
    "spacap_dense_rows_f32": (_i, [_p, _l, _l, _l, _l, _p, _l, _i, _p, _l, _i, _i, _p, _l, _l, _l, _l, _i, _i, _l, _i, _l, _l, _p]),
    "spacap_dense_sum_slices_f32": (_i, [_p, _i, _l, _l, _p, _p]),
    "spacap_dense_wgrad_small_f32": (_i, [_p, _l, _p, _l, _l, _l, _l, _l, _i, _i, _p, _p, _p]),
    "spacap_dense_wgrad_blocks_slabs": (_i, [_l]),
    "spacap_dense_wgrad_blocks_f32": (_i, [_p, _l, _p, _l, _l, _i, _i, _i, _i, _p, _p]),
    "spacap_dense_wgrad_tall_slabs": (_i, [_l, _i, _i]),
    "spacap_dense_wgrad_tall_f32": (_i, [_p, _l, _p, _l, _l, _i, _i, _i, _p, _p]),
    "spacap_gemm_bf3_supported": (_i, [_i, _i]),
    "spacap_gemm_bf3_split_w_f32": (_i, [_p, _l, _i, _i, _i, _p, _p]),
    "spacap_gemm_bf3_f32": (_i, [_p, _l, _p, _p, _l, _i, _i, _i, _p, _l, _p]),
    "spacap_gemm_bf3_wgrad_slabs": (_i, [_l, _i, _i]),
    "spacap_gemm_bf3_wgrad_f32": (_i, [_p, _l, _p, _l, _l, _i, _i, _i, _p, _p]),
    "spacap_rel_wide_tail_supported": (_i, [_i]),
    "spacap_rel_wide_tail_nparts": (_i, [_l]),
    "spacap_rel_wide_tail_bwd_f32": (_i, [_p, _p, _p, _l, _i, _i, _p, _p, _p]),
    "spacap_rel_wide_l1_supported": (_i, [_i, _i, _i]),
    "spacap_rel_wide_transpose_f32": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "spacap_rel_wide_l1_fwd_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _p, _p]),
    "spacap_rel_wide_l1_bwd_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p]),
    "spacap_sa_l3bwd_part_floats": (_l, [_i, _i]),
    "spacap_sa_l3bwd_dw_f32": (_i, [_p, _i, _p, _p, _i, _i, _p, _p, _p]),
    "spacap_sa_wgrad_pool_supported": (_i, [_i, _i, _i]),
    "spacap_sa_wgrad_pool_parts": (_i, [_l, _i, _i, _i]),
    "spacap_sa_wgrad_pool_f32": (_i, [_p, _p, _i, _p, _p, _p, _l, _i, _i, _p, _p]),
    "spacap_sa_bwd_finalize_f32": (_i, [_p, _i, _l, _p, _p, _p, _p, _p]),
    "spacap_sa_dgrad_f32": (_i, [_p, _p, _i, _p, _p, _p, _p, _p, _l, _i, _i, _p, _p, _p]),
    "spacap_sa_dgrad_l1_f32": (_i, [_p] * 10 + [_f] + [_i] * 6 + [_p, _p, _p]),
    "spacap_sa_wgrad_f32": (_i, [_p, _p, _i, _p, _p, _p, _p, _l, _i, _i, _p, _p]),
    "spacap_sa_l1_bwd_f32": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _f, _i, _i, _i, _i, _i, _p, _p, _i, _p]),
    "spacap_sa_rows_scatter_workspace_bytes": (ctypes.c_size_t, [_i, _i, _l]),
    "spacap_sa_rows_scatter_f32": (_i, [_p, _p, _i, _i, _l, _i, _p, _p, _p]),
    "spacap_sa_rows_index_f32": (_i, [_p, _i, _i, _l, _p, _p]),
    "spacap_sa_rows_gather_f32": (_i, [_p, _i, _i, _l, _i, _p, _p, _p]),
    "spacap_sa_drel_sums_f32": (_i, [_p, _i, _i, _i, _i, _p, _p, _p, _p]),
    "spacap_sa_dw1_assemble_f32": (_i, [_p, _i, _p, _i, _i, _i, _p, _p]),
    "spacap_relu_dropout_fwd_f32": (_i, [_p, _l, _f, _u64, _p, _p, _p]),
    "spacap_relu_dropout_bwd_f32": (_i, [_p, _p, _l, _f, _p, _p]),
    "spacap_dropout_add_fwd_f32": (_i, [_p, _p, _l, _f, _u64, _p, _p, _p]),
    "spacap_dropout_add_bwd_f32": (_i, [_p, _l, _f, _u64, _p, _p, _p]),
    "spacap_scene_aug_doubles": (_i, []),
    "spacap_scene_sample_augment_f32": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p]),
    "spacap_scene_sample_augment_map_f32": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p]),
    "spacap_scene_gather_rows_f32": (_i, [_p, _p, _i, _i, _i, _p, _i, _i, _p]),
    "spacap_l2norm_rows_fwd_f32": (_i, [_p, _l, _i, _p, _p, _p]),
    "spacap_l2norm_rows_bwd_f32": (_i, [_p, _p, _p, _l, _i, _p, _p]),
    "spacap_cap_loss_fwd_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p]),
    "spacap_cap_loss_bwd_f32": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p]),
    "spacap_scene_votes_workspace_bytes": (ctypes.c_size_t, [_i, _i]),
    "spacap_scene_votes_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p]),
    "spacap_det_npart": (_i, []),
    "spacap_det_losses_fwd_f32": (_i, [_p] * 16 + [_i] * 8 + [_f] * 4 + [_p] * 9 + [_p]),
    "spacap_det_losses_bwd_f32": (_i, [_p] * 5 + [_i] * 6 + [_p] * 3 + [_p]),
    "spacap_rel_loss_nparts": (_l, [_i, _i]),
    "spacap_rel_loss_fwd_f32": (_i, [_p] * 7 + [_i] * 3 + [_p] * 3 + [_p]),
    "spacap_rel_loss_bwd_f32": (_i, [_p] * 3 + [_i] * 2 + [_p, _p]),
    "spacap_sum_slabs_f32": (_i, [_p, _i, _l, _p, _p]),
    "spacap_sum_slabs_batched_f32": (_i, [_p, _p, _p, _p, _i, _p]),
    "spacap_adam_flat_f32": (_i, [_p, _p, _p, _p, _l, _f, _f, _f, _f, _f, _p, _f, _p, _p]),
    "spacap_linear_dgrad_mask_f32": (_i, [_p, _p, _p, _f, _l, _i, _i, _p, _p]),
    "spacap_rel_tail_fwd_f32": (_i, [_p, _p, _p, _p, _p, _l, _p, _p, _p]),
    "spacap_rel_tail_bwd_nparts": (_i, [_l]),
    "spacap_rel_tail_bwd_f32": (_i, [_p, _p, _p, _l, _p, _p, _p]),
    "spacap_conv1x1_wgrad_slabs": (_i, [_i, _i, _i, _i]),
    "spacap_conv1x1_wgrad_f32": (_i, [_p, _p, _i, _i, _i, _i, _p, _p]),
    "spacap_conv1x1_wgrad_slabs_batched": (_i, [_i, _i, _i, _i]),
    "spacap_conv1x1_wgrad_batched_f32": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p]),
    "spacap_linear_rows_supported": (_i, [_l, _i, _i]),
    "spacap_linear_rows_f32": (_i, [_p, _p, _p, _l, _i, _i, _i, _p, _p]),
    "spacap_linear_wgrad_slabs": (_i, [_l, _i, _i]),
    "spacap_linear_wgrad_f32": (_i, [_p, _p, _l, _i, _i, _i, _p, _p]),
    "spacap_linear_wgrad_nslab_f32": (_i, [_p, _p, _l, _i, _i, _i, _i, _p, _p]),
    "spacap_linear_wgrad_slabs_batched": (_i, [_l, _i, _i]),
    "spacap_linear_wgrad_batched_f32": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _p]),
    "spacap_layernorm_fwd_f32": (_i, [_p, _p, _p, _l, _i, _f, _p, _p, _p]),
    "spacap_layernorm_bwd_workspace_bytes": (ctypes.c_size_t, [_l, _i]),
    "spacap_layernorm_bwd_f32": (_i, [_p, _p, _p, _p, _l, _i, _f, _p, _p, _p, _p, _p]),
    "spacap_layernorm_bwd_add_f32": (_i, [_p, _p, _p, _p, _p, _l, _i, _f, _p, _p, _p, _p, _p]),
    "spacap_proposal_decode_fwd_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "spacap_proposal_decode_bwd_f32": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p]),
    "spacap_loss_tail_fwd_f32": (_i, [_p, _p, _p, _p, _p, _p, _i, _p, _p, _p]),
    "spacap_loss_tail_bwd_f32": (_i, [_p, _p, _p, _p, _p]),
    "spacap_tf_rows_f32": (_i, [_p, _p]),
    "spacap_tf_rows_parts": (_i, [_l]),
    "spacap_tf_ffn1_f32": (_i, [_p, _p, _p, _l, _i, _f, _u64, _p, _p, _p]),
    "spacap_tf_ffn_f32": (_i, [_i, _p, _p, _p, _p, _p, _l, _i, _f, _u64, _p, _p, _p, _p]),
    "spacap_tf_ffn_pieces_elems": (_l, [_i]),
    "spacap_tf_ffn_split_f32": (_i, [_p, _p, _p, _i, _i, _p]),
    "spacap_tf_ffn_bf3_f32": (_i, [_i, _p, _p, _p, _p, _l, _i, _f, _u64, _p, _p, _p, _p]),
    "spacap_decode_attn_f32": (_i, [_p, _p, _p, _l, _i, _i, _i, _i, _f, _p, _p]),
    "spacap_decode_word_workspace_bytes": (ctypes.c_size_t, [_l, _i]),
    "spacap_decode_word_f32": (_i, [_p, _p, _p, _l, _i, _p, _f, _p, _p, _i, _i, _p, _p, _p]),
    "spacap_tf_gemm_splits": (_i, [_l, _i, _i]),
    "spacap_tf_gemm_f32": (_i, [_p, _p, _l, _i, _i, _i, _i, _p, _p]),
    "spacap_tf_dgrad_mask_f32": (_i, [_p, _p, _p, _f, _l, _i, _i, _p, _p]),
    "spacap_mha_fwd_f32": (_i, [_p, _p, _p] + [_l] * 9 + [_p, _l, _l, _p, _l, _l, _l]
                           + [_i] * 5 + [_f, _f, _u64, _p, _p, _p, _p, _p]),
    "spacap_mha_bwd_workspace_bytes": (ctypes.c_size_t, [_i, _i, _i]),
    "spacap_mha_bwd_delta_f32": (_i, [_p, _p, _p] + [_l] * 9 + [_p, _l, _l, _p, _l, _l, _l]
                                 + [_i] * 5 + [_f, _f, _u64, _p, _p, _p, _p, _p, _p, _p, _l, _p]),
    "spacap_mha_bwd_f32": (_i, [_p, _p, _p] + [_l] * 9 + [_p, _l, _l, _p, _l, _l, _l]
                           + [_i] * 5 + [_f, _f, _u64, _p, _p, _p, _p, _p, _p, _p, _p, _l, _p]),
}


class TfRowsArgs(ctypes.Structure):
    """``spacap_tf_rows_args`` of include/spacap_hip.h (same field order; ctypes applies the C layout rules)."""
    _fields_ = [("mode", _i), ("R", _l), ("a1", _p), ("w1", _p), ("bias1", _p), ("k1", _i), ("drop_p", _f), ("eps", _f),
                ("seed", _u64), ("seed_dev", _p), ("res", _p), ("x_out", _p), ("ln_a", _p), ("ln_b", _p), ("n_out", _p),
                ("stats", _p), ("x_ln", _p), ("g", _p), ("part", _p), ("w2", _p), ("bias2", _p), ("out2", _p), ("n2", _i), ("nparts", _i), ("attn_out", _p), ("delta_out", _p), ("lq", _i)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C spacap3d_amd/csrc`). "
            "spacap3d_amd has no CPU or PyTorch fallback for its native operators.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    got = lib.spacap_abi_version()
    if got != ABI_VERSION:
        raise ImportError(f"libspacap_hip.so ABI {got} != expected {ABI_VERSION}: rebuild the extension")
    return lib


lib = _load()


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib.spacap_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")


_DEFERRED = None   # the active deferred_slab_sums() block, if any


class deferred_slab_sums:
    """While active, ``sum_slabs(part, deferrable=True)`` only allocates its result and queues the reduction;
    ``flush()`` then runs ALL queued reductions in one launch (spacap_sum_slabs_batched_f32).  For a training step's
    backward pass: the ~70 weight-gradient slab sums are only read by the optimizer, so they can wait until the
    backward is over (the engine wraps ``loss.backward()`` in this and flushes before it touches the gradients).
    Call sites mark a sum deferrable only when its result is handed to autograd as a leaf gradient untouched."""

    def __enter__(self):
        global _DEFERRED
        self._prev = _DEFERRED
        self.items = []   # (part, out): slab sums to run
        self.jobs = []    # (g2, x2, with_bias, part): weight gradients (linear_wgrad_partials) that fill queued partials
        self.conv_jobs = []   # (g, x, (B, CO, CI, N), part): 1x1-convolution weight gradients (conv1x1_wgrad_partials)
        _DEFERRED = self
        return self

    def __exit__(self, *exc):
        global _DEFERRED
        _DEFERRED = self._prev
        if exc[0] is None:
            self.flush()
        return False

    def has_job_for(self, part):
        """True when ``part`` is a queued, NOT YET COMPUTED weight-gradient partial."""
        ptr = part.data_ptr()
        return any(j[3].data_ptr() == ptr for j in self.jobs) or any(j[3].data_ptr() == ptr for j in self.conv_jobs)

    def outputs(self):
        """The result tensors of the queued slab sums (still unfilled until ``flush``)."""
        return [o for _, o in self.items]

    def flush(self):
        self.run_jobs()
        self.run_sums()

    def run_jobs(self):
        """Run the queued weight-gradient kernels now (fills their ``part`` buffers)."""
        import torch
        jobs, self.jobs = self.jobs, []
        if jobs:   # first the weight gradients themselves (one launch), then the sums over their slabs
            by_dev = {}
            for j in jobs:
                by_dev.setdefault(j[0].device, []).append(j)
            for dev, group in by_dev.items():
                k = len(group)
                arr = lambda ct, vals: (ct * k)(*vals)
                with torch.cuda.device(dev):
                    check(lib.spacap_linear_wgrad_batched_f32(
                        arr(ctypes.c_void_p, [j[0].data_ptr() for j in group]), arr(ctypes.c_void_p, [j[1].data_ptr() for j in group]),
                        arr(ctypes.c_long, [j[0].shape[0] for j in group]), arr(ctypes.c_int, [j[0].shape[1] for j in group]),
                        arr(ctypes.c_int, [j[1].shape[1] for j in group]), arr(ctypes.c_int, [int(j[2]) for j in group]),
                        arr(ctypes.c_int, [j[3].shape[0] for j in group]),
                        arr(ctypes.c_void_p, [j[3].data_ptr() for j in group]), k, torch.cuda.current_stream(dev).cuda_stream),
                        "spacap_linear_wgrad_batched_f32")
        cjobs, self.conv_jobs = self.conv_jobs, []
        if cjobs:
            by_dev = {}
            for j in cjobs:
                by_dev.setdefault(j[0].device, []).append(j)
            for dev, group in by_dev.items():
                k = len(group)
                arr = lambda ct, vals: (ct * k)(*vals)
                with torch.cuda.device(dev):
                    check(lib.spacap_conv1x1_wgrad_batched_f32(
                        arr(ctypes.c_void_p, [j[0].data_ptr() for j in group]), arr(ctypes.c_void_p, [j[1].data_ptr() for j in group]),
                        arr(ctypes.c_int, [j[2][0] for j in group]), arr(ctypes.c_int, [j[2][1] for j in group]),
                        arr(ctypes.c_int, [j[2][2] for j in group]), arr(ctypes.c_int, [j[2][3] for j in group]),
                        arr(ctypes.c_int, [j[3].shape[0] for j in group]), arr(ctypes.c_int, [int(j[4]) for j in group]),
                        arr(ctypes.c_void_p, [j[3].data_ptr() for j in group]),
                        k, torch.cuda.current_stream(dev).cuda_stream), "spacap_conv1x1_wgrad_batched_f32")

    def run_sums(self):
        import torch
        items, self.items = self.items, []
        by_dev = {}
        for part, out in items:
            by_dev.setdefault(part.device, []).append((part, out))
        for dev, group in by_dev.items():
            k = len(group)
            parts = (ctypes.c_void_p * k)(*[p.data_ptr() for p, _ in group])
            outs = (ctypes.c_void_p * k)(*[o.data_ptr() for _, o in group])
            ns = (ctypes.c_long * k)(*[o.numel() for _, o in group])
            nsl = (ctypes.c_int * k)(*[p.shape[0] for p, _ in group])
            with torch.cuda.device(dev):
                check(lib.spacap_sum_slabs_batched_f32(parts, outs, ns, nsl, k, torch.cuda.current_stream(dev).cuda_stream),
                      "spacap_sum_slabs_batched_f32")


# weight.data_ptr() -> a view of a trainer's flat gradient bucket covering [dW | db] of that Linear layer: the slab sums
# of the Linear-layer gradients then land directly where the all-reduce / optimizer read them, and the per-step gradient
# pack has nothing left to copy for them.  The table is ACTIVE ONLY inside a ``grad_slots(table)`` block -- the engine
# opens one around the backward of a step whose gradients autograd will ASSIGN (``p.grad is None`` everywhere).  Outside
# it every weight gradient is an ordinary fresh tensor: a backward run with ``.grad`` already set (accumulation,
# ``zero_grad(set_to_none=False)``, gradient checks) must never be handed a view of the buffer it accumulates into.
GRAD_SLOTS = {}


class grad_slots:
    def __init__(self, table):
        self.table = table or {}

    def __enter__(self):
        global GRAD_SLOTS
        self._prev, GRAD_SLOTS = GRAD_SLOTS, self.table
        return self

    def __exit__(self, *exc):
        global GRAD_SLOTS
        GRAD_SLOTS = self._prev
        return False


def grad_slot(weight, numel):
    t = GRAD_SLOTS.get(weight.data_ptr())
    return t if (t is not None and t.numel() == numel and t.device == weight.device) else None


def sum_slabs(part, deferrable=False, out=None):
    """part (nslab, ...) float32 contiguous -> sum over dim 0 in ascending order (csrc/elementwise.hip); falls back to
    torch.sum when the row size is not a multiple of 4.  ``deferrable``: see ``deferred_slab_sums``.  ``out``: optional
    destination (float32, contiguous, same number of elements, 16-byte aligned)."""
    import torch
    n = part[0].numel()
    if out is not None and (out.numel() != n or out.data_ptr() % 16 or not out.is_contiguous()):
        out = None
    if part.shape[0] == 1:
        if out is not None and not (_DEFERRED is not None and _DEFERRED.has_job_for(part)):
            out.copy_(part[0].reshape(-1))
            return out.view(part.shape[1:])
        return part[0]   # (inside a deferred block a queued job fills it at the flush: still the leaf gradient)
    can_defer = deferrable and _DEFERRED is not None and part.is_cuda and part.data_ptr() % 16 == 0 and n % 4 == 0 \
        and part.dtype == torch.float32 and part.is_contiguous()
    if not can_defer and _DEFERRED is not None and part.is_cuda and _DEFERRED.has_job_for(part):
        # `part` was only QUEUED by linear_wgrad_partials / conv1x1_wgrad_partials and this call reduces it at once
        # (odd row size, unaligned pointer, caller did not mark the sum deferrable): compute it first
        _DEFERRED.run_jobs()
    if not part.is_cuda or n % 4 or part.dtype != torch.float32 or not part.is_contiguous():
        return part.sum(0)
    if can_defer:
        if out is None:
            with torch.cuda.device(part.device):
                out = torch.empty(part.shape[1:], dtype=torch.float32, device=part.device)
        else:
            out = out.view(part.shape[1:])
        _DEFERRED.items.append((part, out))
        return out
    with torch.cuda.device(part.device):
        out = torch.empty(part.shape[1:], dtype=torch.float32, device=part.device) if out is None else out.view(part.shape[1:])
        check(lib.spacap_sum_slabs_f32(part.data_ptr(), part.shape[0], n, out.data_ptr(),
                                       torch.cuda.current_stream(part.device).cuda_stream), "spacap_sum_slabs_f32")
    return out


def linear_wgrad_partials(g2, x2, with_bias, deferrable=False):
    """Per-slab partials of dW = g2^T x2 (+ db = column sums of g2) for g2 (R, CK), x2 (R, CP) dense float32 on the GPU:
    (nslab, CK*CP [+ CK]) to be summed over dim 0, or None when the shape has no kernel.  Inside a
    ``deferred_slab_sums`` block with ``deferrable=True`` the kernel is only queued: all queued weight gradients run as
    one launch at the flush, before the slab sums -- the caller must then hand the result to ``sum_slabs(...,
    deferrable=True)`` untouched."""
    import torch
    R, CK = g2.shape
    CP = x2.shape[1]
    nslab = int(lib.spacap_linear_wgrad_slabs(R, CK, CP))
    if nslab == 0:
        return None
    with torch.cuda.device(g2.device):
        if deferrable and _DEFERRED is not None:
            # the batch fills the chip: fewer, longer slabs (less to write and to add up)
            nb = int(lib.spacap_linear_wgrad_slabs_batched(R, CK, CP))
            part = torch.empty(nb, CK * CP + (CK if with_bias else 0), dtype=torch.float32, device=g2.device)
            _DEFERRED.jobs.append((g2, x2, bool(with_bias), part))
            return part
        part = torch.empty(nslab, CK * CP + (CK if with_bias else 0), dtype=torch.float32, device=g2.device)
        check(lib.spacap_linear_wgrad_f32(g2.data_ptr(), x2.data_ptr(), R, CK, CP, 1 if with_bias else 0, part.data_ptr(),
                                          torch.cuda.current_stream(g2.device).cuda_stream), "spacap_linear_wgrad_f32")
    return part


def conv1x1_wgrad_partials(g, x, B, CO, CI, N, deferrable=False, with_bias=False):
    """Per-slab partials (nslab, CO*CI) of the weight gradient of a 1x1 convolution on channel-major g (B,CO,N...) and
    x (B,CI,N...); queued inside a ``deferred_slab_sums`` block with ``deferrable=True`` (see linear_wgrad_partials).
    ``with_bias`` (only honoured when the job is queued; check the row length): rows are (CO*CI + CO rounded up to 4) and the
    tail holds the bias gradient's partial sums."""
    import torch
    with torch.cuda.device(g.device):
        if deferrable and _DEFERRED is not None:
            nb = int(lib.spacap_conv1x1_wgrad_slabs_batched(B, CO, CI, N))
            part = torch.empty(nb, CO * CI + (((CO + 3) // 4) * 4 if with_bias else 0), dtype=torch.float32, device=g.device)
            _DEFERRED.conv_jobs.append((g, x, (B, CO, CI, N), part, bool(with_bias)))
            return part
        nslab = int(lib.spacap_conv1x1_wgrad_slabs(B, CO, CI, N))
        part = torch.empty(nslab, CO * CI, dtype=torch.float32, device=g.device)
        check(lib.spacap_conv1x1_wgrad_f32(g.data_ptr(), x.data_ptr(), B, CO, CI, N, part.data_ptr(),
                                           torch.cuda.current_stream(g.device).cuda_stream), "spacap_conv1x1_wgrad_f32")
    return part


def copy_batched(dsts, srcs):
    """``dst.copy_(src)`` for lists of contiguous same-shape, same-dtype CUDA tensors on one device as ONE launch
    (csrc/elementwise.hip: copy_batched_kernel); falls back to torch._foreach_copy_ for anything else."""
    import torch
    ok = len(dsts) == len(srcs) and len(dsts) > 0 and all(
        d.is_cuda and s.is_cuda and d.device == s.device and d.dtype == s.dtype and d.shape == s.shape and d.is_contiguous()
        and s.is_contiguous() for d, s in zip(dsts, srcs))
    if not ok:
        if dsts:
            torch._foreach_copy_(list(dsts), list(srcs), non_blocking=True)
        return
    n = len(dsts)
    PA, LA = ctypes.c_void_p * n, ctypes.c_long * n
    src = PA(*[s.data_ptr() for s in srcs])
    dst = PA(*[d.data_ptr() for d in dsts])
    nb = LA(*[d.numel() * d.element_size() for d in dsts])
    dev = dsts[0].device
    with torch.cuda.device(dev):
        check(lib.spacap_copy_batched(src, dst, nb, n, torch.cuda.current_stream(dev).cuda_stream), "spacap_copy_batched")

