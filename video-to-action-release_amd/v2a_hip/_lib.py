"""ctypes binding of libv2a_hip.so (the C ABI declared in include/v2a.h).

The product path has NO fallback: if the shared library is missing or a symbol is absent this
module raises at import time, and every wrapper raises on a non-zero return code.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# V2A_HIP_LIB: debug hook -- another build of the same library (csrc/Makefile `asan`); the product loads the in-tree .so
LIB_PATH = os.environ.get("V2A_HIP_LIB") or os.path.join(_HERE, "libv2a_hip.so")

P = ctypes.c_void_p
I = ctypes.c_int
F = ctypes.c_float
D = ctypes.c_double
SZ = ctypes.c_size_t
U64 = ctypes.c_uint64
LL = ctypes.c_longlong

# name -> (restype, argtypes).  Order/meaning mirrors include/v2a.h exactly.
SIGNATURES = {
    "v2a_set_precision": (I, [I]),
    "v2a_set_half_format": (I, [I]),
    "v2a_get_half_format": (I, []),
    "v2a_get_precision": (I, []),
    "v2a_set_policy_half": (I, [I]),
    "v2a_get_policy_half": (I, []),
    "v2a_debug_force_tile": (I, [I, I]),
    "v2a_debug_set_parity_classes": (I, [I]),
    "v2a_debug_set_smallk": (I, [I]),
    "v2a_conv2d_x3m_eligible": (I, [I, I, I, I]),
    "v2a_debug_set_maps_kernel": (I, [I]),
    "v2a_debug_force_wgrad_plan": (I, [I, I, I]),
    "v2a_conv2d_workspace_bytes": (SZ, [I, I, I]),
    "v2a_conv2d_fwd": (I, [P] * 8 + [I] * 19 + [P, SZ, P]),
    "v2a_conv2d_wgrad_workspace_bytes": (SZ, [I, I, I]),
    "v2a_conv2d_wgrad": (I, [P, P, P, P, P] + [I] * 17 + [P, SZ, P]),
    "v2a_conv2d_wgrad_h_workspace_bytes": (SZ, [I, I, I]),
    "v2a_conv2d_wgrad_h": (I, [P, P, P, P, P] + [I] * 17 + [P, SZ, P]),
    "v2a_pack_weight": (I, [P, P, I, I, I, I, I, P]),
    "v2a_wgrad_item_bytes": (I, []),
    "v2a_conv2d_wgrad_deferred": (I, [P, P, P, P, P] + [I] * 17 + [P, SZ, P, P, P, P]),
    "v2a_conv2d_wgrad_h_deferred": (I, [P, P, P, P, P] + [I] * 17 + [P, SZ, P, P, P, P]),
    "v2a_wgrad_reduce_multi": (I, [P, P, I, P]),
    "v2a_conv2d_wgrad_describe": (I, [P] * 8 + [I] * 18 + [P, SZ, P, P, P, P, P, P, P, P]),
    "v2a_wgrad_multi_max": (I, []),
    "v2a_wgrad_family": (I, [I]),
    "v2a_conv2d_wgrad_multi": (I, [P, P, P, I, P]),
    "v2a_pack_chunk_elems": (I, []),
    "v2a_debug_wgrad_dma": (I, [I]),
    "v2a_conv2d_plan": (I, [I, I, I, P, P, P]),
    "v2a_conv2d_wgrad_plan": (I, [I, I, I, P, P, P]),
    "v2a_pack_weights_multi": (I, [P, P, I, I, P]),
    "v2a_groupnorm_workspace_bytes": (SZ, [I, I, I, I]),
    "v2a_groupnorm_fwd": (I, [P, P, I, P, P, P, P, I, P, P, P, I, I, I, I, F, I, P, SZ, P]),
    "v2a_groupnorm_bwd": (I, [P] * 5 + [I] + [P] * 9 + [I, I, I, I, I, I, P, SZ, P]),
    "v2a_groupnorm_fwd_t": (I, [P, P, I, P, P, P, P, I, P, P, P, P, I, I, I, I, F, I, P, SZ, P]),
    "v2a_groupnorm_bwd_t": (I, [P] * 5 + [I] + [P] * 10 + [I, I, I, I, I, I, P, SZ, P]),
    "v2a_act_fwd": (I, [P, P, SZ, I, P]),
    "v2a_act_bwd": (I, [P, P, P, SZ, I, P]),
    "v2a_axpy": (I, [P, P, P, F, SZ, P]),
    "v2a_scale_by_device_scalar": (I, [P, SZ, P, P]),
    "v2a_copy2d": (I, [P, P, I, I, I, I, I, P]),
    "v2a_colsum": (I, [P, P, I, I, I, P]),
    "v2a_sincos_embed": (I, [P, P, I, I, I, P]),
    "v2a_add_noise": (I, [P, P, P, P, P, I, I, P, P, I, P]),
    "v2a_mse_loss": (I, [P, P, P, P, I, P]),
    "v2a_mse_loss_scaled": (I, [P, P, P, P, I, P, P]),
    "v2a_policy_sched_step": (I, [P, P, P, P, I, F, F, F, F, F, I, P]),
    "v2a_gn_param_grads_multi": (I, [P, P, I, P]),
    "v2a_groupnorm_takes_slabs": (I, [I, I, I]),
    "v2a_groupnorm_takes_post": (I, [I, I, I]),
    "v2a_groupnorm_fwd_s": (I, [P, P, I, P, P, P, P, I, P, P, SZ, P, P, I, I, I, I, F, I, P, I, SZ, P, P, P, I, SZ, P, P, SZ, P]),
    "v2a_groupnorm_bwd_s": (I, [P] * 5 + [I] + [P] * 5 + [SZ] + [P] * 5 + [I, I, I, I, I, I, P, I, SZ, P, P, P, SZ, P]),
    "v2a_policy_persist_op_bytes": (SZ, []),
    "v2a_policy_persist_args_bytes": (SZ, []),
    "v2a_policy_persist_waves_per_wg": (I, []),
    "v2a_policy_persist_lds_bytes": (SZ, [I, I, I, I, I, I, I, I]),
    "v2a_policy_persist_launch": (I, [P, I, SZ, P]),
    "v2a_dp_max_world": (I, []),
    "v2a_dp_slots": (I, []),
    "v2a_dp_signal_bytes": (SZ, []),
    "v2a_dp_arena_alloc": (I, [P, SZ]),
    "v2a_dp_arena_free": (I, [P]),
    "v2a_dp_signal_alloc": (I, [P]),
    "v2a_dp_signal_free": (I, [P]),
    "v2a_dp_errword_alloc": (I, [P]),
    "v2a_dp_errword_free": (I, [P]),
    "v2a_dp_ipc_export": (I, [P, P, P, P]),
    "v2a_dp_ipc_open": (I, [P, P]),
    "v2a_dp_ipc_close": (I, [P]),
    "v2a_dp_allreduce_direct": (I, [P, P, I, I, SZ, SZ, I, ctypes.c_uint32, P, I, I, P]),
    "v2a_h5_open": (I, [ctypes.c_char_p, P]),
    "v2a_h5_close": (None, [P]),
    "v2a_h5_last_error": (ctypes.c_char_p, [P]),
    "v2a_h5_exists": (I, [P, ctypes.c_char_p]),
    "v2a_h5_list": (ctypes.c_long, [P, ctypes.c_char_p, P, SZ]),
    "v2a_h5_dataset_info": (I, [P, ctypes.c_char_p, P, P, P, P, P, P]),
    "v2a_h5_read": (I, [P, ctypes.c_char_p, P, SZ]),
    "v2a_unnormalize_action": (I, [P, P, I, P, P, I, P]),
    "v2a_nchw_to_nhwc_f32": (I, [P, P, I, I, I, I, P]),
    "v2a_nchw_to_nhwc_u8": (I, [P, P, I, I, I, I, P]),
    "v2a_nchw_to_nhwc4p": (I, [P, I, P, I, I, I, I, I, P]),
    "v2a_nhwc_to_nchw_f32": (I, [P, P, I, I, I, P]),
    "v2a_video_pack": (I, [P, P, P, I, I, I, SZ, SZ, I, P]),
    "v2a_mha_fwd": (I, [P, P, P, P, P, I, I, I, I, I, I, I, I, F, U64, U64, P]),
    "v2a_mha_bwd": (I, [P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, F, U64, U64, P]),
    "v2a_dropout": (I, [P, P, SZ, F, U64, U64, P]),
    "v2a_video_qsample": (I, [P, P, P, P, P, P, I, SZ, I, P]),
    "v2a_video_loss_workspace_bytes": (SZ, [I]),
    "v2a_video_loss_fwd": (I, [P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, P, SZ, P]),
    "v2a_video_loss_bwd": (I, [P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, P]),
    "v2a_video_denoise_step": (I, [P, P, P, P, P, I, I, I, F, F, F, F, F, F, F, F, I, I, I, P]),
    "v2a_video_denoise_row_bytes": (I, []),
    "v2a_video_denoise_step2": (I, [P, P, P, P, P, I, I, I, I, I, P, P, I, I, P, P]),
    "v2a_video_sampler_advance": (I, [P, P, P, I, I, P]),
    "v2a_emb_linear_multi_max": (I, []),
    "v2a_emb_linear_multi": (I, [P, I, I, P, P, P, P, I, P]),
    "v2a_philox_normal": (I, [P, SZ, U64, P, U64, P]),
    "v2a_philox_normal_rows": (I, [P, I, SZ, P, U64, P]),
    "v2a_philox_randint": (I, [P, I, I, U64, P, U64, P]),
    "v2a_advance_counter": (I, [P, U64, P]),
    "v2a_debug_timestamp": (I, [P, P]),
    "v2a_conv2d_p3_eligible": (I, [I, I, I, I, I]),
    "v2a_conv2d_fwd_p3": (I, [P, SZ, P, SZ, P, SZ, P, P, P, P] + [I] * 14 + [P, P, SZ, P]),
    "v2a_split3_f32": (I, [P, P, SZ, SZ, P]),
    "v2a_set_f32_conv_mode": (I, [I]),
    "v2a_get_f32_conv_mode": (I, []),
    "v2a_attention_fwd": (I, [P, P, I, I, I, I, P]),
    "v2a_perceiver_attention_bwd": (I, [P] * 9 + [I, I, I, I, I, F, P]),
    "v2a_layernorm_bwd": (I, [P, P, P, P, P, I, I, F, P]),
    "v2a_bcast_rows": (I, [P, P, I, I, I, F, P]),
    "v2a_attention_bwd": (I, [P, P, P, P, I, I, I, I, P]),
    "v2a_sumpool2x2": (I, [P, P, I, I, I, I, P]),
    "v2a_colsum_batched_workspace_bytes": (SZ, [I, I, I]),
    "v2a_colsum_batched": (I, [P, P, I, I, I, I, P, SZ, P]),
    "v2a_perceiver_attention": (I, [P, P, P, P, P, I, I, I, I, I, F, P]),
    "v2a_layernorm": (I, [P, P, P, P, I, I, F, P]),
    "v2a_mean_rows": (I, [P, P, I, I, I, P]),
    "v2a_maxpool3x3s2_fwd": (I, [P, P, P, I, I, I, I, P]),
    "v2a_maxpool3x3s2_bwd": (I, [P, P, P, I, I, I, I, P]),
    "v2a_spatial_softmax_fwd": (I, [P, P, P, I, I, I, I, P]),
    "v2a_spatial_softmax_bwd": (I, [P, P, P, P, I, I, I, I, P]),
    "v2a_opt_chunk_elems": (I, []),
    "v2a_opt_state_bytes": (SZ, []),
    "v2a_opt_state_init": (I, [P, D, D, D, D, D, D, D, D, D, D, I, I]),
    "v2a_conv2d_h_workspace_bytes": (SZ, [I, I, I]),
    "v2a_conv2d_fwd_h": (I, [P, P, P, P, P, P, P, P, P, P] + [I] * 17 + [P, P, SZ, P]),
    "v2a_groupnorm_h_workspace_bytes": (SZ, [I, I, I]),
    "v2a_groupnorm_fwd_h": (I, [P, P, I, P, P, P, P, P, P, P, I, I, I, I, F, I, P, SZ, P]),
    "v2a_attention_fwd_h": (I, [P, P, I, I, I, I, P]),
    "v2a_conv2d_dma_f32_workspace_bytes": (SZ, [I, I, I]),
    "v2a_conv2d_h_can_emit_stats": (I, [I, I, I]),
    "v2a_conv2d_fwd_dma_f32": (I, [P, P, P, P, P, P, P, P] + [I] * 17 + [P, P, SZ, P]),
    "v2a_conv2d_fwd_dma_f32_d": (I, [P, P, P, P, P, P, P, P] + [I] * 17 + [P, P, SZ, P]),
    "v2a_conv2d_fwd_window_f32": (I, [P, P, P, P, P] + [I] * 12 + [P, SZ, P]),
    "v2a_conv2d_h_splits": (I, [I, I, I]),
    "v2a_conv2d_fwd_h_d": (I, [P, P, P, P, P, P, P, P] + [I] * 17 + [P, P, SZ, P]),
    "v2a_conv2d_h2_eligible": (I, [I, I, I, I, I]),
    "v2a_conv2d_dma_f32_can_emit_stats": (I, [I, I, I]),
    "v2a_groupnorm_fwd_st": (I, [P, P, I, P, P, P, P, P, I, I, I, I, F, I, P, P, P, SZ, P]),
    "v2a_conv2d_h3_eligible": (I, [I] * 13),
    "v2a_conv2d_t3_eligible": (I, [I] * 13),
    "v2a_conv2d_x3t_eligible": (I, [I] * 7),
    "v2a_conv2d_x3p_eligible": (I, [I] * 5),
    "v2a_conv2d_x3p_ups4_eligible": (I, [I] * 5),
    "v2a_conv2d_hp_ups4_eligible": (I, [I] * 5),
    "v2a_conv2d_fwd_hp_ups4": (I, [P, P, P, P, P, I, I, I, I, I, P]),
    "v2a_pack_weight_ups4": (I, [P, P, I, I, P]),
    "v2a_conv2d_fwd_x3p_ups4": (I, [P, P, P, P, P, I, I, I, I, I, P]),
    "v2a_conv2d_fwd_x3p_gn": (I, [P, P, P, P, P, I, I, I, P, P, P, P, I, I, I, I, I, P]),
    "v2a_groupnorm_stats_f32": (I, [P, P, P, I, I, I, I, F, P, SZ, P]),
    "v2a_groupnorm_prep_h": (I, [P, P, I, P, P, P, P, P, P, I, I, I, I, F, P, P, SZ, P]),
    "v2a_groupnorm_apply_h": (I, [P, P, I, P, P, I, I, I, I, P]),
    "v2a_conv2d_fwd_h3_gn": (I, [P, P, I, P, I, I, P, P, P, P, P, P, I, I, I, I, I, I, P, P]),
    "v2a_conv2d_fwd_t3": (I, [P] * 7 + [I] * 6 + [P, P]),
    "v2a_conv2d_fwd_h3": (I, [P, P, P, P, P, P, P, I, I, I, I, I, I, I, P, P]),
    "v2a_conv2d_fwd_h2": (I, [P, P, P, P, P, P, P, P] + [I] * 16 + [P, P]),
    "v2a_pack_weight_h": (I, [P, P, I, I, I, P]),
    "v2a_cast_f32_bf16": (I, [P, P, SZ, P]),
    "v2a_cast_f32_h": (I, [P, P, SZ, I, P]),
    "v2a_cast_h_f32": (I, [P, P, SZ, I, P]),
    "v2a_pad_cast_f32_bf16": (I, [P, P, SZ, I, I, P]),
    "v2a_cast_bf16_f32": (I, [P, P, SZ, P]),
    "v2a_opt_state_peek": (I, [P, P, P, P, P]),
    "v2a_opt_state_counters": (I, [P, P, P, P]),
    "v2a_opt_state_set_counters": (I, [P, LL, LL, I, D]),
    "v2a_opt_state_set_scaler": (I, [P, D, D, D, I, I]),
    "v2a_opt_state_scaler": (I, [P, P, P, P, P, P, P, P, P]),
    "v2a_opt_state_loss_scale_offset": (SZ, []),
    "v2a_opt_step": (I, [P, P, I, P, P, I, P]),
    "v2a_opt_step_packed": (I, [P, P, I, P, P, I, P, I, I, P]),
    "v2a_opt_presum": (I, [P, P, I, I, P, P]),
    "v2a_opt_scale_grads": (I, [P, P, I, F, P]),
    "v2a_replay_sample_indices": (I, [P, P, P, I, I, I, P, P]),
    "v2a_replay_count_uniform_below": (I, [P, I, D]),
    "v2a_mt_seed_numpy": (I, [P, ctypes.c_uint32]),
    "v2a_mt_seed_python": (I, [P, P, I]),
    "v2a_replay_gather": (I, [P, I, P, P, P, P, P, I, I, I, I, I, I, I, P]),
}


class V2AError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise V2AError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C video-to-action-release_amd/csrc`). There is no CPU fallback for the product path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise V2AError(f"libv2a_hip.so does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()

_ERR = {-1: "bad argument", -2: "kernel launch failed", -3: "workspace too small", -4: "episode shorter than act_len + 1"}


def check(rc, what=""):
    if rc != 0:
        raise V2AError(f"{what}: error {rc} ({_ERR.get(rc, 'unknown')})")
