"""ctypes loader for libfrlhip.so (the C-ABI boundary declared in include/frl_hip.h).

The product path has no CPU or eager-PyTorch fallback: if the shared library is missing or a call
fails, an exception is raised.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# FRL_HIP_LIB_TAG=x loads libfrlhip_x.so (an A/B build made with FRL_BUILD_TAG=x python build.py); there is no other fallback
LIB_PATH = os.path.join(_HERE, "libfrlhip_%s.so" % os.environ["FRL_HIP_LIB_TAG"] if os.environ.get("FRL_HIP_LIB_TAG") else "libfrlhip.so")

F32, BF16, F16 = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_SIGMOID = 0, 1, 2

_lib = None


class FrlHipError(RuntimeError):
    pass


# name -> (restype, argtypes); kept in sync with include/frl_hip.h (tests check every symbol resolves)
P, I, L, F, S, D = c_void_p, c_int, c_int64, c_float, c_size_t, ctypes.c_double
SIGNATURES = {
    "frl_version": (c_int, []),
    "frl_last_error": (c_char_p, []),
    "frl_device_arch": (c_int, [c_char_p, I]),
    "frl_kernel_timing_enable": (c_int, [I]),
    "frl_kernel_timing_report": (c_int, [c_char_p, I]),
    "frl_pack_cache_table_bytes": (S, []),
    "frl_pack_cache_create": (c_int, [P, S]),
    "frl_pack_cache_destroy": (c_int, [I]),
    "frl_pack_cache_activate": (c_int, [I]),
    "frl_pack_cache_images": (c_int, [I]),
    "frl_pack_cache_refresh": (c_int, [I, P]),
    "frl_conv_workspace_bytes": (S, [I, I, I]),
    "frl_conv1x1_fwd": (c_int, [P, P, P, P, L, I, I, I, I, P, S, P]),
    "frl_conv1x1_bwd_data": (c_int, [P, P, I, P, P, L, I, I, I, P, S, P]),
    "frl_conv1x1_bwd_data_add": (c_int, [P, P, I, P, P, P, L, I, I, I, P, S, P]),
    "frl_wgrad_set_max_workgroups": (c_int, [I]),
    "frl_conv3x3_wgrad_force_generic": (c_int, [I]),
    "frl_conv3x3_tile32": (c_int, [I]),
    "frl_conv1x1_bwd_weight_workspace_bytes": (S, [L, I, I]),
    "frl_conv1x1_bwd_weight": (c_int, [P, P, I, P, P, P, L, I, I, I, P, S, P]),
    "frl_conv_tap_bwd_weight": (c_int, [P, P, I, P, P, L, L, P, L, I, I, I, I, I, I, P, S, I, P]),
    "frl_vq_workspace_bytes": (S, [L, I, I]),
    "frl_vq_assign_fwd": (c_int, [P, P, L, I, I, P, P, P, P, I, P, S, P]),
    "frl_vq_prepared_bytes": (S, [I, I]),
    "frl_vq_prepare": (c_int, [P, L, I, I, I, P, S, P]),
    "frl_vq_stream_tiles": (c_int, [I]),
    "frl_vq_assign_fwd_prepared": (c_int, [P, P, P, L, I, I, P, P, P, P, I, P, S, P]),
    "frl_vq_bwd": (c_int, [P, P, P, P, P, P, P, F, L, I, I, P, P, P, I, P, S, P]),
    "frl_vq_ema_update": (c_int, [P, P, I, I, F, F, P, P, P, P, P]),
    "frl_vq_revive_dead_codes": (c_int, [P, P, L, P, L, I, I, ctypes.c_uint64, P, P, P, I, P]),
    "frl_groupnorm_fwd": (c_int, [P, P, P, P, P, P, I, I, I, I, F, I, I, P]),
    "frl_groupnorm_bwd_workspace_bytes": (S, [I, I, I]),
    "frl_groupnorm_bwd": (c_int, [P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, P, S, P]),
    "frl_scalar_combine": (c_int, [P, P, I, P, P, P]),
    "frl_scalar_fanout": (c_int, [P, P, I, P, P]),
    "frl_scalar_combine_dev": (c_int, [P, P, P, I, P, P, P]),
    "frl_scalar_combine_aux": (c_int, [P, P, P, P, I, P, P, P, P]),
    "frl_scalar_fanout_dev": (c_int, [P, P, P, I, P, P]),
    "frl_encoder2_supported": (c_int, [I, I, I, I, I, I, I]),
    "frl_encoder2_workspace_bytes": (S, [I]),
    "frl_encoder2_fwd": (c_int, [P, P, P, P, P, P, P, P, P, I, I, F, P, S, P]),
    "frl_encoder2_bwd": (c_int, [P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, P, S, P]),
    "frl_decoder_mse_fused_supported": (c_int, [I, I, I, I]),
    "frl_decoder_mse_workspace_bytes": (S, [L, I]),
    "frl_decoder_mse_fwd": (c_int, [P, P, P, P, P, P, P, P, P, L, I, P, S, P]),
    "frl_decoder_mse_bwd": (c_int, [P, P, P, P, P, P, P, P, P, P, P, P, P, P, L, I, P, S, P]),
    "frl_mse_workspace_bytes": (S, []),
    "frl_mse_fwd": (c_int, [P, P, P, L, I, P, I, P, S, P]),
    "frl_mse_bwd": (c_int, [P, P, P, P, P, L, I, P, I, P]),
    "frl_film_modulate_fwd": (c_int, [P, P, P, P, L, I, L, I, I, P]),
    "frl_film_modulate_bwd": (c_int, [P, P, P, P, P, P, L, I, L, I, I, P]),
    "frl_film_fused_supported": (c_int, [I, I, I, I]),
    "frl_film_fused_workspace_bytes": (S, []),
    "frl_film_fused_fwd": (c_int, [P] * 13 + [I, I, I, P, S, P]),
    "frl_film_fused_bwd": (c_int, [P] * 20 + [I, I, I, P, S, P]),
    "frl_gate_blend_fwd": (c_int, [P, P, P, F, P, P, L, I, P]),
    "frl_gate_blend_bwd": (c_int, [P, P, P, P, F, P, P, L, I, P]),
    "frl_mean_time_fwd": (c_int, [P, P, L, I, L, I, P]),
    "frl_add": (c_int, [P, P, F, P, L, I, P]),
    "frl_normalize_tiles": (c_int, [P, I, P, P, P, I, P, L, I, P]),
    "frl_gather_locations_fwd": (c_int, [P, L, L, L, I, I, I, P, L, P, I, P]),
    "frl_segment_sum_rows": (c_int, [P, P, P, L, I, P, L, I, P]),
    "frl_infonce_fwd": (c_int, [P, I, P, P, P, L, P, L, F, I, P, P, P, P, P, P]),
    "frl_infonce_pair_grads": (c_int, [P, I, P, P, P, P, L, L, F, I, P, P, P]),
    "frl_mutual_knn_max_points": (S, [I]),
    "frl_mutual_knn": (c_int, [P, I, I, P, P, F, I, P, P, P]),
    "frl_host_parallel_copy": (c_int, [P, P, S, I]),
    "frl_normalize_chunk_tiles": (c_int, [P, I, I, I, I, I, P, I, I, P, P, I, P, P]),
    "frl_conv3x3_fwd": (c_int, [P, P, P, P, I, I, I, I, I, I, I, P, S, P]),
    "frl_conv3x3_fwd_gate_blend": (c_int, [P, P, P, P, P, P, P, I, I, I, I, I, I, P, S, P]),
    "frl_conv3x3_bwd_data": (c_int, [P, P, I, P, P, I, I, I, I, I, I, P, S, P]),
    "frl_conv3x3_bwd_data_fused": (c_int, [P, P, I, P, P, P, P, P, I, I, I, I, I, I, P, S, P]),
    "frl_conv3x3_bwd_weight_workspace_bytes": (S, [I, I, I, I, I]),
    "frl_conv3x3_bwd_weight": (c_int, [P, P, I, P, P, P, I, I, I, I, I, I, P, S, I, P]),
    "frl_sobel_fwd": (c_int, [P, P, I, I, I, I, I, P]),
    "frl_sobel_bwd": (c_int, [P, P, I, I, I, I, I, P]),
    "frl_sobel_bwd_add": (c_int, [P, P, P, I, I, I, I, I, P]),
    "frl_edge_smooth_stencil_fwd": (c_int, [P, P, P, P, P, P, P, I, I, I, I, I, I, I, P]),
    "frl_edge_smooth_stencil_bwd": (c_int, [P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, P]),
    "frl_smooth_heads_supported": (c_int, [I, I, I, I]),
    "frl_smooth_heads_workspace_bytes": (S, [L]),
    "frl_smooth_heads_bwd_scratch_bytes": (S, [L]),
    "frl_smooth_heads_force_gather": (c_int, [I]),
    "frl_smooth_heads_fwd": (c_int, [P, P, P, P, P, P, P, P, I, I, I, I, P, S, P]),
    "frl_smooth_heads_bwd": (c_int, [P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, S, I, I, I, I, P, S, P]),
    "frl_tcn_block_fwd": (c_int, [P, P, P, P, P, P, P, P, P, P, L, I, I, I, I, I, I, F, I, P, S, P]),
    "frl_tcn_block_bwd_data": (c_int, [P, P, P, P, P, L, I, I, I, I, I, I, P, S, P]),
    "frl_tcn_block_bwd_workspace_bytes": (S, [L, I]),
    "frl_tcn_block_bwd_fused_supported": (c_int, [I, I, I, I, I, I]),
    "frl_tcn_block_bwd_fused_workspace_bytes": (S, [L]),
    "frl_tcn_block_bwd_fused": (c_int, [P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, L, I, I, I, I, F, P, S, P]),
    "frl_adamw_workspace_bytes": (S, []),
    "frl_adamw_clip_step": (c_int, [P, I, P, P, I, F, F, D, D, F, I, P, P, P, P, P, S, P]),
    "frl_multi_tensor_scale_copy": (c_int, [P, I, P, P, I, F, P]),
    "frl_tcn_hot_supported": (c_int, [I, I, I, I, I, I, I]),
    "frl_tcn_hot_fwd_workspace_bytes": (S, []),
    "frl_tcn_hot_bwd_workspace_bytes": (S, [L]),
    "frl_tcn_hot_fwd": (c_int, [P, P, P, P, P, P, P, P, P, L, I, I, F, P, S, P]),
    "frl_tcn_hot_bwd": (c_int, [P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, L, I, I, F, P, S, P]),
    "frl_tcn_hot_force_generic_tiles": (c_int, [I]),
    "frl_tcn_hot_bwd_variant": (c_int, [I]),
    "frl_decoder_mse_bwd_subgroups": (c_int, [I]),
    "frl_tcn_hot_bwd4_share": (c_int, [I, I]),
    "frl_tcn_chain_static_tiles": (c_int, [I]),
    "frl_conv3x3_bwd_data_outmask": (c_int, [P, P, I, P, P, P, I, I, I, I, I, I, I, P, S, P]),
    "frl_gate_blend_bwd_masked": (c_int, [P, P, P, P, F, P, P, L, I, I, P]),
    "frl_smooth_heads_bwd_masked": (c_int, [P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, S, I, I, I, I, I, P, S, P]),
    "frl_tcn_hot_bwd_head_supported": (c_int, [L, I, I]),
    "frl_tcn_hot_bwd_head_workspace_bytes": (S, [L]),
    "frl_tcn_hot_bwd_head": (c_int, [P, P, P, I, P, P, P, P, P, P, P, P, P, P, P, P, P, L, I, I, F, P, S, P]),
    "frl_defer_begin": (c_int, []),
    "frl_defer_pending": (c_int, []),
    "frl_defer_destinations": (c_int, [P, I]),
    "frl_defer_flush": (c_int, [P]),
    "frl_defer_abort": (c_int, []),
    "frl_tcn_hot_bwd_nodx_supported": (c_int, [L, I]),
    "frl_tcn_chain_fwd_workspace_bytes": (S, []),
    "frl_tcn_chain_fwd": (c_int, [P] * 13 + [L, I, I, F, P, S, P]),
    "frl_tcn_block_bwd": (c_int, [P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, L, I, I, I, I, I, I, F, I, P, S, P]),
}


def load():
    """Loads the library once; raises FrlHipError when it is absent (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FrlHipError(
            f"{LIB_PATH} not found: build it with `python vq-vae_amd/build.py` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # PyTorch bundles its own libamdhip64 (same soname): it must be loaded FIRST so that libfrlhip.so binds to the
    # runtime that owns torch's device pointers and streams.  Loading ours first leaves two HIP runtimes in the process
    # and every launch fails with hipErrorNoDevice.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the header and library drift apart
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().frl_last_error()
        raise FrlHipError(f"{what} failed with code {rc}: {msg.decode() if msg else ''}")
