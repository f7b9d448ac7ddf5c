"""ctypes binding of libavdf_hip.so (the C ABI declared in include/avdf.h).

The library is built in-tree by `audio_visual_deepfake_detection_amd.csrc.build` (hipcc,
--offload-arch=gfx950).  There is NO fallback: if the shared object is missing or a symbol
cannot be resolved, importing this module raises.
"""
import ctypes
import os

# torch FIRST: PyTorch-ROCm ships its own libamdhip64 (torch/lib) while libavdf_hip.so is linked against the system one
# (/opt/rocm/lib).  Whichever is mapped first serves both (same SONAME) -- but only torch's copy is the runtime that owns the
# devices torch hands us pointers from.  Loading this library before torch left the process with a HIP runtime that
# reports "no ROCm-capable device" at the first kernel launch (seen on the GPU box: build() then smoke() in one process).
import torch  # noqa: F401,E402

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AVDF_LIB_PATH") or os.path.join(_HERE, "libavdf_hip.so")     # AVDF_LIB_PATH: A/B builds (tools)

P, I, F, L = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int64

# name -> (restype, argtypes); must match include/avdf.h
SIGNATURES = {
    "avdf_last_error": (ctypes.c_char_p, []),
    "avdf_version": (I, []),
    "avdf_conv1d_fwd": (I, [P, P, P, P, P, P, I, I, I, I, I, I, I, I, P]),
    "avdf_conv1d_fwd_ex": (I, [P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, P]),
    "avdf_conv1d_fwd_strided": (I, [P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, L, L, L, I, P]),
    "avdf_softmax_alibi_heads": (I, [P, I, I, I, L, F, P, P]),
    "avdf_gemm_mode": (I, [I]),
    "avdf_gemm_override": (I, [I, I, I]),
    "avdf_tuned_extra": (I, [P]),
    "avdf_conv1d_fwd_grouped": (I, [I, P, P, P, P, P, P, I, I, I, I, I, I, I, I, P]),
    "avdf_conv1d_bwd_data_grouped": (I, [I, P, P, P, P, P, I, I, I, I, I, I, I, I, P]),
    "avdf_conv1d_fwd_residual": (I, [P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, P]),
    "avdf_conv1d_bwd_pair_grouped": (I, [I, P, P, P, P, P, P, P, P, I, I, I, I, I, P]),
    "avdf_conv1d_bwd_pair_dact": (I, [I, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, P]),
    "avdf_fpn_tail_fwd": (I, [I, P, P, P, P, P, P, P, P, P, P, P, P, I, I, F, P]),
    "avdf_fpn_tail_bwd": (I, [I, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, P]),
    "avdf_bwd_pair_mode": (I, [I]),
    "avdf_conv1d_bwd_weight_grouped": (I, [I, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, P]),
    "avdf_conv1d_bwd_data": (I, [P, P, P, P, P, I, I, I, I, I, I, I, I, P]),
    "avdf_conv1d_bwd_weight_ws": (L, [I, I, I, I, I]),
    "avdf_conv1d_bwd_weight": (I, [P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, P]),
    "avdf_layernorm_fwd": (I, [P, P, P, P, P, P, I, I, F, I, P, P, I, P]),
    "avdf_layernorm_bwd_ws": (L, [I, I]),
    "avdf_layernorm_bwd": (I, [P, P, P, P, P, P, P, P, P, P, I, I, I, I, P]),
    "avdf_layernorm_fwd_rs": (I, [P, P, P, P, P, P, I, I, F, I, P, P]),
    "avdf_layernorm_bwd_rs": (I, [P, P, P, P, P, P, P, P, P, P, I, I, I, I, P, P]),
    "avdf_dwconv_fwd": (I, [P, P, P, P, I, I, I, I, I, P]),
    "avdf_dwconv_bwd_ws": (L, [I, I, I]),
    "avdf_dwconv_bwd": (I, [P, P, P, P, P, P, P, I, I, I, I, I, I, P]),
    "avdf_local_attn_fwd": (I, [P, P, P, P, P, P, I, I, I, I, I, F, P]),
    "avdf_local_attn_bwd_ws": (L, [I, I, I, I]),
    "avdf_local_attn_bwd": (I, [P, P, P, P, P, P, P, P, P, I, I, I, I, I, F, P]),
    "avdf_local_attn_fwd_ex": (I, [P, P, P, P, P, P, P, P, I, I, I, I, I, F, P]),
    "avdf_local_attn_bwd_ex": (I, [P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, F, P]),
    "avdf_global_attn_fwd": (I, [P, P, P, P, P, P, I, I, I, I, I, F, P]),
    "avdf_global_attn_bwd_ws": (L, [I, I, I, I]),
    "avdf_global_attn_bwd": (I, [P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, F, P]),
    "avdf_maxpool3s2_fwd": (I, [P, P, I, I, I, P]),
    "avdf_maxpool3s2_bwd": (I, [P, P, P, I, I, I, P]),
    "avdf_affine_residual_fwd": (I, [P, P, P, P, P, P, P, I, I, I, P]),
    "avdf_affine_residual_bwd_ws": (L, [I, I, I]),
    "avdf_affine_residual_bwd_rows": (I, [I, I, I]),
    "avdf_affine_residual_bwd": (I, [P, P, P, P, P, P, P, P, P, P, I, I, I, I, P]),
    "avdf_upsample_fwd": (I, [P, P, I, I, I, I, P]),
    "avdf_upsample_bwd": (I, [P, P, I, I, I, I, P]),
    "avdf_downsample_fwd": (I, [P, P, I, I, I, I, P]),
    "avdf_downsample_bwd": (I, [P, P, I, I, I, I, P]),
    "avdf_add_upsampled": (I, [P, P, P, I, I, I, I, P]),
    "avdf_bias_mask_fwd": (I, [P, P, P, P, L, I, P]),
    "avdf_masked_colsum_ws": (L, [I]),
    "avdf_masked_colsum": (I, [P, P, P, P, L, I, I, P]),
    "avdf_l1_mean_ws": (L, []),
    "avdf_l1_mean": (I, [P, P, P, P, P, P, I, L, F, P]),
    "avdf_count_real": (I, [P, P, I, P]),
    "avdf_l1_mean_dev": (I, [P, P, P, P, P, P, P, P, I, L, P]),
    "avdf_scale_rows": (I, [P, P, P, I, I, P]),
    "avdf_transpose": (I, [P, P, I, I, I, P]),
    "avdf_instnorm_lrelu_fwd": (I, [P, P, P, P, I, I, I, F, F, P]),
    "avdf_instnorm_lrelu_bwd": (I, [P, P, P, P, P, I, I, I, F, P]),
    "avdf_maxavgpool_fwd": (I, [P, P, I, I, I, P]),
    "avdf_maxavgpool_bwd": (I, [P, P, P, I, I, I, P]),
    "avdf_smallconv_fwd": (I, [P, P, P, P, P, P, I, I, I, I, I, P]),
    "avdf_smallconv_bwd_ws": (L, [I, I, I, I, I]),
    "avdf_smallconv_bwd": (I, [P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, P]),
    "avdf_interp_linear_batched": (I, [P, P, I, L, L, I, I, I, I, I, P]),
    "avdf_interp_linear": (I, [P, P, I, I, I, I, I, P]),
    "avdf_loc_losses_ws": (L, [I, I]),
    "avdf_loc_assign": (I, [P, P, P, P, P, P, I, I, I, F, I, P]),
    "avdf_loc_losses": (I, [P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, F, F, F, F, P]),
    "avdf_focal_loss_sum": (I, [P, P, P, P, L, P]),
    "avdf_decode_nms": (I, [P, P, P, P, P, P, P, P, I, I, F, F, F, F, F, I, P]),
    "avdf_nms1d_host": (I, [P, P, L, F, P, P]),
    "avdf_frame_window": (I, [P, P, P, L, I, I, I, P]),
    "avdf_power_spectrum": (I, [P, P, L, I, P]),
    "avdf_logmel_norm": (I, [P, P, L, I, F, F, F, P]),
    "avdf_conv2d_c1_bn_relu_pool": (I, [P, P, P, P, P, I, I, I, P]),
    "avdf_mel_im2col": (I, [P, P, I, I, I, P]),
    "avdf_maxpool2x2": (I, [P, P, I, I, I, I, P]),
    "avdf_softnms1d_host": (I, [P, P, L, F, F, F, I, P, P, P]),
    "avdf_batched_nms_host": (I, [P, P, L, F, F, L, F, P, P, P, P]),
    "avdf_flash_attn_alibi": (I, [P, P, P, P, P, I, I, I, I, I, L, F, P]),
    "avdf_wave_conv_c1": (I, [P, P, P, L, L, I, I, I, P]),
    "avdf_grouped_conv_pos": (I, [P, P, P, P, I, I, I, I, P]),
    "avdf_grouped_conv1d_fwd": (I, [P, P, P, P, I, I, I, I, P]),
    "avdf_residual_ln_fwd": (I, [P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, F, P]),
    "avdf_residual_ln_bwd_ws": (L, [I, I, I]),
    "avdf_residual_ln_bwd": (I, [P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, P]),
    "avdf_chain_fwd": (I, [P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, F, P]),
    "avdf_chain_fwd_map": (I, [P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, F, P]),
    "avdf_chain_bwd_ws": (L, [I, I, I]),
    "avdf_chain_bwd_rows": (I, [I, I]),
    "avdf_chain_bwd": (I, [P, P, P, P, P, P, P, P, P, P, P, I, P, I, I, I, I, I, P]),
    "avdf_chain_bwd_res": (I, [P, P, P, P, P, P, P, P, P, P, P, P, I, P, I, I, I, I, I, P]),
    "avdf_chain_bwd_map": (I, [P, P, P, P, P, P, P, P, P, P, P, P, I, P, P, I, I, I, I, I, P]),
    "avdf_chain_bwd_folds": (I, [I, I]),
    "avdf_reduce_segments": (I, [P, I, I, I, P, P, P, P, I, P]),
    "avdf_reduce_segments_batched": (I, [I, P, P, P, P, P, P, P, P, P]),
    "avdf_conv3d_k3": (I, [P, P, P, P, I, I, I, I, I, I, I, P]),
    "avdf_conv3d_mode": (I, [I]),
    "avdf_video_prep": (I, [P, P, I, I, I, I, I, P]),
    "avdf_maxpool_hw": (I, [P, P, L, I, I, I, I, P]),
    "avdf_softmax_alibi": (I, [P, I, I, F, F, P]),
    "avdf_gap_concat_fwd": (I, [P, P, P, I, P, P, I, I, P]),
    "avdf_gap_concat_bwd": (I, [P, P, P, I, I, I, P]),
    "avdf_head_out_fwd": (I, [P, P, P, I, P, I, I, I, P]),
    "avdf_head_out_bwd": (I, [P, P, P, P, P, P, I, P, I, I, I, P]),
    "avdf_mask_pyramid": (I, [P, P, I, I, I, I, P]),
    "avdf_keep_from_uniform": (I, [P, P, P, L, I, I, F, P]),
    "avdf_sumsq_ws": (L, []),
    "avdf_sumsq": (I, [P, P, P, L, P]),
    "avdf_adamw_ema": (I, [P, P, P, P, P, P, P, L, F, F, F, F, F, F, P]),
    "avdf_adamw_ema_seg": (I, [P, P, P, P, P, P, P, L, I, P, P, F, F, F, F, F, I, P]),
    "avdf_sumsq_ex": (I, [P, P, P, L, I, P]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is mandatory (no CPU fallback). Build it with "
            "`python -m audio_visual_deepfake_detection_amd.csrc.build` or `__graft_entry__.build()`.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing -> loud failure
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


class AvdfError(RuntimeError):
    pass


def check(rc, what):
    """Mirror of the reference's error behaviour: Python exceptions (the C ABI returns int codes)."""
    if rc != 0:
        msg = lib.avdf_last_error()
        raise AvdfError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")
