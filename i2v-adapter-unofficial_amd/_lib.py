"""ctypes binding of libi2v_hip.so (C ABI declared in include/i2v_hip.h).

There is NO fallback: if the shared library is missing, or a kernel reports an error, the caller gets an
exception.  Build the library with `python __graft_entry__.py build` (hipcc --offload-arch=gfx950).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# I2V_LIB_PATH selects another build of the same ABI (same-box A/B of two kernels, tools/ab_bench.sh); the in-tree
# library is never overwritten by tooling
LIB_PATH = os.environ.get("I2V_LIB_PATH") or os.path.join(_HERE, "libi2v_hip.so")
ABI_VERSION = 9

I2V_EPI_NONE, I2V_EPI_GELU, I2V_EPI_GEGLU = 0, 1, 2
I2V_STORE_ROWMAJOR, I2V_STORE_ROWPERM, I2V_STORE_VT, I2V_STORE_VT_T = 0, 1, 2, 3
I2V_A_PLAIN, I2V_A_CONV3X3 = 0, 1


class HipLibraryError(RuntimeError):
    pass


class GemmParams(C.Structure):
    _fields_ = [
        ("a", C.c_void_p), ("lda", C.c_int64),
        ("a2", C.c_void_p), ("lda2", C.c_int64),
        ("k_split", C.c_int32), ("a_mode", C.c_int32),
        ("w", C.c_void_p), ("ldw", C.c_int64),
        ("bias", C.c_void_p),
        ("residual", C.c_void_p), ("ldr", C.c_int64),
        ("rowvec", C.c_void_p), ("ld_rowvec", C.c_int64), ("rows_per_vec", C.c_int32), ("rowvec_period", C.c_int32),
        ("ln_wsum", C.c_void_p), ("ln_eps", C.c_float),
        ("c", C.c_void_p), ("ldc", C.c_int64),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("epilogue", C.c_int32), ("store_mode", C.c_int32),
        ("frames", C.c_int32), ("hw", C.c_int32),
        ("vt_len", C.c_int32), ("vt_ld", C.c_int32),
        ("out_scale", C.c_float),
        ("n_img", C.c_int32), ("in_h", C.c_int32), ("in_w", C.c_int32), ("cin", C.c_int32),
        ("out_h", C.c_int32), ("out_w", C.c_int32), ("stride", C.c_int32), ("upsample", C.c_int32),
        ("asym_pad", C.c_int32), ("conv_kblock", C.c_int32),
        ("w_batch_stride", C.c_int64), ("rows_per_w", C.c_int32),
        ("a_perm_frames", C.c_int32), ("a_perm_hw", C.c_int32),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
        ("c_is_f32", C.c_int32),
        ("gn_partial", C.c_void_p), ("gn_groups", C.c_int32),
        ("residual_lo", C.c_void_p), ("c_lo", C.c_void_p),
    ]


class AttnParams(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("q_row_stride", C.c_int64), ("q_batch_stride", C.c_int64),
        ("k", C.c_void_p), ("k_row_stride", C.c_int64), ("k_batch_stride", C.c_int64),
        ("vt", C.c_void_p), ("vt_row_stride", C.c_int64), ("vt_batch_stride", C.c_int64),
        ("o", C.c_void_p), ("o_row_stride", C.c_int64), ("o_batch_stride", C.c_int64),
        ("batch_q", C.c_int32), ("kv_group", C.c_int32), ("heads", C.c_int32), ("head_dim", C.c_int32),
        ("lq", C.c_int32), ("lk", C.c_int32),
        ("scale", C.c_float), ("accumulate", C.c_int32), ("acc_scale", C.c_float),
        ("lse", C.c_void_p),
    ]


class AttnBwdParams(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("q_row_stride", C.c_int64), ("q_batch_stride", C.c_int64),
        ("qt", C.c_void_p), ("qt_row_stride", C.c_int64), ("qt_batch_stride", C.c_int64),
        ("k", C.c_void_p), ("k_row_stride", C.c_int64), ("k_batch_stride", C.c_int64),
        ("kt", C.c_void_p), ("kt_row_stride", C.c_int64), ("kt_batch_stride", C.c_int64),
        ("v", C.c_void_p), ("v_row_stride", C.c_int64), ("v_batch_stride", C.c_int64),
        ("dout", C.c_void_p), ("do_row_stride", C.c_int64), ("do_batch_stride", C.c_int64),
        ("doutt", C.c_void_p), ("dot_row_stride", C.c_int64), ("dot_batch_stride", C.c_int64),
        ("lse", C.c_void_p), ("delta", C.c_void_p),
        ("dq", C.c_void_p), ("dq_row_stride", C.c_int64), ("dq_batch_stride", C.c_int64),
        ("dk", C.c_void_p), ("dk_row_stride", C.c_int64), ("dk_batch_stride", C.c_int64),
        ("dv", C.c_void_p), ("dv_row_stride", C.c_int64), ("dv_batch_stride", C.c_int64),
        ("batch_q", C.c_int32), ("kv_group", C.c_int32), ("heads", C.c_int32), ("head_dim", C.c_int32),
        ("lq", C.c_int32), ("lk", C.c_int32),
        ("scale", C.c_float), ("kv_partitions", C.c_int32), ("dkv_partial", C.c_void_p),
    ]


class TAttnParams(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("q_row_stride", C.c_int64),
        ("k", C.c_void_p), ("k_row_stride", C.c_int64),
        ("vt", C.c_void_p), ("vt_ld", C.c_int32),
        ("o", C.c_void_p), ("o_row_stride", C.c_int64),
        ("n_pixels", C.c_int32), ("frames", C.c_int32), ("heads", C.c_int32), ("head_dim", C.c_int32),
        ("scale", C.c_float),
    ]


class MotionAttnParams(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("ldx", C.c_int64),
        ("gamma", C.c_void_p),
        ("shift", C.c_void_p), ("ld_shift", C.c_int64),
        ("w_qkv", C.c_void_p),
        ("out", C.c_void_p), ("ldo", C.c_int64),
        ("rows", C.c_int64),
        ("channels", C.c_int32), ("heads", C.c_int32), ("head_dim", C.c_int32), ("frames", C.c_int32),
        ("eps", C.c_float), ("scale", C.c_float),
        ("w_o", C.c_void_p), ("b_o", C.c_void_p),
    ]


class CrossAttnFusedParams(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("ldx", C.c_int64),
        ("gamma", C.c_void_p), ("beta", C.c_void_p),
        ("w_q", C.c_void_p),
        ("ctx_frag", C.c_void_p),
        ("out", C.c_void_p), ("ldo", C.c_int64),
        ("rows", C.c_int64), ("rows_per_ctx", C.c_int64),
        ("channels", C.c_int32), ("heads", C.c_int32), ("head_dim", C.c_int32), ("ctx_len", C.c_int32),
        ("eps", C.c_float), ("scale", C.c_float),
        ("ip_frag", C.c_void_p), ("ip_len", C.c_int32), ("ip_scale", C.c_float),
        ("w_o", C.c_void_p), ("b_o", C.c_void_p),
    ]


class FfFusedParams(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("ldx", C.c_int64),
        ("gamma", C.c_void_p), ("beta", C.c_void_p),
        ("w1", C.c_void_p), ("b1", C.c_void_p),
        ("w2", C.c_void_p), ("b2", C.c_void_p),
        ("out", C.c_void_p), ("ldo", C.c_int64),
        ("rows", C.c_int64),
        ("channels", C.c_int32), ("inner", C.c_int32),
        ("eps", C.c_float),
        ("w3", C.c_void_p), ("b3", C.c_void_p),
        ("res2", C.c_void_p), ("ld_res2", C.c_int64),
        ("perm_frames", C.c_int32), ("perm_hw", C.c_int32),
        ("res2_lo", C.c_void_p), ("out_lo", C.c_void_p),
    ]


class LnQkvParams(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("ldx", C.c_int64),
        ("gamma", C.c_void_p), ("beta", C.c_void_p),
        ("w", C.c_void_p),
        ("qk", C.c_void_p), ("ld_qk", C.c_int64),
        ("vt", C.c_void_p), ("vt_batch_stride", C.c_int64), ("vt_row_stride", C.c_int64),
        ("rows", C.c_int64), ("rows_per_image", C.c_int64),
        ("channels", C.c_int32), ("n_qk", C.c_int32),
        ("eps", C.c_float),
        ("x_image_stride", C.c_int64),
    ]


class UnetConfig(C.Structure):
    _fields_ = [
        ("in_channels", C.c_int32), ("out_channels", C.c_int32),
        ("block_out_channels", C.c_int32 * 4),
        ("layers_per_block", C.c_int32), ("num_attention_heads", C.c_int32),
        ("cross_attention_dim", C.c_int32), ("norm_num_groups", C.c_int32),
        ("motion_max_seq_length", C.c_int32), ("motion_num_attention_heads", C.c_int32),
        ("use_motion_mid_block", C.c_int32), ("ip_num_tokens", C.c_int32),
    ]


class UnetPlan(C.Structure):
    _fields_ = [
        ("batch", C.c_int32), ("frames", C.c_int32), ("height", C.c_int32), ("width", C.c_int32),
        ("ctx_len", C.c_int32), ("has_ip", C.c_int32),
    ]


class GnParams(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("c1", C.c_int32),
        ("x2", C.c_void_p), ("c2", C.c_int32),
        ("gamma", C.c_void_p), ("beta", C.c_void_p),
        ("y", C.c_void_p),
        ("n_img", C.c_int32), ("hw", C.c_int32), ("groups", C.c_int32), ("frames_per_stat", C.c_int32),
        ("eps", C.c_float), ("silu", C.c_int32),
        ("out_perm", C.c_int32), ("frames", C.c_int32),
        ("workspace", C.c_void_p),
        ("gpartial_in", C.c_void_p), ("gpartial_rows", C.c_int32),
    ]


class LnParams(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("ldx", C.c_int64),
        ("gamma", C.c_void_p), ("beta", C.c_void_p),
        ("pe", C.c_void_p), ("ld_pe", C.c_int64), ("pe_period", C.c_int32),
        ("y", C.c_void_p), ("ldy", C.c_int64),
        ("rows", C.c_int32), ("C", C.c_int32),
        ("eps", C.c_float),
        ("x_rows_per_batch", C.c_int32), ("x_batch_stride", C.c_int64),
    ]


# every symbol include/i2v_hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SIGNATURES = {
    "i2v_abi_version": (C.c_int, []),
    "i2v_last_error": (C.c_char_p, []),
    "i2v_gemm_f16": (C.c_int, [C.POINTER(GemmParams), _P]),
    "i2v_gemm_ln_supported": (C.c_int, [C.POINTER(GemmParams)]),
    "i2v_gemm_batch_supported": (C.c_int, [C.POINTER(GemmParams)]),
    "i2v_gemm_workspace_bytes": (C.c_int64, [C.POINTER(GemmParams)]),
    "i2v_gemm_gn_partial_rows": (C.c_int32, [C.POINTER(GemmParams)]),
    "i2v_attention_f16": (C.c_int, [C.POINTER(AttnParams), _P]),
    "i2v_temporal_attention_f16": (C.c_int, [C.POINTER(TAttnParams), _P]),
    "i2v_motion_attn_supported": (C.c_int32, [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "i2v_motion_attn_pack_rows": (C.c_int32, [C.c_int32, C.c_int32]),
    "i2v_motion_attn_f16": (C.c_int, [C.POINTER(MotionAttnParams), _P]),
    "i2v_cross_attn_fused_supported": (C.c_int32, [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64]),
    "i2v_cross_attn_fused_pack_rows": (C.c_int32, [C.c_int32, C.c_int32]),
    "i2v_cross_attn_fused_ctx_elems": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "i2v_cross_attn_fused_f16": (C.c_int, [C.POINTER(CrossAttnFusedParams), _P]),
    "i2v_ln_qkv_supported": (C.c_int32, [C.c_int64, C.c_int32, C.c_int32, C.c_int64]),
    "i2v_ln_qkv_f16": (C.c_int, [C.POINTER(LnQkvParams), _P]),
    "i2v_ff_fused_supported": (C.c_int32, [C.c_int64, C.c_int32, C.c_int32]),
    "i2v_ff_fused_tail_supported": (C.c_int32, [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "i2v_ff_fused_f16": (C.c_int, [C.POINTER(FfFusedParams), _P]),
    "i2v_groupnorm_workspace_bytes": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "i2v_groupnorm_f16": (C.c_int, [C.POINTER(GnParams), _P]),
    "i2v_groupnorm_fold_f16": (C.c_int, [C.POINTER(GnParams), _P, C.c_int64, _P, C.c_int32, _P, _P, _P]),
    "i2v_layernorm_f16": (C.c_int, [C.POINTER(LnParams), _P]),
    "i2v_softmax_rows_f16": (C.c_int, [_P, C.c_int64, _P, C.c_int64, C.c_int32, C.c_int32, C.c_float, _P]),
    "i2v_nchw_to_tokens": (C.c_int, [_P, C.c_int32, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "i2v_tokens_to_nchw": (C.c_int, [_P, C.c_int32, C.c_int64, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "i2v_timestep_embedding": (C.c_int, [_P, _P, C.c_int32, _P, C.c_int32, C.c_int32, _P]),
    "i2v_silu_f16": (C.c_int, [_P, _P, C.c_int64, _P]),
    "i2v_repeat_rows_f16": (C.c_int, [_P, _P, C.c_int64, C.c_int64, C.c_int32, _P]),
    "i2v_copy3d_f16": (C.c_int, [_P, C.c_int64, C.c_int64, _P, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                 C.c_int64, _P]),
    "i2v_pack_ctx_fragments_elems": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "i2v_pack_ctx_fragments_f16": (C.c_int, [_P, C.c_int64, _P, C.c_int64, C.c_int64, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "i2v_ddim_prep": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "i2v_gaussian_sample_f32": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P]),
    "i2v_first_frame_prior_f32": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                            C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _P]),
    "i2v_attention_lse_f32": (C.c_int, [C.POINTER(AttnParams), _P, _P]),
    "i2v_attention_bwd_f16": (C.c_int, [C.POINTER(AttnBwdParams), _P]),
    "i2v_transpose_f16": (C.c_int, [_P, C.c_int64, C.c_int64, _P, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _P]),
    "i2v_rowdot_heads_f32": (C.c_int, [_P, C.c_int64, _P, C.c_int64, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _P]),
    "i2v_layernorm_bwd_f16": (C.c_int, [_P, C.c_int64, _P, C.c_int64, _P, _P, C.c_int64, _P, C.c_int64, C.c_int32, C.c_int32,
                                        C.c_float, _P]),
    "i2v_geglu_bwd_f16": (C.c_int, [_P, C.c_int64, _P, C.c_int64, _P, C.c_int64, C.c_int64, C.c_int32, _P]),
    "i2v_geglu_f16": (C.c_int, [_P, C.c_int64, _P, C.c_int64, C.c_int64, C.c_int32, _P]),
    "i2v_colsum_f32": (C.c_int, [_P, C.c_int64, _P, C.c_int64, C.c_int32, _P]),
    "i2v_colsum_prod_f32": (C.c_int, [_P, C.c_int64, _P, C.c_int64, _P, C.c_int64, C.c_int32, _P]),
    "i2v_colsum_workspace_bytes": (C.c_int64, [C.c_int64, C.c_int32]),
    "i2v_colsum_det_f32": (C.c_int, [_P, C.c_int64, _P, C.c_int64, _P, C.c_int64, C.c_int32, _P, _P]),
    "i2v_masked_mse_grad_f16": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_float, _P]),
    "i2v_masked_mse_grad_f32": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_float, _P]),
    "i2v_groupnorm_bwd_workspace_bytes": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "i2v_groupnorm_bwd_f16": (C.c_int, [C.POINTER(GnParams), _P, _P, _P, _P]),
    "i2v_add_f16": (C.c_int, [_P, _P, _P, C.c_int64, _P]),
    "i2v_permute_rows_f16": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "i2v_zero_insert2x_f16": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _P]),
    "i2v_sum_pool2x_f16": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _P]),
    "i2v_sumsq_f32": (C.c_int, [_P, C.c_int64, _P, _P]),
    "i2v_adamw_f32": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int32,
                                C.c_float, _P, C.c_float, _P]),
    "i2v_adamw_guarded_f32": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                        C.c_float, C.c_float, _P, C.c_int32, _P, _P, _P, _P]),
    "i2v_axpby_f32": (C.c_int, [_P, _P, C.c_float, C.c_float, C.c_int64, _P]),
    "i2v_select_row_f16": (C.c_int, [_P, C.c_int64, C.c_int32, _P, _P, C.c_int32, _P]),
    "i2v_ddim_cfg_step": (C.c_int, [_P, _P, C.c_int32, C.c_int64, _P, C.c_int32, _P, C.c_float, C.c_int32, C.c_int32,
                                    C.c_int32, C.c_int32, C.c_int32, _P]),
    # the model handle: configuration, weight registry, plan, one captured step
    "i2v_unet_create": (C.c_int, [C.POINTER(UnetConfig), C.POINTER(_P)]),
    "i2v_unet_destroy": (C.c_int, [_P]),
    "i2v_unet_set_weight": (C.c_int, [_P, C.c_char_p, _P, C.c_int32, C.c_int32, C.POINTER(C.c_int64)]),
    "i2v_unet_get_weight": (C.c_int, [_P, C.c_char_p, C.POINTER(_P), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                      C.POINTER(C.c_int64)]),
    "i2v_unet_num_weights": (C.c_int64, [_P]),
    "i2v_unet_plan": (C.c_int, [_P, C.POINTER(UnetPlan)]),
    "i2v_unet_set_plan": (C.c_int, [_P, C.c_char_p, C.c_int64]),
    "i2v_unet_activation_bytes": (C.c_int64, [_P]),
    "i2v_unet_plan_launches": (C.c_int32, [_P]),
    "i2v_unet_plan_num_keys": (C.c_int32, [_P]),
    "i2v_unet_plan_key": (C.c_char_p, [_P, C.c_int32]),
    "i2v_unet_set_workspace": (C.c_int, [_P, _P, C.c_int64]),
    "i2v_unet_forward": (C.c_int, [_P, _P, _P, _P, _P, _P, _P]),
    "i2v_unet_abort_capture": (C.c_int, [_P]),
    "i2v_unet_run": (C.c_int, [_P, C.POINTER(_P), C.c_int32, _P]),
    "i2v_unet_capture_step": (C.c_int, [_P, _P]),
    "i2v_unet_end_capture": (C.c_int, [_P]),
    "i2v_unet_replay_step": (C.c_int, [_P, _P]),
    "i2v_unet_has_step": (C.c_int32, [_P]),
}

_lib = None


def load():
    """Load libi2v_hip.so and bind every exported symbol.  Raises HipLibraryError when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} not found: the HIP extension is not built.  Run `python __graft_entry__.py build` "
            "(needs hipcc, cross-compiles gfx950 without a GPU).  There is no CPU fallback for this path.")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise HipLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipLibraryError(f"{LIB_PATH} does not export `{name}` declared in include/i2v_hip.h") from e
        fn.restype = res
        fn.argtypes = args
    v = lib.i2v_abi_version()
    if v != ABI_VERSION:
        raise HipLibraryError(f"{LIB_PATH} has ABI version {v}, binding expects {ABI_VERSION}")
    _lib = lib
    return lib


def check(status: int, what: str):
    if status != 0:
        msg = load().i2v_last_error()
        raise HipLibraryError(f"{what} failed with status {status}: {msg.decode() if msg else ''}")
