"""ctypes signatures of the C ABI declared in include/visiondk.h.

`bind(cdll)` attaches argtypes/restype for every exported entry point and returns the names it bound;
`tests/test_abi.py` checks this table against the header and the shared object's symbol table.
"""
from __future__ import annotations

import ctypes as C

P = C.c_void_p
I32 = C.c_int32
I64 = C.c_int64
F32 = C.c_float
SZ = C.c_size_t
PSZ = C.POINTER(C.c_size_t)



class GemmDesc(C.Structure):
    """VdkGemmDesc of include/visiondk.h"""
    _fields_ = [("A", P), ("lda", I64), ("B", P), ("ldb", I64), ("C", P), ("ldc", I64),
                ("M", I32), ("N", I32), ("K", I32), ("c_dtype", I32),
                ("bias", P), ("residual", P), ("ldr", I64), ("act", I32), ("aux", P), ("ldaux", I64),
                ("alpha", F32), ("splitk", I32), ("row_group", I32), ("trans", I32), ("a_row_group", I32), ("conv", P), ("a_colsum", P), ("c_colsum", P), ("ab_dtype", I32), ("col_scale", P), ("row_scale", P), ("rows_per_scale", I32)]


class ConvGeom(C.Structure):
    """VdkConvGeom of include/visiondk.h"""
    _fields_ = [("Cin", I32), ("H", I32), ("W", I32), ("OH", I32), ("OW", I32), ("KH", I32), ("KW", I32), ("stride", I32), ("pad", I32), ("transposed", I32), ("rows", I32)]


class VitConfig(C.Structure):
    """VdkVitConfig of include/visiondk.h"""
    _fields_ = [("batch", I32), ("img_size", I32), ("patch_size", I32), ("in_chans", I32), ("dim", I32), ("depth", I32),
                ("heads", I32), ("mlp_dim", I32), ("num_classes", I32), ("ln_eps", F32), ("no_class_token", I32), ("fp8", I32), ("fp8_w", P), ("fp8_state", P), ("operand", I32), ("pre_norm", I32)]


class GemmF32Desc(C.Structure):
    _fields_ = [("A", P), ("lda", I64), ("B", P), ("ldb", I64), ("C", P), ("ldc", I64), ("M", I32), ("N", I32), ("K", I32), ("bias", P), ("residual", P),
                ("ldr", I64), ("act", I32), ("alpha", C.c_float), ("col_scale", P), ("b_kmajor", I32), ("batch1", I32), ("batch2", I32), ("sa1", I64), ("sa2", I64),
                ("sb1", I64), ("sb2", I64), ("sc1", I64), ("sc2", I64), ("a_kmajor", I32), ("k_total", I64)]


class ConvNextConfig(C.Structure):
    _fields_ = [("batch", I32), ("img_size", I32), ("in_chans", I32), ("depths", I32 * 4), ("dims", I32 * 4), ("ln_eps", C.c_float), ("num_classes", I32), ("operand", I32)]


class ResNetConfig(C.Structure):
    _fields_ = [("batch", I32), ("img_size", I32), ("in_chans", I32), ("widths", I32 * 4), ("depths", I32 * 4), ("num_classes", I32), ("bn_eps", C.c_float),
                ("bn_momentum", C.c_float), ("mid", I32 * 4), ("stem_width", I32), ("operand_dtype", I32)]


class SwinConfig(C.Structure):
    """VdkSwinConfig of include/visiondk.h"""
    _fields_ = [("batch", I32), ("img_size", I32), ("in_chans", I32), ("embed_dim", I32), ("depths", I32 * 4), ("heads", I32 * 4), ("num_classes", I32), ("ln_eps", C.c_float),
                ("operand", I32), ("drop_path", P)]


class MarginHead(C.Structure):
    """VdkMarginHead of include/visiondk.h"""
    _fields_ = [("mode", I32), ("scale", F32), ("margin", F32), ("margin_am", F32), ("mv_weight", F32), ("row_margin", P)]

    def __deepcopy__(self, memo):      # ModelEMA deep-copies the model (models/ema.py:22) and the head module keeps one of these; row_margin is a per-call pointer
        return MarginHead(self.mode, self.scale, self.margin, self.margin_am, self.mv_weight, None)


F16_ = 2   # VDK_F16
EUNSUPPORTED = -4   # VDK_EUNSUPPORTED
HEAD_ARCFACE, HEAD_CIRCLE, HEAD_MV_AM, HEAD_MV_ARC = 0, 1, 2, 3
GRAD_READY_FN = C.CFUNCTYPE(None, P, I64, I64)
STAT_SYNC_FN = C.CFUNCTYPE(None, P, P, I64)   # vdk_stat_sync_fn(user, stats, n)

BF16, F32_ = 0, 1
ACT_NONE, ACT_GELU, ACT_DGELU, ACT_GELU_SAVE_GRAD, ACT_MUL_AUX = 0, 1, 2, 3, 4

# name -> (restype, [argtypes])
SIGNATURES: dict[str, tuple] = {
    "vdk_last_error": (C.c_char_p, []),
    "vdk_is_device_build": (C.c_int, []),
    "vdk_abi_version": (C.c_int, []),
    # hot path B
    "vdk_l2norm_rows": (C.c_int, [P, P, I64, I32, F32, P]),
    "vdk_cbir_workspace_bytes": (C.c_int, [I64, I32, I64, PSZ]),
    "vdk_cbir_search": (C.c_int, [P, I64, P, I64, I32, I32, I64, P, P, I64, P, SZ, P]),
    "vdk_cbir_prepare_gallery": (C.c_int, [P, I64, I32, P, P, P, P]),
    "vdk_cbir_fast_workspace_bytes": (C.c_int, [I64, I32, I64, PSZ]),
    "vdk_cbir_search_fast": (C.c_int, [P, I64, P, P, P, I64, I32, I32, I64, P, P, I64, P, SZ, P]),
    "vdk_cbir_fast2_workspace_bytes": (C.c_int, [I64, I32, I32, I64, PSZ]),
    "vdk_cbir_search_fast2": (C.c_int, [P, I64, P, I32, P, P, I64, I32, I32, I64, P, P, I64, I32, P, P, SZ, P]),
    "vdk_cbir_merge_topk": (C.c_int, [P, P, I32, I64, I32, P, P, P, SZ, P]),
    # hot path A: dense ops
    "vdk_gemm_splitk_workspace_bytes": (C.c_int, [I32, I32, I32, PSZ]),
    "vdk_gemm_streamk_workspace_bytes": (C.c_int, [PSZ]),
    "vdk_gemm_bf16_nt": (C.c_int, [C.POINTER(GemmDesc), P, SZ, P]),
    "vdk_gemm_a_colsum_rows": (C.c_int, [I32, I32, I32]),
    "vdk_gemm_c_colsum_rows": (C.c_int, [I32, I32, I32]),
    "vdk_gemm_reserve_cus": (C.c_int, [C.c_int32]),
    "vdk_gemm_reserved_cus": (C.c_int, []),
    "vdk_quant_fp8": (C.c_int, [P, I32, I64, P, P, I32, P, P]),
    "vdk_fp8_scale_update": (C.c_int, [P, P, P, I32, I32, F32, P]),
    "vdk_gemm_fp8_nt": (C.c_int, [P, I32, I32, P, P, P]),
    "vdk_gemm_fp8_nt_q8": (C.c_int, [P, I32, I32, P, P, P, I64, I32, P, P, P]),
    "vdk_prof_begin": (C.c_int, [I32]),
    "vdk_prof_pause": (C.c_int, [I32]),
    "vdk_prof_end": (C.c_int, [C.POINTER(C.c_double), C.POINTER(I64), C.POINTER(C.c_double)]),
    "vdk_prof_bytes": (C.c_int, [C.POINTER(C.c_double)]),
    "vdk_transpose_bf16": (C.c_int, [P, I64, I32, I32, P, I64, I32, I32, P, P]),
    "vdk_attention_fwd": (C.c_int, [P, I64, P, I64, P, I32, I32, I32, I32, F32, P]),
    "vdk_attention_bwd": (C.c_int, [P, I64, P, P, I64, P, P, I64, P, I32, I32, I32, I32, F32, P]),
    "vdk_attention_fwd_dt": (C.c_int, [P, I64, P, I64, P, I32, I32, I32, I32, F32, I32, P]),
    "vdk_attention_bwd_dt": (C.c_int, [P, I64, P, P, I64, P, P, I64, P, I32, I32, I32, I32, F32, I32, P]),
    "vdk_layernorm_fwd": (C.c_int, [P, I64, I32, I32, P, P, F32, P, I64, I32, P, P, P]),
    "vdk_layernorm_fwd_q8": (C.c_int, [P, I64, I32, I32, P, P, C.c_float, P, I64, P, P, P, I64, I32, P, P, P]),
    "vdk_layernorm_bwd_workspace_bytes": (C.c_int, [I32, I32, PSZ]),
    "vdk_layernorm_bwd": (C.c_int, [P, I64, I32, P, I64, P, P, P, P, I64, I32, I32, P, I64, P, I64, P, P, P, SZ, P]),
    "vdk_batchnorm1d_fwd": (C.c_int, [P, I64, I32, I32, P, P, F32, F32, I32, P, P, P, I64, P, P, P]),
    "vdk_batchnorm1d_bwd": (C.c_int, [P, I64, P, I64, I32, I32, P, P, P, P, I64, P, P, P]),
    "vdk_reduce_rows_f32": (C.c_int, [P, I64, I32, I64, P, F32, P]),
    "vdk_colsum_bf16_workspace_bytes": (C.c_int, [I32, I32, PSZ]),
    "vdk_colsum_bf16": (C.c_int, [P, I64, I32, I32, P, P, SZ, P]),
    "vdk_softmax_ce": (C.c_int, [P, I64, I32, I32, P, P, F32, F32, F32, P, P, I64, P, I64, P]),
    "vdk_bce_logits": (C.c_int, [P, I64, P, I64, I32, I32, F32, F32, F32, P, P, I64, P, I64, P]),
    "vdk_softmax_ce_amp": (C.c_int, [P, I64, I32, I32, P, P, F32, F32, F32, P, P, P, I64, I32, P, I64, P]),
    "vdk_bce_logits_amp": (C.c_int, [P, I64, P, I64, I32, I32, F32, P, F32, F32, P, P, I64, I32, P, I64, P]),
    "vdk_patchify_bf16": (C.c_int, [P, I32, I32, I32, I32, I32, P, I32, P]),
    "vdk_cls_rows": (C.c_int, [P, I64, I32, I32, P, P, P]),
    "vdk_cast_f32_bf16": (C.c_int, [P, P, I64, P]),
    "vdk_cast_f32_f16": (C.c_int, [P, P, I64, P]),
    "vdk_transpose_cast_f32_bf16": (C.c_int, [P, I64, I32, I32, P, I64, I32, P]),
    "vdk_sumsq_workspace_bytes": (C.c_int, [PSZ]),
    "vdk_sumsq_f32": (C.c_int, [P, I64, P, P, SZ, P]),
    "vdk_sgd_step": (C.c_int, [P, P, P, P, P, I64, F32, F32, F32, F32, P, F32, F32, I32, P]),
    "vdk_sgd_step_graph": (C.c_int, [P, P, P, P, P, I64, P, F32, P, F32, P]),
    "vdk_sgd_step_amp": (C.c_int, [P, P, P, P, P, I32, I64, F32, F32, F32, F32, P, P, F32, F32, I32, P]),
    "vdk_loss_scale_update": (C.c_int, [P, P, F32, F32, I32, P]),
    "vdk_mixup": (C.c_int, [P, P, F32, I32, I64, P, P]),
    "vdk_sam_first_step": (C.c_int, [P, P, P, I64, F32, I32, P, P, SZ, P]),
    "vdk_ohem_mask": (C.c_int, [P, I64, I32, I32, P, I32, F32, I64, P, P, P]),
    "vdk_topk_rows": (C.c_int, [P, I64, I32, I32, I32, P, P, P]),
    # collectives for hosts without a process group (csrc/comm.hip, RCCL by dlopen)
    "vdk_comm_unique_id": (C.c_int, [P]),
    "vdk_comm_init": (C.c_int, [P, I32, I32, C.POINTER(P)]),
    "vdk_comm_destroy": (C.c_int, [P]),
    "vdk_comm_rank": (C.c_int, [P]),
    "vdk_comm_world": (C.c_int, [P]),
    "vdk_allreduce_bucket": (C.c_int, [P, P, I64, I64, P]),
    "vdk_comm_finish": (C.c_int, [P, P]),
    "vdk_comm_trace": (C.c_int, [P, I32]),
    "vdk_comm_trace_close_last": (C.c_int, [P]),
    "vdk_comm_mark": (C.c_int, [P, P]),
    "vdk_comm_trace_read": (C.c_int, [P, P, P, I32, C.POINTER(I32), P, I32, C.POINTER(I32)]),
    "vdk_comm_stream": (C.c_void_p, [P]),
    "vdk_allgather": (C.c_int, [P, P, P, I64, P]),
    # margin-softmax heads
    "vdk_margin_cos_pass": (C.c_int, [P, I32, P, I64, P, I64, I32, I32, I32, I32, I32, P, P, P, P, P, F32, F32, P, I64, P]),
    "vdk_margin_rowstat": (C.c_int, [P, I64, P, I32, I32, F32, P, P, P]),
    "vdk_margin_target_cos_direct": (C.c_int, [P, I64, P, I64, I32, I32, P, P, P]),
    "vdk_attn_pool_fwd": (C.c_int, [P, P, I64, I32, I32, I32, F32, P, I64, P, P]),
    "vdk_attn_pool_bwd": (C.c_int, [P, P, I64, P, P, I64, I32, I32, I32, F32, P, I64, P, P]),
    "vdk_attn_pool_fwd_dt": (C.c_int, [P, P, I64, I32, I32, I32, F32, P, I64, P, I32, P]),
    "vdk_attn_pool_bwd_dt": (C.c_int, [P, P, I64, P, P, I64, I32, I32, I32, F32, P, I64, P, I32, P]),
    "vdk_colnorm_fwd": (C.c_int, [P, I64, I32, I32, I32, F32, P, P, I64, I32, P]),
    "vdk_colnorm_fwd_dt": (C.c_int, [P, I64, I32, I32, I32, F32, P, P, I64, I32, I32, P]),
    "vdk_rownorm_fwd_dt": (C.c_int, [P, I32, I32, I32, F32, P, P, P, P, I32, I32, P]),
    "vdk_colnorm_bwd": (C.c_int, [P, I64, P, P, I64, I32, I32, P, I64, P]),
    "vdk_rownorm_fwd": (C.c_int, [P, I32, I32, I32, F32, P, P, P, P, I32, P]),
    "vdk_rownorm_bwd": (C.c_int, [P, P, P, I64, I32, I32, P, P]),
    "vdk_margin_ce": (C.c_int, [C.POINTER(MarginHead), P, I64, I32, I32, P, F32, F32, P, I64, P, P, I64, P]),
    "vdk_margin_ce_amp": (C.c_int, [C.POINTER(MarginHead), P, I64, I32, I32, P, F32, F32, P, P, I64, P, P, I64, I32, P]),
    "vdk_margin_ce_f32": (C.c_int, [C.POINTER(MarginHead), P, I64, I32, I32, P, C.c_float, C.c_float, P, P, I64, P]),
    "vdk_margin_target_cos": (C.c_int, [P, I64, I32, I32, I64, P, P, P]),
    "vdk_margin_stats": (C.c_int, [C.POINTER(MarginHead), P, I64, I32, I32, I64, P, P, P, P]),
    "vdk_margin_grad": (C.c_int, [C.POINTER(MarginHead), P, I64, I32, I32, I64, I64, P, P, P, P, F32, F32, P, I64, P]),
    "vdk_margin_bwd": (C.c_int, [C.POINTER(MarginHead), P, I64, I32, I32, P, P, I64, P, I64, P]),
    # native ViT engine
    "vdk_gemm_f32_nt": (C.c_int, [C.POINTER(GemmF32Desc), P]),
    "vdk_softmax_rows_f32": (C.c_int, [P, I64, I64, I32, C.c_float, P]),
    "vdk_window_attention_fwd": (C.c_int, [P, I64, P, I64, P, P, P, I32, I64, I32, I32, I32, F32, P, P, SZ, P]),
    "vdk_window_attention_fwd_workspace_bytes": (C.c_int, [I32, I32, PSZ]),
    "vdk_relpos_bias_table_grad": (C.c_int, [P, P, I32, I32, I32, I32, P, P]),
    "vdk_window_attention_bwd_workspace_bytes": (C.c_int, [I64, I32, I32, PSZ]),
    "vdk_window_attention_bwd": (C.c_int, [P, I64, P, P, I64, P, P, P, I32, I64, I32, I32, I32, F32, P, P, I64, P, P, SZ, P]),
    "vdk_gelu_f32": (C.c_int, [P, P, I64, P]),
    "vdk_dgelu_f32": (C.c_int, [P, P, I64, P]),
    "vdk_rowscale_f32": (C.c_int, [P, P, P, I64, I64, P]),
    "vdk_colsum_f32_workspace_bytes": (C.c_int, [I64, I32, PSZ]),
    "vdk_colsum_f32": (C.c_int, [P, I64, I64, I32, P, P, SZ, P]),
    "vdk_depth_to_space2_f32": (C.c_int, [P, P, I32, I32, I32, I32, P]),
    "vdk_patchify_f32": (C.c_int, [P, I32, I32, I32, I32, I32, P, P]),
    "vdk_space_to_depth2_f32": (C.c_int, [P, P, I32, I32, I32, I32, P]),
    "vdk_dwconv7_fwd": (C.c_int, [P, P, P, P, P, P, I32, I32, I32, I32, I32, P]),
    "vdk_dwconv7_wgrad_workspace_bytes": (C.c_int, [I32, I32, I32, I32, PSZ]),
    "vdk_dwconv7_wgrad": (C.c_int, [P, P, P, P, I32, I32, I32, I32, P, SZ, P]),
    "vdk_dwconv7_weight_prep": (C.c_int, [P, P, I32, P]),
    "vdk_space_to_depth2_bf16": (C.c_int, [P, P, I32, I32, I32, I32, I32, P]),
    "vdk_conv2x2_weight_prep": (C.c_int, [P, P, P, I32, I32, P]),
    "vdk_conv2x2_wgrad_unpermute": (C.c_int, [P, P, I32, I32, P]),
    "vdk_layerscale_weight_prep": (C.c_int, [P, P, P, P, P, P, I32, I32, P]),
    "vdk_layerscale_grad": (C.c_int, [P, P, P, P, P, P, P, P, I32, I32, P]),
    "vdk_vit_param_count": (C.c_int, [C.POINTER(VitConfig), C.POINTER(I64), C.POINTER(I32), C.POINTER(I64)]),
    "vdk_vit_param_info": (C.c_int, [C.POINTER(VitConfig), I32, C.c_char_p, I32, C.POINTER(I64), C.POINTER(I64),
                                     C.POINTER(I64), C.POINTER(I32)]),
    "vdk_vit_fp8_update": (C.c_int, [C.POINTER(VitConfig), P]),
    "vdk_vit_workspace_bytes": (C.c_int, [C.POINTER(VitConfig), PSZ]),
    "vdk_vit_refresh_weights": (C.c_int, [C.POINTER(VitConfig), P, P, P, I32, P]),
    "vdk_vit_forward": (C.c_int, [C.POINTER(VitConfig), P, P, P, P, SZ, P, P]),
    "vdk_vit_backward": (C.c_int, [C.POINTER(VitConfig), P, P, P, P, P, SZ, P, P, P, P, P]),
    "vdk_swin_param_count": (C.c_int, [C.POINTER(SwinConfig), C.POINTER(I64), C.POINTER(I32), C.POINTER(I64)]),
    "vdk_swin_param_info": (C.c_int, [C.POINTER(SwinConfig), I32, C.c_char_p, I32, C.POINTER(I64), C.POINTER(I64), C.POINTER(I64), C.POINTER(I32)]),
    "vdk_swin_workspace_bytes": (C.c_int, [C.POINTER(SwinConfig), PSZ]),
    "vdk_swin_refresh_weights": (C.c_int, [C.POINTER(SwinConfig), P, P, P, I32, P]),
    "vdk_swin_forward": (C.c_int, [C.POINTER(SwinConfig), P, P, P, P, SZ, P, P]),
    "vdk_swin_backward": (C.c_int, [C.POINTER(SwinConfig), P, P, P, P, P, SZ, P, P, P, P]),
    "vdk_conv_weight_prep": (C.c_int, [P, P, P, I32, I32, I32, I32, I32, P]),
    "vdk_conv_wgrad_unpermute": (C.c_int, [P, P, I32, I32, I32, I32, I32, P]),
    "vdk_nchw_to_nhwc_bf16": (C.c_int, [P, P, I32, I32, I32, I32, I32, P]),
    "vdk_im2col_bf16": (C.c_int, [P, P, I32, I32, I32, I32, I32, I32, I32, I32, I32, I32, P]),
    "vdk_bn_rows_workspace_bytes": (C.c_int, [I64, I32, PSZ]),
    "vdk_bn_act_fwd": (C.c_int, [P, I64, I32, P, P, C.c_float, C.c_float, I32, P, P, P, P, I32, P, P, P, P, P, SZ, P, P, P]),
    "vdk_bn_act_bwd": (C.c_int, [P, P, P, I64, I32, P, P, P, P, P, P, P, P, SZ, P, P, P]),
    "vdk_bn_rows_bwd": (C.c_int, [P, P, I64, I32, P, P, P, P, P, P, P, SZ, P, P, P]),
    "vdk_maxpool3s2_fwd": (C.c_int, [P, P, P, I32, I32, I32, I32, P]),
    "vdk_maxpool3s2_bwd": (C.c_int, [P, P, P, P, I32, I32, I32, I32, P]),
    "vdk_avgpool_fwd": (C.c_int, [P, P, I32, I32, I32, I32, P]),
    "vdk_avgpool_bwd": (C.c_int, [P, I64, P, I32, I32, I32, P]),
    "vdk_avgpool_rows_f32_fwd": (C.c_int, [P, P, I32, I32, I32, P]),
    "vdk_scale_dev_f32": (C.c_int, [P, I64, P, I32, P]),
    "vdk_avgpool_rows_f32_bwd": (C.c_int, [P, P, P, I32, I32, I32, P]),
    "vdk_preprocess_workspace_bytes": (C.c_int, [I32, I32, I32, PSZ]),
    "vdk_preprocess_resize_pad_normalize": (C.c_int, [P, P, P, I32, I32, I32, F32, F32, F32, F32, F32, F32, P, P, P, SZ, P]),
    "vdk_vit_workspace_f32_bytes": (C.c_int, [C.POINTER(VitConfig), PSZ]),
    "vdk_vit_forward_f32": (C.c_int, [C.POINTER(VitConfig), P, P, P, SZ, P, P]),
    "vdk_convnext_workspace_f32_bytes": (C.c_int, [C.POINTER(ConvNextConfig), PSZ]),
    "vdk_convnext_forward_f32": (C.c_int, [C.POINTER(ConvNextConfig), P, P, P, P, SZ, P, P]),
    "vdk_convnext_train_f32_workspace_bytes": (C.c_int, [C.POINTER(ConvNextConfig), PSZ]),
    "vdk_convnext_forward_train_f32": (C.c_int, [C.POINTER(ConvNextConfig), P, P, P, P, SZ, P, P]),
    "vdk_convnext_backward_train_f32": (C.c_int, [C.POINTER(ConvNextConfig), P, P, P, P, SZ, P, P, P, P]),
    "vdk_resnet_ops_format": (C.c_int, [I32]),
    "vdk_resnet_param_count": (C.c_int, [C.POINTER(ResNetConfig), C.POINTER(I64), C.POINTER(I32), C.POINTER(I64), C.POINTER(I32), PSZ]),
    "vdk_resnet_param_info": (C.c_int, [C.POINTER(ResNetConfig), I32, I32, C.c_char_p, I32, C.POINTER(I64), C.POINTER(I64), C.POINTER(I64), C.POINTER(I32)]),
    "vdk_resnet_workspace_bytes": (C.c_int, [C.POINTER(ResNetConfig), PSZ]),
    "vdk_resnet_refresh_weights": (C.c_int, [C.POINTER(ResNetConfig), P, P, P, I32, P]),
    "vdk_resnet_forward": (C.c_int, [C.POINTER(ResNetConfig), P, P, P, P, P, I32, P, SZ, P, P, P, P]),
    "vdk_resnet_backward": (C.c_int, [C.POINTER(ResNetConfig), P, P, P, P, P, SZ, P, P, P, P, P, P]),
    "vdk_convnext_param_count": (C.c_int, [C.POINTER(ConvNextConfig), C.POINTER(I64), C.POINTER(I32), PSZ]),
    "vdk_convnext_param_info": (C.c_int, [C.POINTER(ConvNextConfig), I32, C.c_char_p, I32, C.POINTER(I64), C.POINTER(I64), C.POINTER(I64), C.POINTER(I32)]),
    "vdk_convnext_workspace_bytes": (C.c_int, [C.POINTER(ConvNextConfig), PSZ]),
    "vdk_convnext_refresh_weights": (C.c_int, [C.POINTER(ConvNextConfig), P, P, P, I32, P]),
    "vdk_convnext_forward": (C.c_int, [C.POINTER(ConvNextConfig), P, P, P, P, P, SZ, P, P]),
    "vdk_convnext_backward": (C.c_int, [C.POINTER(ConvNextConfig), P, P, P, P, P, SZ, P, P, P, P]),
}


class VdkError(RuntimeError):
    pass


# tuning / debug / test knobs (csrc/vdk_internal.h): exported by the library, not part of the ABI a host binds
INTERNAL = {
    "vdk_gemm_streamk_grid": (C.c_int, [I32]),
    "vdk_gemm_force_kernel": (C.c_int, [I32]),
    "vdk_gemm_force_band_cw": (C.c_int, [I32]),
    "vdk_gemm_last_kernel": (C.c_int, []),
    "vdk_debug_occupy_cus": (C.c_int, [C.c_int32, C.c_int64, C.c_void_p]),
    "vdk_gemm_debug_stamps": (C.c_int, [P]),
    "vdk_attention_force_legacy": (C.c_int, [I32]),
    "vdk_debug_reduce_rows_job": (C.c_int, [P, I64, I32, I64, P, F32, P]),
}


def bind(lib: C.CDLL) -> list[str]:
    bound = []
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
        bound.append(name)
    for name, (res, args) in INTERNAL.items():      # (knobs: bound when present)
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype = res
            fn.argtypes = args
    return bound


def check(lib: C.CDLL, rc: int, what: str) -> None:
    if rc != 0:
        msg = lib.vdk_last_error()
        raise VdkError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")
