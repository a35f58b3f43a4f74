"""ctypes binding of libvtgb.so (include/vtgb.h).  There is no fallback: if the HIP library is
missing or a call fails, the product path raises."""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  -- FIRST: libvtgb.so must bind to the HIP runtime PyTorch has loaded (one runtime,
#                             one device context per process); loading it before torch gives it a second one

HERE = os.path.dirname(os.path.abspath(__file__))
# VTGB_LIB: another build of the same library (same-box A/B runs: tools/ab_*.sh) -- the product path never sets it
LIB_PATH = os.environ.get("VTGB_LIB") or os.path.join(HERE, "libvtgb.so")

F32, BF16 = 0, 1
BF16X3 = 2      # RAFT entry points only: split-bf16 operands (include/vtgb.h)
F16C8 = 3       # vtgb_raft_update only: fp16 main product + two fp8 correction products (include/vtgb.h)
MAP_A, MAP_B = 0, 1
POOL_MEAN, POOL_CONCAT = 0, 1
TGB_MODE = {"text": 0, "vision": 0, "fusion": 1, "multi_modal": 2}
EPI_STORE, EPI_GELU, EPI_RESID_F32, EPI_STORE_F32 = 0, 1, 2, 3
VIT_NW_GLOBAL, VIT_NW_LAYER = 6, 18
QF_NW_GLOBAL, QF_NW_LAYER = 4, 32
TGB_NW_GLOBAL, TGB_NW_LAYER = 17, 26

EXPORTS = [
    "vtgb_version", "vtgb_last_error", "vtgb_pack_bf16", "vtgb_span_select", "vtgb_span_to_frames",
    "vtgb_gather_frames", "vtgb_vit_patch_kpad", "vtgb_vit_workspace_bytes", "vtgb_vit_forward",
    "vtgb_qformer_workspace_bytes", "vtgb_qformer_forward", "vtgb_pool_project_workspace_bytes",
    "vtgb_pool_project", "vtgb_tgb_workspace_bytes", "vtgb_tgb_forward", "vtgb_gemm", "vtgb_attention",
    "vtgb_layernorm", "vtgb_prof_enable", "vtgb_prof_reset", "vtgb_prof_summary", "vtgb_prof_executed_flops",
    "vtgb_llm_rmsnorm", "vtgb_llm_rope_cache", "vtgb_llm_rope_cache_prefill", "vtgb_llm_decode_attention", "vtgb_llm_silu_mul",
    "vtgb_llm_attention_rows", "vtgb_llm_gated_act", "vtgb_llm_rmsnorm_parts", "vtgb_llm_rope_cache_parts", "vtgb_gemm_skinny_splits",
    "vtgb_gemm_skinny_workspace_bytes", "vtgb_gemm_skinny", "vtgb_pack_skinny_weight_bytes", "vtgb_pack_skinny_weight",
    "vtgb_raft_update_workspace_bytes", "vtgb_raft_update", "vtgb_raft_encoder_workspace_bytes", "vtgb_raft_encoder",
    "vtgb_raft_corr_workspace_bytes", "vtgb_raft_corr", "vtgb_preprocess_frames", "vtgb_concat_text_io", "vtgb_shifted_ce_forward", "vtgb_shifted_ce_backward",
    "vtgb_comm_unique_id", "vtgb_comm_init", "vtgb_comm_destroy", "vtgb_allreduce_f32",
    "vtgb_attn_train_forward", "vtgb_attn_train_backward",
    "vtgb_gemm_train", "vtgb_gemm_train_workspace_bytes", "vtgb_col_sum_parts", "vtgb_col_sum_f32", "vtgb_layernorm_train_partials", "vtgb_layernorm_train_forward", "vtgb_layernorm_train_backward",
    "vtgb_gelu_forward", "vtgb_gelu_backward",
    "vtgb_pair_pack", "vtgb_pair_conv", "vtgb_pair_conv_ex",
]
COMM_ID_BYTES = 128

i32, i64, f32, vp, sz = C.c_int32, C.c_int64, C.c_float, C.c_void_p, C.c_size_t


class AttnTrainArgs(C.Structure):
    _fields_ = [("batch", i32), ("heads", i32), ("head_dim", i32), ("s_q", i32), ("s_kv", i32), ("q", vp), ("k", vp), ("v", vp),
                ("q_tok", i64), ("kv_tok", i64), ("q_batch", i64), ("kv_batch", i64), ("key_mask", vp), ("drop", vp), ("scale", f32),
                ("out", vp), ("o_tok", i64), ("o_batch", i64), ("lse", vp), ("dout", vp), ("dq", vp), ("dk", vp), ("dv", vp), ("delta", vp)]


class GemmTrainArgs(C.Structure):
    _fields_ = [("compute", i32), ("M", i32), ("N", i32), ("K", i32), ("a", vp), ("lda", i64), ("a_dtype", i32), ("a_kmajor", i32),
                ("b", vp), ("ldb", i64), ("b_dtype", i32), ("b_kmajor", i32), ("bias", vp), ("out", vp), ("ldo", i64), ("workspace", vp), ("workspace_bytes", sz)]


class LayerNormTrainArgs(C.Structure):
    _fields_ = [("rows", i32), ("D", i32), ("eps", f32), ("x", vp), ("mask", vp), ("resid", vp), ("gamma", vp), ("beta", vp), ("sum", vp),
                ("y", vp), ("mean", vp), ("rstd", vp), ("dy", vp), ("ds", vp), ("dx", vp), ("dgamma", vp), ("dbeta", vp), ("partial", vp)]


class VtgbError(RuntimeError):
    pass


class SpanSelectArgs(C.Structure):
    _fields_ = [("logits", vp), ("noise", vp), ("idx", vp), ("B", i32), ("L", i32), ("draws", i32), ("tau", f32)]


class SpanToFramesArgs(C.Structure):
    _fields_ = [("sel", vp), ("V", vp), ("frame_idx", vp), ("B", i32), ("draws", i32), ("V_all", i32), ("N", i32),
                ("nframe", i32), ("variant", i32)]


class GatherFramesArgs(C.Structure):
    _fields_ = [("pixel_values", vp), ("frame_idx", vp), ("out", vp), ("B", i32), ("N", i32), ("nframe", i32),
                ("frame_elems", i64)]


class PreprocessArgs(C.Structure):
    _fields_ = [("raw", vp), ("frame_idx", vp), ("out", vp), ("T", i32), ("H0", i32), ("W0", i32), ("n_out", i32), ("size", i32),
                ("mean", f32 * 3), ("std", f32 * 3)]


class ConcatTextIoArgs(C.Structure):
    _fields_ = [("input_ids", vp), ("input_atts", vp), ("output_ids", vp), ("output_atts", vp), ("llm_ids", vp), ("llm_atts", vp),
                ("input_len", vp), ("labels", vp), ("pad_id", i64), ("B", i32), ("Li", i32), ("Lo", i32), ("prefix_len", i32)]


class ShiftedCeArgs(C.Structure):
    _fields_ = [("dtype", i32), ("B", i32), ("S", i32), ("V", i32), ("logits", vp), ("labels", vp), ("lse", vp), ("row_loss", vp),
                ("loss", vp), ("grad_out", vp), ("dlogits", vp)]


class VitArgs(C.Structure):
    _fields_ = [("dtype", i32), ("n_frames", i32), ("image", i32), ("patch", i32), ("hidden", i32), ("heads", i32),
                ("mlp", i32), ("layers", i32), ("eps", f32), ("pixel_values", vp), ("weights", C.POINTER(vp)),
                ("out_f32", vp), ("out_act", vp), ("workspace", vp), ("workspace_bytes", sz)]


class QFormerArgs(C.Structure):
    _fields_ = [("dtype", i32), ("n_frames", i32), ("n_query", i32), ("n_text", i32), ("hidden", i32), ("heads", i32),
                ("ffn", i32), ("layers", i32), ("cross_freq", i32), ("enc_tokens", i32), ("enc_hidden", i32),
                ("has_text", i32), ("eps", f32), ("image_embeds", vp), ("query_tokens", vp), ("text_ids", vp),
                ("text_mask", vp), ("image_mask", vp), ("weights", C.POINTER(vp)), ("out_f32", vp), ("workspace", vp),
                ("workspace_bytes", sz)]


class PoolProjectArgs(C.Structure):
    _fields_ = [("dtype", i32), ("n_clips", i32), ("n_query", i32), ("hidden", i32), ("out_dim", i32), ("mode", i32),
                ("query_out", vp), ("widths", C.POINTER(i32)), ("proj_w", vp), ("proj_b", vp), ("out", vp),
                ("workspace", vp), ("workspace_bytes", sz)]


class TgbArgs(C.Structure):
    _fields_ = [("dtype", i32), ("B", i32), ("L", i32), ("n_text", i32), ("hidden", i32), ("heads", i32), ("ffn", i32),
                ("layers", i32), ("fusion_layer", i32), ("mode", i32), ("image", i32), ("patch", i32), ("eps", f32),
                ("of", vp), ("of_mask", vp), ("text_ids", vp), ("text_mask", vp), ("weights", C.POINTER(vp)),
                ("seq_out", vp), ("logits", vp), ("workspace", vp), ("workspace_bytes", sz)]


class RaftUpdateArgs(C.Structure):
    _fields_ = [("dtype", i32), ("n_pairs", i32), ("H8", i32), ("W8", i32), ("iters", i32), ("net", vp), ("inp", vp), ("corr", vp * 4),
                ("weights", C.POINTER(vp)), ("flow_up", vp), ("workspace", vp), ("workspace_bytes", sz), ("corr_f16", i32), ("cnet_nhwc", vp),
                ("flow_init", vp)]


class PairConvArgs(C.Structure):
    _fields_ = [("M", i32), ("N", i32), ("H", i32), ("W", i32), ("KH", i32), ("KW", i32), ("C1", i32), ("a", vp), ("a2", vp), ("weights", vp), ("scale", vp),
                ("bias", vp), ("act", i32), ("out_fmt", i32), ("out", vp), ("ld_out", i32)]


class PairConvExArgs(C.Structure):
    _fields_ = [("conv", PairConvArgs), ("resid", vp), ("ld_resid", i32), ("tail_w", vp), ("tail_out", vp), ("out_f32", vp), ("ld_f32", i32)]


class RaftCorrArgs(C.Structure):
    _fields_ = [("dtype", i32), ("n_pairs", i32), ("H8", i32), ("W8", i32), ("dim", i32), ("pairs_per_clip", i32), ("frames_per_clip", i32),
                ("first_off", i32), ("second_off", i32), ("n_images", i32), ("scale", f32), ("fmap", vp), ("levels", vp * 4),
                ("workspace", vp), ("workspace_bytes", sz)]


class RaftEncoderArgs(C.Structure):
    _fields_ = [("dtype", i32), ("n_images", i32), ("H", i32), ("W", i32), ("norm", i32), ("images", vp), ("weights", C.POINTER(vp)), ("out", vp),
                ("workspace", vp), ("workspace_bytes", sz)]


class GemmSkinnyArgs(C.Structure):
    _fields_ = [("M", i32), ("N", i32), ("K", i32), ("n_splits", i32), ("x", vp), ("ldx", i64), ("w", vp), ("ldw", i64), ("out", vp), ("ldo", i64),
                ("out_dtype", i32), ("w_tiled", i32), ("workspace", vp), ("workspace_bytes", sz), ("defer_reduce", i32)]


class GemmArgs(C.Structure):
    _fields_ = [("dtype", i32), ("M", i32), ("N", i32), ("K", i32), ("epilogue", i32), ("A", vp), ("lda", i64),
                ("W", vp), ("ldw", i64), ("bias", vp), ("resid", vp), ("out", vp), ("ldo", i64)]


class AttentionArgs(C.Structure):
    _fields_ = [("dtype", i32), ("batch", i32), ("heads", i32), ("head_dim", i32), ("s_q", i32), ("s_kv", i32),
                ("q", vp), ("k", vp), ("v", vp), ("q_tok_stride", i64), ("kv_tok_stride", i64), ("q_batch_stride", i64),
                ("kv_batch_stride", i64), ("key_mask", vp), ("rope_q", vp), ("rope_k", vp), ("scale", f32), ("out", vp),
                ("out_tok_stride", i64), ("out_batch_stride", i64), ("causal", i32)]


class LlmAttnRowsArgs(C.Structure):
    _fields_ = [("dtype", i32), ("rows", i32), ("heads", i32), ("head_dim", i32), ("rows_per_batch", i32), ("n_keys", i32), ("t_pad", i32),
                ("scale", f32), ("q", vp), ("q_row", i64), ("k", vp), ("v", vp), ("kv_batch", i64), ("kv_head", i64), ("kv_tok", i64),
                ("bias", vp), ("bias_pos", i64), ("bias_head", i64), ("pos", vp), ("out", vp), ("o_row", i64)]


class LayerNormArgs(C.Structure):
    _fields_ = [("dtype", i32), ("M", i32), ("D", i32), ("eps", f32), ("x", vp), ("gamma", vp), ("beta", vp),
                ("out_f32", vp), ("out_act", vp)]


_lib = None


def lib() -> C.CDLL:
    """Load libvtgb.so; fail loudly if the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VtgbError(f"{LIB_PATH} is missing: build it with `python -m videotgb_amd.build` "
                        "(there is no CPU fallback for the hot path)")
    L = C.CDLL(LIB_PATH)
    L.vtgb_version.restype = C.c_int
    L.vtgb_last_error.restype = C.c_char_p
    L.vtgb_pack_bf16.argtypes = [vp, vp, i64, i64, i64, vp]
    L.vtgb_vit_patch_kpad.argtypes = [i32, i32]
    L.vtgb_vit_patch_kpad.restype = i32
    for name, st in (("span_select", SpanSelectArgs), ("span_to_frames", SpanToFramesArgs),
                     ("gather_frames", GatherFramesArgs), ("vit_forward", VitArgs), ("qformer_forward", QFormerArgs),
                     ("pool_project", PoolProjectArgs), ("tgb_forward", TgbArgs), ("gemm", GemmArgs),
                     ("attention", AttentionArgs), ("layernorm", LayerNormArgs)):
        fn = getattr(L, "vtgb_" + name)
        fn.argtypes = [C.POINTER(st), vp]
        fn.restype = C.c_int
    for name, st in (("vit", VitArgs), ("qformer", QFormerArgs), ("pool_project", PoolProjectArgs), ("tgb", TgbArgs)):
        fn = getattr(L, f"vtgb_{name}_workspace_bytes")
        fn.argtypes = [C.POINTER(st)]
        fn.restype = sz
    L.vtgb_pair_pack.argtypes = [i32, vp, vp, i64, i32, i32, vp]
    L.vtgb_pair_pack.restype = C.c_int
    L.vtgb_pair_conv.argtypes = [C.POINTER(PairConvArgs), vp]
    L.vtgb_pair_conv.restype = C.c_int
    L.vtgb_pair_conv_ex.argtypes = [C.POINTER(PairConvExArgs), vp]
    L.vtgb_pair_conv_ex.restype = C.c_int
    L.vtgb_raft_update.argtypes = [C.POINTER(RaftUpdateArgs), vp]
    L.vtgb_raft_update.restype = C.c_int
    L.vtgb_raft_update_workspace_bytes.argtypes = [C.POINTER(RaftUpdateArgs)]
    L.vtgb_raft_update_workspace_bytes.restype = sz
    L.vtgb_raft_encoder.argtypes = [C.POINTER(RaftEncoderArgs), vp]
    L.vtgb_raft_encoder.restype = C.c_int
    L.vtgb_raft_encoder_workspace_bytes.argtypes = [C.POINTER(RaftEncoderArgs)]
    L.vtgb_raft_encoder_workspace_bytes.restype = sz
    L.vtgb_raft_corr.argtypes = [C.POINTER(RaftCorrArgs), vp]
    L.vtgb_raft_corr.restype = C.c_int
    L.vtgb_raft_corr_workspace_bytes.argtypes = [C.POINTER(RaftCorrArgs)]
    L.vtgb_raft_corr_workspace_bytes.restype = sz
    L.vtgb_llm_rmsnorm.argtypes = [C.c_int, vp, vp, vp, vp, i64, i32, f32, vp]
    L.vtgb_llm_rope_cache.argtypes = [C.c_int, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]
    L.vtgb_llm_rope_cache_prefill.argtypes = [C.c_int, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]
    L.vtgb_llm_decode_attention.argtypes = [C.c_int, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, vp]
    L.vtgb_llm_silu_mul.argtypes = [C.c_int, vp, vp, i64, i32, vp]
    L.vtgb_llm_attention_rows.argtypes = [C.POINTER(LlmAttnRowsArgs), vp]
    L.vtgb_llm_attention_rows.restype = C.c_int
    L.vtgb_llm_gated_act.argtypes = [C.c_int, vp, vp, i64, i32, i32, i32, vp]
    L.vtgb_llm_gated_act.restype = C.c_int
    L.vtgb_llm_rmsnorm_parts.argtypes = [C.c_int, vp, vp, i32, vp, vp, i64, i32, f32, vp]
    L.vtgb_llm_rmsnorm_parts.restype = C.c_int
    L.vtgb_llm_rope_cache_parts.argtypes = [C.c_int, vp, i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]
    L.vtgb_llm_rope_cache_parts.restype = C.c_int
    L.vtgb_gemm_skinny_splits.argtypes = [C.POINTER(GemmSkinnyArgs)]
    L.vtgb_gemm_skinny_splits.restype = i32
    L.vtgb_gemm_skinny.argtypes = [C.POINTER(GemmSkinnyArgs), vp]
    L.vtgb_gemm_skinny.restype = C.c_int
    L.vtgb_gemm_skinny_workspace_bytes.argtypes = [C.POINTER(GemmSkinnyArgs)]
    L.vtgb_gemm_skinny_workspace_bytes.restype = sz
    L.vtgb_pack_skinny_weight_bytes.argtypes = [i32, i32]
    L.vtgb_pack_skinny_weight_bytes.restype = sz
    L.vtgb_pack_skinny_weight.argtypes = [vp, i64, i32, i32, vp, vp]
    L.vtgb_pack_skinny_weight.restype = C.c_int
    for fn in (L.vtgb_attn_train_forward, L.vtgb_attn_train_backward):
        fn.argtypes = [C.POINTER(AttnTrainArgs), vp]
        fn.restype = C.c_int
    L.vtgb_gemm_train.argtypes = [C.POINTER(GemmTrainArgs), vp]
    L.vtgb_gemm_train_workspace_bytes.argtypes = [C.POINTER(GemmTrainArgs)]
    L.vtgb_gemm_train_workspace_bytes.restype = sz
    L.vtgb_col_sum_f32.argtypes = [vp, i64, i32, i32, vp, vp, vp]
    L.vtgb_col_sum_parts.argtypes = [i32]
    L.vtgb_col_sum_parts.restype = i32
    L.vtgb_layernorm_train_partials.argtypes = [i32]
    L.vtgb_layernorm_train_partials.restype = i32
    for fn in (L.vtgb_layernorm_train_forward, L.vtgb_layernorm_train_backward):
        fn.argtypes = [C.POINTER(LayerNormTrainArgs), vp]
    L.vtgb_gelu_forward.argtypes = [vp, vp, i64, vp]
    L.vtgb_gelu_backward.argtypes = [vp, vp, vp, i64, vp]
    for fn in (L.vtgb_gemm_train, L.vtgb_col_sum_f32, L.vtgb_layernorm_train_forward, L.vtgb_layernorm_train_backward, L.vtgb_gelu_forward,
               L.vtgb_gelu_backward):
        fn.restype = C.c_int
    L.vtgb_comm_unique_id.argtypes = [vp]
    L.vtgb_comm_init.argtypes = [C.POINTER(vp), vp, i32, i32]
    L.vtgb_comm_destroy.argtypes = [vp]
    L.vtgb_allreduce_f32.argtypes = [vp, vp, sz, i32, vp]
    for fn in (L.vtgb_comm_unique_id, L.vtgb_comm_init, L.vtgb_comm_destroy, L.vtgb_allreduce_f32):
        fn.restype = C.c_int
    L.vtgb_prof_enable.argtypes = [C.c_int]
    L.vtgb_prof_enable.restype = None
    L.vtgb_prof_reset.restype = None
    L.vtgb_prof_summary.argtypes = [C.c_int, C.POINTER(i64), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.vtgb_prof_summary.restype = C.c_int
    L.vtgb_prof_executed_flops.argtypes = [C.c_int, C.POINTER(C.c_double)]
    L.vtgb_prof_executed_flops.restype = C.c_int
    _lib = L
    return L


def prof_summary(kind: int):
    """(launches, total kernel ms, total algorithmic FLOPs) of one launch kind since the last reset."""
    n, ms, fl = i64(0), C.c_double(0), C.c_double(0)
    check(lib().vtgb_prof_summary(kind, C.byref(n), C.byref(ms), C.byref(fl)))
    return n.value, ms.value, fl.value


def prof_executed_flops(kind: int) -> float:
    f = C.c_double(0)
    check(lib().vtgb_prof_executed_flops(kind, C.byref(f)))
    return f.value


_ERR = {-1: ValueError, -2: VtgbError, -3: VtgbError, -4: NotImplementedError}


def check(rc: int) -> None:
    if rc != 0:
        msg = lib().vtgb_last_error().decode("utf-8", "replace")
        raise _ERR.get(rc, VtgbError)(f"libvtgb: {msg} (code {rc})")
