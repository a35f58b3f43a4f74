"""Tensor-level wrappers over the C ABI (include/vtgb.h).  PyTorch is plumbing here: it owns
device memory and the stream; all arithmetic of the hot path happens inside libvtgb.so.

Every function takes CUDA(=HIP) tensors, launches asynchronously on the current stream and
returns fresh tensors on the same device.  Nothing here falls back to torch ops."""
from __future__ import annotations

import ctypes as C
import logging
import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib as L
from ._lib import BF16, BF16X3, F16C8, F32

_log = logging.getLogger("videotgb_amd")
_ws_logged = set()


def _log_workspace(what: str, need: int, detail: str) -> None:
    """the split-operand RAFT modes take several times the scratch memory of the bf16 mode -- say so once per (stage, size class) at INFO
    level (logging.getLogger('videotgb_amd')), and as a WARNING when a single call asks for more than 64 GB."""
    key = (what, need >> 30)
    if key in _ws_logged:
        return
    _ws_logged.add(key)
    _log.log(logging.WARNING if need > (64 << 30) else logging.INFO, "%s: %.2f GB of workspace (%s)", what, need / 2 ** 30, detail)

Tensor = torch.Tensor


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t: Optional[Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _need_cuda(*ts: Tensor) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise L.VtgbError("videotgb_amd ops need device tensors: the hot path has no CPU implementation")


def dtype_code(dtype) -> int:
    if dtype in (BF16, "bf16", torch.bfloat16):
        return BF16
    if dtype in (F32, "f32", "fp32", torch.float32):
        return F32
    raise ValueError(f"unsupported compute dtype {dtype!r}")


def raft_dtype_code(dtype) -> int:
    """RAFT's modes: "bf16" (bf16 MFMA, a reduced-precision mode the reference does not have), "bf16x3" (split-bf16 operands: fp32
    accuracy on the bf16 matrix cores) and "f32" (fp32 FMAs in the reference's order, the exactness mode)."""
    if dtype in (BF16X3, "bf16x3", "x3"):
        return BF16X3
    if dtype in (F16C8, "f16c8"):
        return F16C8
    return dtype_code(dtype)


def raft_stage_code(code: int) -> int:
    """The mode of RAFT's correlation volume next to an update block at ``code``: at f16c8 the correlation (and the encoders' stem, stride-2 / 1x1
    convolutions and head, include/vtgb.h) runs at bf16x3."""
    return BF16X3 if code == F16C8 else code


def act_dtype(code: int) -> torch.dtype:
    return torch.bfloat16 if code == BF16 else torch.float32


class _Workspace:
    """Scratch buffer handed to the library, one per (device, stream) -- calls on different streams never share scratch, so
    the wrappers are as re-entrant across streams as the C ABI underneath (grown on demand, never shrunk)."""

    def __init__(self, zero: bool = False):
        self.bufs: Dict[Tuple[int, int], Tensor] = {}
        self.zero = zero

    def get(self, nbytes: int, device) -> Tensor:
        idx = device.index if device.index is not None else torch.cuda.current_device()
        key = (idx, torch.cuda.current_stream(idx).cuda_stream)
        b = self.bufs.get(key)
        if b is None or b.numel() < nbytes:
            self.bufs[key] = b = (torch.zeros if self.zero else torch.empty)(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        return b


_ws = _Workspace()


def pack_weight(w: Tensor, code: int, cols_pad: Optional[int] = None) -> Tensor:
    """GEMM weight in the dtype the kernels consume: fp32 -> as is; bf16 -> one-time vtgb_pack_bf16."""
    _need_cuda(w)
    w2 = w.reshape(w.shape[0], -1).contiguous().float()
    if code == F32:
        return w2
    if code == BF16X3:
        raise ValueError("pack_weight: bf16x3 weights are expanded per convolution (split3_k)")
    rows, cols = w2.shape
    if cols_pad is None and cols % 8:
        raise NotImplementedError(f"bf16 GEMM weights need in_features % 8 == 0 (got {cols})")
    cp = cols_pad or cols
    out = torch.empty(rows, cp, dtype=torch.bfloat16, device=w.device)
    L.check(L.lib().vtgb_pack_bf16(w2.data_ptr(), out.data_ptr(), rows, cols, cp, _stream()))
    return out


# ----------------------------------------------------------------------------- K16-K18
def span_select(logits: Tensor, noise: Tensor, tau: float = 0.5) -> Tensor:
    """logits [B, L, 2] fp32, noise [draws, 2B, L] fp32 -> idx [draws, 2B] int64."""
    _need_cuda(logits, noise)
    logits, noise = logits.contiguous().float(), noise.contiguous().float()
    B, Lf, two = logits.shape
    draws = noise.shape[0]
    if two != 2 or tuple(noise.shape) != (draws, 2 * B, Lf):
        raise ValueError(f"span_select: logits {tuple(logits.shape)} / noise {tuple(noise.shape)} mismatch")
    idx = torch.empty(draws, 2 * B, dtype=torch.int64, device=logits.device)
    a = L.SpanSelectArgs(logits.data_ptr(), noise.data_ptr(), idx.data_ptr(), B, Lf, draws, float(tau))
    L.check(L.lib().vtgb_span_select(C.byref(a), _stream()))
    return idx


def span_to_frames(sel: Tensor, V, N: int, nframe: int, variant: str) -> Tensor:
    """sel [draws, 2B] int64; V int or int32 tensor [B] -> frame_idx [B, nframe] int64."""
    _need_cuda(sel)
    sel = sel.contiguous()
    draws, twoB = sel.shape
    B = twoB // 2
    out = torch.empty(B, nframe, dtype=torch.int64, device=sel.device)
    vt = None
    if isinstance(V, Tensor):
        vt = V.to(device=sel.device, dtype=torch.int32).contiguous()
    a = L.SpanToFramesArgs(sel.data_ptr(), _ptr(vt), out.data_ptr(), B, draws, 0 if vt is not None else int(V), N, nframe,
                           {"A": L.MAP_A, "B": L.MAP_B}[variant])
    L.check(L.lib().vtgb_span_to_frames(C.byref(a), _stream()))
    return out


def gather_frames(pixel_values: Tensor, frame_idx: Tensor) -> Tensor:
    """pixel_values [B, N, ...] fp32, frame_idx [B, nframe] int64 -> [B, nframe, ...] fp32."""
    _need_cuda(pixel_values, frame_idx)
    pv = pixel_values.contiguous().float()
    fi = frame_idx.contiguous()
    B, N = pv.shape[:2]
    nframe = fi.shape[1]
    out = torch.empty((B, nframe) + tuple(pv.shape[2:]), dtype=torch.float32, device=pv.device)
    a = L.GatherFramesArgs(pv.data_ptr(), fi.data_ptr(), out.data_ptr(), B, N, nframe, pv[0, 0].numel())
    L.check(L.lib().vtgb_gather_frames(C.byref(a), _stream()))
    return out


CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def preprocess_frames(raw: Tensor, frame_idx: Optional[Tensor] = None, size: int = 224, mean=CLIP_MEAN, std=CLIP_STD) -> Tensor:
    """raw [T, H0, W0, 3] uint8 decoded frames -> [n_out, 3, size, size] fp32: bilinear resize, truncation to uint8, /255,
    (x - mean) / std (get_frames, eval/utils/builder_utils.py:117-128); frame_idx [n_out] int64 picks source frames."""
    _need_cuda(raw)
    if raw.dtype != torch.uint8 or raw.dim() != 4 or raw.shape[-1] != 3:
        raise TypeError("clip tensor should have data type uint8 and shape [T, H, W, 3]. Got %s %s" % (raw.dtype, tuple(raw.shape)))
    raw = raw.contiguous()
    T, H0, W0, _ = raw.shape
    fi = None if frame_idx is None else frame_idx.to(device=raw.device, dtype=torch.int64).contiguous()
    n_out = T if fi is None else fi.numel()
    out = torch.empty(n_out, 3, size, size, dtype=torch.float32, device=raw.device)
    a = L.PreprocessArgs(raw.data_ptr(), None if fi is None else fi.data_ptr(), out.data_ptr(), T, H0, W0, n_out, size,
                         (C.c_float * 3)(*mean), (C.c_float * 3)(*std))
    L.check(L.lib().vtgb_preprocess_frames(C.byref(a), _stream()))
    return out


# ----------------------------------------------------------------------------- building blocks
def gemm(A: Tensor, W: Tensor, bias: Optional[Tensor] = None, epilogue: int = L.EPI_STORE, resid: Optional[Tensor] = None,
         K: Optional[int] = None) -> Tensor:
    """A [M, K] and W [N, K] in bf16 or fp32 -> out [M, N]."""
    _need_cuda(A, W)
    code = dtype_code(A.dtype)
    M, Ka = A.shape
    N = W.shape[0]
    K = K or Ka
    out_dtype = torch.float32 if epilogue in (L.EPI_RESID_F32, L.EPI_STORE_F32) else A.dtype
    out = torch.empty(M, N, dtype=out_dtype, device=A.device)
    a = L.GemmArgs(code, M, N, K, epilogue, A.data_ptr(), A.stride(0), W.data_ptr(), W.stride(0), _ptr(bias), _ptr(resid),
                   out.data_ptr(), N)
    L.check(L.lib().vtgb_gemm(C.byref(a), _stream()))
    return out


class SkinnyWeight:
    """nn.Linear.weight [N, K] (bf16) re-laid for vtgb_gemm_skinny's weight stream (vtgb_pack_skinny_weight; one-time)."""

    def __init__(self, w: Tensor):
        _need_cuda(w)
        w = w.detach()
        assert w.dtype == torch.bfloat16 and w.dim() == 2 and w.stride(1) == 1
        self.N, self.K = w.shape
        nbytes = L.lib().vtgb_pack_skinny_weight_bytes(self.N, self.K)
        if nbytes == 0:
            raise NotImplementedError(f"SkinnyWeight: K={self.K} must be a multiple of 64")
        self.data = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
        L.check(L.lib().vtgb_pack_skinny_weight(w.data_ptr(), w.stride(0), self.N, self.K, self.data.data_ptr(), _stream()))


def gemm_skinny_workspace_bytes(M: int, N: int, K: int, n_splits: int = 0) -> int:
    a = L.GemmSkinnyArgs(M, N, K, n_splits, None, K, None, K, None, N, BF16, 0, None, 0)
    return int(L.lib().vtgb_gemm_skinny_workspace_bytes(C.byref(a)))


def gemm_skinny(x: Tensor, w: Tensor, out: Optional[Tensor] = None, n_splits: int = 0, out_dtype: Optional[torch.dtype] = None,
                workspace: Optional[Tensor] = None, defer_reduce: bool = False):
    """``defer_reduce``: returns (out, n_splits, workspace) and leaves the K-split fragments to the consumer (see below).
    x [M <= 128, K] bf16, w [N, K] bf16 (nn.Linear.weight, or its SkinnyWeight) -> x @ w.T [M, N]: the decode step's projections (vtgb_gemm_skinny:
    the weights stream once, K split over workgroups, fp32 partials added in a fixed order).  ``out`` may be a preallocated
    [M, N] tensor and ``workspace`` a preallocated uint8 scratch (hipGraph capture: fixed addresses, no allocation)."""
    tiled = isinstance(w, SkinnyWeight)
    _need_cuda(x, w.data if tiled else w)
    assert x.dtype == torch.bfloat16 and x.stride(1) == 1 and (tiled or (w.dtype == torch.bfloat16 and w.stride(1) == 1))
    M, K = x.shape
    N = w.N if tiled else w.shape[0]
    assert not tiled or w.K == K
    if out is None:
        out = torch.empty(M, N, dtype=out_dtype or x.dtype, device=x.device)
    a = L.GemmSkinnyArgs(M, N, K, n_splits, x.data_ptr(), x.stride(0), w.data.data_ptr() if tiled else w.data_ptr(), K if tiled else w.stride(0),
                         out.data_ptr(), out.stride(0), dtype_code(out.dtype), 1 if tiled else 0, None, 0, 1 if defer_reduce else 0)
    need = L.lib().vtgb_gemm_skinny_workspace_bytes(C.byref(a))      # 0: no K split (or bad arguments, which the call reports)
    if need:
        ws = workspace if workspace is not None else _ws.get(need, x.device)
        a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
    L.check(L.lib().vtgb_gemm_skinny(C.byref(a), _stream()))
    if defer_reduce:
        # (out, n_splits, fragments): n_splits > 1 -> `out` is NOT written; the consumer (vtgb_llm_rmsnorm_parts / vtgb_llm_rope_cache_parts) adds the
        # fragments left in the workspace
        S = int(L.lib().vtgb_gemm_skinny_splits(C.byref(a))) if need else 1
        return out, S, (ws if need else None)
    return out


def attention(q: Tensor, k: Tensor, v: Tensor, heads: int, scale: float, key_mask: Optional[Tensor] = None,
              rope_q: Optional[Tensor] = None, rope_k: Optional[Tensor] = None, causal: bool = False) -> Tensor:
    """q [B, Sq, H*hd], k/v [B, Skv, H*hd] (any token/batch strides, unit channel stride) -> [B, Sq, H*hd].
    ``causal``: query q attends keys <= q + (Skv - Sq)."""
    _need_cuda(q, k, v)
    code = dtype_code(q.dtype)
    B, Sq, D = q.shape
    Skv = k.shape[1]
    assert k.stride() == v.stride() and q.stride(2) == 1 and k.stride(2) == 1
    out = torch.empty(B, Sq, D, dtype=q.dtype, device=q.device)
    a = L.AttentionArgs(code, B, heads, D // heads, Sq, Skv, q.data_ptr(), k.data_ptr(), v.data_ptr(), q.stride(1), k.stride(1),
                        q.stride(0), k.stride(0), _ptr(key_mask), _ptr(rope_q), _ptr(rope_k), float(scale), out.data_ptr(),
                        out.stride(1), out.stride(0), int(causal))
    L.check(L.lib().vtgb_attention(C.byref(a), _stream()))
    return out


def layernorm(x: Tensor, gamma: Tensor, beta: Tensor, eps: float, out_dtype=torch.float32) -> Tensor:
    _need_cuda(x, gamma, beta)
    x = x.contiguous().float()
    M, D = x.numel() // x.shape[-1], x.shape[-1]
    out = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    f32_out = out_dtype == torch.float32
    a = L.LayerNormArgs(BF16 if out_dtype == torch.bfloat16 else F32, M, D, float(eps), x.data_ptr(), gamma.data_ptr(),
                        beta.data_ptr(), out.data_ptr() if f32_out else None, None if f32_out else out.data_ptr())
    L.check(L.lib().vtgb_layernorm(C.byref(a), _stream()))
    return out


# ----------------------------------------------------------------------------- weight tables
class _WeightTable:
    """Keeps the (packed) device tensors of one stage alive next to the pointer array the ABI takes."""

    def __init__(self, code: int):
        self.code = code
        self.tensors: List[Optional[Tensor]] = []

    def add(self, t: Optional[Tensor], gemm_weight: bool = False, cols_pad: Optional[int] = None, force_f32: bool = False):
        if t is None:
            self.tensors.append(None)
        elif gemm_weight and not force_f32 and self.code in (BF16X3, F16C8):
            self.tensors.append(_bf16_exact(t.reshape(t.shape[0], -1).float()))      # (already split3: bf16 values)
        elif gemm_weight and not force_f32:
            self.tensors.append(pack_weight(t, self.code, cols_pad))
        else:
            self.tensors.append(t.detach().contiguous().float())

    def finish(self):
        self.array = (C.c_void_p * len(self.tensors))(*[_ptr(t) for t in self.tensors])
        return self

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.tensors if t is not None)


class VitWeights(_WeightTable):
    """model.vision_model.* -> the pointer table of vtgb_vit_forward (include/vtgb.h)."""

    def __init__(self, sd: Dict[str, Tensor], prefix: str, code: int, heads: int, eps: float = 1e-6, fold_ln: bool = True):
        super().__init__(code)
        p = prefix
        pw = sd[p + "embeddings.patch_embedding.weight"]
        self.hidden, _, self.patch, _ = pw.shape
        pos = sd[p + "embeddings.position_embedding"]
        self.tokens = pos.shape[1]
        self.image = int(round(math.sqrt(self.tokens - 1))) * self.patch
        self.heads, self.eps = heads, eps
        kpad = L.lib().vtgb_vit_patch_kpad(code, self.patch)
        self.add(pw, True, kpad)
        self.add(sd[p + "embeddings.patch_embedding.bias"])
        self.add(sd[p + "embeddings.class_embedding"].reshape(-1))
        self.add(pos.reshape(self.tokens, self.hidden))
        self.add(sd[p + "post_layernorm.weight"])
        self.add(sd[p + "post_layernorm.bias"])
        self.layers = 0
        while f"{p}encoder.layers.{self.layers}.self_attn.qkv.weight" in sd:
            lp = f"{p}encoder.layers.{self.layers}."
            folded = code == BF16 and fold_ln      # (then the unfolded qkv / fc1 weights are never read: not uploaded -- ADVICE r5)
            self.add(sd[lp + "layer_norm1.weight"]); self.add(sd[lp + "layer_norm1.bias"])
            self.add(None if folded else sd[lp + "self_attn.qkv.weight"], True); self.add(sd[lp + "self_attn.qkv.bias"])
            self.add(sd[lp + "self_attn.projection.weight"], True); self.add(sd[lp + "self_attn.projection.bias"])
            self.add(sd[lp + "layer_norm2.weight"]); self.add(sd[lp + "layer_norm2.bias"])
            self.add(None if folded else sd[lp + "mlp.fc1.weight"], True); self.add(sd[lp + "mlp.fc1.bias"])
            self.add(sd[lp + "mlp.fc2.weight"], True); self.add(sd[lp + "mlp.fc2.bias"])
            # bf16 mode: the two LayerNorms folded into the GEMMs that follow them (include/vtgb.h: +12 .. +17)
            for lnn, lin in (("layer_norm1", "self_attn.qkv"), ("layer_norm2", "mlp.fc1")):
                if code == BF16 and fold_ln:
                    wl, bl = sd[lp + lin + ".weight"].float(), sd[lp + lin + ".bias"].float()
                    gam, bet = sd[lp + lnn + ".weight"].float(), sd[lp + lnn + ".bias"].float()
                    wf = (wl * gam[None, :]).to(torch.bfloat16)
                    self.tensors.append(wf.contiguous())
                    self.add(wf.float().sum(1))
                    self.add((wl.double() @ bet.double() + bl.double()).float())
                else:
                    self.add(None); self.add(None); self.add(None)
            self.mlp = sd[lp + "mlp.fc1.weight"].shape[0]
            self.layers += 1
        self.finish()


class QFormerWeights(_WeightTable):
    """model.qformer.* (InstructBLIP with text branch, or BLIP-2) -> table of vtgb_qformer_forward."""

    def __init__(self, sd: Dict[str, Tensor], prefix: str, code: int, heads: int, cross_freq: int = 2, eps: float = 1e-12):
        super().__init__(code)
        p = prefix
        self.has_text = (p + "embeddings.word_embeddings.weight") in sd
        self.heads, self.cross_freq, self.eps = heads, cross_freq, eps
        if self.has_text:
            self.add(sd[p + "embeddings.word_embeddings.weight"]); self.add(sd[p + "embeddings.position_embeddings.weight"])
            self.add(sd[p + "embeddings.layernorm.weight"]); self.add(sd[p + "embeddings.layernorm.bias"])
        else:
            self.add(None); self.add(None)
            self.add(sd[p + "layernorm.weight"]); self.add(sd[p + "layernorm.bias"])

        def lin(name):
            w = sd.get(name + ".weight")
            self.add(w, True); self.add(sd.get(name + ".bias"))

        def lnp(name):
            self.add(sd.get(name + ".weight")); self.add(sd.get(name + ".bias"))

        self.layers = 0
        while f"{p}encoder.layer.{self.layers}.attention.attention.query.weight" in sd:
            lp = f"{p}encoder.layer.{self.layers}."
            for nm in ("query", "key", "value"):
                lin(f"{lp}attention.attention.{nm}")
            lin(lp + "attention.output.dense"); lnp(lp + "attention.output.LayerNorm")
            for nm in ("query", "key", "value"):
                lin(f"{lp}crossattention.attention.{nm}")
            lin(lp + "crossattention.output.dense"); lnp(lp + "crossattention.output.LayerNorm")
            lin(lp + "intermediate_query.dense"); lin(lp + "output_query.dense"); lnp(lp + "output_query.LayerNorm")
            lin(lp + "intermediate.dense"); lin(lp + "output.dense"); lnp(lp + "output.LayerNorm")
            self.hidden = sd[lp + "attention.attention.query.weight"].shape[0]
            self.ffn = sd[lp + "intermediate_query.dense.weight"].shape[0]
            if (lp + "crossattention.attention.key.weight") in sd:
                self.enc_hidden = sd[lp + "crossattention.attention.key.weight"].shape[1]
            self.layers += 1
        self.finish()


class TgbWeights(_WeightTable):
    """temporal_encoder.* -> table of vtgb_tgb_forward."""

    def __init__(self, sd: Dict[str, Tensor], prefix: str, code: int, heads: int, fusion_layer: int, eps: float = 1e-12):
        super().__init__(code)
        p = prefix
        self.heads, self.fusion_layer, self.eps = heads, fusion_layer, eps
        t = p + "temporal_embeddings."
        pw = sd[t + "projection.weight"]
        self.hidden, _, self.patch, _ = pw.shape
        self.image = int(round(math.sqrt(sd[t + "fc.weight"].shape[1]))) * self.patch
        self.add(sd[p + "embeddings.word_embeddings.weight"]); self.add(sd[p + "embeddings.token_type_embeddings.weight"])
        self.add(sd[p + "embeddings.LayerNorm.weight"]); self.add(sd[p + "embeddings.LayerNorm.bias"])
        self.add(sd[t + "bos"]); self.add(sd[t + "eos"])
        self.add(pw.reshape(self.hidden, -1)); self.add(sd[t + "projection.bias"])
        self.add(sd[t + "fc.weight"].reshape(-1)); self.add(sd[t + "fc.bias"])
        self.add(sd[t + "frame_pos_embed.weight"])
        self.add(sd[t + "ln.weight"]); self.add(sd[t + "ln.bias"])
        self.add(sd[p + "encoder.embed_positions.weight"]); self.add(sd[p + "encoder.c_embed_positions.weight"])
        self.add(sd[p + "mrc_head.weight"]); self.add(sd[p + "mrc_head.bias"])
        self.max_pos = sd[p + "encoder.embed_positions.weight"].shape[0]

        def lin(name):
            self.add(sd.get(name + ".weight"), True); self.add(sd.get(name + ".bias"))

        def lnp(name):
            self.add(sd.get(name + ".weight")); self.add(sd.get(name + ".bias"))

        self.layers = 0
        while f"{p}encoder.layer.{self.layers}.attention.self.query.weight" in sd:
            lp = f"{p}encoder.layer.{self.layers}."
            for att in ("attention", "crossattention"):
                for nm in ("query", "key", "value"):
                    lin(f"{lp}{att}.self.{nm}")
                lin(f"{lp}{att}.output.dense"); lnp(f"{lp}{att}.output.LayerNorm")
            lin(lp + "intermediate.dense"); lin(lp + "output.dense"); lnp(lp + "output.LayerNorm")
            self.ffn = sd[lp + "intermediate.dense.weight"].shape[0]
            self.layers += 1
        self.finish()


# ----------------------------------------------------------------------------- stages
def vit_forward(w: VitWeights, pixel_values: Tensor, want_f32: bool = True, want_act: bool = False):
    """pixel_values [n, 3, image, image] fp32 -> last_hidden_state [n, tokens, hidden] (fp32 and/or compute dtype)."""
    if pixel_values is None:
        raise ValueError("You have to specify pixel_values")
    _need_cuda(pixel_values)
    pv = pixel_values.contiguous().float()
    n = pv.shape[0]
    if tuple(pv.shape[1:]) != (3, w.image, w.image):
        raise ValueError(f"vit: pixel_values {tuple(pv.shape)} does not match image size {w.image}")
    dev = pv.device
    out32 = torch.empty(n, w.tokens, w.hidden, dtype=torch.float32, device=dev) if want_f32 else None
    outa = torch.empty(n, w.tokens, w.hidden, dtype=act_dtype(w.code), device=dev) if want_act else None
    a = L.VitArgs(w.code, n, w.image, w.patch, w.hidden, w.heads, w.mlp, w.layers, float(w.eps), pv.data_ptr(),
                  C.cast(w.array, C.POINTER(C.c_void_p)), _ptr(out32), _ptr(outa), None, 0)
    need = L.lib().vtgb_vit_workspace_bytes(C.byref(a))
    ws = _ws.get(need, dev)
    a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
    L.check(L.lib().vtgb_vit_forward(C.byref(a), _stream()))
    return out32, outa


def qformer_forward(w: QFormerWeights, query_tokens: Tensor, image_embeds: Tensor, text_ids: Optional[Tensor] = None,
                    text_mask: Optional[Tensor] = None, image_mask: Optional[Tensor] = None) -> Tensor:
    """image_embeds [n, tokens, enc_hidden] in the compute dtype -> query rows [n, n_query, hidden] fp32."""
    _need_cuda(image_embeds, query_tokens)
    ie = image_embeds.contiguous()
    if ie.dtype != act_dtype(w.code):
        ie = ie.to(act_dtype(w.code))
    n, enc_tokens, enc_hidden = ie.shape
    qt = query_tokens.reshape(-1, query_tokens.shape[-1]).contiguous().float()
    nq = qt.shape[0]
    nt = 0
    if w.has_text and text_ids is not None:
        text_ids = text_ids.contiguous().to(torch.int64)
        nt = text_ids.shape[1]
        if text_mask is not None:
            text_mask = text_mask.contiguous().to(torch.int64)
    else:
        text_ids = text_mask = None
    if image_mask is not None:
        image_mask = image_mask.contiguous().to(torch.int64)
    out = torch.empty(n, nq, w.hidden, dtype=torch.float32, device=ie.device)
    a = L.QFormerArgs(w.code, n, nq, nt, w.hidden, w.heads, w.ffn, w.layers, w.cross_freq, enc_tokens, enc_hidden,
                      1 if (w.has_text and nt > 0) else 0, float(w.eps), ie.data_ptr(), qt.data_ptr(), _ptr(text_ids),
                      _ptr(text_mask), _ptr(image_mask), C.cast(w.array, C.POINTER(C.c_void_p)), out.data_ptr(), None, 0)
    need = L.lib().vtgb_qformer_workspace_bytes(C.byref(a))
    ws = _ws.get(need, ie.device)
    a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
    L.check(L.lib().vtgb_qformer_forward(C.byref(a), _stream()))
    return out


def pool_project(query_out: Tensor, widths: Sequence[int], proj_w: Tensor, proj_b: Tensor, mode: str, code: int) -> Tensor:
    """query_out [sum(widths), n_query, hidden] fp32; proj_w packed for `code` -> LLM prefix tokens (fp32)."""
    _need_cuda(query_out, proj_w)
    if mode not in ("mean", "concat"):
        raise ValueError(f"INVALID POOL MODE: {mode}")
    qo = query_out.contiguous().float()
    total, nq, hidden = qo.shape
    if sum(widths) != total:
        raise ValueError(f"pool_project: widths sum {sum(widths)} != {total} frames")
    n_clips = len(widths)
    out_dim = proj_w.shape[0]
    if mode == "mean":
        out = torch.empty(n_clips, nq, out_dim, dtype=torch.float32, device=qo.device)
    else:
        out = torch.empty(total * nq, out_dim, dtype=torch.float32, device=qo.device)
    wd = (C.c_int32 * n_clips)(*[int(x) for x in widths])
    a = L.PoolProjectArgs(code, n_clips, nq, hidden, out_dim, L.POOL_MEAN if mode == "mean" else L.POOL_CONCAT, qo.data_ptr(),
                          C.cast(wd, C.POINTER(C.c_int32)), proj_w.data_ptr(), proj_b.data_ptr(), out.data_ptr(), None, 0)
    need = L.lib().vtgb_pool_project_workspace_bytes(C.byref(a))
    ws = _ws.get(need, qo.device)
    a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
    L.check(L.lib().vtgb_pool_project(C.byref(a), _stream()))
    if mode == "concat":
        if len(set(widths)) != 1:
            raise ValueError("concat pooling needs equal widths")
        out = out.reshape(n_clips, -1, out_dim)
    return out


def tgb_forward(w: TgbWeights, of: Tensor, of_mask: Tensor, text_ids: Tensor, text_mask: Tensor, mode: str):
    """of [B, L, 2, image, image] fp32 -> (sequence_output [B, L+2, hidden], logits [B, L, 2]) fp32."""
    if mode not in L.TGB_MODE:
        raise ValueError(f"INVALID MODE: {mode}")
    _need_cuda(of, of_mask, text_ids, text_mask)
    of = of.contiguous().float()
    B, Lf = of.shape[:2]
    if tuple(of.shape[2:]) != (2, w.image, w.image):
        raise ValueError(f"tgb: flow {tuple(of.shape)} does not match image size {w.image}")
    if Lf + 2 > w.max_pos:
        raise ValueError(f"tgb: {Lf} flow frames exceed the {w.max_pos} position table")
    of_mask = of_mask.contiguous().to(torch.int64)
    text_ids = text_ids.contiguous().to(torch.int64)
    text_mask = text_mask.contiguous().to(torch.int64)
    seq = torch.empty(B, Lf + 2, w.hidden, dtype=torch.float32, device=of.device)
    logits = torch.empty(B, Lf, 2, dtype=torch.float32, device=of.device)
    a = L.TgbArgs(w.code, B, Lf, text_ids.shape[1], w.hidden, w.heads, w.ffn, w.layers, w.fusion_layer, L.TGB_MODE[mode],
                  w.image, w.patch, float(w.eps), of.data_ptr(), of_mask.data_ptr(), text_ids.data_ptr(), text_mask.data_ptr(),
                  C.cast(w.array, C.POINTER(C.c_void_p)), seq.data_ptr(), logits.data_ptr(), None, 0)
    need = L.lib().vtgb_tgb_workspace_bytes(C.byref(a))
    ws = _ws.get(need, of.device)
    a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
    L.check(L.lib().vtgb_tgb_forward(C.byref(a), _stream()))
    return seq, logits


# ----------------------------------------------------------------------------- RAFT (a2 / f1)
def conv_k_order(w: Tensor) -> Tensor:
    """[co, kh, kw, ci] (ci % 64 == 0) -> [co, K] in the implicit-GEMM kernels' K order: 64-channel chunk major,
    tap minor (the taps of one channel slab are gathered back to back, see gemm.hip / conv_f32.hip)."""
    co, kh, kw, ci = w.shape
    return w.reshape(co, kh * kw, ci // 64, 64).permute(0, 2, 1, 3).reshape(co, -1).contiguous()


def _bf16_parts(w: Tensor) -> Tuple[Tensor, Tensor]:
    """w fp32 -> (hi, lo) as fp32 tensors holding bf16 values: hi = bf16(w), lo = bf16(w - hi)."""
    hi = w.to(torch.bfloat16).float()
    return hi, (w - hi).to(torch.bfloat16).float()


def split3(w: Tensor, sources: Optional[Sequence[int]] = None) -> Tensor:
    """[co, kh, kw, ci] fp32 -> [co, kh, kw, 3 ci] for the bf16x3 contraction: per source of the (virtual) channel concatenation the
    channel blocks [Wh | Wh | Wl], which meet the activation pair read as [hi | lo | hi] (GemmDesc::conv_wrap)."""
    ci = w.shape[-1]
    out, c0 = [], 0
    for c in (sources or [ci]):
        hi, lo = _bf16_parts(w[..., c0:c0 + c])
        out += [hi, hi, lo]
        c0 += c
    assert c0 == ci
    return torch.cat(out, -1)


H8_LO_SCALE = 2048.0      # csrc/pair_h8.h


def h8_weight_scale(w: Tensor) -> Tuple[float, int]:
    """(sw, E8M0 byte of 2^-11 / sw): sw = the largest power of two with max |w| sw <= 448 (e4m3's largest finite value)."""
    import math
    m = float(w.abs().max())
    e = 0 if m == 0.0 else math.floor(math.log2(448.0 / m))
    e = max(min(e, 100), -100)
    return 2.0 ** e, 127 - 11 - e


def h8_conv_pack(w: Tensor, sw: float, sources: Optional[Sequence[int]] = None) -> Tensor:
    """[co, kh, kw, ci] fp32 -> [co, K] bf16-typed 16-bit units for the f16c8 contraction (csrc/pair_h8.h, gemm_h8.hip): per source of the (virtual)
    channel concatenation the K order is [fp16(w): taps x C | correction bytes: taps x C], each half 64-channel chunk major / tap minor; the
    correction bytes of channels 4g .. 4g+3 are (Wh8 x 4, Wl' x 4) -- under the activations' (xl' x 4, xh8 x 4)."""
    co, kh, kw, ci = w.shape
    out, c0 = [], 0
    for c in (sources or [ci]):
        ws = w[..., c0:c0 + c].float()
        wh = ws.to(torch.float16)
        wl = ws - wh.float()
        h8 = (ws * sw).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8)                      # [co, kh, kw, c]
        l8 = (wl * (sw * H8_LO_SCALE)).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8)
        corr = torch.stack([h8.reshape(co, kh, kw, c // 4, 4), l8.reshape(co, kh, kw, c // 4, 4)], -2)      # [.., c/4, {Wh8, Wl'}, 4]
        corr16 = corr.reshape(co, kh, kw, 2 * c).contiguous().view(torch.int16)                             # [co, kh, kw, c] 16-bit units
        out += [conv_k_order(wh.view(torch.int16)), conv_k_order(corr16)]
        c0 += c
    assert c0 == ci
    return torch.cat(out, 1).contiguous().view(torch.bfloat16)


def flow_tail_pack(w2: Tensor) -> Tensor:
    """FlowHead.conv2 as the 1x1 tail of conv1's f16c8 launch: w2 [32 (tap * 2 + c, zero-padded), 256] fp32 -> [wn 4][i 4][lane 64][e 4][ob 2] fp32 with
    the value W2[channel = 64 wn + 16 i + 4 (lane >> 4) + e][out = 16 ob + (lane & 15)] -- eight consecutive floats per (wn, i, lane)."""
    t = w2.float().t().contiguous()                                    # [256 channels, 32 outs]
    t = t.reshape(4, 4, 4, 4, 2, 16)                                   # [wn, i, fg, e, ob, fr]
    return t.permute(0, 1, 2, 5, 3, 4).contiguous().reshape(-1)        # [wn, i, fg, fr, e, ob]


def _bf16_exact(w: Tensor) -> Tensor:
    """fp32 holding bf16 values -> bf16 (exact)."""
    return w.contiguous().to(torch.bfloat16)


def pair_pack(x: Tensor, fmt: int = F16C8, ld_pair: Optional[int] = None) -> Tensor:
    """x [M, C] fp32 -> pair rows (include/vtgb.h vtgb_pair_pack) as an int16 tensor [M, 2 * ld_pair]."""
    _need_cuda(x)
    x = x.contiguous().float()
    M, Cc = x.shape
    ld = ld_pair or Cc
    out = torch.empty(M, 2 * ld, dtype=torch.int16, device=x.device)
    L.check(L.lib().vtgb_pair_pack(fmt, x.data_ptr(), out.data_ptr(), M, Cc, ld, _stream()))
    return out


def pair_unpack(rows: Tensor, C_: int, fmt: int = F16C8) -> Tensor:
    """The fp32 values pair rows stand for where they are read back element-wise: f16c8 xh + xl' 2^-11 (csrc/pair_h8.h), bf16x3 hi + lo."""
    M, two_ld = rows.shape
    ld = two_ld // 2
    if fmt == BF16X3:
        v = rows.view(torch.bfloat16).float()
        return v[:, :C_] + v[:, ld:ld + C_]
    hi = rows[:, :C_].contiguous().view(torch.float16).float()
    lo = rows[:, ld:].contiguous().view(torch.uint8).reshape(M, ld // 4, 2, 4)[:, :, 0].reshape(M, ld)[:, :C_].contiguous()
    return hi + lo.view(torch.float8_e5m2).float() / H8_LO_SCALE


def pair_conv(a: Tensor, w: Tensor, sw: float, H: int, W: int, a2: Optional[Tensor] = None, bias: Optional[Tensor] = None, relu: bool = False,
              out_fmt: int = F16C8, ld_out: Optional[int] = None, resid: Optional[Tensor] = None, tail_w: Optional[Tensor] = None,
              out_f32: bool = False) -> Tensor:
    """One 'same' convolution over f16c8 pair rows (include/vtgb.h vtgb_pair_conv / vtgb_pair_conv_ex): a [M, 2 C1] int16 pair rows (pair_pack),
    w [co, kh, kw, ci] fp32 with ci = C1 (x 2 with a2), sw from h8_weight_scale(w) -> pair rows [M, 2 * ld_out] int16.  resid (f16c8 pair rows of the
    output's layout): the ResidualBlock tail relu(resid + act(conv)); tail_w ([32, 256] fp32, co == 256): returns the [M, 32] fp32 products with it
    instead; out_f32: returns fp32 rows [M, ld_out] instead."""
    _need_cuda(a, w)
    co, kh, kw, ci = w.shape
    C1 = a.shape[1] // 2
    assert ci == C1 * (2 if a2 is not None else 1)
    _, byte = h8_weight_scale(w)
    packed = h8_conv_pack(w, sw, [C1, C1] if a2 is not None else None)
    scale = torch.tensor([byte], dtype=torch.int32, device=a.device)
    ld = ld_out or ((co + 3) // 4 * 4)
    out = torch.zeros(a.shape[0], 2 * ld, dtype=torch.int16, device=a.device)
    bias_t = None if bias is None else bias.contiguous().float()
    args = L.PairConvArgs(a.shape[0], co, H, W, kh, kw, C1, a.data_ptr(), _ptr(a2), packed.data_ptr(), scale.data_ptr(), _ptr(bias_t),
                          1 if relu else 0, out_fmt, out.data_ptr(), ld)
    if resid is None and tail_w is None and not out_f32:
        L.check(L.lib().vtgb_pair_conv(C.byref(args), _stream()))
        torch.cuda.current_stream().synchronize()      # (packed / scale are temporaries of this call)
        return out
    tail_p = tail_out = f32 = None
    if tail_w is not None:
        tail_p = flow_tail_pack(tail_w.to(a.device))
        tail_out = torch.zeros(a.shape[0], 32, dtype=torch.float32, device=a.device)
    if out_f32:
        f32 = torch.zeros(a.shape[0], ld, dtype=torch.float32, device=a.device)
    ex = L.PairConvExArgs(args, _ptr(resid), ld, _ptr(tail_p), _ptr(tail_out), _ptr(f32), ld)
    L.check(L.lib().vtgb_pair_conv_ex(C.byref(ex), _stream()))
    torch.cuda.current_stream().synchronize()
    return tail_out if tail_w is not None else f32 if out_f32 else out


class RaftWeights(_WeightTable):
    """of_extractor.update_block.* -> the packed table of vtgb_raft_update ([C_out, KH, KW, C_in] in the compute dtype)."""

    def __init__(self, sd: Dict[str, Tensor], prefix: str = "update_block.", code: int = BF16, hoist_inp: Optional[bool] = None):
        super().__init__(code)
        p = prefix
        if code in (BF16X3, F16C8):
            self._init_x3(sd, p, h8=(code == F16C8))
            return
        # bf16 mode: split the loop-invariant `inp` channels (128..255) out of the GRU convolutions (include/vtgb.h [26..29])
        self.hoist_inp = (code == BF16) if hoist_inp is None else (hoist_inp and code == BF16)

        def conv(name, cin_pad=None, channels=None):
            w = sd[p + name + ".weight"].float()
            if channels is not None:
                w = w[:, channels]
            co, ci, kh, kw = w.shape
            w = w.permute(0, 2, 3, 1)                       # [co, kh, kw, ci]
            if cin_pad and cin_pad != ci:
                w = torch.nn.functional.pad(w, (0, cin_pad - ci))
            return conv_k_order(w)

        def add_conv(name, cin_pad=None):
            self.add(conv(name, cin_pad), True)
            self.add(sd[p + name + ".bias"])

        add_conv("encoder.convc1", 384)
        add_conv("encoder.convc2")
        if code == BF16:
            # convf1 on the matrix cores: k = tap * 4 + {x, y, x, y} (flow head | flow remainder), 49 taps padded to 56
            wf = sd[p + "encoder.convf1.weight"].float().reshape(128, 2, 49).permute(0, 2, 1)         # [co, tap, c]
            wf = torch.nn.functional.pad(torch.cat([wf, wf], 2), (0, 0, 0, 7))                        # [co, 56, 4]
            self.add(wf.reshape(128, 224).contiguous(), True)
        else:
            # exactness mode: plain fp32 FMAs over the 98 taps; [98, 128], k = c * 49 + ky * 7 + kx
            self.add(sd[p + "encoder.convf1.weight"].float().reshape(128, 98).t().contiguous(), True)
        self.add(sd[p + "encoder.convf1.bias"])
        add_conv("encoder.convf2")
        add_conv("encoder.conv")
        dyn = (list(range(0, 128)) + list(range(256, 384))) if self.hoist_inp else None     # [h | motion + flow]
        for sfx in ("1", "2"):
            self.add(torch.cat([conv("gru.convz" + sfx, channels=dyn), conv("gru.convr" + sfx, channels=dyn)], 0), True)
            self.add(torch.cat([sd[p + "gru.convz" + sfx + ".bias"], sd[p + "gru.convr" + sfx + ".bias"]], 0))
            self.add(conv("gru.convq" + sfx, channels=dyn), True)
            self.add(sd[p + "gru.convq" + sfx + ".bias"])
        add_conv("flow_head.conv1")
        # flow_head.conv2 as a GEMM with the taps on the output side: row tap*2 + o = w[o, :, ky, kx]; 18 rows padded to 32
        w2 = sd[p + "flow_head.conv2.weight"].float().permute(2, 3, 0, 1).reshape(18, 256)
        self.add(torch.nn.functional.pad(w2, (0, 0, 0, 14)).contiguous(), True)
        self.add(sd[p + "flow_head.conv2.bias"])
        add_conv("mask.0")
        add_conv("mask.2")
        inp = list(range(128, 256))
        for sfx in ("1", "2"):
            if self.hoist_inp:
                self.add(torch.cat([conv("gru.convz" + sfx, channels=inp), conv("gru.convr" + sfx, channels=inp)], 0), True)
                self.add(conv("gru.convq" + sfx, channels=inp), True)
            else:
                self.add(None); self.add(None)
        self.finish()

    def _init_x3(self, sd: Dict[str, Tensor], p: str, h8: bool = False) -> None:
        """The bf16x3 table (raft_x3.hip): every MFMA convolution as [C_out, taps, 3 C_in] in the kernels' K order with the channel blocks
        [Wh | Wh | Wl] per source; convf1 as in the fp32 mode; the mask head's 0.25 (update.py:143) folded into mask.2 (a power of two: exact).
        ``h8`` (VTGB_F16C8): the nine large convolutions and convf2 in the f16c8 operand format instead (h8_conv_pack) and entry [30] = their scale bytes."""
        self.hoist_inp = True
        scales = []

        def raw(name, cin_pad=None, scale=1.0, channels=None):
            w = sd[p + name + ".weight"].float() * scale
            if channels is not None:
                w = w[:, channels]
            co, ci, kh, kw = w.shape
            w = w.permute(0, 2, 3, 1)
            if cin_pad and cin_pad != ci:
                w = torch.nn.functional.pad(w, (0, cin_pad - ci))
            return w

        def conv(name, cin_pad=None, sources=None, scale=1.0, channels=None):
            return _bf16_exact(conv_k_order(split3(raw(name, cin_pad, scale, channels), sources)))

        def big(ws, sources=None):
            """one large convolution (the rows of `ws` stacked along C_out) in the mode's operand format"""
            if not h8:
                return torch.cat([_bf16_exact(conv_k_order(split3(w, sources))) for w in ws], 0).contiguous()
            sw, byte = h8_weight_scale(torch.cat([w.reshape(-1) for w in ws]))
            scales.append(byte)
            return torch.cat([h8_conv_pack(w, sw, sources) for w in ws], 0).contiguous()

        def add_big(name, cin_pad=None):
            self.tensors.append(big([raw(name, cin_pad)]))
            self.add(sd[p + name + ".bias"].float())

        def add_conv(name, cin_pad=None, scale=1.0):
            self.tensors.append(conv(name, cin_pad, scale=scale))
            self.add(sd[p + name + ".bias"].float() * scale)

        add_big("encoder.convc1", 384)
        add_big("encoder.convc2")
        self.add(sd[p + "encoder.convf1.weight"].float().reshape(128, 98).t().contiguous())
        self.add(sd[p + "encoder.convf1.bias"])
        late_scales = []
        if h8:      # (round 6, later: convf2 joins the f16c8 convolutions; its scale byte is entry [9] of the table's scale vector)
            wf2 = raw("encoder.convf2")
            sw, byte = h8_weight_scale(wf2)
            late_scales.append(byte)
            self.tensors.append(h8_conv_pack(wf2, sw))
            self.add(sd[p + "encoder.convf2.bias"].float())
        else:
            add_conv("encoder.convf2")
        add_big("encoder.conv")
        dyn = list(range(0, 128)) + list(range(256, 384))     # [h | motion + flow]: two pair sources of 128 channels
        for sfx in ("1", "2"):
            self.tensors.append(big([raw("gru.convz" + sfx, channels=dyn), raw("gru.convr" + sfx, channels=dyn)], sources=[128, 128]))
            self.add(torch.cat([sd[p + "gru.convz" + sfx + ".bias"], sd[p + "gru.convr" + sfx + ".bias"]], 0))
            self.tensors.append(big([raw("gru.convq" + sfx, channels=dyn)], sources=[128, 128]))
            self.add(sd[p + "gru.convq" + sfx + ".bias"])
        add_big("flow_head.conv1")
        w2 = sd[p + "flow_head.conv2.weight"].float().permute(2, 3, 0, 1).reshape(18, 256)
        w2 = torch.nn.functional.pad(w2, (0, 0, 0, 14)).reshape(32, 1, 1, 256)
        if h8:      # (round 6, later) fp32, in the order the flow-head epilogue's lanes read it (csrc/gemm_h8.hip EPI_FTAIL)
            self.add(flow_tail_pack(w2.reshape(32, 256)))
        else:
            self.tensors.append(_bf16_exact(conv_k_order(split3(w2))))
        self.add(sd[p + "flow_head.conv2.bias"])
        add_big("mask.0")
        add_conv("mask.2", scale=0.25)
        inp = list(range(128, 256))                           # the loop-invariant third: start maps, once per call
        for sfx in ("1", "2"):
            self.tensors.append(torch.cat([conv("gru.convz" + sfx, channels=inp), conv("gru.convr" + sfx, channels=inp)], 0).contiguous())
            self.tensors.append(conv("gru.convq" + sfx, channels=inp))
        if h8:      # [30]: E8M0 bytes of 2^-11 / sw of convc1, convc2, conv, zr1, q1, zr2, q2, flow_head.conv1, mask.0, convf2 (include/vtgb.h)
            assert len(scales) == 9
            self.tensors.append(torch.tensor(scales + late_scales, dtype=torch.int32, device=self.tensors[0].device))
        self.finish()


def raft_update(w: RaftWeights, net: Optional[Tensor], inp: Optional[Tensor], pyramid: Sequence[Tensor], iters: int = 20,
                cnet_nhwc: Optional[Tensor] = None, hw: Optional[Tuple[int, int]] = None, flow_init: Optional[Tensor] = None) -> Tensor:
    """net/inp [n, 128, H8, W8] fp32 (tanh / relu applied) -- or ``cnet_nhwc`` [n, H8*W8, 256], the context encoder's
    pixel-major output, with ``hw=(H8, W8)`` (tanh / relu are then applied inside); pyramid: 4 levels
    [n*H8*W8, 1, h, w] fp32 or fp16 -> flow_up [n, 2, 8H8, 8W8].  ``flow_init`` [n, 2, H8, W8] as in xraft.py:131-132."""
    if cnet_nhwc is not None:
        _need_cuda(cnet_nhwc, *pyramid)
        cnet_nhwc = cnet_nhwc.contiguous().float()
        n, (H8, W8) = cnet_nhwc.shape[0], hw
    else:
        _need_cuda(net, inp, *pyramid)
        net, inp = net.contiguous().float(), inp.contiguous().float()
        n, _, H8, W8 = net.shape
    if len(pyramid) != 4:
        raise ValueError("raft_update: the correlation pyramid has 4 levels")
    half = all(t.dtype == torch.float16 for t in pyramid)
    lv = [t.contiguous() if half else t.contiguous().float() for t in pyramid]
    dev = lv[0].device
    if flow_init is not None:
        flow_init = flow_init.contiguous().float()
    out = torch.empty(n, 2, 8 * H8, 8 * W8, dtype=torch.float32, device=dev)
    a = L.RaftUpdateArgs(w.code, n, H8, W8, iters, None if net is None else net.data_ptr(), None if inp is None else inp.data_ptr(),
                         (C.c_void_p * 4)(*[t.data_ptr() for t in lv]), C.cast(w.array, C.POINTER(C.c_void_p)), out.data_ptr(), None, 0,
                         1 if half else 0, None if cnet_nhwc is None else cnet_nhwc.data_ptr(), _ptr(flow_init))
    need = L.lib().vtgb_raft_update_workspace_bytes(C.byref(a))
    _log_workspace("raft_update", need, f"{n} frame pairs of {H8} x {W8} coarse pixels, dtype code {w.code}; lower flow_clips_per_call to shrink it")
    ws = _ws.get(need, dev)
    a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
    L.check(L.lib().vtgb_raft_update(C.byref(a), _stream()))
    return out


def raft_corr(fmap: Tensor, n_pairs: int, H8: int, W8: int, pairs_per_clip: int, frames_per_clip: int, first_off: int, second_off: int,
              code: int = BF16) -> List[Tensor]:
    """CorrBlock.__init__ (corr.py:12-27, :52-60): fmap [n_images, H8*W8, 256] fp32 (raft_encoder layout) -> the 4-level pyramid,
    level l [n_pairs*H8*W8, 1, H8 >> l, W8 >> l] in fp16 (bf16 mode) or fp32 (exactness mode).  Pair n correlates images
    (n // pairs_per_clip) * frames_per_clip + n % pairs_per_clip + first_off / + second_off."""
    _need_cuda(fmap)
    fmap = fmap.contiguous().float()
    n_images = fmap.numel() // (H8 * W8 * 256)
    odt = torch.float16 if code == BF16 else torch.float32      # (bf16x3: split-bf16 products, fp32 levels)
    lv, h, w = [], H8, W8
    for _ in range(4):
        lv.append(torch.empty(n_pairs * H8 * W8, 1, h, w, dtype=odt, device=fmap.device))
        h, w = h // 2, w // 2
    done = 0
    while done < n_pairs:   # the kernel's grid takes at most 65535 pairs; chunks must start on a clip boundary
        per = n_pairs - done
        if per > 65535:
            per = max(65535 // pairs_per_clip, 1) * pairs_per_clip
            if per > 65535:
                raise NotImplementedError("raft_corr: more than 65535 pairs per clip")
        img0 = (done // pairs_per_clip) * frames_per_clip
        fm = fmap.view(n_images, -1)[img0:]
        lvs = [t[done * H8 * W8:] for t in lv]
        a = L.RaftCorrArgs(code, per, H8, W8, 256, pairs_per_clip, frames_per_clip, first_off, second_off, n_images - img0, 1.0 / 16.0,
                           fm.data_ptr(), (C.c_void_p * 4)(*[t.data_ptr() for t in lvs]), None, 0)
        need = L.lib().vtgb_raft_corr_workspace_bytes(C.byref(a))
        if need == 0:
            L.check(L.lib().vtgb_raft_corr(C.byref(a), _stream()))   # raises with the library's message
        ws = _ws.get(need, fmap.device)
        a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
        L.check(L.lib().vtgb_raft_corr(C.byref(a), _stream()))
        done += per
    return lv


class RaftEncoderWeights(_WeightTable):
    """of_extractor.{fnet,cnet}.* -> the packed table of vtgb_raft_encoder.  batch_norm=True folds the eval-mode
    BatchNorm2d that follows each convolution (extractor.py:20-24,124) into the convolution's weight and bias."""

    def __init__(self, sd: Dict[str, Tensor], prefix: str, batch_norm: bool, code: int = BF16):
        super().__init__(code)
        self.batch_norm = batch_norm
        p = prefix

        def folded(conv, bn):
            w, b = sd[p + conv + ".weight"].float(), sd[p + conv + ".bias"].float()
            if batch_norm:
                g = sd[p + bn + ".weight"].float() / torch.sqrt(sd[p + bn + ".running_var"].float() + 1e-5)
                w = w * g.view(-1, 1, 1, 1)
                b = (b - sd[p + bn + ".running_mean"].float()) * g + sd[p + bn + ".bias"].float()
            return w, b

        x3 = code in (BF16X3, F16C8)
        h8_scales = [0] * 12      # VTGB_F16C8: the stride-1 3x3 convolutions on f16c8 operands (csrc/raft_enc.hip), entry [40] = their scale bytes

        def packed(w, cin_pad, cout_pad=None, h8=None):
            co, ci, kh, kw = w.shape
            w = w.permute(0, 2, 3, 1)
            if cin_pad != ci:
                w = torch.nn.functional.pad(w, (0, cin_pad - ci))
            if h8 is not None:      # (slot of the scale byte: 2 * block + conv)
                if batch_norm and cout_pad and cout_pad != co:
                    w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, 0, 0, cout_pad - co))
                sw, byte = h8_weight_scale(w)
                h8_scales[h8] = byte
                return h8_conv_pack(w, sw)
            w = conv_k_order(split3(w) if x3 else w)
            if batch_norm and cout_pad and cout_pad != co:        # cnet stores activations straight from the GEMM: padded channels = 0
                w = torch.nn.functional.pad(w, (0, 0, 0, cout_pad - co))
            return w

        def pbias(b, cout_pad):
            return torch.nn.functional.pad(b, (0, cout_pad - b.numel())) if batch_norm and cout_pad != b.numel() else b

        # stem as a 4x1 convolution over the space-to-depth image (raft_enc.hip): [co][tY][dX, py, px, c | pad to 64],
        # ky = 2 tY + py - 1, kx = 2 dX + px - 1.  fp32 mode: the kernel packs 2*(x/255)-1 itself, weights as they are.
        # bf16 mode: the kernel packs x - 127.5 as a bf16 pair (hi | lo channel chunks), so w' = w * 2/255 in BOTH chunks, b' = b
        w, b = folded("conv1", "norm1")
        if code != F32:
            w = w * (2.0 / 255.0)
        wp = torch.zeros(64, 4, 4, 2, 2, 3, dtype=torch.float32, device=w.device)
        for tY in range(4):
            for py in range(2):
                ky = 2 * tY + py - 1
                if ky < 0:
                    continue
                for dX in range(4):
                    for px in range(2):
                        kx = 2 * dX + px - 1
                        if kx >= 0:
                            wp[:, tY, dX, py, px, :] = w[:, :, ky, kx]
        wp = torch.nn.functional.pad(wp.reshape(64, 4, 48), (0, 16))                    # [co, tY, 64]
        if code == BF16:
            wp = torch.stack([wp, wp], 1)                                               # K order: 64-channel chunk major, tap minor
        if x3:                                                                          # [co, tY, 1, 64] -> blocks [Wh | Wh | Wl], K order
            wp = conv_k_order(split3(wp.reshape(64, 4, 1, 64)))
        self.add(wp.reshape(64, -1).contiguous(), True); self.add(b)
        cin_pad = 64
        for l, (li, c, cpad) in enumerate((("layer1", 64, 64), ("layer2", 96, 128), ("layer3", 128, 128))):
            for bi in range(2):
                bp = f"{li}.{bi}."
                blk = 2 * l + bi
                w1, b1 = folded(bp + "conv1", bp + "norm1")
                w2, b2 = folded(bp + "conv2", bp + "norm2")
                # f16c8: every stride-1 3x3 (raw 16-bit patterns: appended as they are); the stride-2 conv1 of layer2.0 / layer3.0 stays bf16x3
                if code == F16C8 and (l == 0 or bi == 1):
                    self.tensors.append(packed(w1, cin_pad, cpad, 2 * blk)); self.add(pbias(b1, cpad))
                else:
                    self.add(packed(w1, cin_pad, cpad), True); self.add(pbias(b1, cpad))
                if code == F16C8:
                    self.tensors.append(packed(w2, cpad, cpad, 2 * blk + 1)); self.add(pbias(b2, cpad))
                else:
                    self.add(packed(w2, cpad, cpad), True); self.add(pbias(b2, cpad))
                if (p + bp + "downsample.0.weight") in sd:
                    wd, bd = folded(bp + "downsample.0", bp + "norm3")
                    self.add(packed(wd, cin_pad, cpad), True); self.add(pbias(bd, cpad))
                else:
                    self.add(None); self.add(None)
                cin_pad = cpad
        wh = sd[p + "conv2.weight"].float().reshape(256, 128)
        self.add(conv_k_order(split3(wh.reshape(256, 1, 1, 128))) if x3 else wh.contiguous(), True)
        self.add(sd[p + "conv2.bias"])
        if code == F16C8:
            assert len(self.tensors) == 40
            self.tensors.append(torch.tensor(h8_scales, dtype=torch.int32, device=self.tensors[0].device))
        self.finish()


def raft_encoder(w: RaftEncoderWeights, images: Tensor, max_images: int = 384) -> Tensor:
    """images [n, 3, H, W] fp32 (RAFT's 0..255 convention) -> NHWC features [n, H/8 * W/8, 256] fp32."""
    _need_cuda(images)
    images = images.contiguous().float()
    n, _, H, W = images.shape
    out = torch.empty(n, (H // 8) * (W // 8), 256, dtype=torch.float32, device=images.device)
    # max_images frames per launch sequence.  Workspace per frame at 224 x 224: 7.2 MB at bf16, 14.4 MB at fp32 / bf16x3 / f16c8 (4 bytes per channel of
    # every activation: pairs) = 5.5 GB for 384 frames.  Larger chunks are faster (per frame, f16c8 fnet: 39.2 us at 96, 36.8 at 192, 35.3 at 384, 34.7 at
    # 768 frames -- the small late stages fill the 256 CUs better; a chunk that fits the 256 MB Infinity Cache is NOT: tools/exp/enc_chunk.py), so the
    # pair modes no longer halve the chunk (round 5 did, for memory; the fp32 exactness mode still does).
    if w.code == F32:
        max_images = max(max_images // 2, 1)
    for i0 in range(0, n, max_images):
        chunk = images[i0:i0 + max_images]
        a = L.RaftEncoderArgs(w.code, chunk.shape[0], H, W, 1 if w.batch_norm else 0, chunk.data_ptr(), C.cast(w.array, C.POINTER(C.c_void_p)),
                              out[i0:i0 + max_images].data_ptr(), None, 0)
        need = L.lib().vtgb_raft_encoder_workspace_bytes(C.byref(a))
        _log_workspace("raft_encoder", need, f"{chunk.shape[0]} frames of {H} x {W} per chunk, dtype code {w.code}; max_images sets the chunk")
        ws = _ws.get(need, images.device)
        a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
        L.check(L.lib().vtgb_raft_encoder(C.byref(a), _stream()))
    return out
