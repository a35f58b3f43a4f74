"""The LoRA training step of config C5: twins of the loss-side pieces of
``src/models/LSTP_Vicuna_IVT_module.py`` around the frozen video->prefix path.

* ``concat_text_input_output`` + label masking (:692-718, :284-291)  -> ``vtgb_concat_text_io`` (HIP, integer)
* shifted cross-entropy (:297-299, :325-326)                        -> ``vtgb_shifted_ce_*`` (HIP, fwd + bwd)
* LoRA (``peft.get_peft_model(LoraConfig(CAUSAL_LM, r=8, lora_alpha=32, lora_dropout=0.1))``, :183-186; peft 0.4.0 is
  third-party and not vendored by the reference): ``LoraLinear`` keeps its parameter names
  (``q_proj.weight``, ``q_proj.lora_A.default.weight``, ``q_proj.lora_B.default.weight``) and arithmetic
* ``freeze_weights`` (:682-690), AdamW + the cosine schedule as the reference configures it (:634-679)
* gradient exchange: one flat bucket, one all-reduce per optimizer step (``dist.FlatGradBucket``; RCCL on the GPU box)
* the trainable set is the reference's (``freeze_weights`` :682-690 freezes RAFT, the vision tower and the TGB only): the
  Q-Former, ``query_tokens`` and ``language_projection`` train together with the adapters.  ``prefix_with_grad`` (and, for the
  SF flavours, ``tgb_with_grad``) build the TRAINING forward as an autograd graph whose linear layers run forward, dgrad and
  wgrad on the library's own GEMM kernel (``_HipLinear``) and whose attentions run on ``train_attn.hip`` forward and backward
  (``_HipAttention``); LayerNorm with its residual add and dropout mask (``_HipLayerNorm``) and GELU (``_HipGelu``) are own kernels forward
  and backward too -- what is left to torch is indexing (embedding gathers, ``cat``, slices, the TGB's rotary shuffle) and the mean
  pool.  The fused inference stages (vtgb_qformer_forward,
  vtgb_tgb_forward) keep no activations and have no dropout: they serve eval.  Dropout (Q-Former / TGB hidden and
  attention-probability dropout 0.1; LoRA dropout inside ``LoraLinear``) enters as injectable masks (``Dropout``).

The language model itself, autograd through it and AdamW stay PyTorch, as they are third-party in the reference.
"""
import ctypes as C
import math
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import torch
from torch import Tensor, nn

from . import _lib as L
from .ops import _need_cuda, _stream, dtype_code


# ----------------------------------------------------------------------------- tokens and labels
def concat_text_input_output(input_ids: Tensor, input_atts: Tensor, output_ids: Tensor, output_atts: Tensor,
                             pad_token_id: Optional[int] = None, prefix_len: int = 0):
    """LSTPModule.concat_text_input_output (LSTP_Vicuna_IVT_module.py:692-718) on the device, without the per-row host
    syncs: returns ``(llm_tokens, input_part_targets_len)`` like the reference (the lengths as an int64 tensor), plus
    ``labels`` [B, prefix_len + L] (:284-291) when ``pad_token_id`` is given."""
    _need_cuda(input_ids, input_atts, output_ids, output_atts)
    ii, ia = input_ids.long().contiguous(), input_atts.long().contiguous()
    oi, oa = output_ids.long().contiguous(), output_atts.long().contiguous()
    B, Li = ii.shape
    Lo = oi.shape[1]
    dev = ii.device
    ids = torch.empty(B, Li + Lo - 1, dtype=torch.int64, device=dev)
    atts = torch.empty_like(ids)
    lens = torch.empty(B, dtype=torch.int64, device=dev)
    labels = torch.empty(B, prefix_len + Li + Lo - 1, dtype=torch.int64, device=dev) if pad_token_id is not None else None
    a = L.ConcatTextIoArgs(ii.data_ptr(), ia.data_ptr(), oi.data_ptr(), oa.data_ptr(), ids.data_ptr(), atts.data_ptr(), lens.data_ptr(),
                           None if labels is None else labels.data_ptr(), 0 if pad_token_id is None else int(pad_token_id), B, Li, Lo, prefix_len)
    L.check(L.lib().vtgb_concat_text_io(C.byref(a), _stream()))
    out = ({"input_ids": ids, "attention_mask": atts}, lens)
    return out + (labels,) if labels is not None else out


class _ShiftedCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits: Tensor, labels: Tensor):
        _need_cuda(logits, labels)
        lg = logits.contiguous()
        lb = labels.long().contiguous()
        B, S, V = lg.shape
        lse = torch.empty(B, S - 1, dtype=torch.float32, device=lg.device)
        row = torch.empty_like(lse)
        loss = torch.empty(2, dtype=torch.float32, device=lg.device)
        a = L.ShiftedCeArgs(dtype_code(lg.dtype), B, S, V, lg.data_ptr(), lb.data_ptr(), lse.data_ptr(), row.data_ptr(), loss.data_ptr(), None, None)
        L.check(L.lib().vtgb_shifted_ce_forward(C.byref(a), _stream()))
        ctx.save_for_backward(lg, lb, lse, loss)
        return loss[0]

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        lg, lb, lse, loss = ctx.saved_tensors
        B, S, V = lg.shape
        g = grad_out.reshape(1).float().contiguous()
        d = torch.empty_like(lg)
        a = L.ShiftedCeArgs(dtype_code(lg.dtype), B, S, V, lg.data_ptr(), lb.data_ptr(), lse.data_ptr(), None, loss.data_ptr(), g.data_ptr(), d.data_ptr())
        L.check(L.lib().vtgb_shifted_ce_backward(C.byref(a), _stream()))
        return d, None


def shifted_cross_entropy(logits: Tensor, labels: Tensor) -> Tensor:
    """``CrossEntropyLoss(reduction="mean")(logits[..., :-1, :], labels[..., 1:])`` with ignore_index -100
    (LSTP_Vicuna_IVT_module.py:297-299, :325-326); logits [B, S, V] fp32 or bf16, labels [B, S] -> fp32 scalar."""
    if logits.dim() != 3 or labels.shape != logits.shape[:2]:
        raise ValueError(f"Expected logits [B, S, V] and labels [B, S], got {tuple(logits.shape)} and {tuple(labels.shape)}")
    return _ShiftedCE.apply(logits, labels)


# ----------------------------------------------------------------------------- own-kernel autograd building blocks
def _operand(t: Tensor) -> Tensor:
    """A GEMM operand as the kernel can read it in place: fp32 or bf16, 2-D, unit stride along its rows (any leading dimension)."""
    if t.dtype not in (torch.float32, torch.bfloat16):
        t = t.float()
    if t.stride(1) != 1 or t.stride(0) < t.shape[1]:
        t = t.contiguous()
    return t


GEMM_FLOPS = [0.0]      # 2 M N K of every gemm_train launch since the caller last reset it (tools/train_bench.py: achieved TFLOP/s)


def gemm_train(a: Tensor, a_kmajor: bool, b: Tensor, b_kmajor: bool, bias: Optional[Tensor], code: int) -> Tensor:
    """out [M, N] fp32 = op(a) . op(b) (+ bias) on vtgb_gemm_train, the operands read where they lie.  ``a`` is stored [M, K]
    (``a_kmajor`` False) or [K, M] (True); ``b`` is stored [N, K] or [K, N].  ``code``: L.BF16 (operands rounded to bf16 inside the
    kernel, MFMA, fp32 accumulation) or L.F32 (fp32 FMA, exactness mode)."""
    _need_cuda(a, b)
    a, b = _operand(a), _operand(b)
    K, M = (a.shape[0], a.shape[1]) if a_kmajor else (a.shape[1], a.shape[0])
    Kb, N = (b.shape[0], b.shape[1]) if b_kmajor else (b.shape[1], b.shape[0])
    if K != Kb:
        raise ValueError(f"gemm_train: contraction lengths differ ({K} and {Kb})")
    bias = None if bias is None else bias.float().contiguous()
    if M == 0 or N == 0:
        return torch.empty(M, N, dtype=torch.float32, device=a.device)
    if K == 0:
        z = torch.zeros(M, N, dtype=torch.float32, device=a.device)
        return z if bias is None else z + bias
    out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    g = L.GemmTrainArgs(code, M, N, K, a.data_ptr(), a.stride(0), dtype_code(a.dtype), int(a_kmajor), b.data_ptr(), b.stride(0),
                        dtype_code(b.dtype), int(b_kmajor), None if bias is None else bias.data_ptr(), out.data_ptr(), out.stride(0), None, 0)
    need = L.lib().vtgb_gemm_train_workspace_bytes(C.byref(g))
    if need:
        ws = torch.empty(need, dtype=torch.uint8, device=a.device)
        g.workspace, g.workspace_bytes = ws.data_ptr(), need
    L.check(L.lib().vtgb_gemm_train(C.byref(g), _stream()))
    GEMM_FLOPS[0] += 2.0 * M * N * K
    return out


def col_sum(x: Tensor) -> Tensor:
    """Column sums of an fp32 matrix in a fixed order (vtgb_col_sum_f32): the bias gradient of a linear layer."""
    _need_cuda(x)
    x = x.float()
    if x.shape[0] == 0 or x.shape[1] == 0:          # (a rank whose clips all have width 0: torch's sum accepted it)
        return torch.zeros(x.shape[1], dtype=torch.float32, device=x.device)
    if x.stride(1) != 1 or x.stride(0) < x.shape[1]:      # a row-broadcast gradient (strides (0, 1)) is not a matrix the kernel can address
        x = x.contiguous()
    out = torch.empty(x.shape[1], dtype=torch.float32, device=x.device)
    part = torch.empty(L.lib().vtgb_col_sum_parts(x.shape[0]), x.shape[1], dtype=torch.float32, device=x.device)
    L.check(L.lib().vtgb_col_sum_f32(x.data_ptr(), x.stride(0), x.shape[0], x.shape[1], out.data_ptr(), part.data_ptr(), _stream()))
    return out


class _HipLinear(torch.autograd.Function):
    """y = x W^T + b with forward, dgrad and wgrad on vtgb_gemm_train (no torch.matmul, and no transposed or converted copy of an
    operand):  dX [M, K] = dY [M, N] . W [N, K] (W read k-major);  dW [N, K] = dY^T . X (both read k-major);  db = column sums of dY."""

    @staticmethod
    def forward(ctx, x: Tensor, w: Tensor, b: Optional[Tensor], code: int):
        x2 = x.reshape(-1, x.shape[-1])
        ctx.save_for_backward(x2, w)
        ctx.code, ctx.has_bias, ctx.shape = code, b is not None, x.shape
        return gemm_train(x2, False, w, False, b, code).reshape(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, gy: Tensor):
        x2, w = ctx.saved_tensors
        g2 = gy.reshape(-1, gy.shape[-1])
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = gemm_train(g2, False, w, True, None, ctx.code).reshape(ctx.shape)
        if ctx.needs_input_grad[1]:
            gw = gemm_train(g2, True, x2, True, None, ctx.code)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = col_sum(g2)
        return gx, gw, gb, None


class _HipLayerNorm(torch.autograd.Function):
    """y = LayerNorm(x * mask + resid) on vtgb_layernorm_train_forward / _backward: the post-LN residual sites (``LayerNorm(dropout(dense(h))
    + input)``: BertSelfOutput / BertOutput, xinstructblip.py:707, :788, xropebert.py:542-582) in one pass each way; ``mask`` is the
    multiplicative dropout mask (or None), ``resid`` the residual (or None: a plain LayerNorm)."""

    @staticmethod
    def forward(ctx, x: Tensor, resid: Optional[Tensor], mask: Optional[Tensor], gamma: Tensor, beta: Tensor, eps: float):
        _need_cuda(x)
        D = x.shape[-1]
        x2 = x.reshape(-1, D).float().contiguous()
        r2 = None if resid is None else resid.expand_as(x).reshape(-1, D).float().contiguous()
        m2 = None if mask is None else mask.expand_as(x).reshape(-1, D).float().contiguous()
        g, b = gamma.float().contiguous(), beta.float().contiguous()
        rows = x2.shape[0]
        y = torch.empty_like(x2)
        s = torch.empty_like(x2) if (r2 is not None or m2 is not None) else None
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        a = L.LayerNormTrainArgs(rows, D, float(eps), x2.data_ptr(), None if m2 is None else m2.data_ptr(), None if r2 is None else r2.data_ptr(),
                                 g.data_ptr(), b.data_ptr(), None if s is None else s.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                 None, None, None, None, None, None)
        L.check(L.lib().vtgb_layernorm_train_forward(C.byref(a), _stream()))
        ctx.save_for_backward(x2 if s is None else s, m2, g, mean, rstd)
        ctx.eps, ctx.shape, ctx.has_resid = float(eps), x.shape, resid is not None
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, gy: Tensor):
        s, m2, g, mean, rstd = ctx.saved_tensors
        rows, D = s.shape
        dy = gy.reshape(rows, D).float().contiguous()
        ds = torch.empty_like(s)
        dx = torch.empty_like(s) if m2 is not None else None
        dgamma = torch.empty(D, dtype=torch.float32, device=s.device)
        dbeta = torch.empty_like(dgamma)
        partial = torch.empty(L.lib().vtgb_layernorm_train_partials(rows), 2, D, dtype=torch.float32, device=s.device)
        a = L.LayerNormTrainArgs(rows, D, ctx.eps, None, None if m2 is None else m2.data_ptr(), None, g.data_ptr(), None, s.data_ptr(), None,
                                 mean.data_ptr(), rstd.data_ptr(), dy.data_ptr(), ds.data_ptr(), None if dx is None else dx.data_ptr(),
                                 dgamma.data_ptr(), dbeta.data_ptr(), partial.data_ptr())
        L.check(L.lib().vtgb_layernorm_train_backward(C.byref(a), _stream()))
        ds = ds.view(ctx.shape)
        return (ds if dx is None else dx.view(ctx.shape)), (ds if ctx.has_resid else None), None, dgamma, dbeta, None


class _HipGelu(torch.autograd.Function):
    """F.gelu (erf form) forward and backward on vtgb_gelu_forward / _backward."""

    @staticmethod
    def forward(ctx, x: Tensor):
        _need_cuda(x)
        x = x.float().contiguous()
        y = torch.empty_like(x)
        L.check(L.lib().vtgb_gelu_forward(x.data_ptr(), y.data_ptr(), x.numel(), _stream()))
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, gy: Tensor):
        (x,) = ctx.saved_tensors
        gy = gy.float().contiguous()
        dx = torch.empty_like(x)
        L.check(L.lib().vtgb_gelu_backward(x.data_ptr(), gy.data_ptr(), dx.data_ptr(), x.numel(), _stream()))
        return dx


def layer_norm(x: Tensor, gamma: Tensor, beta: Tensor, eps: float, resid: Optional[Tensor] = None, mask: Optional[Tensor] = None) -> Tensor:
    if x.numel() == 0:      # no rows (F.layer_norm accepted them): nothing to normalise; keeps the graph connected with zero gradients
        return x + 0.0 * (gamma.sum() + beta.sum())
    return _HipLayerNorm.apply(x, resid, mask, gamma, beta, eps)


def gelu(x: Tensor) -> Tensor:
    return _HipGelu.apply(x) if x.numel() else x


class _HipAttention(torch.autograd.Function):
    """dropout(softmax(q k^T * scale + key_mask)) v per head on vtgb_attn_train_forward / _backward (train_attn.hip).
    q [B, Sq, D], k / v [B, Skv, D] fp32; key_mask additive [B, Skv] or None; drop [B, H, Sq, Skv] multiplicative mask or None."""

    @staticmethod
    def _args(q, k, v, heads, scale, key_mask, drop, out, lse):
        B, Sq, D = q.shape
        return L.AttnTrainArgs(B, heads, D // heads, Sq, k.shape[1], q.data_ptr(), k.data_ptr(), v.data_ptr(), q.stride(1), k.stride(1), q.stride(0),
                               k.stride(0), None if key_mask is None else key_mask.data_ptr(), None if drop is None else drop.data_ptr(), float(scale),
                               out.data_ptr(), out.stride(1), out.stride(0), lse.data_ptr(), None, None, None, None, None)

    @staticmethod
    def forward(ctx, q: Tensor, k: Tensor, v: Tensor, heads: int, scale: float, key_mask: Optional[Tensor], drop: Optional[Tensor]):
        _need_cuda(q, k, v)
        q, k, v = q.float().contiguous(), k.float().contiguous(), v.float().contiguous()
        key_mask = None if key_mask is None else key_mask.float().contiguous()
        drop = None if drop is None else drop.float().contiguous()
        out = torch.empty_like(q)
        lse = torch.empty(q.shape[0], heads, q.shape[1], dtype=torch.float32, device=q.device)
        a = _HipAttention._args(q, k, v, heads, scale, key_mask, drop, out, lse)
        L.check(L.lib().vtgb_attn_train_forward(C.byref(a), _stream()))
        ctx.save_for_backward(q, k, v, out, lse, key_mask, drop)
        ctx.heads, ctx.scale = heads, scale
        return out

    @staticmethod
    def backward(ctx, go: Tensor):
        q, k, v, out, lse, key_mask, drop = ctx.saved_tensors
        go = go.float().contiguous()
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        delta = torch.empty_like(lse)
        a = _HipAttention._args(q, k, v, ctx.heads, ctx.scale, key_mask, drop, out, lse)
        a.dout, a.dq, a.dk, a.dv, a.delta = go.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), delta.data_ptr()
        L.check(L.lib().vtgb_attn_train_backward(C.byref(a), _stream()))
        return dq, dk, dv, None, None, None, None


class Dropout:
    """The reference's dropout sites as injectable masks (like the Gumbel noise of the sampler): ``masks[name]`` (multiplicative:
    0 or 1 / (1 - p)) when given, else a fresh Bernoulli mask from ``generator`` when ``p > 0``, else the identity.  ``drawn``
    records every mask by site name, so a step can be replayed (and a test can hand the same masks to a torch reference)."""

    def __init__(self, p: float = 0.0, masks: Optional[Dict[str, Tensor]] = None, generator: Optional[torch.Generator] = None):
        self.p, self.masks, self.generator, self.drawn = float(p), masks, generator, {}

    def mask(self, name: str, shape, device) -> Optional[Tensor]:
        if self.masks is not None:
            m = self.masks.get(name)
            if m is not None:
                self.drawn[name] = m
            return m
        if self.p <= 0.0:
            return None
        m = (torch.rand(tuple(shape), device=device, generator=self.generator) >= self.p).float() / (1.0 - self.p)
        self.drawn[name] = m
        return m

    def __call__(self, name: str, x: Tensor) -> Tensor:
        m = self.mask(name, x.shape, x.device)
        return x if m is None else x * m


_NO_DROPOUT = Dropout(0.0)


# ----------------------------------------------------------------------------- differentiable prefix (Q-Former + pooling + projection)
def _qformer_graph(sd: Dict[str, Tensor], query_tokens: Tensor, image_embeds: Tensor, heads: int, input_ids: Optional[Tensor], text_mask: Optional[Tensor],
                   cross_freq: int, eps: float, code: int = L.F32, drop: Dropout = _NO_DROPOUT) -> Tensor:
    """The Q-Former as an autograd graph over the LIVE parameters ``sd`` (names relative to ``model.qformer.``):
    InstructBlipQFormerModel.forward xinstructblip.py:1122-1242 (``input_ids`` given: embeddings :1018-1046, text FFN branch)
    / Blip2QFormerModel.forward xblip2.py:1063-1174.  Every linear layer runs forward, dgrad and wgrad on the library's GEMM
    kernel (_HipLinear: operands read in place, no transposed / converted copies), every attention on train_attn.hip
    (_HipAttention), every ``LayerNorm(dropout(dense(.)) + input)`` site and every GELU on train_ops.hip.  Dropout sites: embeddings (:1045), attention probabilities (:679), attention output (:707), FFN output (:788).
    -> [n, n_query, hidden]."""
    n = image_embeds.shape[0]
    q = query_tokens.expand(n, -1, -1)
    nq = q.shape[1]

    def lin(name, x):
        return _HipLinear.apply(x, sd[name + ".weight"], sd[name + ".bias"], code)

    def ln(name, x, resid=None, site=None):
        """LayerNorm(dropout_site(x) + resid) in one kernel each way"""
        mask = None if site is None else drop.mask(site, x.shape, x.device)
        return layer_norm(x, sd[name + ".weight"], sd[name + ".bias"], eps, resid, mask)

    def attn(ap, site, hidden, kv, key_mask):
        qq, kk, vv = lin(ap + "attention.query", hidden), lin(ap + "attention.key", kv), lin(ap + "attention.value", kv)
        dm = drop.mask(site + ".probs", (hidden.shape[0], heads, hidden.shape[1], kv.shape[1]), hidden.device)
        ctx = _HipAttention.apply(qq, kk, vv, heads, 1.0 / math.sqrt(qq.shape[-1] // heads), key_mask, dm)
        return ln(ap + "output.LayerNorm", lin(ap + "output.dense", ctx), hidden, site + ".out")

    def ffn(lp, site, inter, outp, x):
        return ln(lp + outp + ".LayerNorm", lin(lp + outp + ".dense", gelu(lin(lp + inter + ".dense", x))), x, site)

    if input_ids is not None:
        lt = input_ids.shape[1]
        emb = sd["embeddings.word_embeddings.weight"][input_ids] + sd["embeddings.position_embeddings.weight"][:lt][None]
        x = drop("embeddings", ln("embeddings.layernorm", torch.cat([q, emb], dim=1)))
        m = torch.cat([torch.ones(n, nq, device=q.device), (text_mask if text_mask is not None else torch.ones(n, lt, device=q.device)).float()], 1)
        self_mask = (1.0 - m) * -10000.0                      # additive, per key (xinstructblip.py:1119)
    else:
        x = drop("embeddings", ln("layernorm", q))
        self_mask = None
    image_embeds = image_embeds.float()
    i = 0
    while f"encoder.layer.{i}.attention.attention.query.weight" in sd:
        lp = f"encoder.layer.{i}."
        att = attn(lp + "attention.", f"layer.{i}.attention", x, x, self_mask)
        qa = att[:, :nq]
        if i % cross_freq == 0:
            qa = attn(lp + "crossattention.", f"layer.{i}.crossattention", qa, image_embeds, None)
        out = ffn(lp, f"layer.{i}.ffn_query", "intermediate_query", "output_query", qa)
        if att.shape[1] > nq:
            out = torch.cat([out, ffn(lp, f"layer.{i}.ffn_text", "intermediate", "output", att[:, nq:])], dim=1)
        x = out
        i += 1
    return x[:, :nq]


def _pool_project_graph(query_out: Tensor, widths: Sequence[int], w: Tensor, b: Tensor, mode: str, code: int = L.F32) -> Tensor:
    """eval/utils/model.py:186-195 / LSTP_Vicuna_IVT_module.py:244-249 (mean over ragged widths; width 0 -> zero row) or
    LSTP_module.py:477-481 (concat); the projection on the library's GEMM (forward and backward)."""
    if mode == "mean":
        rows, idx = [], 0
        for wd in widths:
            rows.append(query_out[idx:idx + wd].mean(0) if wd > 0 else torch.zeros_like(query_out[0]))
            idx += wd
        return _HipLinear.apply(torch.stack(rows), w, b, code)
    y = _HipLinear.apply(query_out, w, b, code)
    return y.reshape(len(widths), -1, y.shape[-1])


def prefix_params(pm) -> Tuple[List[str], List[nn.Parameter]]:
    """The prefix-side trainable parameters of ``self.model`` in a fixed order: qformer.*, query_tokens, language_projection.*."""
    names = ["qformer." + n for n, _ in pm.qformer.named_parameters()] + ["query_tokens", "language_projection.weight", "language_projection.bias"]
    params = [p for _, p in pm.qformer.named_parameters()] + [pm.query_tokens, pm.language_projection.weight, pm.language_projection.bias]
    return names, params


def prefix_with_grad(pm, image_embeds: Tensor, input_ids: Optional[Tensor], text_mask: Optional[Tensor], widths: Sequence[int], mode: str = "mean",
                     compute_dtype=None, dropout: Optional[Dropout] = None) -> Tensor:
    """``language_model_inputs`` [n_clips, P, H] from the frozen vision tower's ``image_embeds`` [sum(widths), tokens, enc] WITH
    gradients to the Q-Former, ``query_tokens`` and ``language_projection``: the training forward is the autograd graph above
    (own GEMM / attention kernels forward and backward), not the fused inference stage -- which has no dropout and keeps no
    activations.  ``compute_dtype``: "f32" (exactness mode; default = the model's) or "bf16" GEMM operands.  ``dropout``: the
    reference trains with p = 0.1 at the Q-Former's dropout sites; pass ``Dropout(0.1, generator=...)`` (or recorded masks)."""
    names, params = prefix_params(pm)
    sd = dict(zip(names, params))
    code = dtype_code(compute_dtype) if compute_dtype is not None else pm.qformer.code
    qf = pm.qformer
    q = _qformer_graph({k[len("qformer."):]: v for k, v in sd.items() if k.startswith("qformer.")}, sd["query_tokens"], image_embeds.detach(),
                       qf.cfg.heads, input_ids, text_mask, qf.cfg.cross_freq, qf.cfg.eps, code, dropout or _NO_DROPOUT)
    return _pool_project_graph(q, widths, sd["language_projection.weight"], sd["language_projection.bias"], mode, code)


# ----------------------------------------------------------------------------- differentiable Temporal Grounding Bridge (SF flavours)
def _rope(table_rows: Tensor, x: Tensor) -> Tensor:
    """apply_rotary_position_embeddings xropebert.py:335-377 on [B, S, H, hd]: interleaved pairs, table = [sin half | cos half]."""
    sin, cos = table_rows.chunk(2, dim=-1)
    sin_pos = torch.stack([sin, sin], dim=-1).reshape(table_rows.shape)[None, :, None, :]
    cos_pos = torch.stack([cos, cos], dim=-1).reshape(table_rows.shape)[None, :, None, :]
    rot = torch.stack([-x[..., 1::2], x[..., ::2]], dim=-1).reshape(x.shape)
    return x * cos_pos + rot * sin_pos


def _tgb_graph(sd: Dict[str, Tensor], of: Tensor, of_mask: Tensor, text_ids: Tensor, text_mask: Tensor, mode: str, heads: int, fusion_layer: int,
               eps: float, code: int = L.F32, drop: Dropout = _NO_DROPOUT) -> Tuple[Tensor, Tensor]:
    """RopeBertModel.forward with ``encoder_embeds=of`` (xropebert.py:1048-1169) as an autograd graph over the live parameters
    ``sd`` (names relative to ``temporal_encoder.``) -> (sequence_output [B, L+2, D], logits [B, L, 2]).
    TemporalOFEmbedding (:103-129): the k16 / s16 patch convolution and the 196 -> 1 linear commute, so the patches are reduced
    with the fc weights first (elementwise) and ONE GEMM follows -- the order the HIP forward uses."""
    F = nn.functional
    b, l, c, hh, ww = of.shape
    te = "temporal_embeddings."
    pw = sd[te + "projection.weight"]
    ps = pw.shape[-1]
    d = pw.shape[0]

    def lin(name, x):
        return _HipLinear.apply(x, sd[name + ".weight"], sd[name + ".bias"], code)

    def ln(name, x, e=eps, resid=None, site=None):
        mask = None if site is None else drop.mask(site, x.shape, x.device)
        return layer_norm(x, sd[name + ".weight"], sd[name + ".bias"], e, resid, mask)

    patches = of.float().reshape(b * l, c, hh // ps, ps, ww // ps, ps).permute(0, 2, 4, 1, 3, 5).reshape(b * l, -1, c * ps * ps)   # [BL, 196, 512]
    fcw, fcb = sd[te + "fc.weight"].reshape(-1), sd[te + "fc.bias"].reshape(())
    red = (patches * fcw[None, :, None]).sum(1)                                                                                   # [BL, 512]
    x = _HipLinear.apply(red, pw.reshape(d, -1), None, code) + sd[te + "projection.bias"] * fcw.sum() + fcb
    x = x.view(b, l, d)
    x = torch.cat([sd[te + "bos"].expand(b, 1, -1), x, torch.zeros(b, 1, d, device=x.device)], dim=1)
    ends = of_mask.sum(dim=1) - 1
    onehot = F.one_hot(ends, x.shape[1]).to(x.dtype)[..., None]                                                                   # x[b, ends[b]] = eos
    x = x * (1.0 - onehot) + sd[te + "eos"][None, None] * onehot
    x = x + sd[te + "frame_pos_embed.weight"][: x.shape[1]][None]
    x = drop("temporal_embeddings", ln(te + "ln", x, 1e-5))
    t = sd["embeddings.word_embeddings.weight"][text_ids] + sd["embeddings.token_type_embeddings.weight"][0]
    t = drop("embeddings", ln("embeddings.LayerNorm", t))
    s_len = x.shape[1]
    self_mask = (1.0 - of_mask.float()) * -10000.0                                       # :1044-1045
    cross_mask = (1.0 - text_mask.float()) * torch.finfo(torch.float32).min              # :1127
    pos = sd["encoder.embed_positions.weight"][:s_len]
    cpos = sd["encoder.c_embed_positions.weight"][: t.shape[1]]

    def attn(ap, site, hidden, kv, key_mask, qpos, kpos):
        hd = hidden.shape[-1] // heads
        qq = _rope(qpos, lin(ap + "self.query", hidden).view(*hidden.shape[:2], heads, hd)).reshape(hidden.shape)
        kk = _rope(kpos, lin(ap + "self.key", kv).view(*kv.shape[:2], heads, hd)).reshape(kv.shape[0], kv.shape[1], -1)
        vv = lin(ap + "self.value", kv)
        dm = drop.mask(site + ".probs", (hidden.shape[0], heads, hidden.shape[1], kv.shape[1]), hidden.device)
        ctx = _HipAttention.apply(qq, kk, vv, heads, 1.0 / math.sqrt(hd), key_mask, dm)
        return ln(ap + "output.LayerNorm", lin(ap + "output.dense", ctx), eps, hidden, site + ".out")

    n_layers = 0
    while f"encoder.layer.{n_layers}.attention.self.query.weight" in sd:
        n_layers += 1
    if mode in ("vision", "text"):
        lo, hi = 0, fusion_layer
    elif mode == "fusion":
        lo, hi = fusion_layer, n_layers
    elif mode == "multi_modal":
        lo, hi = 0, n_layers
    else:
        raise ValueError(f"INVALID MODE: {mode}")
    for i in range(lo, hi):
        lp = f"encoder.layer.{i}."
        a = attn(lp + "attention.", f"layer.{i}.attention", x, x, self_mask, pos, pos)
        if i >= fusion_layer:
            a = attn(lp + "crossattention.", f"layer.{i}.crossattention", a, t, cross_mask, pos, cpos)
        hmid = gelu(lin(lp + "intermediate.dense", a))
        x = ln(lp + "output.LayerNorm", lin(lp + "output.dense", hmid), eps, a, f"layer.{i}.ffn")
    logits = lin("mrc_head", x[:, 1:-1])
    return x, logits


def tgb_with_grad(temporal_encoder, of: Tensor, of_mask: Tensor, text_ids: Tensor, text_mask: Tensor, mode: str, compute_dtype=None,
                  dropout: Optional[Dropout] = None) -> Tuple[Tensor, Tensor]:
    """``self.temporal_encoder(encoder_embeds=of, ...)`` WITH gradients to the TGB's parameters -- what LSTP_SF_module.py:276-298
    needs: the self-refinement MRC loss is a loss on the TGB's span logits and is what trains the sampler.  Same graph
    construction as ``prefix_with_grad`` (own GEMM / attention kernels forward and backward)."""
    sd = dict(temporal_encoder.named_parameters())
    sd.update({k: v for k, v in temporal_encoder.named_buffers()})
    code = dtype_code(compute_dtype) if compute_dtype is not None else temporal_encoder.code
    cfg = temporal_encoder.cfg
    return _tgb_graph(sd, of, of_mask, text_ids, text_mask, mode, cfg.heads, cfg.fusion_layer, cfg.eps, code, dropout or _NO_DROPOUT)


def enable_prefix_training(pm) -> List[nn.Parameter]:
    """requires_grad = True for what the reference leaves trainable on the prefix side (everything of ``self.model`` but the
    vision tower; LSTP_Vicuna_IVT_module.py:682-690) -- the stages register their parameters frozen for inference."""
    _, params = prefix_params(pm)
    params = params + list(pm.temporal_projection.parameters())          # dead weight, but in the reference's optimizer all the same
    for p in params:
        p.requires_grad = True
    return params


# ----------------------------------------------------------------------------- LoRA
class _Adapter(nn.ModuleDict):
    """{"default": Linear} -- gives the parameters peft's names (``lora_A.default.weight``)."""


class LoraLinear(nn.Linear):
    """peft 0.4.0 ``tuners.lora.Linear``: the wrapped projection keeps ``weight`` / ``bias`` (frozen); the update is
    ``lora_B(lora_A(dropout(x))) * (lora_alpha / r)``; A ~ kaiming_uniform(a=sqrt(5)), B = 0."""

    def __init__(self, base: nn.Linear, r: int = 8, lora_alpha: int = 32, lora_dropout: float = 0.1):
        super().__init__(base.in_features, base.out_features, bias=base.bias is not None, device=base.weight.device, dtype=base.weight.dtype)
        self.weight, self.bias = base.weight, base.bias
        self.weight.requires_grad = False
        if self.bias is not None:
            self.bias.requires_grad = False
        self.r, self.lora_alpha, self.scaling = r, lora_alpha, lora_alpha / r
        self.lora_dropout = nn.ModuleDict({"default": nn.Dropout(p=lora_dropout) if lora_dropout > 0.0 else nn.Identity()})
        kw = dict(bias=False, device=base.weight.device, dtype=torch.float32)
        self.lora_A = _Adapter({"default": nn.Linear(base.in_features, r, **kw)})
        self.lora_B = _Adapter({"default": nn.Linear(r, base.out_features, **kw)})
        nn.init.kaiming_uniform_(self.lora_A["default"].weight, a=math.sqrt(5))
        nn.init.zeros_(self.lora_B["default"].weight)

    def forward(self, x: Tensor) -> Tensor:
        result = nn.functional.linear(x, self.weight, self.bias)
        a = self.lora_A["default"]
        xd = self.lora_dropout["default"](x.to(a.weight.dtype))
        return result + (self.lora_B["default"](a(xd)) * self.scaling).to(result.dtype)


def apply_lora(language_model: nn.Module, r: int = 8, lora_alpha: int = 32, lora_dropout: float = 0.1,
               target_modules: Sequence[str] = ("q_proj", "v_proj")) -> List[nn.Parameter]:
    """``get_peft_model(language_model, LoraConfig(task_type=CAUSAL_LM, r=8, lora_alpha=32, lora_dropout=0.1))``
    (LSTP_Vicuna_IVT_module.py:183-186): peft's default targets for Llama are q_proj and v_proj; every other
    parameter of the language model is frozen.  Returns the trainable (adapter) parameters
    (Vicuna-7B: 32 layers x 2 x (4096*8 + 8*4096) = 4,194,304)."""
    for p in language_model.parameters():
        p.requires_grad = False
    for parent in list(language_model.modules()):
        for name, child in list(parent.named_children()):
            if name in target_modules and isinstance(child, nn.Linear) and not isinstance(child, LoraLinear):
                setattr(parent, name, LoraLinear(child, r, lora_alpha, lora_dropout))
    return [p for n, p in language_model.named_parameters() if "lora_" in n]


def freeze_weights(module: nn.Module) -> None:
    """LSTPModule.freeze_weights (LSTP_Vicuna_IVT_module.py:682-690): RAFT, the vision tower and the TGB are frozen."""
    for name in ("of_extractor", "temporal_encoder"):
        sub = getattr(module, name, None)
        if sub is not None:
            for p in sub.parameters():
                p.requires_grad = False
    vm = getattr(getattr(module, "model", None), "vision_model", None)
    if vm is not None:
        for p in vm.parameters():
            p.requires_grad = False


# ----------------------------------------------------------------------------- optimizer
def cosine_schedule_lambda(num_warmup_steps: int, num_training_steps: int, num_cycles: float = 0.5):
    """The lr lambda of transformers.get_cosine_schedule_with_warmup (4.36.0)."""
    def f(step: int) -> float:
        if step < num_warmup_steps:
            return float(step) / float(max(1, num_warmup_steps))
        progress = float(step - num_warmup_steps) / float(max(1, num_training_steps - num_warmup_steps))
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * progress)))
    return f


def configure_optimizers(params: Iterable[nn.Parameter], lr: float = 1e-4, weight_decay: float = 0.0, max_steps: int = -1,
                         warmup_ratio: float = 0.1, scheduler: Optional[str] = "cosine") -> Dict[str, object]:
    """LSTPModule.configure_optimizers (:634-679): AdamW over the trainable set; "cosine" = transformers' warmup-cosine
    with ``warmup = int(max_steps * ratio)`` where ``max_steps = trainer.max_steps`` (-1 when the run is bounded by
    epochs, so warmup = 0 as in the reference); stepped once per EPOCH (``"interval": "epoch"``)."""
    opt = torch.optim.AdamW([p for p in params if p.requires_grad], lr=lr, weight_decay=weight_decay)
    if scheduler is None:
        return {"optimizer": opt}
    if scheduler != "cosine":
        raise NotImplementedError("UNKONWN SCHEDULER")
    warmup = int(max_steps * warmup_ratio)
    sch = torch.optim.lr_scheduler.LambdaLR(opt, cosine_schedule_lambda(warmup, max_steps))
    return {"optimizer": opt, "lr_scheduler": {"scheduler": sch, "monitor": "val/score", "interval": "epoch", "frequency": 1}}


# ----------------------------------------------------------------------------- the step
class LoraTrainStep:
    """One C5 micro-batch (LSTP_Vicuna_IVT_module.py:191-413): frozen vision tower (HIP, no grad) -> Q-Former + pooling +
    projection (``prefix_with_grad``: trainable, as in the reference) -> [prefix | question+answer embeddings] -> language model
    with LoRA -> shifted CE (HIP) -> backward; every ``accumulate_grad_batches`` micro-batches ALL trainable gradients (Q-Former
    185.7 M + projections + query tokens + 4.19 M adapter parameters = the reference's ~785 MB fp32) are summed over the
    ranks in one flat bucket and AdamW steps (configs/experiment/LSTP_instructblipvicuna7b_ivtinstruct.yaml:34
    accumulate_grad_batches=4).  ``train_prefix=False`` keeps the round-1 adapters-only step (not reference-equivalent)."""

    def __init__(self, lstp, pad_token_id: int, lr: float = 1e-4, weight_decay: float = 0.0, accumulate_grad_batches: int = 4,
                 train_prefix: bool = True, use_rccl: bool = True):
        from .dist import FlatGradBucket
        self.m = lstp
        self.lm = lstp.model.language_model
        self.pad_token_id = pad_token_id
        self.params = apply_lora(self.lm)
        freeze_weights(lstp)
        self.train_prefix = train_prefix
        if train_prefix:
            self.params = enable_prefix_training(lstp.model) + self.params
        cfg = configure_optimizers(self.params, lr=lr, weight_decay=weight_decay)
        self.optimizer, self.scheduler = cfg["optimizer"], cfg["lr_scheduler"]["scheduler"]
        # gradients are views of one flat buffer; with more than one rank the exchange runs over RCCL through the C ABI
        # (vtgb_allreduce_f32), segment by segment under the last micro-batch's backward
        from .dist import RcclComm, rank_world
        comm = RcclComm(self.params[0].device) if (rank_world()[1] > 1 and self.params[0].is_cuda and use_rccl) else None
        self.bucket = FlatGradBucket(self.params, comm=comm)
        self.accumulate = accumulate_grad_batches
        self.micro = 0

    def prefix(self, frames: Tensor, qformer_text: Optional[Tensor], qformer_mask: Optional[Tensor], widths: Sequence[int], pool: str = "mean") -> Tensor:
        """frames [sum(widths), 3, H, W] (already selected, as the IV / IVT datasets deliver them) -> language_model_inputs."""
        pm = self.m.model
        with torch.no_grad():
            img = pm.vision_model(pixel_values=frames, return_dict=True, act_output=True).last_hidden_state
        ids = mask = None
        if qformer_text is not None:
            rep = torch.as_tensor(list(widths), device=frames.device)
            ids, mask = torch.repeat_interleave(qformer_text, rep, 0), torch.repeat_interleave(qformer_mask, rep, 0)
        if self.train_prefix:
            return prefix_with_grad(pm, img, ids, mask, widths, pool)
        with torch.no_grad():
            from . import ops
            q = ops.qformer_forward(pm.qformer.table(), pm.query_tokens[0], img, ids, mask, None)
            return pm.language_projection.pool(q, widths, pool)

    def loss(self, language_model_inputs: Tensor, question: Tensor, question_mask: Tensor, answer: Tensor, answer_mask: Tensor) -> Tensor:
        """language_model_inputs [B, P, H] (the projected Q-Former prefix, LSTP_Vicuna_IVT_module.py:255-260) + tokens -> loss."""
        B, P, _ = language_model_inputs.shape
        llm_tokens, _, labels = concat_text_input_output(question, question_mask, answer, answer_mask, self.pad_token_id, P)
        emb = self.lm.get_input_embeddings()(llm_tokens["input_ids"])
        inputs_embeds = torch.cat([language_model_inputs.to(emb.dtype), emb], dim=1)
        attention_mask = torch.cat([torch.ones(B, P, dtype=torch.long, device=emb.device), llm_tokens["attention_mask"]], dim=1)
        logits = self.lm(inputs_embeds=inputs_embeds, attention_mask=attention_mask)[0]
        return shifted_cross_entropy(logits, labels)

    def step(self, language_model_inputs: Tensor, question: Tensor, question_mask: Tensor, answer: Tensor, answer_mask: Tensor) -> Tuple[Tensor, bool]:
        """One micro-batch from a prefix (``self.prefix(...)`` output: gradients flow into the Q-Former / projection when
        ``train_prefix``; a plain tensor is treated as a constant)."""
        loss = self.loss(language_model_inputs, question, question_mask, answer, answer_mask)
        last = (self.micro + 1) % self.accumulate == 0
        if last:
            self.bucket.arm(average=True)            # this backward completes the window: segments go out as they become final
        (loss / self.accumulate).backward()
        self.micro += 1
        stepped = False
        if last:
            self.bucket.all_reduce(average=True)     # DDP semantics: mean over ranks; waits for the segments in flight
            self.optimizer.step()
            self.bucket.zero_()                      # (the gradients are views of the bucket: one memset)
            stepped = True
        return loss.detach(), stepped

    def step_frames(self, frames: Tensor, qformer_text, qformer_mask, widths: Sequence[int], question: Tensor, question_mask: Tensor, answer: Tensor,
                    answer_mask: Tensor) -> Tuple[Tensor, bool]:
        """The whole micro-batch of LSTPModule.forward (LSTP_Vicuna_IVT_module.py:191-335): frames -> prefix -> loss -> backward."""
        return self.step(self.prefix(frames, qformer_text, qformer_mask, widths), question, question_mask, answer, answer_mask)
