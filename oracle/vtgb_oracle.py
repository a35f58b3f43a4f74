"""CPU oracle for the VideoTGB video->LLM-prefix hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this file, and only as the checker.  The
product path (``videotgb_amd``) never imports it and has no CPU fallback.

This is an independent restatement -- plain fp32 PyTorch CPU ops for the
floating-point stages, exact integer / IEEE rules for the integer stages -- of
what the reference's Python computes.  Every function cites the reference
file:line (relative to /root/reference) it follows.  The restatement is pinned
against outputs of the reference itself (imported in the build container by
``tests/golden/make_golden.py``; fixtures committed under ``tests/golden``):
parity is PINNED for every stage up to ``inputs_embeds``; the LLM step is
third-party ``transformers`` on both sides.

All functions take a flat ``sd`` (state_dict: name -> fp32 tensor) using the
reference's checkpoint key names plus a key prefix, so one seeded state_dict
drives the reference, this oracle and the HIP build alike.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# ----------------------------------------------------------------------------
# small helpers
# ----------------------------------------------------------------------------
def _lin(sd: SD, name: str, x: Tensor) -> Tensor:
    return F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))


def _ln(sd: SD, name: str, x: Tensor, eps: float) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], eps)


def _count_layers(sd: SD, prefix: str) -> int:
    n = 0
    while any(k.startswith(f"{prefix}{n}.") for k in sd):
        n += 1
    return n


def _heads(x: Tensor, nh: int) -> Tensor:
    b, s, d = x.shape
    return x.view(b, s, nh, d // nh).permute(0, 2, 1, 3)


def _unheads(x: Tensor) -> Tensor:
    b, nh, s, hd = x.shape
    return x.permute(0, 2, 1, 3).reshape(b, s, nh * hd)


# ----------------------------------------------------------------------------
# a9/a10: EVA-ViT-g vision tower
# ----------------------------------------------------------------------------
def vit_embed(sd: SD, p: str, pixel_values: Tensor) -> Tensor:
    """InstructBlipVisionEmbeddings.forward, src/models/components/xinstructblip.py:113-122
    (Blip2VisionEmbeddings, xblip2.py:89 is identical)."""
    w = sd[p + "embeddings.patch_embedding.weight"]
    b = sd[p + "embeddings.patch_embedding.bias"]
    x = F.conv2d(pixel_values, w, b, stride=w.shape[-1])
    x = x.flatten(2).transpose(1, 2)
    cls = sd[p + "embeddings.class_embedding"].expand(x.shape[0], 1, -1)
    x = torch.cat([cls, x], dim=1)
    return x + sd[p + "embeddings.position_embedding"][:, : x.shape[1], :]


def vit_layer(sd: SD, lp: str, x: Tensor, num_heads: int, eps: float) -> Tensor:
    """InstructBlipEncoderLayer.forward xinstructblip.py:233-269 with
    InstructBlipAttention.forward :162-204 and InstructBlipMLP.forward :216-220."""
    b, s, d = x.shape
    h = _ln(sd, lp + "layer_norm1", x, eps)
    qkv = _lin(sd, lp + "self_attn.qkv", h)                      # :172 bias = (q_bias, 0, v_bias)
    qkv = qkv.reshape(b, s, 3, num_heads, d // num_heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    scores = torch.matmul(q, k.transpose(-1, -2)) * ((d // num_heads) ** -0.5)   # :180-182
    probs = torch.softmax(scores, dim=-1)
    ctx = torch.matmul(probs, v).permute(0, 2, 1, 3).reshape(b, s, d)
    x = _lin(sd, lp + "self_attn.projection", ctx) + x           # :200,:257
    h = _ln(sd, lp + "layer_norm2", x, eps)
    h = _lin(sd, lp + "mlp.fc1", h)
    h = F.gelu(h)                                                # exact erf GELU (ACT2FN["gelu"])
    h = _lin(sd, lp + "mlp.fc2", h)
    return h + x


def vit_forward(sd: SD, p: str, pixel_values: Tensor, num_heads: int, eps: float = 1e-6,
                return_all: bool = False):
    """InstructBlipVisionModel.forward xinstructblip.py:515-558 -> last_hidden_state
    (post_layernorm applied to all tokens, :545)."""
    if pixel_values is None:
        raise ValueError("You have to specify pixel_values")       # :532-533
    x = vit_embed(sd, p, pixel_values.float())
    hs = [x]
    for i in range(_count_layers(sd, p + "encoder.layers.")):
        x = vit_layer(sd, f"{p}encoder.layers.{i}.", x, num_heads, eps)
        hs.append(x)
    out = _ln(sd, p + "post_layernorm", x, eps)
    return (out, hs) if return_all else out


# ----------------------------------------------------------------------------
# a11: Q-Former (InstructBLIP with text branch / BLIP-2 queries only)
# ----------------------------------------------------------------------------
def _bert_attn(sd: SD, ap: str, hidden: Tensor, kv_src: Tensor, add_mask: Optional[Tensor],
               nh: int, eps: float, self_key: str) -> Tensor:
    """Multi-head attention + BertSelfOutput (dense, residual, post-LN).
    xinstructblip.py:611-694 + :698-709 (Q-Former);  scores / sqrt(head_dim), additive mask."""
    q = _heads(_lin(sd, f"{ap}{self_key}.query", hidden), nh)
    k = _heads(_lin(sd, f"{ap}{self_key}.key", kv_src), nh)
    v = _heads(_lin(sd, f"{ap}{self_key}.value", kv_src), nh)
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(q.shape[-1])
    if add_mask is not None:
        s = s + add_mask
    ctx = _unheads(torch.matmul(torch.softmax(s, dim=-1), v))
    out = _lin(sd, ap + "output.dense", ctx)
    return _ln(sd, ap + "output.LayerNorm", out + hidden, eps)


def _ffn(sd: SD, lp: str, inter: str, outp: str, x: Tensor, eps: float) -> Tensor:
    h = F.gelu(_lin(sd, lp + inter + ".dense", x))
    h = _lin(sd, lp + outp + ".dense", h)
    return _ln(sd, lp + outp + ".LayerNorm", h + x, eps)


def qformer_forward(sd: SD, p: str, query_tokens: Tensor, image_embeds: Tensor, num_heads: int,
                    input_ids: Optional[Tensor] = None, text_mask: Optional[Tensor] = None,
                    image_mask: Optional[Tensor] = None, cross_freq: int = 2,
                    eps: float = 1e-12) -> Tensor:
    """InstructBlipQFormerModel.forward xinstructblip.py:1122-1242 (input_ids given) or
    Blip2QFormerModel.forward xblip2.py:1063-1174 (input_ids None).  Returns the full
    sequence output [B, 32(+Lt), H]; callers slice [:, :32]."""
    b = image_embeds.shape[0]
    q = query_tokens.expand(b, -1, -1)
    nq = q.shape[1]
    if input_ids is not None:
        # InstructBlipQFormerEmbeddings.forward :1018-1046
        lt = input_ids.shape[1]
        emb = sd[p + "embeddings.word_embeddings.weight"][input_ids]
        emb = emb + sd[p + "embeddings.position_embeddings.weight"][:lt][None]
        x = torch.cat([q, emb], dim=1)
        x = _ln(sd, p + "embeddings.layernorm", x, eps)
        m = torch.cat([torch.ones(b, nq, dtype=torch.float32),
                       (text_mask if text_mask is not None else torch.ones(b, lt)).float()], dim=1)
    else:
        x = _ln(sd, p + "layernorm", q, eps)                       # xblip2.py:1108
        m = torch.ones(b, nq, dtype=torch.float32)
    self_mask = (1.0 - m)[:, None, None, :] * -10000.0            # :1118-1119
    if image_mask is None:
        cross_mask = None
    else:                                                          # HF invert_attention_mask
        cross_mask = (1.0 - image_mask.float())[:, None, None, :] * torch.finfo(torch.float32).min
    for i in range(_count_layers(sd, p + "encoder.layer.")):
        lp = f"{p}encoder.layer.{i}."
        att = _bert_attn(sd, lp + "attention.", x, x, self_mask, num_heads, eps, "attention")
        qa = att[:, :nq]
        if i % cross_freq == 0:                                   # :802-806, :842-855
            qa = _bert_attn(sd, lp + "crossattention.", qa, image_embeds, cross_mask, num_heads,
                            eps, "attention")
        out = _ffn(sd, lp, "intermediate_query", "output_query", qa, eps)   # :857-862
        if att.shape[1] > nq:                                     # :864-871
            out_t = _ffn(sd, lp, "intermediate", "output", att[:, nq:], eps)
            out = torch.cat([out, out_t], dim=1)
        x = out
    return x


# ----------------------------------------------------------------------------
# a12: frame pooling + language_projection
# ----------------------------------------------------------------------------
def pool_project(sd: SD, proj: str, query_out: Tensor, widths: Sequence[int], mode: str) -> Tensor:
    """mean: eval/utils/model.py:186-195 (and ragged ``widths`` of
    src/models/LSTP_Vicuna_IVT_module.py:244-248; width 0 -> zeros);
    concat: src/models/LSTP_module.py:477-481.  query_out is [sum(widths), 32, H]."""
    nq, h = query_out.shape[1], query_out.shape[2]
    if mode == "mean":
        pooled = torch.zeros(len(widths), nq, h, dtype=query_out.dtype)
        idx = 0
        for i, w in enumerate(widths):
            if w > 0:
                pooled[i] = query_out[idx:idx + w].mean(0)
            idx += w
        return _lin(sd, proj, pooled)
    if mode == "concat":
        y = _lin(sd, proj, query_out)
        assert len(set(widths)) == 1
        return y.reshape(len(widths), -1, y.shape[-1])
    raise ValueError(f"INVALID POOL MODE: {mode}")


# ----------------------------------------------------------------------------
# a3-a7a: Temporal Grounding Bridge (RoPE-BERT)
# ----------------------------------------------------------------------------
def rope_table(n_pos: int, dim: int) -> Tensor:
    """BertSinusoidalPositionalEmbedding._init_weight xropebert.py:149-164:
    float64 numpy -> fp32; [:, :dim/2] = sin, [:, dim/2:] = cos."""
    pos = np.arange(n_pos, dtype=np.float64)[:, None]
    j = np.arange(dim)
    enc = pos / np.power(10000, 2 * (j // 2) / dim)[None, :]
    out = np.zeros((n_pos, dim), dtype=np.float32)
    half = dim // 2
    out[:, :half] = np.sin(enc[:, 0::2]).astype(np.float32)
    out[:, half:] = np.cos(enc[:, 1::2]).astype(np.float32)
    return torch.from_numpy(out)


def apply_rope(table_rows: Tensor, x: Tensor) -> Tensor:
    """apply_rope / apply_rotary_position_embeddings xropebert.py:335-377: interleaved pairs."""
    sin, cos = table_rows.chunk(2, dim=-1)
    sin_pos = torch.stack([sin, sin], dim=-1).reshape(table_rows.shape)
    cos_pos = torch.stack([cos, cos], dim=-1).reshape(table_rows.shape)
    rot = torch.stack([-x[..., 1::2], x[..., ::2]], dim=-1).reshape(x.shape)
    return x * cos_pos + rot * sin_pos


def tgb_flow_embed(sd: SD, p: str, of: Tensor, of_mask: Tensor) -> Tensor:
    """TemporalOFEmbedding.forward xropebert.py:103-129."""
    b, l, c, h, w = of.shape
    tp = p + "temporal_embeddings."
    pw = sd[tp + "projection.weight"]
    x = F.conv2d(of.reshape(-1, c, h, w).float(), pw, sd[tp + "projection.bias"], stride=pw.shape[-1])
    x = x.flatten(2)                                   # [B*L, D, 196]
    x = _lin(sd, tp + "fc", x).view(b, l, -1)          # Linear(196 -> 1)
    d = x.shape[-1]
    x = torch.cat([sd[tp + "bos"].expand(b, 1, -1), x, torch.zeros(b, 1, d)], dim=1)
    ends = of_mask.sum(dim=1) - 1
    x[torch.arange(b), ends] = sd[tp + "eos"]
    x = x + sd[tp + "frame_pos_embed.weight"][: x.shape[1]][None]
    return F.layer_norm(x, (d,), sd[tp + "ln.weight"], sd[tp + "ln.bias"], 1e-5)


def tgb_text_embed(sd: SD, p: str, input_ids: Tensor, eps: float = 1e-12) -> Tensor:
    """RopeBertEmbeddings.forward xropebert.py:190-208 (no absolute position add)."""
    e = sd[p + "embeddings.word_embeddings.weight"][input_ids]
    e = e + sd[p + "embeddings.token_type_embeddings.weight"][0]
    return _ln(sd, p + "embeddings.LayerNorm", e, eps)


def _tgb_attn(sd: SD, ap: str, hidden: Tensor, kv_src: Tensor, add_mask: Tensor, nh: int,
              q_pos: Tensor, k_pos: Tensor, eps: float) -> Tensor:
    """RopeBertSelfAttention.forward xropebert.py:243-332 + BertSelfOutput :542-553."""
    q = _heads(_lin(sd, ap + "self.query", hidden), nh)
    k = _heads(_lin(sd, ap + "self.key", kv_src), nh)
    v = _heads(_lin(sd, ap + "self.value", kv_src), nh)
    q = apply_rope(q_pos[None, None], q)
    k = apply_rope(k_pos[None, None], k)
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(q.shape[-1]) + add_mask
    ctx = _unheads(torch.matmul(torch.softmax(s, dim=-1), v))
    out = _lin(sd, ap + "output.dense", ctx)
    return _ln(sd, ap + "output.LayerNorm", out + hidden, eps)


def tgb_mode_layers(mode: str, fusion_layer: int, n_layers: int) -> Tuple[int, int]:
    """BertEncoder.forward mode switch xropebert.py:621-634."""
    if mode in ("vision", "text"):
        return 0, fusion_layer
    if mode == "fusion":
        return fusion_layer, n_layers
    if mode == "multi_modal":
        return 0, n_layers
    raise ValueError(f"INVALID MODE: {mode}")


def tgb_forward(sd: SD, p: str, of: Tensor, of_mask: Tensor, text_ids: Tensor, text_mask: Tensor,
                mode: str, num_heads: int, fusion_layer: int, eps: float = 1e-12,
                return_all: bool = False):
    """RopeBertModel.forward xropebert.py:1048-1169 with encoder_embeds=of
    -> (sequence_output [B, L+2, D], logits [B, L, 2])."""
    x = tgb_flow_embed(sd, p, of, of_mask)
    t = tgb_text_embed(sd, p, text_ids, eps)
    s = x.shape[1]
    self_mask = (1.0 - of_mask.float())[:, None, None, :] * -10000.0          # :1044-1045
    cross_mask = (1.0 - text_mask.float())[:, None, None, :] * torch.finfo(torch.float32).min  # :1127
    pos = sd[p + "encoder.embed_positions.weight"][:s]                         # :615
    cpos = sd[p + "encoder.c_embed_positions.weight"][: t.shape[1]]            # :616
    n_layers = _count_layers(sd, p + "encoder.layer.")
    lo, hi = tgb_mode_layers(mode, fusion_layer, n_layers)
    hs = [x]
    for i in range(lo, hi):
        lp = f"{p}encoder.layer.{i}."
        a = _tgb_attn(sd, lp + "attention.", x, x, self_mask, num_heads, pos, pos, eps)
        if i >= fusion_layer:                                                  # :442, :466-510
            a = _tgb_attn(sd, lp + "crossattention.", a, t, cross_mask, num_heads, pos, cpos, eps)
        x = _ffn(sd, lp, "intermediate", "output", a, eps)
        hs.append(x)
    logits = _lin(sd, p + "mrc_head", x[:, 1:-1])                              # :1164
    return (x, logits, hs) if return_all else (x, logits)


# ----------------------------------------------------------------------------
# a7: Gumbel top-k span selection; a8: span -> frame index map + gather
# ----------------------------------------------------------------------------
def gumbel_noise(shape, generator: torch.Generator) -> Tensor:
    """g = -log(E), E ~ Exp(1): what F.gumbel_softmax draws (Appendix B)."""
    return -torch.empty(shape, dtype=torch.float32).exponential_(generator=generator).log()


def span_select(logits: Tensor, noise: Tensor, tau: float = 0.5) -> Tensor:
    """eval/utils/model.py:101-113.  logits [B, L, 2]; noise [draws, 2B, L] (injected).
    Returns idx [draws, 2B] int64: rows 0..B-1 = start, B..2B-1 = end.
    softmax is monotone, so argmax(softmax((l+g)/tau)) == first argmax of (l+g)/tau."""
    start, end = logits.split(1, dim=-1)
    cat = torch.cat([start, end], dim=0).squeeze(-1)        # [2B, L]
    out = []
    for d in range(noise.shape[0]):
        y = (cat + noise[d]) / tau
        out.append(torch.argmax(torch.softmax(y, dim=1), dim=1))
    return torch.stack(out)


def _f32(x) -> np.float32:
    return np.float32(x)


def span_to_frames(starts: Sequence[int], ends: Sequence[int], V: int, N: int, nframe: int,
                   variant: str) -> List[int]:
    """eval/utils/model.py:124-150 (variant 'A': :135) and :337-365 (variant 'B': :350).
    starts/ends: the per-draw indices for ONE clip.  Rounding rules: SURVEY Appendix B."""
    cand = set()
    for s, e in zip(starts, ends):
        s, e = int(s), int(e)
        if s >= V or e >= V or (s == 0 and e == 0):
            # python-int branch: float64 arithmetic
            s64, e64 = 0, V - 1
            if variant == "A":
                a, b = int(s64 / V * N), int(e64 / V * N)
            else:
                a, b = int(s64 * (N - 1) / (V - 1)), int(e64 * (N - 1) / (V - 1))
        elif variant == "A":
            # 0-dim int64 tensor / python int -> float32 true-divide, then float32 multiply
            a = int(_f32(_f32(s) / _f32(V)) * _f32(N))
            b = int(_f32(_f32(e) / _f32(V)) * _f32(N))
        else:
            # int64 multiply, one float32 divide
            a = int(_f32(s * (N - 1)) / _f32(V - 1))
            b = int(_f32(e * (N - 1)) / _f32(V - 1))
        cand |= set(range(a, b))
    cand = sorted(cand)
    return subsample_candidates(cand, N, nframe)


def subsample_candidates(cand: List[int], N: int, nframe: int) -> List[int]:
    """eval/utils/model.py:140-149: empty -> range(N); duplicate-double; linspace midpoint."""
    if cand == []:
        cand = list(range(N))
    while len(cand) < nframe:
        cand = [xx for x in cand for xx in (x, x)]
    if len(cand) > nframe:
        intv = np.linspace(start=0, stop=len(cand), num=nframe + 1).astype(int)
        cand = [cand[(intv[x] + intv[x + 1] - 1) // 2] for x in range(len(intv) - 1)]
    assert len(cand) == nframe
    return cand


def sample_frames(num_frames: int, video_len: int, sample: str = "uniform", fix_start: float = -1) -> List[int]:
    """src/data/components/util.py:20-34 (deterministic branches).  NB eval's get_frames calls
    sample_frames(32, vlen, "uniform", 1.) (eval/utils/builder_utils.py:138): fix_start=1 >= 0
    wins over 'uniform', so that caller picks intv[i]+1, not the midpoint."""
    if num_frames >= video_len:
        return list(range(video_len))
    intv = np.linspace(start=0, stop=video_len, num=num_frames + 1).astype(int)
    if sample == "rand":
        raise NotImplementedError("random branch is not part of the deterministic oracle")
    if fix_start >= 0:
        return [int(intv[i]) + int(fix_start) for i in range(len(intv) - 1)]
    if sample == "uniform":
        return [int((intv[i] + intv[i + 1] - 1) // 2) for i in range(len(intv) - 1)]
    raise NotImplementedError


def candidate_frame_ids(vlen: int, n_cand: int = 32) -> List[int]:
    """eval/utils/builder_utils.py:131-139: duplicate-double to >= n_cand, then sample_frames(.., 'uniform', 1.)."""
    indices = list(range(vlen))
    while len(indices) < n_cand:
        indices = [f for ind in indices for f in (ind, ind)]
    ids = sample_frames(n_cand, len(indices), "uniform", 1.0)
    return [indices[i] for i in ids]


CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def preprocess_frames(raw: Tensor, size: int = 224) -> Tensor:
    """The transform chain of get_frames (eval/utils/builder_utils.py:118-128) on decoded frames raw [T, H0, W0, 3] uint8:
    read_videos_av's layout change (:86: THWC -> CTHW float), ResizeVideo = F.interpolate(bilinear, align_corners=False)
    (src/gadgets/functional_video.py:33-41), ToUint8 (transforms.py:217-218), ToTHWC + ToTensorVideo = float / 255
    (functional_video.py:76-90), NormalizeVideo (:93-110), then CTHW -> TCHW (builder_utils.py:127)."""
    clip = raw.permute(3, 0, 1, 2).float()                                            # C T H W
    clip = F.interpolate(clip, size=(size, size), mode="bilinear", align_corners=False)
    clip = clip.to(torch.uint8)
    clip = clip.permute(1, 2, 3, 0)                                                    # T H W C
    clip = clip.float().permute(3, 0, 1, 2) / 255.0                                    # C T H W
    mean = torch.as_tensor(CLIP_MEAN, dtype=clip.dtype)
    std = torch.as_tensor(CLIP_STD, dtype=clip.dtype)
    clip = (clip - mean[:, None, None, None]) / std[:, None, None, None]
    return clip.permute(1, 0, 2, 3).contiguous()                                       # T C H W


def get_frames(raw: Tensor, size: int = 224) -> Tuple[Tensor, Tensor]:
    """eval/utils/builder_utils.py:117-144 from decoded frames: (frames [32, 3, S, S], flow_frames [T, 3, S, S])."""
    flow_frames = preprocess_frames(raw, size)
    return flow_frames[candidate_frame_ids(flow_frames.shape[0])], flow_frames


def gather_frames(pixel_values: Tensor, idx: Tensor) -> Tensor:
    """eval/utils/model.py:122,151: index_select into a zero-initialised fp32 buffer.
    pixel_values [B, N, 3, H, W]; idx [B, nframe] -> [B, nframe, 3, H, W] fp32."""
    out = torch.zeros((pixel_values.shape[0], idx.shape[1]) + tuple(pixel_values.shape[2:]),
                      dtype=torch.float32)
    for j in range(pixel_values.shape[0]):
        out[j] = torch.index_select(pixel_values[j], 0, idx[j]).float()
    return out


# ----------------------------------------------------------------------------
# a2: RAFT optical flow (functional restatement)
# ----------------------------------------------------------------------------
def _conv(sd: SD, name: str, x: Tensor, stride=1, padding=0) -> Tensor:
    return F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"], stride=stride, padding=padding)


def _raft_norm(sd: SD, name: str, x: Tensor, kind: str) -> Tensor:
    if kind == "instance":       # nn.InstanceNorm2d defaults: no affine, no running stats
        return F.instance_norm(x, eps=1e-5)
    return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"],
                        sd[name + ".weight"], sd[name + ".bias"], training=False, eps=1e-5)


def _raft_resblock(sd: SD, p: str, x: Tensor, kind: str, stride: int) -> Tensor:
    """ResidualBlock.forward raft_utils/extractor.py:48-56."""
    y = F.relu(_raft_norm(sd, p + "norm1", _conv(sd, p + "conv1", x, stride, 1), kind))
    y = F.relu(_raft_norm(sd, p + "norm2", _conv(sd, p + "conv2", y, 1, 1), kind))
    if stride != 1:
        x = _raft_norm(sd, p + "norm3", _conv(sd, p + "downsample.0", x, stride, 0), kind)
    return F.relu(x + y)


def raft_encoder(sd: SD, p: str, x: Tensor, kind: str) -> Tensor:
    """BasicEncoder.forward raft_utils/extractor.py:161-189."""
    x = F.relu(_raft_norm(sd, p + "norm1", _conv(sd, p + "conv1", x, 2, 3), kind))
    for li, stride in (("layer1", 1), ("layer2", 2), ("layer3", 2)):
        x = _raft_resblock(sd, f"{p}{li}.0.", x, kind, stride)
        x = _raft_resblock(sd, f"{p}{li}.1.", x, kind, 1)
    return _conv(sd, p + "conv2", x)


def raft_corr_pyramid(fmap1: Tensor, fmap2: Tensor, levels: int = 4) -> List[Tensor]:
    """CorrBlock.__init__/corr raft_utils/corr.py:12-27,52-60."""
    b, d, h, w = fmap1.shape
    corr = torch.matmul(fmap1.view(b, d, h * w).transpose(1, 2), fmap2.view(b, d, h * w))
    corr = corr / torch.sqrt(torch.tensor(d).float())
    corr = corr.reshape(b * h * w, 1, h, w)
    pyr = [corr]
    for _ in range(levels - 1):
        corr = F.avg_pool2d(corr, 2, stride=2)
        pyr.append(corr)
    return pyr


def raft_corr_lookup(pyr: List[Tensor], coords: Tensor, radius: int = 4) -> Tensor:
    """CorrBlock.__call__ raft_utils/corr.py:29-50 with bilinear_sampler utils.py:58-72.
    NB the reference's delta is stack(meshgrid(dy, dx)) added to (x, y) coords: the first
    channel (x) receives the *row* offset of the 9x9 window -- replicated as is."""
    r = radius
    coords = coords.permute(0, 2, 3, 1)
    b, h1, w1, _ = coords.shape
    dx = torch.linspace(-r, r, 2 * r + 1)
    dy = torch.linspace(-r, r, 2 * r + 1)
    delta = torch.stack(torch.meshgrid(dy, dx, indexing="ij"), dim=-1).view(1, 2 * r + 1, 2 * r + 1, 2)
    out = []
    for i, corr in enumerate(pyr):
        cl = coords.reshape(b * h1 * w1, 1, 1, 2) / 2 ** i + delta
        hh, ww = corr.shape[-2:]
        xg = 2 * cl[..., 0:1] / (ww - 1) - 1
        yg = 2 * cl[..., 1:2] / (hh - 1) - 1
        samp = F.grid_sample(corr, torch.cat([xg, yg], dim=-1), align_corners=True)
        out.append(samp.view(b, h1, w1, -1))
    return torch.cat(out, dim=-1).permute(0, 3, 1, 2).contiguous().float()


def _raft_update(sd: SD, p: str, net: Tensor, inp: Tensor, corr: Tensor, flow: Tensor, want_mask: bool):
    """BasicUpdateBlock.forward update.py:134-144; BasicMotionEncoder :88-97; SepConvGRU :50-65."""
    e = p + "encoder."
    cor = F.relu(_conv(sd, e + "convc1", corr))
    cor = F.relu(_conv(sd, e + "convc2", cor, 1, 1))
    flo = F.relu(_conv(sd, e + "convf1", flow, 1, 3))
    flo = F.relu(_conv(sd, e + "convf2", flo, 1, 1))
    out = F.relu(_conv(sd, e + "conv", torch.cat([cor, flo], dim=1), 1, 1))
    x = torch.cat([inp, out, flow], dim=1)
    g = p + "gru."
    for sfx, pad in (("1", (0, 2)), ("2", (2, 0))):
        hx = torch.cat([net, x], dim=1)
        z = torch.sigmoid(_conv(sd, g + "convz" + sfx, hx, 1, pad))
        r = torch.sigmoid(_conv(sd, g + "convr" + sfx, hx, 1, pad))
        q = torch.tanh(_conv(sd, g + "convq" + sfx, torch.cat([r * net, x], dim=1), 1, pad))
        net = (1 - z) * net + z * q
    delta = _conv(sd, p + "flow_head.conv2", F.relu(_conv(sd, p + "flow_head.conv1", net, 1, 1)), 1, 1)
    mask = None
    if want_mask:
        mask = 0.25 * _conv(sd, p + "mask.2", F.relu(_conv(sd, p + "mask.0", net, 1, 1)))
    return net, mask, delta


def raft_forward(sd: SD, p: str, image1: Tensor, image2: Tensor, iters: int = 20, flow_init: Optional[Tensor] = None) -> Tensor:
    """RAFT.forward xraft.py:102-156, test_mode=True -> last flow_up [B, 2, H, W].
    Only the last iteration's upsample is materialised (earlier ones are discarded by
    the reference, :154-156)."""
    image1 = (2 * (image1 / 255.0) - 1.0).contiguous()
    image2 = (2 * (image2 / 255.0) - 1.0).contiguous()
    b = image1.shape[0]
    f = raft_encoder(sd, p + "fnet.", torch.cat([image1, image2], dim=0), "instance")
    fmap1, fmap2 = f[:b].float(), f[b:].float()
    pyr = raft_corr_pyramid(fmap1, fmap2)
    c = raft_encoder(sd, p + "cnet.", image1, "batch")
    net, inp = torch.tanh(c[:, :128]), torch.relu(c[:, 128:])
    n, _, h, w = image1.shape
    ys, xs = torch.meshgrid(torch.arange(h // 8), torch.arange(w // 8), indexing="ij")
    coords0 = torch.stack([xs, ys], dim=0).float()[None].repeat(n, 1, 1, 1)   # coords_grid utils.py:75-78
    coords1 = coords0.clone()
    if flow_init is not None:                                                   # xraft.py:131-132
        coords1 = coords1 + flow_init
    mask = None
    for it in range(iters):
        corr = raft_corr_lookup(pyr, coords1)
        net, mask, delta = _raft_update(sd, p + "update_block.", net, inp, corr, coords1 - coords0,
                                        want_mask=(it == iters - 1))
        coords1 = coords1 + delta
    flow = coords1 - coords0
    # upsample_flow xraft.py:88-99
    hh, ww = flow.shape[-2:]
    m = torch.softmax(mask.view(n, 1, 9, 8, 8, hh, ww), dim=2)
    up = F.unfold(8 * flow, [3, 3], padding=1).view(n, 2, 9, 1, 1, hh, ww)
    up = torch.sum(m * up, dim=2).permute(0, 1, 4, 2, 5, 3)
    return up.reshape(n, 2, 8 * hh, 8 * ww)


def raft_clip_flow(sd: SD, p: str, flow_frames: Tensor, iters: int = 20) -> Tensor:
    """eval/utils/model.py:76-84: per clip, flow between consecutive frames, last flow
    repeated so the sequence has T entries.  flow_frames [B, T, 3, H, W] -> [B, T, 2, H, W].
    InputPadder (xraft.py:30-48) pads to a multiple of 8 with replicate; 224 % 8 == 0 -> no-op."""
    outs = []
    for ff in flow_frames:
        ht, wd = ff.shape[-2:]
        pad_ht = (((ht // 8) + 1) * 8 - ht) % 8
        pad_wd = (((wd // 8) + 1) * 8 - wd) % 8
        pad = [pad_wd // 2, pad_wd - pad_wd // 2, pad_ht // 2, pad_ht - pad_ht // 2]
        if any(pad):
            ff = F.pad(ff, pad, mode="replicate")
        fl = raft_forward(sd, p, ff[:-1], ff[1:], iters)
        outs.append(torch.cat([fl, fl[-1:]], dim=0))
    return torch.stack(outs)


# ----------------------------------------------------------------------------
# end-to-end prefix: eval/utils/model.py LSTP.generate / LSTP_blip2.generate up to inputs_embeds
# ----------------------------------------------------------------------------
def lstp_prefix(sd: SD, *, arch: str, frames: Tensor, nframe: int, sampler_ids: Tensor,
                sampler_mask: Tensor, noise: Tensor, vit_heads: int, qf_heads: int,
                tgb_heads: int, fusion_layer: int, of: Optional[Tensor] = None,
                flow_frames: Optional[Tensor] = None, qformer_ids: Optional[Tensor] = None,
                qformer_mask: Optional[Tensor] = None, raft_iters: int = 20, pool: str = "mean"):
    """arch 'instructblip' -> LSTP.generate (eval/utils/model.py:47-214: multi_modal, V=T, map A);
    arch 'blip2' -> LSTP_blip2.generate (:266-425: fusion, V=T, map B).
    Either ``of`` (precomputed flow, the batch["of"] contract of src/models/LSTP_SF_module.py:476)
    or ``flow_frames`` (RAFT inline) is given.  Returns a dict of every stage boundary."""
    b = sampler_ids.shape[0]
    n = frames.shape[0] // b
    pix = frames.view(b, n, *frames.shape[1:])
    if of is None:
        of = raft_clip_flow(sd, "of_extractor.", flow_frames, raft_iters)
    T = of.shape[1]
    of_mask = torch.ones(b, T + 2, dtype=torch.long)
    mode, variant = ("multi_modal", "A") if arch == "instructblip" else ("fusion", "B")
    seq, logits = tgb_forward(sd, "temporal_encoder.", of, of_mask, sampler_ids, sampler_mask,
                              mode, tgb_heads, fusion_layer)
    sel = span_select(logits, noise)                      # [draws, 2B]
    idx = torch.tensor([span_to_frames(sel[:, j].tolist(), sel[:, b + j].tolist(), T, n, nframe, variant)
                        for j in range(b)], dtype=torch.long)
    sampled = gather_frames(pix, idx).view(b * nframe, *frames.shape[1:])
    img = vit_forward(sd, "model.vision_model.", sampled, vit_heads)
    if arch == "instructblip":
        qi = torch.repeat_interleave(qformer_ids, nframe, 0)
        qm = torch.repeat_interleave(qformer_mask, nframe, 0)
        qo = qformer_forward(sd, "model.qformer.", sd["model.query_tokens"], img, qf_heads, qi, qm,
                             torch.ones(img.shape[:2]))
    else:
        qo = qformer_forward(sd, "model.qformer.", sd["model.query_tokens"], img, qf_heads,
                             image_mask=torch.ones(img.shape[:2]))
    nq = sd["model.query_tokens"].shape[1]
    qo = qo[:, :nq]
    prefix = pool_project(sd, "model.language_projection", qo, [nframe] * b, pool)
    return dict(of=of, tgb_seq=seq, tgb_logits=logits, sel=sel, cand_index=idx, sampled=sampled,
                image_embeds=img, query_out=qo, prefix=prefix)


# ----------------------------------------------------------------------------
# a14: loss side of the LoRA training step (config C5)
# ----------------------------------------------------------------------------
def concat_text_input_output(input_ids: Tensor, input_atts: Tensor, output_ids: Tensor, output_atts: Tensor):
    """src/models/LSTP_Vicuna_IVT_module.py:692-718: per row, the answer (without its first token) is spliced in
    right after the question's real tokens, the question's padding moves to the end."""
    ids, atts, lens = [], [], []
    for i in range(input_ids.size(0)):
        n = int(input_atts[i].sum())
        lens.append(n)
        ids.append(torch.cat([input_ids[i][:n], output_ids[i][1:], input_ids[i][n:]]))
        atts.append(torch.cat([input_atts[i][:n], output_atts[i][1:], input_atts[i][n:]]))
    return {"input_ids": torch.stack(ids), "attention_mask": torch.stack(atts)}, lens


def lm_labels(llm_ids: Tensor, input_len: Sequence[int], pad_id: int, prefix_len: int) -> Tensor:
    """LSTP_Vicuna_IVT_module.py:284-291: pad -> -100, the question part -> -100, -100 for the visual prefix."""
    labels = llm_ids.masked_fill(llm_ids == pad_id, -100)
    for i, l in enumerate(input_len):
        labels[i][:l] = -100
    empty = torch.full((llm_ids.shape[0], prefix_len), -100, dtype=torch.long)
    return torch.cat([empty, labels], dim=1)


def shifted_cross_entropy(logits: Tensor, labels: Tensor) -> Tensor:
    """LSTP_Vicuna_IVT_module.py:297-299, :325-326: position t scores token t+1; mean over targets != -100."""
    V = logits.shape[-1]
    shift_logits = logits[..., :-1, :].contiguous()
    shift_labels = labels[..., 1:].contiguous()
    return F.cross_entropy(shift_logits.view(-1, V).float(), shift_labels.view(-1), reduction="mean")


def lora_linear(x: Tensor, weight: Tensor, bias: Optional[Tensor], lora_A: Tensor, lora_B: Tensor, alpha: float, r: int) -> Tensor:
    """peft 0.4.0 (pinned, requirement.txt:233; not vendored) tuners/lora.py Linear.forward in eval mode (dropout
    off): F.linear(x, W, b) + lora_B(lora_A(x)) * (alpha / r)."""
    return F.linear(x, weight, bias) + F.linear(F.linear(x, lora_A), lora_B) * (alpha / r)


def cosine_schedule_lambda(step: int, num_warmup_steps: int, num_training_steps: int, num_cycles: float = 0.5) -> float:
    """transformers.get_cosine_schedule_with_warmup's lr lambda (pinned 4.36.0), as configured by
    LSTP_Vicuna_IVT_module.py:661-665 (max_steps = trainer.max_steps, which is -1 under max_epochs)."""
    if step < num_warmup_steps:
        return float(step) / float(max(1, num_warmup_steps))
    progress = float(step - num_warmup_steps) / float(max(1, num_training_steps - num_warmup_steps))
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * progress)))


# ----------------------------------------------------------------------------
# f4: self-refinement pseudo labels
# ----------------------------------------------------------------------------
def rouge_n_list(gold: List[str], pred: List[str], ignore=(",", ".")) -> List[float]:
    """src/gadgets/my_metrics.py:131-157 (list form, ignore given): per pair hits / counted gold tokens, then divided
    by len(gold) -- the number of PAIRS (:154-155)."""
    out = []
    for g, p in zip(gold, pred):
        g, p = g.split(), p.split()
        toks = [t for t in g if t not in ignore]
        r = (sum(1 for t in toks if t in p) / len(toks)) if toks else 0
        out.append(r / len(gold) if len(gold) > 0 else r)
    return out


def monotone_span(score: Sequence[float]) -> Tuple[int, int]:
    """src/models/LSTP_SF_module.py:246-261 restated as the definition it implements: over all windows [s, e] of the
    scores, maximise (e - s + 1) * min(score[s..e]); the stack pops candidates in increasing right edge, so among
    equal areas the first one popped wins; if no window has positive area the default (0, n - 1) stays."""
    n = len(score)
    bs, best = 0, (0, n - 1)
    sc = [0] + list(score) + [0]
    stack: List[int] = []
    for i in range(len(sc)):
        while stack and sc[stack[-1]] > sc[i]:
            tmp = stack.pop()
            area = (i - stack[-1] - 1) * sc[tmp]
            if area > bs:
                bs, best = area, (stack[-1], i - 2)
        stack.append(i)
    return best


def pseudo_span_targets(scores: Tensor, flow_lengths: Sequence[int]) -> Tuple[List[int], List[int]]:
    """LSTP_SF_module.py:246-265."""
    b, n = scores.shape
    st, en = [], []
    for i in range(b):
        s, e = monotone_span(scores[i].tolist())
        st.append(int(s / (n - 1) * (flow_lengths[i] - 1)))
        en.append(int(e / (n - 1) * (flow_lengths[i] - 1)))
    return st, en


def mrc_loss(of_logits: Tensor, start_targets: Tensor, end_targets: Tensor) -> Tensor:
    """LSTP_SF_module.py:285-298."""
    L_ = of_logits.shape[1]
    s = F.cross_entropy(of_logits[..., 0], start_targets.clamp(0, L_), ignore_index=L_)
    e = F.cross_entropy(of_logits[..., 1], end_targets.clamp(0, L_), ignore_index=L_)
    return (s + e) / 2
