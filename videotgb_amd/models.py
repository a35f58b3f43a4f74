"""Host-side mirror of the reference's module interface for the hot path.

The classes keep the reference's attribute names, call signatures, return shapes, error
messages and state_dict keys, so a Lightning checkpoint's
``state_dict`` loads with ``strict=True`` and the reference's drivers (eval/inference.py,
src/models/*_module.py) can call them unchanged -- but every forward goes through the C ABI
of libvtgb.so (videotgb_amd.ops).  There is no torch fallback: on a machine without the HIP
library or without a GPU the forwards raise.

  VisionModel        <- InstructBlipVisionModel / Blip2VisionModel   (xinstructblip.py:498-558)
  QFormer            <- InstructBlipQFormerModel / Blip2QFormerModel (xinstructblip.py:1049-1242, xblip2.py:988-1174)
  LanguageProjection <- nn.Linear language_projection                (xinstructblip.py:1266)
  TemporalEncoder    <- RopeBertModel                                (xropebert.py:929-1178)
  Raft               <- RAFT                                        (xraft.py:51-156, raft_utils/*)
  LSTP / LSTP_blip2  <- eval/utils/model.py:19-235 / :238-445
"""
from __future__ import annotations

import os
from types import SimpleNamespace
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops, synth
from ._lib import VtgbError

Tensor = torch.Tensor


class ModelOutput(tuple):
    """Minimal stand-in for HF's BaseModelOutput: ``out.last_hidden_state`` and ``out[0]``."""

    def __new__(cls, last_hidden_state, pooler_output=None):
        self = super().__new__(cls, (last_hidden_state, pooler_output))
        self.last_hidden_state = last_hidden_state
        self.pooler_output = pooler_output
        return self


class ParamTree(nn.Module):
    """Registers parameters/buffers under the reference's dotted names so state_dict keys match."""

    _BUFFERS = ("position_ids", "running_mean", "running_var", "num_batches_tracked")

    def __init__(self, shapes: synth.Shapes, strip: str = ""):
        super().__init__()
        for key, shape in shapes.items():
            assert key.startswith(strip), (key, strip)
            parts = key[len(strip):].split(".")
            mod = self
            for p in parts[:-1]:
                if p not in mod._modules:
                    mod.add_module(p, nn.Module())
                mod = mod._modules[p]
            if parts[-1] in self._BUFFERS:
                dt = torch.long if parts[-1] in ("position_ids", "num_batches_tracked") else torch.float32
                buf = torch.zeros(shape, dtype=dt)
                if parts[-1] == "position_ids":
                    buf = torch.arange(shape[-1]).expand(shape).clone()
                mod.register_buffer(parts[-1], buf)
            else:
                mod.register_parameter(parts[-1], nn.Parameter(torch.zeros(shape), requires_grad=False))


class _Stage(nn.Module):
    """Base of the HIP-backed stages: owns a ParamTree and the packed weight table built from it."""

    def __init__(self, shapes: synth.Shapes, strip: str, compute_dtype="bf16"):
        super().__init__()
        tree = ParamTree(shapes, strip)
        for name, child in list(tree._modules.items()):
            self.add_module(name, child)
        for name, p in list(tree._parameters.items()):
            self.register_parameter(name, p)
        for name, b in list(tree._buffers.items()):
            self.register_buffer(name, b)
        self.code = ops.dtype_code(compute_dtype)
        self._table_cache = None
        self._table_key = None

    # The packed (bf16) weight table is a cache of the parameters: keyed on (in-place version counter, storage address) of EVERY
    # parameter and buffer, trainable or frozen -- an optimizer step, a parent's load_state_dict (nn.Module calls the child's
    # _load_from_state_dict, not its load_state_dict), Lightning restoring a checkpoint, or `param.data = ...` all invalidate it
    # and nothing else does: no per-forward repack, no stale table (round-2 / round-3 ADVICE).  A tuple compare per forward.
    def _versions(self):
        return tuple((p._version, p.data_ptr()) for p in self.parameters()) + tuple((b._version, b.data_ptr()) for b in self.buffers())

    @property
    def _table(self):
        if self._table_cache is not None and self._versions() != self._table_key:
            self._table_cache = None
        return self._table_cache

    @_table.setter
    def _table(self, value):
        self._table_cache = value
        self._table_key = self._versions() if value is not None else None

    def _trainable(self) -> bool:
        return any(p.requires_grad for p in self.parameters())

    def _apply(self, fn, *a, **k):
        self._table = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._table = None
        return super().load_state_dict(*a, **k)

    def set_compute_dtype(self, compute_dtype):
        self.code = ops.dtype_code(compute_dtype)
        self._table = None
        return self

    def _sd(self) -> Dict[str, Tensor]:
        return {k: v for k, v in self.state_dict().items()}

    @property
    def device(self):
        return next(self.parameters()).device


class VisionModel(_Stage):
    def __init__(self, cfg: synth.VitCfg, compute_dtype="bf16"):
        super().__init__(synth.vit_shapes(cfg, ""), "", compute_dtype)
        self.cfg = cfg

    def table(self) -> ops.VitWeights:
        if self._table is None:
            self._table = ops.VitWeights(self._sd(), "", self.code, self.cfg.heads, self.cfg.eps)
        return self._table

    @torch.no_grad()
    def forward(self, pixel_values: Optional[Tensor] = None, output_attentions=None, output_hidden_states=None,
                return_dict: Optional[bool] = None, act_output: bool = False):
        if pixel_values is None:
            raise ValueError("You have to specify pixel_values")
        out32, outa = ops.vit_forward(self.table(), pixel_values, want_f32=not act_output, want_act=act_output)
        h = outa if act_output else out32
        return ModelOutput(h, h[:, 0])


class QFormer(_Stage):
    """Returns the query rows [n, n_query, hidden]; the reference returns [n, n_query + Lt, hidden] and every
    caller slices [:, :n_query] (eval/utils/model.py:176), which is a no-op on this output."""

    def __init__(self, cfg: synth.QFormerCfg, compute_dtype="bf16"):
        super().__init__(synth.qformer_shapes(cfg, ""), "", compute_dtype)
        self.cfg = cfg

    def table(self) -> ops.QFormerWeights:
        if self._table is None:
            self._table = ops.QFormerWeights(self._sd(), "", self.code, self.cfg.heads, self.cfg.cross_freq, self.cfg.eps)
        return self._table

    @torch.no_grad()
    def forward(self, input_ids=None, attention_mask=None, position_ids=None, query_embeds=None, head_mask=None,
                encoder_hidden_states=None, encoder_attention_mask=None, return_dict=None, **_):
        if input_ids is None and query_embeds is None:
            raise ValueError("You have to specify query_embeds when input_ids is None")
        if encoder_hidden_states is None:
            raise ValueError("encoder_hidden_states must be given for cross-attention layers")
        nq = query_embeds.shape[1]
        text_mask = None
        if input_ids is not None and attention_mask is not None:
            text_mask = attention_mask[:, nq:]
        # (a given encoder_attention_mask goes to the kernel as is -- no host-side "is it all ones" read; the path's own
        # callers pass None for the all-ones case)
        q = ops.qformer_forward(self.table(), query_embeds[0], encoder_hidden_states, input_ids, text_mask,
                                encoder_attention_mask)
        return ModelOutput(q, q[:, 0])


class LanguageProjection(nn.Module):
    """nn.Linear(768 -> LLM hidden) whose forward is the MFMA GEMM (fp32 output)."""

    def __init__(self, in_features: int, out_features: int, compute_dtype="bf16"):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(out_features, in_features), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(out_features), requires_grad=False)
        self.code = ops.dtype_code(compute_dtype)
        self._packed = None
        self._packed_key = None

    def _apply(self, fn, *a, **k):
        self._packed = None
        return super()._apply(fn, *a, **k)

    def packed(self) -> Tensor:
        key = (self.weight._version, self.weight.data_ptr(), self.code)     # repacked when (and only when) the weight changed
        if self._packed is None or self._packed_key != key:
            self._packed = ops.pack_weight(self.weight.data, self.code)
            self._packed_key = key
        return self._packed

    @torch.no_grad()
    def forward(self, x: Tensor) -> Tensor:
        shp = x.shape
        y = ops.pool_project(x.reshape(1, -1, shp[-1]).float(), [1], self.packed(), self.bias.data, "concat", self.code)
        return y.reshape(*shp[:-1], -1)

    @torch.no_grad()
    def pool(self, query_out: Tensor, widths: Sequence[int], mode: str) -> Tensor:
        """Fused frame pooling + projection (K11): mean -> [n_clips, 32, H], concat -> [n_clips, w*32, H]."""
        return ops.pool_project(query_out, widths, self.packed(), self.bias.data, mode, self.code)


class TemporalEncoder(_Stage):
    def __init__(self, cfg: synth.TgbCfg, compute_dtype="bf16"):
        super().__init__(synth.tgb_shapes(cfg, ""), "", compute_dtype)
        self.cfg = cfg
        with torch.no_grad():
            self.encoder.embed_positions.weight.copy_(synth.rope_table(cfg.max_pos, cfg.hidden // cfg.heads))
            self.encoder.c_embed_positions.weight.copy_(synth.rope_table(cfg.max_pos, cfg.hidden // cfg.heads))

    def table(self) -> ops.TgbWeights:
        if self._table is None:
            self._table = ops.TgbWeights(self._sd(), "", self.code, self.cfg.heads, self.cfg.fusion_layer, self.cfg.eps)
        return self._table

    @torch.no_grad()
    def forward(self, input_ids=None, attention_mask=None, token_type_ids=None, head_mask=None, inputs_embeds=None,
                encoder_embeds=None, encoder_hidden_states=None, encoder_attention_mask=None, return_dict=None,
                mode="multi_modal", **_):
        if encoder_embeds is None:
            raise ValueError("You have to specify either input_ids or inputs_embeds or encoder_embeds")
        if mode not in ("vision", "text", "fusion", "multi_modal"):
            raise ValueError(f"INVALID MODE: {mode}")
        b, l = encoder_embeds.shape[:2]
        if attention_mask is None:
            attention_mask = torch.ones(b, l + 2, dtype=torch.long, device=encoder_embeds.device)
        if encoder_attention_mask is None:
            encoder_attention_mask = torch.ones_like(encoder_hidden_states)
        return ops.tgb_forward(self.table(), encoder_embeds, attention_mask, encoder_hidden_states, encoder_attention_mask, mode)


class InputPadder:
    """Pads images such that dimensions are divisible by 8 (xraft.py:30-48)."""

    def __init__(self, dims, mode="sintel"):
        self.ht, self.wd = dims[-2:]
        pad_ht = (((self.ht // 8) + 1) * 8 - self.ht) % 8
        pad_wd = (((self.wd // 8) + 1) * 8 - self.wd) % 8
        if mode == "sintel":
            self._pad = [pad_wd // 2, pad_wd - pad_wd // 2, pad_ht // 2, pad_ht - pad_ht // 2]
        else:
            self._pad = [pad_wd // 2, pad_wd - pad_wd // 2, 0, pad_ht]

    def pad(self, x):
        return F.pad(x, self._pad, mode="replicate") if any(self._pad) else x

    def unpad(self, x):
        ht, wd = x.shape[-2:]
        c = [self._pad[2], ht - self._pad[3], self._pad[0], wd - self._pad[1]]
        return x[..., c[0]:c[1], c[2]:c[3]]


class Raft(nn.Module):
    """RAFT-large (xraft.py:51-156) with the reference's parameter names; every forward runs in libvtgb.so:
    vtgb_raft_encoder (fnet, cnet) -> vtgb_raft_corr (all-pairs correlation + pyramid) -> vtgb_raft_update (refinement
    loop, mask head, convex upsample).  ``compute_dtype``: "bf16" = MFMA implicit-GEMM convolutions (a reduced-precision
    mode the reference does not have), "f32" = fp32 FMAs, the reference's arithmetic (xraft.py:118-119), "bf16x3" / "f16c8" = the reference's fp32
    accuracy on the matrix cores (split operands; include/vtgb.h).
    Only the last iteration's upsampled flow is materialised (``test_mode=True``, what every caller on the path uses)."""

    def __init__(self, compute_dtype="bf16"):
        super().__init__()
        tree = ParamTree(synth.raft_shapes(""), "")
        for name, child in list(tree._modules.items()):
            self.add_module(name, child)
        self.code = ops.raft_dtype_code(compute_dtype)
        self._table = None

    def _apply(self, fn, *a, **k):
        self._table = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, state_dict, *a, **k):
        self._table = None
        if any(key.startswith("module.") for key in state_dict):     # DataParallel checkpoint (raft_utils/utils.py:85-90)
            state_dict = dp_state_to_normal(state_dict)
        return super().load_state_dict(state_dict, *a, **k)

    def set_compute_dtype(self, compute_dtype):
        self.code = ops.raft_dtype_code(compute_dtype)
        self._table = None
        return self

    def _hip_tables(self):
        if self._table is None:
            sd = {k: v for k, v in self.state_dict().items()}
            # (f16c8: the update block and the encoders' stride-1 3x3 convolutions on fp16 + fp8-correction operands, the other encoder stages and the correlation at bf16x3)
            self._table = (ops.RaftWeights(sd, "update_block.", self.code), ops.RaftEncoderWeights(sd, "fnet.", False, self.code),
                           ops.RaftEncoderWeights(sd, "cnet.", True, self.code))
        return self._table

    @torch.no_grad()
    def forward_clips(self, frames: Tensor, iters: int = 20) -> Tensor:
        """Whole clips: frames [B, T, 3, H, W] -> flow [B, T-1, 2, H, W] between consecutive frames.
        fnet runs once per distinct frame (the reference encodes cat(image1, image2), i.e. every inner frame
        twice, with identical results since InstanceNorm is per image); cnet on frames[:, :-1]."""
        upd, fw, cw = self._hip_tables()
        b, t, _, h, w = frames.shape
        h8, w8 = h // 8, w // 8
        fmap = ops.raft_encoder(fw, frames.reshape(b * t, 3, h, w))                           # [b*t, HW, 256]
        cmap = ops.raft_encoder(cw, frames[:, :-1].reshape(b * (t - 1), 3, h, w))             # [n, HW, 256]
        n = b * (t - 1)
        pyr = ops.raft_corr(fmap, n, h8, w8, t - 1, t, 0, 1, ops.raft_stage_code(self.code))
        # net = tanh(cnet[:, :128]), inp = relu(cnet[:, 128:]) (xraft.py:126-127) are taken from the pixel-major cnet output inside
        return ops.raft_update(upd, None, None, pyr, iters, cnet_nhwc=cmap, hw=(h8, w8)).view(b, t - 1, 2, h, w)

    @torch.no_grad()
    def forward(self, image1, image2, iters=20, flow_init=None, upsample=True, test_mode=True):
        """RAFT.forward (xraft.py:102-156): image1 / image2 [N, 3, H, W] (0..255 convention), H and W multiples of 8
        (InputPadder) -> flow_up [N, 2, H, W]; ``test_mode=False``: the list of every iteration's upsampled flow (:146-156; `upsample`
        is as unused as in the reference).  The update kernel keeps only the last iteration's mask head, so the per-iteration list
        re-runs the recurrence with 1, 2, ... iters iterations (identical states: iteration i never sees a later one) -- RAFT is
        frozen in every module of the path, so this form exists for interface parity, not speed.
        COST of ``test_mode=False``: iters * (iters + 1) / 2 update iterations (210 for iters = 20) plus iters mask-head / upsample
        passes, i.e. ~10 x the test-mode call; nothing on the inference or training path uses it."""
        upd, fw, cw = self._hip_tables()
        n, _, h, w = image1.shape
        h8, w8 = h // 8, w // 8
        fmap = ops.raft_encoder(fw, torch.cat([image1, image2], 0))                           # fnet([image1, image2]) (:115)
        cmap = ops.raft_encoder(cw, image1)
        pyr = ops.raft_corr(fmap, n, h8, w8, n, n, 0, n, ops.raft_stage_code(self.code))
        if not test_mode:
            return [ops.raft_update(upd, None, None, pyr, i + 1, cnet_nhwc=cmap, hw=(h8, w8), flow_init=flow_init) for i in range(iters)]
        return ops.raft_update(upd, None, None, pyr, iters, cnet_nhwc=cmap, hw=(h8, w8), flow_init=flow_init)


def dp_state_to_normal(state_dict):
    """raft_utils/utils.py:85-90: strip the DataParallel ``module.`` prefix of a RAFT checkpoint (keys without it are dropped,
    as in the reference)."""
    return {k.replace("module.", ""): v for k, v in state_dict.items() if k.startswith("module")}


def path_cfg_from_hf(hf_config, arch: str) -> synth.PathCfg:
    """InstructBlipConfig / Blip2Config (transformers; what ``*.from_pretrained(base_model_path)`` parses out of the
    directory's config.json, eval/utils/model.py:33,252) -> the dims of the hot-path stages."""
    v, q, t = hf_config.vision_config, hf_config.qformer_config, hf_config.text_config
    n_query = getattr(hf_config, "num_query_tokens", 32)
    llm_hidden = getattr(t, "hidden_size", None) or getattr(t, "d_model")
    return synth.PathCfg(
        arch,
        synth.VitCfg(hidden=v.hidden_size, layers=v.num_hidden_layers, heads=v.num_attention_heads, mlp=v.intermediate_size,
                     image=v.image_size, patch=v.patch_size, eps=v.layer_norm_eps),
        synth.QFormerCfg(hidden=q.hidden_size, layers=q.num_hidden_layers, heads=q.num_attention_heads, ffn=q.intermediate_size,
                         enc_hidden=q.encoder_hidden_size, n_query=n_query, vocab=q.vocab_size, max_pos=q.max_position_embeddings,
                         cross_freq=q.cross_attention_frequency, has_text=(arch == "instructblip"), eps=q.layer_norm_eps),
        synth.TgbCfg(),                                            # BertConfig(fusion_layer=6, encoder_width=768): BERT-base
        llm_hidden)


def load_hf_config(base_model_path: str, arch: str):
    from transformers import Blip2Config, InstructBlipConfig
    return (InstructBlipConfig if arch == "instructblip" else Blip2Config).from_pretrained(base_model_path)


def build_language_model(hf_config, dtype=torch.float32, device=None) -> nn.Module:
    """The third-party LLM of the path, random-init from ``text_config`` exactly as the reference's
    ``InstructBlipForConditionalGeneration(config=...)`` does (xinstructblip.py:1268-1273, xblip2.py:1551-1556): a causal LM
    when ``use_decoder_only_language_model`` else a seq2seq LM (Flan-T5).  The checkpoint's state_dict then fills it."""
    from transformers import AutoModelForCausalLM, AutoModelForSeq2SeqLM
    auto = AutoModelForCausalLM if hf_config.use_decoder_only_language_model else AutoModelForSeq2SeqLM
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        if device is not None:
            with torch.device(device):
                lm = auto.from_config(hf_config.text_config)
        else:
            lm = auto.from_config(hf_config.text_config)
    finally:
        torch.set_default_dtype(prev)
    return lm.eval()


class PathModel(nn.Module):
    """The ``self.model`` object of the reference modules (InstructBlip/Blip2ForConditionalGeneration)
    restricted to what the path touches: vision_model, qformer, query_tokens, language_projection,
    temporal_projection (dead weight, must exist), language_model (third-party HF), config."""

    def __init__(self, cfg: synth.PathCfg, language_model: Optional[nn.Module] = None, compute_dtype="bf16",
                 llm_architectures=None, decoder_only: Optional[bool] = None, hf_config=None):
        super().__init__()
        self.vision_model = VisionModel(cfg.vit, compute_dtype)
        self.qformer = QFormer(cfg.qformer, compute_dtype)
        self.query_tokens = nn.Parameter(torch.zeros(1, cfg.qformer.n_query, cfg.qformer.hidden), requires_grad=False)
        self.language_projection = LanguageProjection(cfg.qformer.hidden, cfg.llm_hidden, compute_dtype)
        self.temporal_projection = nn.Linear(cfg.qformer.hidden, cfg.llm_hidden)
        self.language_model = language_model
        lm_cfg = getattr(language_model, "config", None)
        if hf_config is not None:
            self.config = hf_config                                 # the reference's own config object (transformers)
        else:
            if decoder_only is None:
                decoder_only = not bool(getattr(lm_cfg, "is_encoder_decoder", False))
            if llm_architectures is None:
                llm_architectures = getattr(lm_cfg, "architectures", None) or ["LlamaForCausalLM"]
            self.config = SimpleNamespace(use_decoder_only_language_model=decoder_only,
                                          text_config=SimpleNamespace(architectures=list(llm_architectures),
                                                                      vocab_size=getattr(lm_cfg, "vocab_size", 0)))

    def get_input_embeddings(self):
        return self.language_model.get_input_embeddings()


class _LSTPBase(nn.Module):
    ARCH = "instructblip"
    TGB_MODE = "multi_modal"
    MAP = "A"

    def __init__(self, base_model_path, device="cuda", lora: bool = False, language_model: Optional[nn.Module] = None,
                 compute_dtype="bf16", raft_dtype=None, lm_dtype=None, tgb_cfg: Optional[synth.TgbCfg] = None):
        """Reference signature (eval/utils/model.py:21-45, :240-264): ``LSTP(base_model_path, device, lora=False)`` -- the
        HF config in ``base_model_path`` sizes the vision tower, Q-Former and language model (random init, the checkpoint
        fills them), the TGB is BERT-base with fusion_layer 6, RAFT is RAFT-large.  ``base_model_path`` may also be a
        ``synth.PathCfg`` (``from_cfg``): then ``language_model`` is the caller's.  Keyword extensions: ``compute_dtype``
        ("bf16" / "f32") of the HIP stages, ``raft_dtype`` ("f16c8" by default next to bf16 stages: the reference's fp32 RAFT accuracy on the matrix cores -- update
        block on fp16 + fp8-correction operands, encoders / correlation on split-bf16 operands; "bf16x3" = split-bf16 operands everywhere (round 5's default,
        same accuracy class, 1.2 x slower); "bf16" = the fast REDUCED-PRECISION RAFT; "f32" = the fp32 FMA chain), ``lm_dtype`` of the built LLM
        (default: bf16 with compute_dtype "bf16", else fp32), ``tgb_cfg`` to size the TGB differently from BERT-base (tests)."""
        super().__init__()
        hf_config = None
        if isinstance(base_model_path, synth.PathCfg):
            cfg = base_model_path
        else:
            hf_config = load_hf_config(base_model_path, self.ARCH)
            cfg = path_cfg_from_hf(hf_config, self.ARCH)
            if tgb_cfg is not None:
                cfg.tgb = tgb_cfg
            if language_model is None:
                if lm_dtype is None:
                    lm_dtype = torch.bfloat16 if ops.dtype_code(compute_dtype) == ops.BF16 else torch.float32
                language_model = build_language_model(hf_config, lm_dtype)
        self.cfg = cfg
        self.model = PathModel(cfg, language_model, compute_dtype, hf_config=hf_config)
        self.temporal_encoder = TemporalEncoder(cfg.tgb, compute_dtype)
        # RAFT's mode.  The reference keeps RAFT in fp32 under EVERY Lightning precision (xraft.py:58,113-118: mixed_precision = False), so the
        # default follows that contract: "f16c8" (fp32 accuracy on the matrix cores: tests/test_gpu_raft.py, test_gpu_selection.py hold it to the
        # bf16x3 mode's bounds) next to bf16 stages, "f32" next to fp32 stages.  ``raft_dtype="bf16"`` opts into the faster reduced-precision RAFT the
        # reference does not have (bench.py reports it as the `raft_bf16_fast` companion, never as the headline).
        self.of_extractor = Raft(raft_dtype or self._default_raft_dtype(compute_dtype))
        self._raft_follows = raft_dtype is None
        self.device = device
        self.fell_back = False
        if lora:
            # the reference wraps the LLM with peft (r=8, alpha=32, dropout 0.1, inference mode; eval/utils/model.py:40-44);
            # same adapter arithmetic and parameter names here (videotgb_amd.train.LoraLinear), without peft's
            # ``base_model.model.`` wrapper level -- load_state_dict strips it from a reference checkpoint's keys
            from .train import apply_lora
            apply_lora(self.model.language_model, r=8, lora_alpha=32, lora_dropout=0.1)
            self.model.language_model.eval()

    @staticmethod
    def _default_raft_dtype(compute_dtype):
        return "f16c8" if ops.dtype_code(compute_dtype) == ops.BF16 else "f32"

    @classmethod
    def from_cfg(cls, cfg: synth.PathCfg, device="cuda", language_model: Optional[nn.Module] = None, compute_dtype="bf16", raft_dtype=None):
        return cls(cfg, device, False, language_model, compute_dtype, raft_dtype)

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        """Accepts a reference (Lightning) checkpoint's ``state_dict`` as is: peft's ``language_model.base_model.model.``
        level is flattened, the dead ``position_ids`` buffers older transformers versions persisted are tolerated."""
        lp = "model.language_model.base_model.model."
        if any(k.startswith(lp) for k in state_dict):
            state_dict = {("model.language_model." + k[len(lp):] if k.startswith(lp) else k): v for k, v in state_dict.items()}
        return super().load_state_dict(state_dict, strict=strict, **kw)

    def set_compute_dtype(self, compute_dtype):
        for m in (self.model.vision_model, self.model.qformer, self.temporal_encoder):
            m.set_compute_dtype(compute_dtype)
        self.model.language_projection.code = ops.dtype_code(compute_dtype)
        self.model.language_projection._packed = None
        if self._raft_follows:
            self.of_extractor.set_compute_dtype(self._default_raft_dtype(compute_dtype))
        return self

    # ---- stages -------------------------------------------------------------------------------
    flow_clips_per_call = 16  # RAFT batch (clips); ~1.7 GB of workspace per clip at T = 96, 224 x 224

    @torch.no_grad()
    def flow(self, flow_frames: Tensor, clips_per_call: Optional[int] = None) -> Tensor:
        """eval/utils/model.py:76-84: RAFT between consecutive frames of each clip, last flow repeated.
        The frame pairs of up to ``clips_per_call`` clips go through RAFT in one call (the reference loops
        over clips; pairs are independent, so batching them changes nothing but the launch count)."""
        b, t = flow_frames.shape[:2]
        clips_per_call = clips_per_call or self.flow_clips_per_call
        outs = []
        for c0 in range(0, b, clips_per_call):
            ff = flow_frames[c0:c0 + clips_per_call]
            ff = InputPadder(ff.shape).pad(ff.reshape(-1, *ff.shape[2:])).reshape(ff.shape[0], t, ff.shape[2], -1, ff.shape[4]) \
                if (ff.shape[-1] % 8 or ff.shape[-2] % 8) else ff
            fl = self.of_extractor.forward_clips(ff)
            outs.append(torch.cat([fl, fl[:, -1:]], dim=1))
        return torch.cat(outs, dim=0)

    @torch.no_grad()
    def select_frames(self, pixel_values: Tensor, of: Tensor, sampler_ids: Tensor, sampler_mask: Tensor, nframe: int,
                      noise: Optional[Tensor] = None, of_mask: Optional[Tensor] = None, video_lengths=None):
        """TGB -> Gumbel top-k -> index map -> gather, all on the device (eval/utils/model.py:85-151).
        pixel_values [B, N, 3, H, W]; of [B, T, 2, H, W].  Returns (sampled [B*nframe, 3, H, W], idx [B, nframe], logits)."""
        b, t = of.shape[:2]
        if of_mask is None:
            of_mask = torch.ones(b, t + 2, dtype=torch.long, device=of.device)
        _, logits = self.temporal_encoder(encoder_embeds=of, attention_mask=of_mask, encoder_hidden_states=sampler_ids,
                                          encoder_attention_mask=sampler_mask, mode=self.TGB_MODE)
        if noise is None:   # F.gumbel_softmax's noise: -log(Exp(1)), fresh per draw (Appendix B)
            noise = -torch.empty(2, 2 * b, t, device=of.device).exponential_().log()
        sel = ops.span_select(logits, noise, 0.5)
        v = t if video_lengths is None else video_lengths
        idx = ops.span_to_frames(sel, v, pixel_values.shape[1], nframe, self.MAP)
        sampled = ops.gather_frames(pixel_values, idx)
        return sampled.view(b * nframe, *pixel_values.shape[2:]), idx, logits

    @torch.no_grad()
    def prefix(self, sampled: Tensor, batch_size: int, nframe: int, text_encoding=None, pool: str = "mean") -> Tensor:
        """ViT -> Q-Former -> frame pooling + language_projection (eval/utils/model.py:154-195)."""
        img = self.model.vision_model(pixel_values=sampled, return_dict=True, act_output=True).last_hidden_state
        query_tokens = self.model.query_tokens.expand(img.shape[0], -1, -1)
        if self.ARCH == "instructblip":
            qi = torch.repeat_interleave(text_encoding["qformer_input_ids"], nframe, 0)
            qm = torch.repeat_interleave(text_encoding["qformer_attention_mask"], nframe, 0)
            am = torch.cat([torch.ones(query_tokens.shape[:-1], dtype=torch.long, device=img.device), qm], dim=1)
            qo = self.model.qformer(input_ids=qi, attention_mask=am, query_embeds=query_tokens, encoder_hidden_states=img,
                                    encoder_attention_mask=None, return_dict=True).last_hidden_state
        else:
            qo = self.model.qformer(query_embeds=query_tokens, encoder_hidden_states=img, encoder_attention_mask=None)[0]
        qo = qo[:, : query_tokens.size(1), :]
        return self.model.language_projection.pool(qo, [nframe] * batch_size, pool)

    @staticmethod
    def _graph_plan(lm, inputs_embeds, attention_mask, do_sample, temperature, stopping_criteria, kw):
        """The keyword arguments of ``decode.*Decoder.generate`` for this request, or None when it is outside what the graph decoders reproduce
        (then HF ``generate`` runs it): a Llama / T5 language model on the device, no padding, one beam, neutral penalties; greedy, or (Llama)
        sampling with temperature / top_k / top_p; ``stopping_criteria`` None or KeywordsStoppingCriteria objects (eval/utils/builder_utils.py:320-346:
        recognised by their ``keyword_ids`` / ``keywords`` / ``tokenizer`` attributes)."""
        mt = getattr(lm.config, "model_type", "")
        if not inputs_embeds.is_cuda or not ("llama" in mt or mt == "t5") or not bool((attention_mask != 0).all()):
            return None
        gc = getattr(lm, "generation_config", None)     # HF generate's defaults come from the generation config
        plan = dict(eos_token_id=kw.pop("eos_token_id", getattr(gc, "eos_token_id", None)), pad_token_id=kw.pop("pad_token_id", getattr(gc, "pad_token_id", None)),
                    min_new_tokens=kw.pop("min_new_tokens", 0))
        if kw.pop("num_beams", 1) != 1 or kw.pop("repetition_penalty", 1.0) not in (None, 1.0) or kw.pop("length_penalty", 1.0) not in (None, 1.0):
            return None
        top_k, top_p = kw.pop("top_k", getattr(gc, "top_k", 50)), kw.pop("top_p", getattr(gc, "top_p", 1.0))
        noise, gen = kw.pop("sample_noise", None), kw.pop("generator", None)
        if kw:
            return None
        crit = list(stopping_criteria) if stopping_criteria is not None else []
        if any(not all(hasattr(c, a) for a in ("keyword_ids", "keywords", "tokenizer")) for c in crit):
            return None
        if mt == "t5":
            return plan if not do_sample and not crit else None
        if do_sample:
            plan.update(do_sample=True, temperature=temperature, top_k=top_k, top_p=top_p, sample_noise=noise, generator=gen)
        if crit:
            if inputs_embeds.shape[0] != 1:
                raise AssertionError("Only support batch size 1 (yet)")      # (the reference's criteria)
            from .decode import keyword_stop_plan
            plan.update(keyword_stop_plan(crit))
        return plan

    # ---- the reference entry point ---------------------------------------------------------------
    @torch.no_grad()
    def generate(self, frames, flow_frames, nframe, text_encoding, sampler_text_encoding, do_sample=True, temperature=0.2,
                 max_new_tokens=1024, use_cache=True, stopping_criteria=None, of: Optional[Tensor] = None,
                 noise: Optional[Tensor] = None, pool: str = "mean", return_stages: bool = False, fast_decode="auto",
                 **gen_kwargs):
        """eval/utils/model.py:48-235 (LSTP) / :267-445 (LSTP_blip2).  Extensions: ``of`` supplies a
        precomputed flow (the batch["of"] contract of the LightningModules), ``noise`` injects the
        Gumbel noise, ``pool`` selects mean (eval) or concat (LightningModules) pooling.
        ``fast_decode``: "auto" (default) decodes on libvtgb.so's kernels (videotgb_amd.decode: hipGraph-replayed steps over the same HF weights)
        whenever the request is one that decoder reproduces -- which includes the reference's own eval call (eval/inference.py:98-109: sampling at
        temperature 0.2, ``use_cache=False`` [no effect on the ids], ``KeywordsStoppingCriteria``) -- and through HF ``generate`` otherwise; True
        insists (TypeError outside the envelope); False always takes HF ``generate``.  ``sample_noise`` [max_new_tokens, B] injects the sampler's
        uniform numbers (tests)."""
        sampler_ids = sampler_text_encoding["input_ids"]
        batch_size = sampler_ids.shape[0]
        pixel_values = frames
        num_frames = pixel_values.size(0) // batch_size
        pixel_values = pixel_values.view(batch_size, num_frames, *pixel_values.shape[1:])
        if of is None:
            of = self.flow(flow_frames)
        # NB the reference wraps RAFT+TGB in a bare ``except:`` and silently falls back to the full
        # span (eval/utils/model.py:114-116).  A kernel failure must not be masked: we raise.
        sampled, idx, logits = self.select_frames(pixel_values, of, sampler_ids, sampler_text_encoding["attention_mask"],
                                                  nframe, noise)
        lm_inputs = self.prefix(sampled, batch_size, nframe, text_encoding, pool)
        lm = self.model.language_model
        lm_dtype = next(lm.parameters()).dtype
        lm_inputs = lm_inputs.to(lm_dtype)
        lm_mask = torch.ones(lm_inputs.size()[:-1], dtype=torch.long, device=lm_inputs.device)
        attention_mask = torch.cat([lm_mask, text_encoding["attention_mask"]], dim=1)
        inputs_embeds = self.model.get_input_embeddings()(text_encoding["input_ids"])
        inputs_embeds = torch.cat([lm_inputs, inputs_embeds.to(lm_dtype)], dim=1)
        plan = None if fast_decode is False else self._graph_plan(lm, inputs_embeds, attention_mask, do_sample, temperature, stopping_criteria, dict(gen_kwargs))
        if fast_decode is True and plan is None:      # (nothing is dropped silently -- and nothing is decoded first)
            raise TypeError(f"generate(fast_decode=True): outside the graph decoder's envelope (do_sample={do_sample}, stopping_criteria="
                            f"{type(stopping_criteria).__name__}, kwargs {sorted(gen_kwargs)}); call with fast_decode=False or \"auto\"")
        if plan is not None:
            from .decode import make_decoder, weights_key
            if getattr(self, "_decoder", None) is None or self._decoder.lm is not lm or self._decoder.key != weights_key(lm):
                self._decoder = make_decoder(lm)      # Llama (causal) or T5 (seq2seq, LSTP_blip2)
            outputs = self._decoder.generate(inputs_embeds, max_new_tokens, **plan)
        else:
            outputs = lm.generate(inputs_embeds=inputs_embeds, attention_mask=attention_mask, do_sample=do_sample,
                                  temperature=temperature, max_new_tokens=max_new_tokens, use_cache=use_cache,
                                  stopping_criteria=stopping_criteria, **gen_kwargs)
        if self.model.config.text_config.architectures[0] == "LLaMAForCausalLM":
            outputs[outputs == 0] = 2
        cand_index = idx[-1]
        if return_stages:
            return outputs, cand_index, dict(of=of, tgb_logits=logits, frame_idx=idx, sampled=sampled, prefix=lm_inputs,
                                             inputs_embeds=inputs_embeds)
        return outputs, cand_index


class LSTP(_LSTPBase):
    """eval/utils/model.py:19-235 (InstructBLIP; TGB mode multi_modal, index map A, V = T)."""
    ARCH, TGB_MODE, MAP = "instructblip", "multi_modal", "A"


class LSTP_blip2(_LSTPBase):
    """eval/utils/model.py:238-445 (BLIP-2; TGB mode fusion, index map B, V = T)."""
    ARCH, TGB_MODE, MAP = "blip2", "fusion", "B"


def load_synth(model: _LSTPBase, seed: int = 0, device="cuda", with_raft: bool = True) -> Dict[str, Tensor]:
    """Load the seeded synthetic state_dict (videotgb_amd.synth) into an LSTP module, strictly
    for every hot-path key (the LLM, third-party, keeps its own weights)."""
    sd = synth.path_state_dict(model.cfg, seed, with_raft)
    own = {k: v for k, v in model.state_dict().items() if not k.startswith("model.language_model.")}
    missing = set(own) - set(sd)
    extra = set(sd) - set(own)
    assert not missing and not extra, (sorted(missing)[:5], sorted(extra)[:5])
    model.load_state_dict(sd, strict=False)
    model.to(device)
    return sd
