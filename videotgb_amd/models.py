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
  Raft               <- RAFT (host torch ops on the GPU for now) (xraft.py:51-156)
  LSTP / LSTP_blip2  <- eval/utils/model.py:19-235 / :238-445
"""
from __future__ import annotations

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
        self._table = None

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
        if isinstance(encoder_attention_mask, Tensor) and bool((encoder_attention_mask != 0).all()):
            encoder_attention_mask = None      # all-ones: nothing to mask (every caller on the path)
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

    def _apply(self, fn, *a, **k):
        self._packed = None
        return super()._apply(fn, *a, **k)

    def packed(self) -> Tensor:
        if self._packed is None:
            self._packed = ops.pack_weight(self.weight.data, self.code)
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
    """RAFT-large with the reference's parameter names.  First pass: stock
    PyTorch-ROCm ops (MIOpen convolutions, rocBLAS all-pairs correlation) on the GPU, in
    ``raft_dtype`` (fp32 by default as in the reference, xraft.py:118-119).  Only the last
    iteration's upsampled flow is materialised."""

    def __init__(self, raft_dtype=torch.float32, hip_update: bool = False):
        super().__init__()
        tree = ParamTree(synth.raft_shapes(""), "")
        for name, child in list(tree._modules.items()):
            self.add_module(name, child)
        self.raft_dtype = raft_dtype
        self.hip_update = hip_update     # run the 20 refinement iterations in libvtgb.so (bf16 MFMA implicit-GEMM convs)
        self.channels_last = False       # NHWC activations for the MIOpen encoder convolutions
        self.hip_encoders = False        # fnet / cnet in libvtgb.so as well (forward_clips)
        self._table = None

    def _apply(self, fn, *a, **k):
        self._table = None
        return super()._apply(fn, *a, **k)

    def _hip_tables(self):
        if self._table is None:
            sd = {k: v for k, v in self.state_dict().items()}
            self._table = (ops.RaftWeights(sd), ops.RaftEncoderWeights(sd, "fnet.", False), ops.RaftEncoderWeights(sd, "cnet.", True))
        return self._table

    @torch.no_grad()
    def forward_clips(self, frames: Tensor, iters: int = 20) -> Tensor:
        """All-HIP path for whole clips: frames [B, T, 3, H, W] -> flow [B, T-1, 2, H, W] between consecutive frames.
        fnet runs once per distinct frame (the reference encodes cat(image1, image2), i.e. every inner frame
        twice, with identical results since InstanceNorm is per image); cnet on frames[:, :-1]."""
        upd, fw, cw = self._hip_tables()
        b, t, _, h, w = frames.shape
        h8, w8 = h // 8, w // 8
        fmap = ops.raft_encoder(fw, frames.reshape(b * t, 3, h, w)).view(b, t, h8 * w8, 256)
        cmap = ops.raft_encoder(cw, frames[:, :-1].reshape(b * (t - 1), 3, h, w))             # [n, HW, 256]
        n = b * (t - 1)
        # all-pairs correlation (corr.py:52-60) as ONE half-precision batched GEMM (fp32 accumulation in the MFMA; the
        # features are O(1): 11 significant bits on inputs and outputs against the bf16 features the lookup emits),
        # a tenth of the fp32 GEMM's time and half the volume's traffic
        fh = fmap.to(torch.float16)
        corr = torch.matmul(fh[:, :-1].reshape(n, h8 * w8, 256), fh[:, 1:].reshape(n, h8 * w8, 256).transpose(1, 2))
        pyr = ops.raft_corr_pyramid(corr.view(n * h8 * w8, h8 * w8), h8, w8)                   # / sqrt(256) + 3 avg-pools, fp16
        # net = tanh(cnet[:, :128]), inp = relu(cnet[:, 128:]) (xraft.py:126-127) are taken from the pixel-major cnet output inside
        return ops.raft_update(upd, None, None, pyr, iters, cnet_nhwc=cmap, hw=(h8, w8)).view(b, t - 1, 2, h, w)

    def _c(self, name, x, stride=1, padding=0):
        m = self.get_submodule(name)
        w = m.weight.to(x.dtype)
        if self.channels_last:
            w = w.contiguous(memory_format=torch.channels_last)
        return F.conv2d(x, w, m.bias.to(x.dtype), stride=stride, padding=padding)

    def _norm(self, name, x, kind):
        if kind == "instance":
            return F.instance_norm(x, eps=1e-5)
        m = self.get_submodule(name)
        return F.batch_norm(x, m.running_mean.to(x.dtype), m.running_var.to(x.dtype), m.weight.to(x.dtype),
                            m.bias.to(x.dtype), training=False, eps=1e-5)

    def _res(self, p, x, kind, stride):
        y = F.relu(self._norm(p + "norm1", self._c(p + "conv1", x, stride, 1), kind))
        y = F.relu(self._norm(p + "norm2", self._c(p + "conv2", y, 1, 1), kind))
        if stride != 1:
            x = self._norm(p + "norm3", self._c(p + "downsample.0", x, stride, 0), kind)
        return F.relu(x + y)

    def _encoder(self, p, x, kind):
        x = F.relu(self._norm(p + "norm1", self._c(p + "conv1", x, 2, 3), kind))
        for li, stride in (("layer1", 1), ("layer2", 2), ("layer3", 2)):
            x = self._res(f"{p}{li}.0.", x, kind, stride)
            x = self._res(f"{p}{li}.1.", x, kind, 1)
        return self._c(p + "conv2", x)

    @torch.no_grad()
    def forward(self, image1, image2, iters=20, flow_init=None, upsample=True, test_mode=True):
        dt = self.raft_dtype
        image1 = (2 * (image1.float() / 255.0) - 1.0).contiguous().to(dt)
        image2 = (2 * (image2.float() / 255.0) - 1.0).contiguous().to(dt)
        n, _, h, w = image1.shape
        if self.channels_last:
            image1 = image1.contiguous(memory_format=torch.channels_last)
            image2 = image2.contiguous(memory_format=torch.channels_last)
        f = self._encoder("fnet.", torch.cat([image1, image2], 0), "instance").float()
        fmap1, fmap2 = f[:n], f[n:]
        d, hh, ww = fmap1.shape[1:]
        corr = torch.matmul(fmap1.view(n, d, hh * ww).transpose(1, 2), fmap2.view(n, d, hh * ww))
        corr = (corr / torch.sqrt(torch.tensor(d).float())).reshape(n * hh * ww, 1, hh, ww)
        pyr = [corr]
        for _ in range(3):
            corr = F.avg_pool2d(corr, 2, stride=2)
            pyr.append(corr)
        c = self._encoder("cnet.", image1, "batch")
        net, inp = torch.tanh(c[:, :128]), torch.relu(c[:, 128:])
        if self.hip_update and flow_init is None:
            return ops.raft_update(self._hip_tables()[0], net.float(), inp.float(), pyr, iters)
        ys, xs = torch.meshgrid(torch.arange(hh, device=image1.device), torch.arange(ww, device=image1.device), indexing="ij")
        coords0 = torch.stack([xs, ys], 0).float()[None].repeat(n, 1, 1, 1)
        coords1 = coords0.clone() if flow_init is None else coords0 + flow_init
        r = 4
        dx = torch.linspace(-r, r, 2 * r + 1, device=image1.device)
        delta = torch.stack(torch.meshgrid(dx, dx, indexing="ij"), dim=-1).view(1, 2 * r + 1, 2 * r + 1, 2)
        u = "update_block."
        mask = None
        for it in range(iters):
            cp = coords1.permute(0, 2, 3, 1).reshape(n * hh * ww, 1, 1, 2)
            outs = []
            for i, cv in enumerate(pyr):                      # CorrBlock.__call__ corr.py:29-50
                cl = cp / 2 ** i + delta
                hc, wc = cv.shape[-2:]
                grid = torch.cat([2 * cl[..., 0:1] / (wc - 1) - 1, 2 * cl[..., 1:2] / (hc - 1) - 1], dim=-1)
                outs.append(F.grid_sample(cv, grid, align_corners=True).view(n, hh, ww, -1))
            cfeat = torch.cat(outs, dim=-1).permute(0, 3, 1, 2).contiguous().to(dt)
            flow = (coords1 - coords0).to(dt)
            cor = F.relu(self._c(u + "encoder.convc1", cfeat))
            cor = F.relu(self._c(u + "encoder.convc2", cor, 1, 1))
            flo = F.relu(self._c(u + "encoder.convf1", flow, 1, 3))
            flo = F.relu(self._c(u + "encoder.convf2", flo, 1, 1))
            mf = F.relu(self._c(u + "encoder.conv", torch.cat([cor, flo], 1), 1, 1))
            x = torch.cat([inp, mf, flow], 1)
            for sfx, pad in (("1", (0, 2)), ("2", (2, 0))):
                hx = torch.cat([net, x], 1)
                z = torch.sigmoid(self._c(u + "gru.convz" + sfx, hx, 1, pad))
                rr = torch.sigmoid(self._c(u + "gru.convr" + sfx, hx, 1, pad))
                q = torch.tanh(self._c(u + "gru.convq" + sfx, torch.cat([rr * net, x], 1), 1, pad))
                net = (1 - z) * net + z * q
            dflow = self._c(u + "flow_head.conv2", F.relu(self._c(u + "flow_head.conv1", net, 1, 1)), 1, 1)
            coords1 = coords1 + dflow.float()
            if it == iters - 1:
                mask = 0.25 * self._c(u + "mask.2", F.relu(self._c(u + "mask.0", net, 1, 1))).float()
        flow = coords1 - coords0
        m = torch.softmax(mask.view(n, 1, 9, 8, 8, hh, ww), dim=2)
        up = F.unfold(8 * flow, [3, 3], padding=1).view(n, 2, 9, 1, 1, hh, ww)
        up = torch.sum(m * up, dim=2).permute(0, 1, 4, 2, 5, 3)
        return up.reshape(n, 2, 8 * hh, 8 * ww)


class PathModel(nn.Module):
    """The ``self.model`` object of the reference modules (InstructBlip/Blip2ForConditionalGeneration)
    restricted to what the path touches: vision_model, qformer, query_tokens, language_projection,
    temporal_projection (dead weight, must exist), language_model (third-party HF), config."""

    def __init__(self, cfg: synth.PathCfg, language_model: Optional[nn.Module] = None, compute_dtype="bf16",
                 llm_architectures=("LlamaForCausalLM",), decoder_only: bool = True):
        super().__init__()
        self.vision_model = VisionModel(cfg.vit, compute_dtype)
        self.qformer = QFormer(cfg.qformer, compute_dtype)
        self.query_tokens = nn.Parameter(torch.zeros(1, cfg.qformer.n_query, cfg.qformer.hidden), requires_grad=False)
        self.language_projection = LanguageProjection(cfg.qformer.hidden, cfg.llm_hidden, compute_dtype)
        self.temporal_projection = nn.Linear(cfg.qformer.hidden, cfg.llm_hidden)
        self.language_model = language_model
        self.config = SimpleNamespace(use_decoder_only_language_model=decoder_only,
                                      text_config=SimpleNamespace(architectures=list(llm_architectures),
                                                                  vocab_size=getattr(getattr(language_model, "config", None), "vocab_size", 0)))

    def get_input_embeddings(self):
        return self.language_model.get_input_embeddings()


class _LSTPBase(nn.Module):
    ARCH = "instructblip"
    TGB_MODE = "multi_modal"
    MAP = "A"

    def __init__(self, cfg: synth.PathCfg, device="cuda", language_model: Optional[nn.Module] = None, compute_dtype="bf16",
                 raft_dtype=torch.float32, raft_hip_update: bool = False):
        super().__init__()
        self.cfg = cfg
        self.model = PathModel(cfg, language_model, compute_dtype)
        self.temporal_encoder = TemporalEncoder(cfg.tgb, compute_dtype)
        self.of_extractor = Raft(raft_dtype, raft_hip_update)
        self.device = device
        self.fell_back = False

    def set_compute_dtype(self, compute_dtype):
        for m in (self.model.vision_model, self.model.qformer, self.temporal_encoder):
            m.set_compute_dtype(compute_dtype)
        self.model.language_projection.code = ops.dtype_code(compute_dtype)
        self.model.language_projection._packed = None
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
            if getattr(self.of_extractor, "hip_encoders", False):
                fl = self.of_extractor.forward_clips(ff)
            else:
                i1 = ff[:, :-1].reshape(-1, *ff.shape[2:])
                i2 = ff[:, 1:].reshape(-1, *ff.shape[2:])
                fl = self.of_extractor(i1, i2).view(ff.shape[0], t - 1, 2, ff.shape[3], ff.shape[4])
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

    # ---- the reference entry point ---------------------------------------------------------------
    @torch.no_grad()
    def generate(self, frames, flow_frames, nframe, text_encoding, sampler_text_encoding, do_sample=True, temperature=0.2,
                 max_new_tokens=1024, use_cache=True, stopping_criteria=None, of: Optional[Tensor] = None,
                 noise: Optional[Tensor] = None, pool: str = "mean", return_stages: bool = False, fast_decode: bool = False,
                 **gen_kwargs):
        """eval/utils/model.py:48-235 (LSTP) / :267-445 (LSTP_blip2).  Extensions: ``of`` supplies a
        precomputed flow (the batch["of"] contract of the LightningModules), ``noise`` injects the
        Gumbel noise, ``pool`` selects mean (eval) or concat (LightningModules) pooling."""
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
        if fast_decode and not do_sample and stopping_criteria is None and bool((attention_mask != 0).all()):
            # greedy, unpadded: hipGraph-replayed decode of the same HF weights (videotgb_amd/decode.py)
            from .decode import GreedyDecoder
            if getattr(self, "_decoder", None) is None or self._decoder.lm is not lm:
                self._decoder = GreedyDecoder(lm)
            outputs = self._decoder.generate(inputs_embeds, max_new_tokens)
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
