"""Twins of the reference's LightningModules (the Hydra ``_target_``s of configs/model/*.yaml) over the HIP-backed stages.

Same constructor arguments (``model_name_or_path, sampler_name_or_path, of_extractor_name_or_path, temperature, optimizer,
scheduler, scheduler_params, generate_configs``), same attribute names (``.model`` / ``.temporal_encoder`` / ``.of_extractor`` /
``.processor``), same batch-dict keys (the collate of src/data/components/*_dataset.py), same methods Lightning
calls (``training_step`` -> loss, ``validation_step``, ``test_step``, ``configure_optimizers``), and ``forward(batch) -> (loss,
logits)`` / ``eval_forward(batch) -> output ids`` restating the flavour of each module:

  class here            reference _target_                                   flow source        TGB mode     V          map  pool
  LSTPModule            src.models.LSTP_module.LSTPModule                    of / RAFT(cands)   multi_modal  N + 2      A    concat
  LSTPBlip2Module       src.models.LSTP_blip2_module.LSTPModule              no sampler: range(N) -> midpoint subsample   concat  (C1)
  LSTPSFModule          src.models.LSTP_SF_module.LSTPSFModule               batch["of"]        fusion       of_lengths B    concat
  LSTPSFBlip2Module     src.models.LSTP_SF_blip2_module.LSTPSFModule         batch["of"]        fusion       of_lengths B    concat  (C2)
  LSTPVicunaIV(T)Module src.models.LSTP_Vicuna_IV(T)_module.LSTPModule       frames pre-cut by the dataset; mean over `widths`   (C5 = IVT: LoRA)
  LSTPBlip2IV(T)Module  src.models.LSTP_Blip2_IV(T)_module.LSTPModule        same, Flan-T5

The Lightning / Hydra control plane itself is NOT rebuilt: the base class is ``lightning.LightningModule`` when lightning
is installed and a minimal stand-in otherwise (hparams, log); ``videotgb_amd.dropin.install()`` registers these classes
under the reference's module paths so that ``hydra.utils.instantiate(cfg.model)`` (src/train.py:53) and
``eval/inference.py`` resolve to them unchanged.  Every stage forward goes through libvtgb.so (no torch fallback).
"""
from __future__ import annotations

import glob
import os
from types import SimpleNamespace
from typing import Any, Dict, List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import models, ops, refine, synth, train
from .models import InputPadder

try:  # the control plane is the reference's: use it when present
    from lightning import LightningModule as _Base
except Exception:  # pragma: no cover - lightning is absent in the build image
    class _Base(nn.Module):
        """What the twins use of LightningModule: ``hparams``, ``save_hyperparameters``, ``log`` (a no-op), ``trainer``."""

        def __init__(self):
            super().__init__()
            self.hparams = SimpleNamespace()
            self.trainer = SimpleNamespace(max_steps=-1)
            self.logged: Dict[str, Any] = {}

        def save_hyperparameters(self, logger: bool = False, **kw):
            pass

        def log(self, name, value, **kw):
            self.logged[name] = value


# ------------------------------------------------------------------------------------------ loading
def _load_weight_files(path: str) -> Optional[Dict[str, torch.Tensor]]:
    """state_dict stored next to an HF config.json (model*.safetensors shards or pytorch_model*.bin), or None."""
    if not os.path.isdir(path):
        return None
    sd: Dict[str, torch.Tensor] = {}
    st = sorted(glob.glob(os.path.join(path, "*.safetensors")))
    if st:
        from safetensors.torch import load_file
        for f in st:
            sd.update(load_file(f))
        return sd
    for f in sorted(glob.glob(os.path.join(path, "pytorch_model*.bin"))):
        sd.update(torch.load(f, map_location="cpu"))
    return sd or None


def path_model_from_pretrained(model_name_or_path: str, arch: str, compute_dtype="bf16", lm_dtype=None) -> models.PathModel:
    """``InstructBlipForConditionalGeneration.from_pretrained`` / ``Blip2ForConditionalGeneration.from_pretrained``
    (src/models/LSTP_module.py:108, LSTP_blip2_module.py:108): config.json sizes everything; weight files in the directory are
    loaded when present (HF key names = ours without the ``model.`` prefix; ``temporal_projection`` is new in the reference's
    class and stays freshly initialised, as there)."""
    hf = models.load_hf_config(model_name_or_path, arch)
    cfg = models.path_cfg_from_hf(hf, arch)
    if lm_dtype is None:
        lm_dtype = torch.bfloat16 if ops.dtype_code(compute_dtype) == ops.BF16 else torch.float32
    pm = models.PathModel(cfg, models.build_language_model(hf, lm_dtype), compute_dtype, hf_config=hf)
    sd = _load_weight_files(model_name_or_path)
    if sd is not None:
        msg = pm.load_state_dict(sd, strict=False)
        unexpected = [k for k in msg.unexpected_keys if "position_ids" not in k]
        missing = [k for k in msg.missing_keys if not k.startswith("temporal_projection")]
        if unexpected or missing:
            raise RuntimeError(f"{model_name_or_path}: state_dict mismatch, missing {missing[:5]} unexpected {unexpected[:5]}")
    return pm


def temporal_encoder_from_pretrained(sampler_name_or_path: str, compute_dtype="bf16", state_dict_file: bool = False,
                                     tgb_cfg: Optional[synth.TgbCfg] = None) -> models.TemporalEncoder:
    """``RopeBertModel.from_pretrained(sampler_name_or_path, config=BertConfig(fusion_layer=6, encoder_width=768))``
    (LSTP_module.py:138): BERT-base weights fill the keys they have (``bert.`` prefix stripped), everything the TGB adds
    (temporal embeddings, cross-attention, mrc_head) keeps its fresh BERT-style initialisation.  ``state_dict_file``: the
    IV / IVT modules instead ``torch.load`` a trained sampler checkpoint and load it strictly after ``dp_state_to_normal``
    (LSTP_Vicuna_IVT_module.py:142-146)."""
    te = models.TemporalEncoder(tgb_cfg or synth.TgbCfg(), compute_dtype)
    with torch.no_grad():   # BertPreTrainedModel._init_weights: N(0, 0.02) matrices / embeddings, LayerNorm (1, 0), zero biases
        g = torch.Generator().manual_seed(0)
        for name, p in te.named_parameters():
            if "embed_positions" in name:
                continue
            if name.endswith("LayerNorm.weight") or name.endswith("ln.weight"):
                p.fill_(1.0)
            elif p.dim() >= 2:
                p.normal_(0.0, 0.02, generator=g)
            else:
                p.zero_()
    if state_dict_file:
        sd = models.dp_state_to_normal(torch.load(sampler_name_or_path, map_location="cpu"))
        te.load_state_dict(sd, strict=True)
        return te
    sd = _load_weight_files(sampler_name_or_path)
    if sd is not None:
        own = te.state_dict()
        sd = {k[len("bert."):] if k.startswith("bert.") else k: v for k, v in sd.items()}
        te.load_state_dict({k: v for k, v in sd.items() if k in own and tuple(v.shape) == tuple(own[k].shape)
                            and "embed_positions" not in k}, strict=False)
    return te


def raft_from_checkpoint(of_extractor_name_or_path: str, compute_dtype="bf16") -> models.Raft:
    """``RAFT()`` + ``torch.load`` + ``dp_state_to_normal`` + strict load (LSTP_module.py:151-154)."""
    r = models.Raft(compute_dtype)
    sd = torch.load(of_extractor_name_or_path, map_location="cpu")
    r.load_state_dict(models.dp_state_to_normal(sd), strict=True)
    return r


class _Bleu1:
    """Stand-in for torchmetrics BLEUScore(n_gram=1) used only by validation_step / test_step logging when torchmetrics is
    absent: corpus-level unigram precision with brevity penalty."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.hit = self.total = self.plen = self.tlen = 0

    def __call__(self, preds: List[str], targets: List[str]):
        import math
        for p, t in zip(preds, targets):
            pt, tt = p.split(), (t if isinstance(t, str) else t[0]).split()
            left = list(tt)
            for w in pt:
                if w in left:
                    left.remove(w)
                    self.hit += 1
            self.total += len(pt)
            self.plen += len(pt)
            self.tlen += len(tt)
        self._math = math

    def compute(self) -> float:
        if self.total == 0 or self.hit == 0:
            return 0.0
        bp = 1.0 if self.plen > self.tlen else self._math.exp(1 - self.tlen / max(self.plen, 1))
        return bp * self.hit / self.total


# ------------------------------------------------------------------------------------------ the shared module
class _LSTPLightningBase(_Base):
    ARCH = "instructblip"
    SAMPLER = True              # False: cand_index = range(num_frames) (LSTP_blip2_module.py:254)
    EVAL_RAFT = False           # eval_forward runs RAFT on the candidate frames (LSTP_module.py:379-387)
    TGB_MODE = "multi_modal"
    MAP = "A"
    V_FROM_LENGTHS = False      # V = batch["of_lengths"][j] (SF) instead of num_frames + 2 (LSTP_module.py:423)
    WIDTHS = False              # IV / IVT: frames pre-cut by the dataset, mean over batch["widths"]
    LORA = None                 # IVT: "CAUSAL_LM" / "SEQ_2_SEQ_LM"
    SAMPLER_IS_STATE_DICT = False
    SELF_REFINE = False         # SF: pseudo labels + MRC loss in forward

    def __init__(self, model_name_or_path: str, sampler_name_or_path: str, of_extractor_name_or_path: str, temperature: float = 1.0,
                 optimizer=None, scheduler: Optional[str] = None, scheduler_params: Optional[dict] = None,
                 generate_configs: Optional[dict] = None, compute_dtype="bf16", processor=None, tgb_cfg: Optional[synth.TgbCfg] = None):
        super().__init__()
        self.save_hyperparameters(logger=False)
        if not hasattr(self.hparams, "optimizer"):      # the stand-in base: keep what configure_optimizers reads
            self.hparams.optimizer, self.hparams.scheduler, self.hparams.scheduler_params = optimizer, scheduler, scheduler_params or {}
        self.temperature = temperature
        self.generate_configs = dict(generate_configs or {})
        self.model = path_model_from_pretrained(model_name_or_path, self.ARCH, compute_dtype)
        if processor is None:
            from transformers import AutoProcessor
            processor = AutoProcessor.from_pretrained(model_name_or_path, **({"truncation_side": "left"} if self.ARCH == "instructblip" else {}))
        self.processor = processor
        self.temporal_encoder = temporal_encoder_from_pretrained(sampler_name_or_path, compute_dtype, self.SAMPLER_IS_STATE_DICT, tgb_cfg)
        self.of_extractor = raft_from_checkpoint(of_extractor_name_or_path, compute_dtype)
        print(">>> Load checkpoint for of extractor from", of_extractor_name_or_path)
        self.criterion = nn.CrossEntropyLoss()
        self.val_bleu_score, self.test_bleu_score = _Bleu1(), _Bleu1()
        self.val_score_best = 0.0
        if self.LORA:
            train.apply_lora(self.model.language_model, r=8, lora_alpha=32, lora_dropout=0.1)
        self.freeze_weights()

    # ---- the eval twin's ``prefix`` signature, so refine.frame_answers can drive a LightningModule twin as well
    @torch.no_grad()
    def prefix(self, sampled: torch.Tensor, batch_size: int, nframe: int, text_encoding=None, pool: str = "mean") -> torch.Tensor:
        """models._LSTPBase.prefix: ViT -> Q-Former -> pooling + projection for ``batch_size`` clips of ``nframe`` frames each;
        ``text_encoding["qformer_input_ids"]`` holds one instruction per clip (InstructBLIP)."""
        img = self.model.vision_model(pixel_values=sampled, return_dict=True, act_output=True).last_hidden_state
        batch = {}
        if self.ARCH == "instructblip":
            batch = {"qformer_text": text_encoding["qformer_input_ids"], "qformer_text_attention_mask": text_encoding["qformer_attention_mask"]}
        return self._prefix_nograd(batch, img, [nframe] * batch_size, pool)

    @torch.no_grad()
    def _flow_of_candidates(self, pixel_values: torch.Tensor) -> torch.Tensor:
        """LSTP_module.py:379-387: RAFT between consecutive CANDIDATE frames of each clip, last flow repeated."""
        b, n = pixel_values.shape[:2]
        ff = pixel_values
        if ff.shape[-1] % 8 or ff.shape[-2] % 8:
            ff = InputPadder(ff.shape).pad(ff.reshape(-1, *ff.shape[2:])).reshape(b, n, ff.shape[2], -1, ff.shape[4])
        fl = self.of_extractor.forward_clips(ff)
        return torch.cat([fl, fl[:, -1:]], dim=1)

    @torch.no_grad()
    def _sample(self, batch, pixel_values: torch.Tensor, nframe: int, of=None, of_mask=None, noise=None):
        """Frame selection of one flavour -> (sampled [B*nframe, 3, H, W], frame_idx [B, nframe], of_logits or None)."""
        b, n = pixel_values.shape[:2]
        if not self.SAMPLER:
            # no sampler: cand_index = range(num_frames), then the common duplicate / midpoint-subsample rule
            # (LSTP_blip2_module.py:254-266), on the device through the same index-map kernel
            idx = _full_range_subsample(n, nframe, b, pixel_values.device)
            return ops.gather_frames(pixel_values, idx).view(b * nframe, *pixel_values.shape[2:]), idx, None
        if of_mask is None:
            of_mask = torch.ones(b, of.shape[1] + 2, dtype=torch.long, device=of.device)
        _, logits = self.temporal_encoder(encoder_embeds=of, attention_mask=of_mask, encoder_hidden_states=batch["sampler_question"],
                                          encoder_attention_mask=batch["sampler_question_attention_mask"], mode=self.TGB_MODE)
        if noise is None:
            noise = -torch.empty(2, 2 * b, of.shape[1], device=of.device).exponential_().log()
        sel = ops.span_select(logits, noise, 0.5)
        if self.V_FROM_LENGTHS:
            v = torch.as_tensor(list(batch["of_lengths"]), dtype=torch.int32, device=of.device)
        else:
            v = n + 2
        idx = ops.span_to_frames(sel, v, n, nframe, self.MAP)
        return ops.gather_frames(pixel_values, idx).view(b * nframe, *pixel_values.shape[2:]), idx, logits

    def _prefix(self, batch, sampled: torch.Tensor, batch_size: int, nframe: int) -> torch.Tensor:
        """ViT -> Q-Former -> pooling + language_projection of one flavour -> language_model_inputs [B, P, H].  With autograd on
        and the prefix side trainable (``freeze_weights``) the result carries gradients to the Q-Former / query tokens /
        projection (train.prefix_with_grad: HIP forward, PyTorch-recompute backward); the vision tower is frozen."""
        with torch.no_grad():
            img = self.model.vision_model(pixel_values=sampled, return_dict=True, act_output=True).last_hidden_state
        widths = list(batch["widths"]) if self.WIDTHS else [nframe] * batch_size
        pool = "mean" if self.WIDTHS else "concat"
        if torch.is_grad_enabled() and self.model.query_tokens.requires_grad:
            qi = qm = None
            if self.ARCH == "instructblip":
                rep = torch.as_tensor(widths, device=img.device)
                qi = torch.repeat_interleave(batch["qformer_text"], rep, 0)
                qm = torch.repeat_interleave(batch["qformer_text_attention_mask"], rep, 0)
            return train.prefix_with_grad(self.model, img, qi, qm, widths, pool, dropout=self._dropout())
        with torch.no_grad():
            return self._prefix_nograd(batch, img, widths, pool)

    DROPOUT_P = 0.1             # hidden_dropout_prob / attention_probs_dropout_prob of the Q-Former and BERT configs the reference builds

    def _dropout(self) -> Optional["train.Dropout"]:
        """Training mode (``self.training``): the reference's dropout sites draw fresh masks (seedable through ``dropout_generator``);
        eval mode -- what every parity fixture is recorded in -- has none."""
        if not self.training:
            return None
        return train.Dropout(self.DROPOUT_P, generator=getattr(self, "dropout_generator", None))

    def _prefix_nograd(self, batch, img, widths, pool):
        query_tokens = self.model.query_tokens.expand(img.shape[0], -1, -1)
        if self.ARCH == "instructblip":
            rep = torch.as_tensor(widths, device=img.device)
            qi = torch.repeat_interleave(batch["qformer_text"], rep, 0)
            qm = torch.repeat_interleave(batch["qformer_text_attention_mask"], rep, 0)
            am = torch.cat([torch.ones(query_tokens.shape[:-1], dtype=torch.long, device=img.device), qm], dim=1)
            qo = self.model.qformer(input_ids=qi, attention_mask=am, query_embeds=query_tokens, encoder_hidden_states=img,
                                    encoder_attention_mask=None, return_dict=True).last_hidden_state
        else:
            qo = self.model.qformer(query_embeds=query_tokens, encoder_hidden_states=img, encoder_attention_mask=None)[0]
        qo = qo[:, : query_tokens.size(1), :]
        return self.model.language_projection.pool(qo, widths, pool)

    def _has_frames(self, batch) -> bool:
        fr = batch["frames"]
        return not (isinstance(fr, list) and len(fr) == 0)

    # ---- the reference's entry points
    def _lm_inputs(self, batch, noise=None, train: bool = False):
        """Everything up to ``language_model_inputs`` for forward / eval_forward of this flavour."""
        batch_size = batch["answer"].shape[0]
        nframe = batch["nframe"]
        if self.WIDTHS:
            if not self._has_frames(batch):
                return None, None, None                             # text-only batch (LSTP_Vicuna_IVT_module.py:342)
            return self._prefix(batch, batch["frames"], batch_size, nframe), None, None
        pixel_values = batch["frames"]
        num_frames = pixel_values.size(0) // batch_size
        pixel_values = pixel_values.view(batch_size, num_frames, *pixel_values.shape[1:])
        of = of_mask = None
        if self.SAMPLER:
            if self.EVAL_RAFT and not train:
                of = self._flow_of_candidates(pixel_values)
            else:
                of, of_mask = batch["of"], batch["of_mask"]
        sampled, idx, logits = self._sample(batch, pixel_values, nframe, of, of_mask, noise)
        return self._prefix(batch, sampled, batch_size, nframe), idx, logits

    @torch.no_grad()
    def eval_forward(self, batch, noise: Optional[torch.Tensor] = None, return_stages: bool = False):
        """``eval_forward`` of the flavour (e.g. src/models/LSTP_module.py:370-513).  ``noise`` injects the Gumbel noise
        [2, 2B, L] (tests); ``return_stages`` additionally returns (frame_idx, of_logits, language_model_inputs)."""
        lm_inputs, idx, logits = self._lm_inputs(batch, noise)
        lm = self.model.language_model
        emb = self.model.get_input_embeddings()(batch["question"])
        if lm_inputs is None:
            attention_mask, inputs_embeddings = batch["question_attention_mask"], emb
        else:
            lm_inputs = lm_inputs.to(emb.dtype)
            mask = torch.ones(lm_inputs.size()[:-1], dtype=torch.long, device=lm_inputs.device)
            attention_mask = torch.cat([mask, batch["question_attention_mask"]], dim=1)
            inputs_embeddings = torch.cat([lm_inputs, emb], dim=1)
        outputs = None
        if getattr(self, "fast_decode", True):       # (round 6: on by default) hipGraph-replayed decode of the same HF weights whenever generate_configs is a
            # configuration decode.graph_generate reproduces exactly (greedy, one beam, no padding, Llama / T5); None -> HF generate below
            from .decode import graph_generate
            outputs = graph_generate(self, lm, inputs_embeddings, attention_mask, self.generate_configs)
        if outputs is None:
            outputs = lm.generate(inputs_embeds=inputs_embeddings, attention_mask=attention_mask, **self.generate_configs)
        if self.model.config.text_config.architectures[0] == "LLaMAForCausalLM":
            outputs[outputs == 0] = 2
        if return_stages:
            return outputs, dict(frame_idx=idx, of_logits=logits, language_model_inputs=lm_inputs, inputs_embeds=inputs_embeddings)
        return outputs

    def forward(self, batch, noise: Optional[torch.Tensor] = None):
        """``forward`` of the flavour -> (loss, logits) (e.g. src/models/LSTP_module.py:183-368): frozen HIP prefix path,
        [prefix | question + answer] through the language model, shifted CE (decoder-only; HIP kernels of train.py) or the
        seq2seq model's own label loss (Flan-T5).  SF flavours add the self-refinement MRC loss (LSTP_SF_module.py:147-298)."""
        lm_inputs, idx, of_logits = self._lm_inputs(batch, noise, train=True)
        lm = self.model.language_model
        pad_id = self.processor.tokenizer.pad_token_id
        if self.model.config.use_decoder_only_language_model:
            p = 0 if lm_inputs is None else lm_inputs.shape[1]
            llm_tokens, _, labels = train.concat_text_input_output(batch["question"], batch["question_attention_mask"], batch["answer"],
                                                                   batch["answer_attention_mask"], pad_id, p)
            emb = lm.get_input_embeddings()(llm_tokens["input_ids"])
            attention_mask = llm_tokens["attention_mask"]
            if lm_inputs is not None:
                emb = torch.cat([lm_inputs.to(emb.dtype), emb], dim=1)
                attention_mask = torch.cat([torch.ones(lm_inputs.shape[:2], dtype=torch.long, device=emb.device), attention_mask], dim=1)
            logits = lm(inputs_embeds=emb, attention_mask=attention_mask)[0]
            loss = train.shifted_cross_entropy(logits, labels)
        else:
            emb = lm.get_input_embeddings()(batch["question"])
            attention_mask = batch["question_attention_mask"]
            if lm_inputs is not None:
                emb = torch.cat([lm_inputs.to(emb.dtype), emb], dim=1)
                attention_mask = torch.cat([torch.ones(lm_inputs.shape[:2], dtype=attention_mask.dtype, device=emb.device), attention_mask], dim=1)
            labels = batch["answer"].masked_fill(pad_id == batch["answer"], -100)
            out = lm(inputs_embeds=emb, attention_mask=attention_mask, labels=labels)
            loss, logits = out[0], out[1]
        if self.SELF_REFINE:
            scores, st, en = refine.self_refine_targets(self, batch, lambda ids: self.processor.batch_decode(ids, skip_special_tokens=True),
                                                        num_frames=batch["frames"].shape[0] // batch["answer"].shape[0])
            # "2) optimize temporal encoder" (LSTP_SF_module.py:275-296): the TGB runs again WITH autograd -- the MRC loss on its
            # span logits is what trains the sampler (train.tgb_with_grad: own GEMM / attention kernels forward and backward)
            if torch.is_grad_enabled() and any(p.requires_grad for p in self.temporal_encoder.parameters()):
                _, of_logits = train.tgb_with_grad(self.temporal_encoder, batch["of"], batch["of_mask"], batch["sampler_question"],
                                                   batch["sampler_question_attention_mask"], self.TGB_MODE, dropout=self._dropout())
            loss = loss + refine.mrc_loss(of_logits, st, en)
        return loss, logits

    def model_step(self, batch):
        loss, logits = self.forward(batch)
        return loss, logits, batch["answer"]

    def eval_model_step(self, batch):
        outputs = self.eval_forward(batch)
        return self.processor.batch_decode(outputs, skip_special_tokens=True), batch["text_answer"]

    def training_step(self, batch, batch_idx: int):
        loss, _, _ = self.model_step(batch)
        self.log("train/loss", loss.detach(), on_step=True, on_epoch=True, prog_bar=True)
        return loss

    def validation_step(self, batch, batch_idx: int) -> None:
        preds, targets = self.eval_model_step(batch)
        self.val_bleu_score(preds, targets)
        self.log("val/score", self.val_bleu_score.compute(), on_step=False, on_epoch=True, prog_bar=True)

    def on_validation_epoch_end(self) -> None:
        self.val_score_best = max(self.val_score_best, self.val_bleu_score.compute())
        self.log("val/score_best", self.val_score_best, sync_dist=True, prog_bar=True)

    def test_step(self, batch, batch_idx: int) -> None:
        preds, targets = self.eval_model_step(batch)
        self.test_bleu_score(preds, targets)
        self.log("test/bleu_score", self.test_bleu_score.compute(), on_step=False, on_epoch=True, prog_bar=True)

    def on_train_start(self) -> None:
        self.val_bleu_score.reset()
        self.val_score_best = 0.0

    def configure_optimizers(self) -> Dict[str, Any]:
        """LSTP_module.py:634-679: ``optimizer(params=self.parameters())`` (a functools.partial from Hydra); "cosine" =
        transformers' warmup-cosine with warmup = int(trainer.max_steps * scheduler_params["warmup_steps"]), per-epoch."""
        optimizer = self.hparams.optimizer(params=[p for p in self.parameters() if p.requires_grad])
        if self.hparams.scheduler is None:
            return {"optimizer": optimizer}
        if self.hparams.scheduler != "cosine":
            raise NotImplementedError("UNKONWN SCHEDULER")
        max_steps = self.trainer.max_steps
        warmup_steps = int(max_steps * self.hparams.scheduler_params["warmup_steps"])
        sch = torch.optim.lr_scheduler.LambdaLR(optimizer, train.cosine_schedule_lambda(warmup_steps, max_steps))
        return {"optimizer": optimizer, "lr_scheduler": {"scheduler": sch, "monitor": "val/score", "interval": "epoch", "frequency": 1}}

    def freeze_weights(self):
        """LSTP_module.py:669-675: RAFT, the vision tower and the language model are frozen (Q-Former, projections and the
        TGB stay trainable); IV / IVT instead freeze the TGB and leave the LLM to peft (LSTP_Vicuna_IVT_module.py:682-690)."""
        train.enable_prefix_training(self.model)      # Q-Former, query tokens, projections: trainable in every flavour
        for p in self.of_extractor.parameters():
            p.requires_grad = False
        for p in self.model.vision_model.parameters():
            p.requires_grad = False
        # the TGB: trainable wherever the reference leaves it so (LSTP_module.py / LSTP_SF_module.py:747-751 freeze only the vision
        # tower and the language model).  In LSTP_module it receives no gradient (selection is an argmax); in the SF flavours the
        # MRC loss reaches it through train.tgb_with_grad.  The sinusoid tables are buffers-by-convention (requires_grad False).
        if self.WIDTHS:
            for p in self.temporal_encoder.parameters():
                p.requires_grad = False
        else:
            for n, p in self.temporal_encoder.named_parameters():
                p.requires_grad = "embed_positions" not in n
            for p in self.model.language_model.parameters():
                p.requires_grad = False

    def concat_text_input_output(self, input_ids, input_atts, output_ids, output_atts):
        """LSTP_module.py:677-700 (device kernel, no per-row host syncs); returns (llm_tokens, input_part_targets_len)."""
        return train.concat_text_input_output(input_ids, input_atts, output_ids, output_atts)


def _full_range_subsample(n: int, nframe: int, b: int, device) -> torch.Tensor:
    """cand_index = range(n) -> duplicate-double while shorter than nframe -> float64 linspace midpoint subsample
    (LSTP_blip2_module.py:254-266), for every clip of the batch; through the span kernel with the full-range fallback span."""
    # a span whose end precedes its start (start 1, end 0) selects nothing, and the index-map rule then falls back to
    # `cand_index = list(range(num_frames))` -- exactly the no-sampler candidate list
    sel = torch.cat([torch.ones(1, b, dtype=torch.int64, device=device), torch.zeros(1, b, dtype=torch.int64, device=device)], dim=1)
    return ops.span_to_frames(sel, n + 2, n, nframe, "A")


# ------------------------------------------------------------------------------------------ the flavours
class LSTPModule(_LSTPLightningBase):
    """src.models.LSTP_module.LSTPModule (configs/model/LSTP_instructblip.yaml)."""
    ARCH, SAMPLER, EVAL_RAFT, TGB_MODE, MAP = "instructblip", True, True, "multi_modal", "A"


class LSTPBlip2Module(_LSTPLightningBase):
    """src.models.LSTP_blip2_module.LSTPModule (configs/model/LSTP_blip2.yaml): the sampler is commented out -- BASELINE C1."""
    ARCH, SAMPLER = "blip2", False


class LSTPSFModule(_LSTPLightningBase):
    """src.models.LSTP_SF_module.LSTPSFModule (configs/model/LSTP_SF_instructblip.yaml)."""
    ARCH, TGB_MODE, MAP, V_FROM_LENGTHS, SELF_REFINE = "instructblip", "fusion", "B", True, True


class LSTPSFBlip2Module(_LSTPLightningBase):
    """src.models.LSTP_SF_blip2_module.LSTPSFModule (configs/model/LSTP_SF_blip2.yaml) -- BASELINE C2."""
    ARCH, TGB_MODE, MAP, V_FROM_LENGTHS, SELF_REFINE = "blip2", "fusion", "B", True, True


class LSTPVicunaIVModule(_LSTPLightningBase):
    """src.models.LSTP_Vicuna_IV_module.LSTPModule."""
    ARCH, SAMPLER, WIDTHS, SAMPLER_IS_STATE_DICT = "instructblip", False, True, True


class LSTPVicunaIVTModule(LSTPVicunaIVModule):
    """src.models.LSTP_Vicuna_IVT_module.LSTPModule (configs/model/LSTP_instructblip_IVT.yaml) -- BASELINE C5 (LoRA)."""
    LORA = "CAUSAL_LM"


class LSTPBlip2IVModule(_LSTPLightningBase):
    """src.models.LSTP_Blip2_IV_module.LSTPModule."""
    ARCH, SAMPLER, WIDTHS, SAMPLER_IS_STATE_DICT = "blip2", False, True, True


class LSTPBlip2IVTModule(LSTPBlip2IVModule):
    """src.models.LSTP_Blip2_IVT_module.LSTPModule."""
    LORA = "SEQ_2_SEQ_LM"


TARGETS = {
    "src.models.LSTP_module.LSTPModule": LSTPModule,
    "src.models.LSTP_blip2_module.LSTPModule": LSTPBlip2Module,
    "src.models.LSTP_SF_module.LSTPSFModule": LSTPSFModule,
    "src.models.LSTP_SF_blip2_module.LSTPSFModule": LSTPSFBlip2Module,
    "src.models.LSTP_Vicuna_IV_module.LSTPModule": LSTPVicunaIVModule,
    "src.models.LSTP_Vicuna_IVT_module.LSTPModule": LSTPVicunaIVTModule,
    "src.models.LSTP_Blip2_IV_module.LSTPModule": LSTPBlip2IVModule,
    "src.models.LSTP_Blip2_IVT_module.LSTPModule": LSTPBlip2IVTModule,
}
