"""-m gpu: the OUTER boundary: reference-signature constructors / loaders and the LightningModule twins, built
from a saved HF config directory + checkpoint files the way the reference's drivers build them, against

  * the tiny e2e vectors recorded from the reference's ``eval.utils.model.LSTP(_blip2).generate`` (load_pretrained_model path);
  * ``tests/golden/tiny_modules.npz``: outputs of the reference's own LightningModule ``eval_forward`` methods
    (src.models.LSTP_module / LSTP_SF_module / LSTP_blip2_module / LSTP_SF_blip2_module), incl. the no-sampler BLIP-2 +
    seq2seq (T5) flavour of BASELINE configs C1 / C2.
fp32 mode: frame indices and generated ids bit-exact, tensors to 2e-4 of their scale."""
import functools
import os

import pytest
import torch

from conftest import deq, full_state_dict, load_golden, write_hf_config

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from videotgb_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


class BE(dict):
    __getattr__ = dict.__getitem__


def close(name, got, ref, tol=2e-4):
    got, ref = got.detach().float().cpu(), ref.float()
    err, scale = (got - ref).abs().max().item(), ref.abs().max().item()
    print(f"[{name}] max|diff|={err:.3e} max|ref|={scale:.3e}")
    assert err <= tol * scale, name


@pytest.mark.parametrize("arch", ["instructblip", "blip2"])
def test_load_pretrained_model_then_generate(dev, tmp_path, arch):
    """eval/inference.py's construction path: load_pretrained_model(ckpt, model_base, sampler_base, device, lora) ->
    model.generate(...) -> (output_ids, cand_index), from a config directory and a Lightning checkpoint on disk."""
    from videotgb_amd import builder_utils, models
    from videotgb_amd.synth import tiny_cfg
    cfg = tiny_cfg(arch)
    cfg.vit.image = 56
    base = write_hf_config(str(tmp_path / f"{arch}-tiny"), arch, cfg)
    probe = models.build_language_model(models.load_hf_config(base, arch))
    ckpt = str(tmp_path / "last.ckpt")
    torch.save({"state_dict": full_state_dict(cfg, probe)}, ckpt)
    model, proc, sproc = builder_utils.load_pretrained_model(ckpt, base, None, dev, lora=False, compute_dtype="f32", load_processors=False,
                                                             tgb_cfg=cfg.tgb)
    assert isinstance(model, models.LSTP if arch == "instructblip" else models.LSTP_blip2) and proc is None
    model.to(dev)
    g = load_golden(f"tiny_{arch}_e2e")
    te = BE(input_ids=g["prompt_ids"].to(dev), attention_mask=g["prompt_mask"].to(dev))
    if arch == "instructblip":
        te["qformer_input_ids"], te["qformer_attention_mask"] = g["qformer_ids"].to(dev), g["qformer_mask"].to(dev)
    se = BE(input_ids=g["sampler_ids"].to(dev), attention_mask=g["sampler_mask"].to(dev))
    ids, cand = model.generate(deq(g, "frames_q8").to(dev), deq(g, "flow_frames_q8").to(dev), int(g["nframe"]), te, se, do_sample=False,
                               temperature=None, max_new_tokens=6, use_cache=False, noise=g["noise"].to(dev))
    assert cand.cpu().tolist() == g["cand_index"].tolist()
    assert ids.cpu().tolist() == g["greedy_ids"].tolist()
    with pytest.raises(ValueError):
        builder_utils.load_pretrained_model(ckpt, str(tmp_path), None, dev, load_processors=False)


def up4(q):
    return q.float().repeat_interleave(4, -2).repeat_interleave(4, -1)


CASES = {"ib": ("LSTPModule", "instructblip", "llama", False), "sf": ("LSTPSFModule", "instructblip", "llama", True),
         "b2": ("LSTPBlip2Module", "blip2", "t5", False), "sfb2": ("LSTPSFBlip2Module", "blip2", "t5", True)}


@pytest.mark.parametrize("tag", list(CASES))
def test_lightning_module_twin_eval_forward(dev, tmp_path, tag):
    """Each twin is constructed with the reference's Hydra kwargs from files on disk (HF config dir, RAFT .pth with
    DataParallel ``module.`` keys), takes the Lightning checkpoint's state_dict strictly, and reproduces what the reference's
    own ``eval_forward`` produced for the same batch dict."""
    from videotgb_amd import models, modules
    from videotgb_amd.synth import path_state_dict, tiny_cfg
    cls_name, arch, llm, uses_of = CASES[tag]
    cfg = tiny_cfg(arch)                                         # ViT at 224: the candidate frames double as RAFT input
    base = write_hf_config(str(tmp_path / f"{arch}-tiny"), arch, cfg, llm)
    sd = full_state_dict(cfg, models.build_language_model(models.load_hf_config(base, arch)))
    raft_pth = str(tmp_path / "raft-things.pth")
    torch.save({"module." + k[len("of_extractor."):]: v for k, v in sd.items() if k.startswith("of_extractor.")}, raft_pth)
    proc = BE(tokenizer=BE(pad_token_id=0), batch_decode=lambda ids, skip_special_tokens=True: [" ".join(map(str, r)) for r in ids.tolist()])
    m = getattr(modules, cls_name)(model_name_or_path=base, sampler_name_or_path=str(tmp_path / "no-bert-weights"),
                                   of_extractor_name_or_path=raft_pth, temperature=1.0,
                                   optimizer=functools.partial(torch.optim.AdamW, lr=1e-4), scheduler="cosine",
                                   scheduler_params={"warmup_steps": 0.1}, generate_configs=dict(do_sample=False, max_new_tokens=6),
                                   compute_dtype="f32", processor=proc, tgb_cfg=cfg.tgb)
    assert m.model.config.use_decoder_only_language_model == (llm == "llama")
    missing, unexpected = m.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    m.to(dev)
    g = load_golden("tiny_modules")
    B = 2
    batch = dict(frames=(up4(g["frames_q8"]) / 48).to(dev), nframe=int(g["nframe"]), of_lengths=g["of_lengths"].tolist(),
                 answer=torch.zeros(B, 1, dtype=torch.long, device=dev), text_answer=[""] * B,
                 sampler_question=g[f"{tag}_sampler_ids"].to(dev), sampler_question_attention_mask=g[f"{tag}_sampler_mask"].to(dev),
                 qformer_text=g[f"{tag}_qformer_ids"].to(dev), qformer_text_attention_mask=g[f"{tag}_qformer_mask"].to(dev),
                 question=g[f"{tag}_question"].to(dev), question_attention_mask=g[f"{tag}_question_mask"].to(dev))
    if uses_of:
        batch.update(of=(up4(g["of_q8"]) / 127).to(dev), of_mask=g["of_mask"].to(dev))
    noise = g[f"{tag}_noise"].to(dev) if f"{tag}_noise" in g else None
    cap = []
    h = m.model.language_model.lm_head.register_forward_hook(lambda mod, i, o: cap.append(o))
    m.fast_decode = False                                        # (HF generate: the hook below captures ITS first-step logits)
    ids, st = m.eval_forward(batch, noise=noise, return_stages=True)
    h.remove()
    assert st["frame_idx"].cpu().tolist() == g[f"{tag}_frame_idx"].tolist()
    if f"{tag}_of_logits" in g:
        close(f"{tag} of_logits", st["of_logits"], g[f"{tag}_of_logits"])
    else:
        assert st["of_logits"] is None                        # no-sampler flavour (LSTP_blip2_module.py:254)
    close(f"{tag} language_model_inputs", st["language_model_inputs"], g[f"{tag}_prefix"])
    close(f"{tag} first-step logits", cap[0][:, -1], g[f"{tag}_first_logits"])
    assert ids.cpu().tolist() == g[f"{tag}_ids"].tolist()
    m.fast_decode = True                                         # (round 6: the twins' default) the graph decoder emits the same ids
    ids2 = m.eval_forward(batch, noise=noise)
    assert ids2.cpu().tolist() == g[f"{tag}_ids"].tolist()
    # Lightning-facing methods
    preds, labels = m.eval_model_step(batch)
    assert len(preds) == B and labels == [""] * B
    opt = m.configure_optimizers()
    assert set(opt) == {"optimizer", "lr_scheduler"} and opt["lr_scheduler"]["interval"] == "epoch"
    trainable = {n.split(".")[0] + "." + n.split(".")[1] for n, p in m.named_parameters() if p.requires_grad}
    assert not any(n.startswith("of_extractor") or n.startswith("model.vision_model") or n.startswith("model.language_model") for n in trainable)
    assert {"model.qformer", "model.query_tokens", "model.language_projection"} <= trainable
    # training forward of the flavour: loss is finite and reaches the Q-Former (HIP forward, recompute backward)
    batch.update(answer=torch.randint(3, 100, (B, 4), device=dev), answer_attention_mask=torch.ones(B, 4, dtype=torch.long, device=dev))
    if not uses_of:
        batch.update(of=(up4(g["of_q8"]) / 127).to(dev)[:, :8], of_mask=torch.ones(B, 10, dtype=torch.long, device=dev))
    m.SELF_REFINE = False                                     # (the SF pseudo-label loop is covered by tests/test_gpu_refine.py)
    loss, logits = m.forward(batch, noise=(g[f"{tag}_noise"].to(dev) if uses_of and f"{tag}_noise" in g else None))
    loss.backward()
    assert torch.isfinite(loss) and m.model.query_tokens.grad is not None and m.model.query_tokens.grad.abs().max() > 0


def test_sf_module_self_refinement_forward(dev, tmp_path):
    """LSTPSFModule.forward with the pseudo-label loop on (src/models/LSTP_SF_module.py:147-298): per-frame answers through the
    twin's own prefix / decoder, ROUGE vs the text answers, monotone-stack span, MRC loss added to the LM loss."""
    from videotgb_amd import models, modules, refine
    from videotgb_amd.synth import tiny_cfg
    cfg = tiny_cfg("instructblip")
    base = write_hf_config(str(tmp_path / "instructblip-tiny"), "instructblip", cfg, "llama")
    sd = full_state_dict(cfg, models.build_language_model(models.load_hf_config(base, "instructblip")))
    raft_pth = str(tmp_path / "raft.pth")
    torch.save({"module." + k[len("of_extractor."):]: v for k, v in sd.items() if k.startswith("of_extractor.")}, raft_pth)
    words = ["w%d" % i for i in range(120)]
    proc = BE(tokenizer=BE(pad_token_id=0), batch_decode=lambda ids, skip_special_tokens=True: [" ".join(words[t] for t in r if t > 2) for r in ids.tolist()])
    m = modules.LSTPSFModule(model_name_or_path=base, sampler_name_or_path=str(tmp_path / "x"), of_extractor_name_or_path=raft_pth, temperature=1.0,
                             optimizer=functools.partial(torch.optim.AdamW, lr=1e-4), scheduler=None, scheduler_params={},
                             generate_configs=dict(do_sample=False, max_new_tokens=4), compute_dtype="f32", processor=proc, tgb_cfg=cfg.tgb)
    m.load_state_dict(sd, strict=True)
    m.to(dev)
    g = load_golden("tiny_modules")
    B = 2
    batch = dict(frames=(up4(g["frames_q8"]) / 48).to(dev), nframe=int(g["nframe"]), of_lengths=g["of_lengths"].tolist(),
                 of=(up4(g["of_q8"]) / 127).to(dev), of_mask=g["of_mask"].to(dev),
                 sampler_question=g["sf_sampler_ids"].to(dev), sampler_question_attention_mask=g["sf_sampler_mask"].to(dev),
                 qformer_text=g["sf_qformer_ids"].to(dev), qformer_text_attention_mask=g["sf_qformer_mask"].to(dev),
                 question=g["sf_question"].to(dev), question_attention_mask=g["sf_question_mask"].to(dev),
                 answer=torch.tensor([[1, 21, 45, 70], [1, 16, 96, 15]], device=dev), answer_attention_mask=torch.ones(B, 4, dtype=torch.long, device=dev),
                 text_answer=["w21 w45 w70", "w16 w96 w15"])
    scores, st, en = refine.self_refine_targets(m, batch, lambda ids: proc.batch_decode(ids), num_frames=8, max_length=32 + 5 + 6)
    assert scores.shape == (B, 8) and st.shape == en.shape == (B,) and bool((st <= en).all())
    loss, logits = m.forward(batch, noise=g["sf_noise"].to(dev))
    assert torch.isfinite(loss) and logits.shape[0] == B
    # round 3: the MRC loss TRAINS the sampler (LSTP_SF_module.py:275-296; freeze_weights :747-751 leaves the TGB trainable): its
    # parameters are in the optimizer, the loss reaches them, and a step changes the span logits the fused inference stage computes
    loss.backward()
    te = m.temporal_encoder
    assert te.mrc_head.weight.requires_grad and te.mrc_head.weight.grad is not None and te.mrc_head.weight.grad.abs().max() > 0
    reached = [n for n, p in te.named_parameters() if p.grad is not None and p.grad.abs().max() > 0]
    assert len(reached) >= 30 and any("crossattention" in n for n in reached) and any("temporal_embeddings.projection" in n for n in reached)
    assert all(p.grad is None for p in m.model.vision_model.parameters()) and all(p.grad is None for p in m.model.language_model.parameters())
    opt_ids = {id(p) for grp in m.configure_optimizers()["optimizer"].param_groups for p in grp["params"]}
    assert id(te.mrc_head.weight) in opt_ids
    with torch.no_grad():
        _, before = te(encoder_embeds=batch["of"], attention_mask=batch["of_mask"], encoder_hidden_states=batch["sampler_question"],
                       encoder_attention_mask=batch["sampler_question_attention_mask"], mode="fusion")
    torch.optim.SGD([p for p in te.parameters() if p.grad is not None], lr=0.5).step()
    with torch.no_grad():                                    # the packed weight table follows the parameters' version counters
        _, after = te(encoder_embeds=batch["of"], attention_mask=batch["of_mask"], encoder_hidden_states=batch["sampler_question"],
                      encoder_attention_mask=batch["sampler_question_attention_mask"], mode="fusion")
    assert (after - before).abs().max() > 1e-4
    # training mode: dropout masks are drawn (seedable), the loss stays finite
    m.train()
    m.model.language_model.eval()
    m.dropout_generator = torch.Generator(device=dev).manual_seed(11)
    loss2, _ = m.forward(batch, noise=g["sf_noise"].to(dev))
    assert torch.isfinite(loss2) and abs(loss2.item() - loss.item()) > 1e-6
