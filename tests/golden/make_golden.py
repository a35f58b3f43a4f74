#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE.

Runs only in the build container (it imports /root/reference, which never travels to
the GPU box).  Nothing of the reference's source is written anywhere: the outputs are
data -- quantised inputs and the reference's fp32 outputs at every stage boundary.

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

Recipe: shim the transformers-4.36 names the reference imports, stub the
unused heavy imports, build the reference modules at a tiny configuration, load OUR seeded
state_dict (videotgb_amd.synth) with strict key checking (which also pins the state_dict
layout of Appendix A), call .eval(), inject recorded Gumbel noise, and record.
"""
import importlib.machinery
import inspect
import os
import sys
import tempfile
import textwrap
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)


# ----------------------------------------------------------------------------- shim
def install_shim():
    import transformers  # noqa
    import transformers.modeling_utils as mu
    import transformers.pytorch_utils as pu
    from transformers import PreTrainedModel
    from transformers.pytorch_utils import apply_chunking_to_forward, prune_linear_layer

    def find_pruneable_heads_and_indices(heads, n_heads, head_size, already_pruned_heads):
        mask = torch.ones(n_heads, head_size)
        heads = set(heads) - already_pruned_heads
        for head in heads:
            head = head - sum(1 if h < head else 0 for h in already_pruned_heads)
            mask[head] = 0
        mask = mask.view(-1).contiguous().eq(1)
        return heads, torch.arange(len(mask))[mask].long()

    for m in (mu, pu):
        for n, f in (("apply_chunking_to_forward", apply_chunking_to_forward),
                     ("prune_linear_layer", prune_linear_layer),
                     ("find_pruneable_heads_and_indices", find_pruneable_heads_and_indices)):
            if not hasattr(m, n):
                setattr(m, n, f)
    if not hasattr(PreTrainedModel, "get_head_mask"):
        PreTrainedModel.get_head_mask = lambda self, head_mask, n, is_attention_chunked=False: [None] * n
    PreTrainedModel.all_tied_weights_keys = {}
    for name in ("sentence_transformers", "peft", "av", "cv2", "decord", "ffmpeg"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
            sys.modules[name].__spec__ = importlib.machinery.ModuleSpec(name, None)   # transformers probes find_spec()
    st = sys.modules["sentence_transformers"]
    st.SentenceTransformer = object
    st.util = object
    for n in ("get_peft_model", "LoraConfig", "TaskType", "PeftModel", "PeftMixedModel"):
        setattr(sys.modules["peft"], n, object)
    dec = sys.modules["decord"]
    dec.cpu = lambda *a, **k: None
    dec.bridge = types.SimpleNamespace(set_bridge=lambda *a, **k: None)
    sys.path.insert(0, REF)


def q8(shape, gen, scale):
    """int8 tensor + its exact fp32 dequantisation (value = q * scale)."""
    q = torch.randint(-127, 128, shape, generator=gen, dtype=torch.int32).to(torch.int8)
    return q, q.float() * scale


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1e3:.0f} kB")


# ----------------------------------------------------------------------------- models
def tiny_text_config(kind, hidden):
    """The tiny third-party LLM of the fixtures: Llama (decoder-only) or T5 (seq2seq: the Flan-T5 hookup of configs C1 / C2)."""
    from transformers import LlamaConfig, T5Config
    if kind == "t5":
        return T5Config(vocab_size=120, d_model=hidden, d_kv=16, d_ff=64, num_layers=2, num_decoder_layers=2, num_heads=2,
                        feed_forward_proj="gated-gelu", tie_word_embeddings=False, decoder_start_token_id=0, pad_token_id=0, eos_token_id=1,
                        architectures=["T5ForConditionalGeneration"])
    return LlamaConfig(hidden_size=hidden, intermediate_size=64, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=2,
                       vocab_size=120, architectures=["LlamaForCausalLM"], bos_token_id=1, eos_token_id=2, pad_token_id=0)


def build_reference(arch, cfg, sd, llm="llama", save_dir=None):
    """Reference eval-side module (eval/utils/model.py LSTP / LSTP_blip2) at the tiny config."""
    from transformers import (Blip2Config, Blip2QFormerConfig, Blip2VisionConfig, InstructBlipConfig,
                              InstructBlipQFormerConfig, InstructBlipVisionConfig, LlamaConfig)
    import eval.utils.model as em
    v, q, t = cfg.vit, cfg.qformer, cfg.tgb
    vkw = dict(hidden_size=v.hidden, intermediate_size=v.mlp, num_hidden_layers=v.layers,
               num_attention_heads=v.heads, image_size=v.image, patch_size=v.patch, layer_norm_eps=v.eps)
    qkw = dict(hidden_size=q.hidden, num_hidden_layers=q.layers, num_attention_heads=q.heads,
               intermediate_size=q.ffn, encoder_hidden_size=q.enc_hidden, vocab_size=q.vocab,
               max_position_embeddings=q.max_pos, cross_attention_frequency=q.cross_freq)
    tc = tiny_text_config(llm, cfg.llm_hidden)
    if arch == "instructblip":
        c = InstructBlipConfig(vision_config=InstructBlipVisionConfig(**vkw).to_dict(),
                               qformer_config=InstructBlipQFormerConfig(**qkw).to_dict(),
                               text_config=tc.to_dict(), num_query_tokens=q.n_query)
        cls = em.LSTP
    else:
        c = Blip2Config(vision_config=Blip2VisionConfig(**vkw).to_dict(),
                        qformer_config=Blip2QFormerConfig(**qkw).to_dict(),
                        text_config=tc.to_dict(), num_query_tokens=q.n_query)
        cls = em.LSTP_blip2
    d = save_dir or tempfile.mkdtemp()
    c.save_pretrained(d)
    # the reference hard-codes BertConfig(fusion_layer=6, encoder_width=768) for the TGB
    # (eval/utils/model.py:35); shrink it for the tiny fixture through the class it calls.
    from transformers import BertConfig
    orig = em.BertConfig
    em.BertConfig = lambda **kw: BertConfig(hidden_size=t.hidden, num_hidden_layers=t.layers,
                                            num_attention_heads=t.heads, intermediate_size=t.ffn,
                                            vocab_size=t.vocab, max_position_embeddings=t.max_pos,
                                            fusion_layer=t.fusion_layer, encoder_width=t.enc_width)
    try:
        ref = cls(d, "cpu")
    finally:
        em.BertConfig = orig
    ref.eval()
    # LLM weights: third-party on both sides; seed them through the same generator
    from videotgb_amd.synth import synth_tensor
    full = dict(sd)
    for k, p in ref.model.language_model.state_dict().items():
        full["model.language_model." + k] = synth_tensor("model.language_model." + k, tuple(p.shape))
    for k in list(full):       # T5 ties encoder / decoder token embeddings to `shared`
        if k.endswith("encoder.embed_tokens.weight") or k.endswith("decoder.embed_tokens.weight"):
            full[k] = full[k.rsplit(".", 3)[0] + ".shared.weight"]
    ref_keys = set(ref.state_dict().keys())
    missing = ref_keys - set(full)
    extra = set(full) - ref_keys
    assert not missing and not extra, f"state_dict layout mismatch: missing={sorted(missing)[:8]} extra={sorted(extra)[:8]}"
    for k, p in ref.state_dict().items():
        assert tuple(p.shape) == tuple(full[k].shape), (k, p.shape, full[k].shape)
    ref.load_state_dict(full, strict=True)
    llm_sd = {k: v for k, v in full.items() if k.startswith("model.language_model.")}
    return ref, tc, llm_sd


class NoiseQueue:
    """Make F.gumbel_softmax draw recorded exponentials."""

    def __init__(self, exps):
        self.exps = list(exps)
        self.used = 0

    def __enter__(self):
        self.orig = torch.Tensor.exponential_
        q = self

        def patched(t, *a, **k):
            e = q.exps[q.used]
            q.used += 1
            return t.copy_(e.reshape(t.shape))
        torch.Tensor.exponential_ = patched
        return self

    def __exit__(self, *a):
        torch.Tensor.exponential_ = self.orig


class BE(dict):
    """HF BatchEncoding stand-in: both ["k"] and .k access (eval/inference.py:76-89)."""
    __getattr__ = dict.__getitem__


def e2e_fixture(arch):
    from videotgb_amd.synth import path_state_dict, tiny_cfg
    cfg = tiny_cfg(arch)
    cfg.vit.image = 56
    sd = path_state_dict(cfg, seed=0)
    ref, tc, llm_sd = build_reference(arch, cfg, sd)
    g = torch.Generator().manual_seed(7)
    T, N, nframe, B = 4, 8, 4, 1
    fq, frames = q8((N, 3, 56, 56), g, 1 / 48)
    ffq, flow_frames = q8((B, T, 3, 224, 224), g, 1 / 48)
    samp_ids = torch.randint(3, cfg.tgb.vocab, (B, 7), generator=g)
    samp_mask = torch.ones_like(samp_ids)
    qf_ids = torch.randint(3, cfg.qformer.vocab, (B, 6), generator=g)
    qf_mask = torch.ones_like(qf_ids)
    prompt = torch.randint(3, tc.vocab_size, (B, 5), generator=g)
    pmask = torch.ones_like(prompt)
    exps = [torch.empty(2 * B, T, 1).exponential_(generator=g) for _ in range(2)]
    noise = torch.stack([-e.log().squeeze(-1) for e in exps])             # [draws, 2B, T]
    cap = {}
    hooks = [
        ref.of_extractor.register_forward_hook(lambda m, i, o: cap.__setitem__("raft_flow", o)),
        ref.temporal_encoder.register_forward_hook(lambda m, i, o: cap.update(tgb_seq=o[0], tgb_logits=o[1])),
        ref.model.vision_model.register_forward_hook(lambda m, i, o: cap.__setitem__("image_embeds", o.last_hidden_state)),
        ref.model.vision_model.register_forward_pre_hook(lambda m, a, k: cap.__setitem__("sampled", k["pixel_values"]), with_kwargs=True),
        ref.model.qformer.register_forward_hook(lambda m, i, o: cap.__setitem__("qformer_seq", o[0])),
        ref.model.language_projection.register_forward_hook(lambda m, i, o: cap.__setitem__("prefix", o)),
    ]
    lm = ref.model.language_model
    orig_gen = lm.generate

    def gen_spy(**kw):
        cap["inputs_embeds"] = kw["inputs_embeds"]
        cap["attention_mask"] = kw["attention_mask"]
        out = orig_gen(**kw, output_scores=True, return_dict_in_generate=True)
        cap["first_logits"] = out.scores[0]
        return out.sequences
    lm.generate = gen_spy
    te = BE(input_ids=prompt, attention_mask=pmask)
    if arch == "instructblip":
        te["qformer_input_ids"] = qf_ids
        te["qformer_attention_mask"] = qf_mask
    with NoiseQueue(exps) as nq:
        ids, cand = ref.generate(frames, flow_frames, nframe, te, BE(input_ids=samp_ids, attention_mask=samp_mask),
                                 do_sample=False, temperature=None, max_new_tokens=6, use_cache=False)
    assert nq.used == 2 and "tgb_logits" in cap, "reference took its bare-except fallback"
    for h in hooks:
        h.remove()
    nq_tok = cfg.qformer.n_query
    save(f"tiny_{arch}_e2e", frames_q8=fq, flow_frames_q8=ffq, q8_scale=np.float32(1 / 48),
         sampler_ids=samp_ids, sampler_mask=samp_mask, qformer_ids=qf_ids, qformer_mask=qf_mask,
         prompt_ids=prompt, prompt_mask=pmask, noise=noise, nframe=nframe,
         raft_flow=cap["raft_flow"], tgb_seq=cap["tgb_seq"], tgb_logits=cap["tgb_logits"],
         cand_index=cand, sampled=cap["sampled"], image_embeds=cap["image_embeds"],
         query_out=cap["qformer_seq"][:, :nq_tok], prefix=cap["prefix"], inputs_embeds=cap["inputs_embeds"],
         first_logits=cap["first_logits"], greedy_ids=ids)
    return ref, cfg, sd


def component_fixtures(ref_ib, cfg_ib, sd_ib, ref_b2, cfg_b2, sd_b2):
    """Per-component vectors with batch > 1, ragged masks and both TGB modes."""
    g = torch.Generator().manual_seed(11)
    # --- TGB: B=2, lengths 12 and 9 (ragged of_mask), text padding, both modes
    B, L = 2, 12
    ofq, of = q8((B, L, 2, 224, 224), g, 1 / 127)
    of_mask = torch.ones(B, L + 2, dtype=torch.long)
    of_mask[1, 9 + 2:] = 0
    tids = torch.randint(3, cfg_ib.tgb.vocab, (B, 8), generator=g)
    tmask = torch.ones_like(tids)
    tmask[1, 5:] = 0
    outs = {}
    with torch.no_grad():
        for mode in ("multi_modal", "fusion", "vision"):
            seq, logits = ref_ib.temporal_encoder(encoder_embeds=of.clone(), attention_mask=of_mask,
                                                  encoder_hidden_states=tids, encoder_attention_mask=tmask, mode=mode)
            outs[f"seq_{mode}"] = seq
            outs[f"logits_{mode}"] = logits
        flow_emb = ref_ib.temporal_encoder.temporal_embeddings(of.clone(), of_mask)
        text_emb = ref_ib.temporal_encoder.embeddings(tids)
    save("tiny_tgb", of_q8=ofq, q8_scale=np.float32(1 / 127), of_mask=of_mask, text_ids=tids, text_mask=tmask,
         flow_embed=flow_emb, text_embed=text_emb, **outs)
    # --- ViT: 3 frames, all hidden states
    pq, pix = q8((3, 3, 56, 56), g, 1 / 48)
    with torch.no_grad():
        vo = ref_ib.model.vision_model(pixel_values=pix, output_hidden_states=True, return_dict=True)
    save("tiny_vit", pixel_q8=pq, q8_scale=np.float32(1 / 48), last_hidden_state=vo.last_hidden_state,
         hidden_0=vo.hidden_states[0], hidden_1=vo.hidden_states[1], hidden_2=vo.hidden_states[2])
    # --- Q-Former InstructBLIP: 3 frames, padded text;  BLIP-2: queries only
    img = vo.last_hidden_state
    qids = torch.randint(3, cfg_ib.qformer.vocab, (3, 7), generator=g)
    qmask = torch.ones_like(qids)
    qmask[1, 4:] = 0
    qmask[2, 6:] = 0
    with torch.no_grad():
        qt = ref_ib.model.query_tokens.expand(3, -1, -1)
        am = torch.cat([torch.ones(3, qt.shape[1], dtype=torch.long), qmask], dim=1)
        qo = ref_ib.model.qformer(input_ids=qids, attention_mask=am, query_embeds=qt, encoder_hidden_states=img,
                                  encoder_attention_mask=torch.ones(img.shape[:2], dtype=torch.long), return_dict=True)
        with torch.no_grad():
            vo2 = ref_b2.model.vision_model(pixel_values=pix, return_dict=True).last_hidden_state
        qt2 = ref_b2.model.query_tokens.expand(3, -1, -1)
        qo2 = ref_b2.model.qformer(query_embeds=qt2, encoder_hidden_states=vo2,
                                   encoder_attention_mask=torch.ones(vo2.shape[:2], dtype=torch.long))[0]
        # pooling variants on the InstructBLIP queries (mean: eval/utils/model.py:186-191; concat: LSTP_module.py:477-478)
        q32 = qo.last_hidden_state[:, : qt.shape[1]]
        mean = ref_ib.model.language_projection(q32.mean(0, keepdim=True))
        concat = ref_ib.model.language_projection(q32).reshape(1, -1, mean.shape[-1])
    save("tiny_qformer", image_embeds=img, qformer_ids=qids, qformer_mask=qmask, seq_instructblip=qo.last_hidden_state,
         image_embeds_blip2=vo2, seq_blip2=qo2, prefix_mean=mean, prefix_concat=concat)
    # --- RAFT alone: 3 frames 128x128 -> 2 flows (64x64 would make the coarsest corr level 1x1: division by W-1 = 0)
    rq, rf = q8((3, 3, 128, 128), g, 1.0)
    with torch.no_grad():
        fl = ref_ib.of_extractor(rf[:-1], rf[1:])
        fl5 = ref_ib.of_extractor(rf[:-1], rf[1:], iters=5)
    save("tiny_raft", frames_q8=rq, q8_scale=np.float32(1.0), flow_iters20=fl, flow_iters5=fl5)


def index_map_code(cls):
    """Compile the reference's own span->frame index-map lines (eval/utils/model.py:124-150 or
    :337-366), sliced out of the live function at run time; nothing of them is stored."""
    lines = textwrap.dedent(inspect.getsource(cls.generate)).split("\n")
    lo = next(i for i, l in enumerate(lines) if l.strip().startswith("video_lengths = [flow_frames.size(1)]"))
    hi = next(i for i, l in enumerate(lines) if l.strip().startswith("sampled_pixel_values[j] = torch.index_select"))
    body = [l for l in lines[lo + 1:hi] if "sampled_pixel_values = torch.zeros" not in l and "pixel_shape" not in l]
    body = [l.replace("device=pixel_values.device", "device='cpu'") for l in body]
    return compile(textwrap.dedent("\n".join(body)), f"<reference index map {cls.__name__}>", "exec")


def integer_tables():
    """Known-answer tables for the integer stages, produced by executing the reference's own lines."""
    import eval.utils.model as em
    from src.data.components.util import sample_frames
    rows, fb = [], []
    rng = np.random.default_rng(5)
    for vflag, cls in ((0, em.LSTP), (1, em.LSTP_blip2)):
        code = index_map_code(cls)
        for V in (4, 8, 32, 34, 64, 96, 100, 255, 256):
            for N in (8, 32):
                for nframe in (2, 4, 8):
                    cases = [(0, 0, 0, 0), (0, V - 1, 0, V - 1), (V - 1, 0, 3 % V, 2 % V), (1, 1, 1, 1),
                             (V, 1, 0, 2 % V), (1 % V, V + 3, 1 % V, 2 % V), (0, 1 % V, 0, 0)]
                    cases += [tuple(int(x) for x in rng.integers(0, V, 4)) for _ in range(40)]
                    for s0, e0, s1, e1 in cases:
                        env = dict(video_lengths=[V], num_frames=N, nframe=nframe, np=np, torch=torch,
                                   cand_start_index=[torch.tensor([s0]), torch.tensor([s1])],
                                   cand_end_index=[torch.tensor([e0]), torch.tensor([e1])])
                        exec(code, env)
                        rows.append([vflag, V, N, nframe, s0, e0, s1, e1] + env["cand_index"].tolist() + [-1] * (8 - nframe))
        # the except-fallback: python-int operands from the start (eval/utils/model.py:115-116)
        for V in (4, 32, 96, 100, 256):
            env = dict(video_lengths=[V], num_frames=32, nframe=8, np=np, torch=torch,
                       cand_start_index=[[0]], cand_end_index=[[V - 1]])
            exec(code, env)
            fb.append([vflag, V, 32, 8] + env["cand_index"].tolist())
    sf = []
    for vlen in (1, 5, 31, 32, 33, 64, 96, 100, 250, 256):
        for n in (4, 8, 32):
            for fix in (-1, 1.0):
                ids = list(sample_frames(n, vlen, "uniform", fix))
                sf.append([vlen, n, int(fix)] + [int(x) for x in ids] + [-1] * (32 - len(ids)))
    save("integer_tables", span_map=np.array(rows, dtype=np.int64), span_map_fallback=np.array(fb, dtype=np.int64),
         sample_frames=np.array(sf, dtype=np.int64))


def fullsize_probes():
    """Full-size (ViT-g / Q-Former / BERT-base TGB) reference outputs reduced to probe elements."""
    from transformers import BertConfig, InstructBlipQFormerConfig, InstructBlipVisionConfig
    from src.models.components.xinstructblip import InstructBlipQFormerModel, InstructBlipVisionModel
    from src.models.components.xropebert import RopeBertModel
    from videotgb_amd.synth import QFormerCfg, TgbCfg, VitCfg, qformer_shapes, synth_state_dict, tgb_shapes, vit_shapes
    g = torch.Generator().manual_seed(21)
    probes = {}
    # ViT-g, 2 frames
    vm = InstructBlipVisionModel(InstructBlipVisionConfig()).eval()
    vsd = synth_state_dict(vit_shapes(VitCfg(), ""), 0)
    vm.load_state_dict(vsd, strict=True)
    pq, pix = q8((2, 3, 224, 224), g, 1 / 48)
    with torch.no_grad():
        img = vm(pixel_values=pix, return_dict=True).last_hidden_state
    pidx = torch.randint(0, img.numel(), (4096,), generator=g)
    probes.update(vit_pixel_q8=pq, vit_probe_idx=pidx, vit_probe_val=img.flatten()[pidx],
                  vit_absmean=img.abs().mean(), vit_shape=np.array(img.shape))
    del vm, vsd
    # Q-Former (InstructBLIP), those 2 frames, 9 text tokens
    qm = InstructBlipQFormerModel(InstructBlipQFormerConfig()).eval()
    qc = QFormerCfg()
    qsd = synth_state_dict(qformer_shapes(qc, ""), 0)
    qm.load_state_dict(qsd, strict=True)
    from videotgb_amd.synth import synth_tensor
    qtok = synth_tensor("model.query_tokens", (1, 32, 768))
    qids = torch.randint(1000, 30000, (2, 9), generator=g)
    am = torch.ones(2, 32 + 9, dtype=torch.long)
    with torch.no_grad():
        qo = qm(input_ids=qids, attention_mask=am, query_embeds=qtok.expand(2, -1, -1), encoder_hidden_states=img,
                encoder_attention_mask=torch.ones(2, 257, dtype=torch.long), return_dict=True).last_hidden_state[:, :32]
    qidx = torch.randint(0, qo.numel(), (2048,), generator=g)
    probes.update(qf_ids=qids, qf_probe_idx=qidx, qf_probe_val=qo.flatten()[qidx], qf_absmean=qo.abs().mean())
    del qm, qsd
    # TGB BERT-base, T=24
    tm = RopeBertModel(BertConfig(fusion_layer=6, encoder_width=768)).eval()
    tsd = synth_state_dict(tgb_shapes(TgbCfg(), ""), 0)
    tm.load_state_dict(tsd, strict=True)
    oq, of = q8((1, 24, 2, 224, 224), g, 1 / 127)
    tids = torch.randint(1000, 30000, (1, 14), generator=g)
    with torch.no_grad():
        for mode in ("multi_modal", "fusion"):
            seq, logits = tm(encoder_embeds=of.clone(), attention_mask=torch.ones(1, 26, dtype=torch.long),
                             encoder_hidden_states=tids, encoder_attention_mask=torch.ones_like(tids), mode=mode)
            probes[f"tgb_logits_{mode}"] = logits
            probes[f"tgb_seq_absmean_{mode}"] = seq.abs().mean()
    probes.update(tgb_of_q8=oq, tgb_text_ids=tids)
    save("full_probes", **probes)


def preprocess_fixture():
    """get_frames' transform chain (eval/utils/builder_utils.py:118-128) executed with the reference's own functions
    (src/gadgets/functional_video.py; the Compose wrappers of transforms.py need torchvision, which is absent, and add
    nothing but the calls below) on small synthetic decoded clips, plus its 32-frame pick for several lengths."""
    import src.gadgets.functional_video as FV
    from src.data.components.util import sample_frames
    g = torch.Generator().manual_seed(77)
    out = {}
    for name, (T, H0, W0) in (("a", (5, 37, 53)), ("b", (3, 240, 320)), ("c", (2, 224, 224))):
        raw = torch.randint(0, 256, (T, H0, W0, 3), generator=g, dtype=torch.uint8)
        clip = raw.permute(3, 0, 1, 2).float()                       # read_videos_av, builder_utils.py:86
        clip = FV.resize(clip, (224, 224), "bilinear")               # ResizeVideo(224)
        clip = clip.to(torch.uint8)                                  # ToUint8
        clip = clip.permute(1, 2, 3, 0)                              # ToTHWC
        clip = FV.to_tensor(clip)                                    # ToTensorVideo
        clip = FV.normalize(clip, (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711))
        flow_frames = clip.permute(1, 0, 2, 3)                       # builder_utils.py:127
        # the full tensor is large; keep the raw input, a strided probe of the output and its checksums
        out[f"raw_{name}"] = raw.numpy()
        out[f"probe_{name}"] = flow_frames[:, :, ::7, ::5].contiguous().numpy()
        out[f"sum_{name}"] = flow_frames.double().sum(dim=(2, 3)).numpy()
    picks = []
    for vlen in (1, 3, 5, 17, 31, 32, 33, 40, 96, 100, 256):
        indices = list(range(vlen))                                  # builder_utils.py:131-139
        n = vlen
        while n < 32:
            indices = [f for ind in indices for f in (ind, ind)]
            n = len(indices)
        ids = sample_frames(32, n, "uniform", 1.)
        picks.append([vlen] + [indices[i] for i in ids])
    out["picks"] = np.array(picks, dtype=np.int64)
    save("preprocess", **out)


def train_fixture():
    """a14: concat_text_input_output (called as the reference's own unbound method), the label-masking lines and the
    shifted cross-entropy lines of LSTPModule.forward (src/models/LSTP_Vicuna_IVT_module.py:284-299, :325-326)
    executed verbatim on synthetic tokens / logits.  The LightningModule's imports that are absent here (lightning,
    torchmetrics) are stubbed with empty classes: only plain methods of the class are used."""
    train_stubs()
    import src.models.LSTP_Vicuna_IVT_module as tm
    return _train_fixture_body(tm)


def train_stubs():
    for name, attrs in (("lightning", ["LightningModule"]), ("torchmetrics", ["MaxMetric", "MeanMetric"]),
                        ("torchmetrics.classification", []), ("torchmetrics.classification.accuracy", ["Accuracy"]),
                        ("torchmetrics.text", []), ("torchmetrics.text.bleu", ["BLEUScore"]), ("torchmetrics.text.bert", ["BERTScore"]),
                        ("torchmetrics.text.rouge", ["ROUGEScore"]), ("torchmetrics.text.perplexity", ["Perplexity"])):
        if name not in sys.modules:
            m = types.ModuleType(name)
            for a_ in attrs:
                setattr(m, a_, type(a_, (), {}))
            m.__spec__ = importlib.machinery.ModuleSpec(name, None)
            sys.modules[name] = m


def _train_fixture_body(tm):
    cls = tm.LSTPModule
    lines = textwrap.dedent(inspect.getsource(cls.forward)).split("\n")
    def block(first, last):
        lo = next(i for i, l in enumerate(lines) if l.strip().startswith(first))
        hi = next(i for i, l in enumerate(lines) if i >= lo and l.strip().startswith(last))
        return textwrap.dedent("\n".join(lines[lo:hi + 1]))
    lab_src = block('labels = llm_tokens["input_ids"].masked_fill(', "labels = torch.cat([empty_labels, labels], dim=1)")
    lab_src = lab_src.replace("self.processor.tokenizer.pad_token_id", "pad_id")
    ce_src = block("shift_logits = logits[..., :-1, :].contiguous()", "shift_labels = labels[..., 1:].contiguous()") + "\n" + \
        block('loss_fct = CrossEntropyLoss(reduction="mean")', "loss = loss_fct(")
    ce_src = ce_src.replace("self.model.config.text_config.vocab_size", "V")
    lab_code, ce_code = compile(lab_src, "<reference labels>", "exec"), compile(ce_src, "<reference shifted CE>", "exec")
    g = torch.Generator().manual_seed(99)
    out = {}
    pad_id = 0
    for case, (B, Li, Lo, prefix, V) in enumerate(((4, 12, 7, 32, 97), (3, 20, 20, 8, 211), (1, 5, 3, 0, 50))):
        qlen = torch.randint(1, Li + 1, (B,), generator=g)
        alen = torch.randint(2, Lo + 1, (B,), generator=g)
        qlen[0], alen[0] = Li, Lo                                         # one row without padding
        q_ids = torch.randint(3, V, (B, Li), generator=g)
        a_ids = torch.randint(3, V, (B, Lo), generator=g)
        a_ids[:, 0] = 1                                                    # BOS, dropped by output_ids[i][1:]
        q_att = (torch.arange(Li)[None] < qlen[:, None]).long()
        a_att = (torch.arange(Lo)[None] < alen[:, None]).long()
        q_ids = q_ids * q_att + pad_id * (1 - q_att)
        a_ids = a_ids * a_att + pad_id * (1 - a_att)
        llm_tokens, input_part_targets_len = cls.concat_text_input_output(None, q_ids, q_att, a_ids, a_att)
        env = dict(torch=torch, llm_tokens=llm_tokens, input_part_targets_len=input_part_targets_len, pad_id=pad_id,
                   language_model_attention_mask=torch.ones(B, prefix, dtype=torch.long))
        exec(lab_code, env)
        labels = env["labels"]
        S = labels.shape[1]
        logits = torch.randn(B, S, V, generator=g) * 3.0
        logits.requires_grad_(True)
        env2 = dict(torch=torch, CrossEntropyLoss=torch.nn.CrossEntropyLoss, logits=logits, labels=labels, V=V)
        exec(ce_code, env2)
        loss = env2["loss"]
        loss.backward()
        out.update({f"q_ids_{case}": q_ids, f"q_att_{case}": q_att, f"a_ids_{case}": a_ids, f"a_att_{case}": a_att,
                    f"llm_ids_{case}": llm_tokens["input_ids"], f"llm_att_{case}": llm_tokens["attention_mask"],
                    f"input_len_{case}": torch.stack([x.reshape(()) for x in input_part_targets_len]), f"labels_{case}": labels,
                    f"logits_{case}": logits.detach(), f"loss_{case}": loss.detach().reshape(1), f"dlogits_{case}": logits.grad,
                    f"meta_{case}": torch.tensor([B, Li, Lo, prefix, V, pad_id])})
    save("train_loss", **out)


def refine_fixture():
    """f4: rouge_n as imported from the reference (src/gadgets/my_metrics.py; torchmetrics stubbed), and the
    monotone-stack / rescale / MRC-loss lines of LSTPSFModule.forward (src/models/LSTP_SF_module.py:239-298) executed
    verbatim on synthetic strings, scores and logits."""
    train_stubs()
    tm_mod = sys.modules["torchmetrics"]
    if not hasattr(tm_mod, "Metric"):
        tm_mod.Metric = type("Metric", (), {})
    from src.gadgets.my_metrics import rouge_n
    import src.models.LSTP_SF_module as sf
    lines = textwrap.dedent(inspect.getsource(sf.LSTPSFModule.forward)).split("\n")
    def block(first, last):
        lo = next(i for i, l in enumerate(lines) if l.strip().startswith(first))
        hi = next(i for i, l in enumerate(lines) if i >= lo and l.strip().startswith(last))
        return textwrap.dedent("\n".join(lines[lo:hi + 1]))
    span_code = compile(block("scores = torch.tensor(scores, dtype=torch.float)", "end_targets = [int(end_targets[ii]"), "<reference span>", "exec")
    loss_code = compile(block("start_logits, end_logits = of_logits.split(1, dim=-1)", "mrc_loss = (start_loss + end_loss) / 2"),
                        "<reference mrc loss>", "exec")
    rng = np.random.default_rng(11)
    vocab = ["a", "man", "is", "riding", "horse", "the", "dog", "runs", ",", ".", "yes", "no", "two", "people", "cooking"]
    out, texts = {}, []
    B, N = 4, 32
    gold = [" ".join(rng.choice(vocab, size=int(rng.integers(1, 7)))) for _ in range(B)]
    pred = [" ".join(rng.choice(vocab, size=int(rng.integers(0, 8)))) for _ in range(B * N)]
    target = [gold[int(idx // N)] for idx in range(len(pred))]
    scores = rouge_n(target, pred)
    flow_lengths = [96, 34, 256, 5]
    env = dict(torch=torch, scores=list(scores), batch_size=B, num_frames=N, batch={"of_lengths": flow_lengths})
    exec(span_code, env)
    out["rouge"] = np.array(scores, dtype=np.float64)
    out["start_targets"] = np.array(env["start_targets"], dtype=np.int64)
    out["end_targets"] = np.array(env["end_targets"], dtype=np.int64)
    out["flow_lengths"] = np.array(flow_lengths, dtype=np.int64)
    # hand-made score rows: plateaus, ties, all-zero, single peak
    rows = [[0.0] * 32, [0.5] * 32, [0.0] * 10 + [0.3] * 5 + [0.0] * 17, [0.1] * 8 + [0.9] + [0.1] * 23,
            list(np.round(rng.random(32), 2)), [0.2, 0.2, 0.8, 0.8, 0.8, 0.1] + [0.0] * 26, list(np.round(rng.random(32), 1))]
    env = dict(torch=torch, scores=[x for r in rows for x in r], batch_size=len(rows), num_frames=32,
               batch={"of_lengths": [96, 96, 40, 32, 100, 7, 256]})
    exec(span_code, env)
    out["rows"] = np.array(rows, dtype=np.float32)
    out["rows_lengths"] = np.array([96, 96, 40, 32, 100, 7, 256], dtype=np.int64)
    out["rows_start"] = np.array(env["start_targets"], dtype=np.int64)
    out["rows_end"] = np.array(env["end_targets"], dtype=np.int64)
    g = torch.Generator().manual_seed(4)
    of_logits = torch.randn(7, 96, 2, generator=g)
    st = torch.tensor([0, 5, 95, 96, 200, 17, 3]); en = torch.tensor([95, 9, 95, 120, 300, 40, 3])
    env = dict(torch=torch, CrossEntropyLoss=torch.nn.CrossEntropyLoss, of_logits=of_logits, start_targets=st.clone(), end_targets=en.clone())
    exec(loss_code, env)
    out.update(of_logits=of_logits, mrc_start=st, mrc_end=en, mrc_loss=env["mrc_loss"].reshape(1))
    save("refine", **out)
    with open(os.path.join(OUT, "refine_text.txt"), "w") as fh:      # the synthetic strings (data, one per line)
        fh.write("\n".join(gold + pred) + "\n")


def bf16_reference_fixtures():
    """What does the REFERENCE produce in its own bf16 mode?  Lightning `precision: bf16` (configs/trainer/default.yaml) is
    torch.autocast(bfloat16) around the modules: linear / matmul / conv in bf16, LayerNorm and softmax in fp32, fp32
    parameters.  Recorded here with torch.autocast("cpu", bfloat16) so that the bf16 tolerances of the GPU tests can be
    stated against the reference's own bf16 numbers, not only against its fp32 numbers:
      * tiny e2e (both flavours): TGB logits, ViT output, Q-Former queries, prefix -- RAFT kept in fp32 as the reference
        does (xraft.py:113-119 wraps it in autocast(enabled=False) / .float());
      * full size: ViT-g probes, Q-Former probes (on a seeded int8 image-token tensor, also recorded in fp32 so the stage
        is tested in isolation), BERT-base TGB logits."""
    from transformers import BertConfig, InstructBlipQFormerConfig, InstructBlipVisionConfig
    from src.models.components.xinstructblip import InstructBlipQFormerModel, InstructBlipVisionModel
    from src.models.components.xropebert import RopeBertModel
    from videotgb_amd.synth import (QFormerCfg, TgbCfg, VitCfg, path_state_dict, qformer_shapes, synth_state_dict, synth_tensor,
                                    tgb_shapes, tiny_cfg, vit_shapes)
    ac = lambda: torch.autocast("cpu", dtype=torch.bfloat16)
    # ---- tiny e2e
    for arch in ("instructblip", "blip2"):
        cfg = tiny_cfg(arch)
        cfg.vit.image = 56
        ref, tc, _ = build_reference(arch, cfg, path_state_dict(cfg, seed=0))
        z = np.load(os.path.join(OUT, f"tiny_{arch}_e2e.npz"))
        sc = float(z["q8_scale"])
        frames, flow_frames = torch.from_numpy(z["frames_q8"]).float() * sc, torch.from_numpy(z["flow_frames_q8"]).float() * sc
        noise = torch.from_numpy(z["noise"])
        exps = [torch.exp(-noise[i]).unsqueeze(-1) for i in range(noise.shape[0])]      # g = -log(E)  ->  E = exp(-g)
        raft_fwd = ref.of_extractor.forward

        def raft_fp32(*a, **k):
            with torch.autocast("cpu", enabled=False):
                return raft_fwd(*[x.float() if torch.is_tensor(x) else x for x in a], **k)
        ref.of_extractor.forward = raft_fp32
        cap = {}
        hooks = [
            ref.temporal_encoder.register_forward_hook(lambda m, i, o: cap.update(tgb_logits=o[1])),
            ref.model.vision_model.register_forward_hook(lambda m, i, o: cap.__setitem__("image_embeds", o.last_hidden_state)),
            ref.model.qformer.register_forward_hook(lambda m, i, o: cap.__setitem__("qformer_seq", o[0])),
            ref.model.language_projection.register_forward_hook(lambda m, i, o: cap.__setitem__("prefix", o)),
        ]
        te = BE(input_ids=torch.from_numpy(z["prompt_ids"]), attention_mask=torch.from_numpy(z["prompt_mask"]))
        if arch == "instructblip":
            te["qformer_input_ids"] = torch.from_numpy(z["qformer_ids"])
            te["qformer_attention_mask"] = torch.from_numpy(z["qformer_mask"])
        se = BE(input_ids=torch.from_numpy(z["sampler_ids"]), attention_mask=torch.from_numpy(z["sampler_mask"]))
        with NoiseQueue(exps) as nq, ac():
            ids, cand = ref.generate(frames, flow_frames, int(z["nframe"]), te, se, do_sample=False, temperature=None, max_new_tokens=6,
                                     use_cache=False)
        assert nq.used == 2 and "tgb_logits" in cap, "reference took its bare-except fallback"
        for h in hooks:
            h.remove()
        save(f"tiny_{arch}_e2e_bf16ref", tgb_logits=cap["tgb_logits"].float(), cand_index=cand, image_embeds=cap["image_embeds"].float(),
             query_out=cap["qformer_seq"][:, :cfg.qformer.n_query].float(), prefix=cap["prefix"].float(), greedy_ids=ids)
    # ---- full size
    g = torch.Generator().manual_seed(21)
    old = np.load(os.path.join(OUT, "full_probes.npz"))
    out = {}
    vm = InstructBlipVisionModel(InstructBlipVisionConfig()).eval()
    vm.load_state_dict(synth_state_dict(vit_shapes(VitCfg(), ""), 0), strict=True)
    pix = torch.from_numpy(old["vit_pixel_q8"]).float() / 48
    with torch.no_grad(), ac():
        img16 = vm(pixel_values=pix, return_dict=True).last_hidden_state.float()
    out["vit_probe_val_bf16ref"] = img16.flatten()[torch.from_numpy(old["vit_probe_idx"])]
    del vm
    qm = InstructBlipQFormerModel(InstructBlipQFormerConfig()).eval()
    qm.load_state_dict(synth_state_dict(qformer_shapes(QFormerCfg(), ""), 0), strict=True)
    qtok = synth_tensor("model.query_tokens", (1, 32, 768))
    g2 = torch.Generator().manual_seed(33)
    iq, img = q8((2, 257, 1408), g2, 1 / 64)                       # image tokens ~ U(-2, 2): the scale of a post-LayerNorm ViT output
    qids = torch.randint(1000, 30000, (2, 9), generator=g2)
    qmask = torch.ones(2, 9, dtype=torch.long)
    qmask[1, 6:] = 0
    am = torch.cat([torch.ones(2, 32, dtype=torch.long), qmask], 1)
    kw = dict(input_ids=qids, attention_mask=am, query_embeds=qtok.expand(2, -1, -1), encoder_hidden_states=img,
              encoder_attention_mask=torch.ones(2, 257, dtype=torch.long), return_dict=True)
    with torch.no_grad():
        q32 = qm(**kw).last_hidden_state[:, :32]
        with ac():
            q16 = qm(**kw).last_hidden_state[:, :32].float()
    qidx = torch.randint(0, q32.numel(), (4096,), generator=g2)
    out.update(qf2_image_q8=iq, qf2_ids=qids, qf2_mask=qmask, qf2_probe_idx=qidx, qf2_probe_val=q32.flatten()[qidx],
               qf2_probe_val_bf16ref=q16.flatten()[qidx], qf2_q8_scale=np.float32(1 / 64))
    del qm
    tm = RopeBertModel(BertConfig(fusion_layer=6, encoder_width=768)).eval()
    tm.load_state_dict(synth_state_dict(tgb_shapes(TgbCfg(), ""), 0), strict=True)
    of = torch.from_numpy(old["tgb_of_q8"]).float() / 127
    tids = torch.from_numpy(old["tgb_text_ids"])
    with torch.no_grad(), ac():
        for mode in ("multi_modal", "fusion"):
            seq, logits = tm(encoder_embeds=of.clone(), attention_mask=torch.ones(1, 26, dtype=torch.long), encoder_hidden_states=tids,
                             encoder_attention_mask=torch.ones_like(tids), mode=mode)
            out[f"tgb_logits_{mode}_bf16ref"] = logits.float()
    save("full_probes_bf16ref", **out)


def up4(q):
    """int8 [.., h, w] -> fp32 [.., 4h, 4w] by exact pixel replication (keeps the committed inputs small)."""
    return q.float().repeat_interleave(4, -2).repeat_interleave(4, -1)


def module_fixtures():
    """The reference's LightningModules run THEIR OWN ``eval_forward`` (called as unbound functions on the reference's
    eval-side model object, which has the attributes they touch: model / temporal_encoder / of_extractor / generate_configs;
    lightning and torchmetrics are stubbed with empty classes, nothing of them is used by these methods):
      * src.models.LSTP_module.LSTPModule.eval_forward        InstructBLIP, RAFT on the candidate frames (replicate-padded
                                                              126 -> 128 is not possible with the TGB's fixed 224 x 224 flow
                                                              input, so 224), multi_modal, V = N + 2, map A, concat
      * src.models.LSTP_SF_module.LSTPSFModule.eval_forward   InstructBLIP, batch["of"] with ragged of_lengths, fusion, map B
      * src.models.LSTP_blip2_module.LSTPModule.eval_forward  BLIP-2 + a seq2seq (T5) language model, no sampler  (config C1)
      * src.models.LSTP_SF_blip2_module.LSTPSFModule.eval_forward  BLIP-2 + T5, batch["of"], fusion, map B          (config C2)
    Inputs are int8 at 56 x 56 and replicated x4 to 224 x 224; outputs: generated ids, selected frames, of_logits, prefix."""
    train_stubs()
    tm_mod = sys.modules["torchmetrics"]
    if not hasattr(tm_mod, "Metric"):
        tm_mod.Metric = type("Metric", (), {})
    import src.models.LSTP_module as m_ib
    import src.models.LSTP_SF_module as m_sf
    import src.models.LSTP_blip2_module as m_b2
    import src.models.LSTP_SF_blip2_module as m_sfb2
    from videotgb_amd.synth import path_state_dict, tiny_cfg
    g = torch.Generator().manual_seed(41)
    B, N, nframe = 2, 8, 4
    fq = torch.randint(-127, 128, (B * N, 3, 56, 56), generator=g, dtype=torch.int32).to(torch.int8)
    frames = up4(fq) / 48
    L = 10
    ofq = torch.randint(-127, 128, (B, L, 2, 56, 56), generator=g, dtype=torch.int32).to(torch.int8)
    of = up4(ofq) / 127
    of_lengths = [L, 7]
    of_mask = torch.ones(B, L + 2, dtype=torch.long)
    of_mask[1, 7 + 2:] = 0
    common = dict(nframe=nframe, of_lengths=of_lengths, answer=torch.zeros(B, 1, dtype=torch.long), text_answer=[""] * B)
    out = dict(frames_q8=fq, of_q8=ofq, of_mask=of_mask, of_lengths=np.array(of_lengths), nframe=nframe)
    for tag, arch, llm, mod, cls_name, uses_of in (("ib", "instructblip", "llama", m_ib, "LSTPModule", False),
                                                  ("sf", "instructblip", "llama", m_sf, "LSTPSFModule", True),
                                                  ("b2", "blip2", "t5", m_b2, "LSTPModule", False),
                                                  ("sfb2", "blip2", "t5", m_sfb2, "LSTPSFModule", True)):
        cfg = tiny_cfg(arch)                                   # vit.image stays 224: the TGB's flow input is 224 x 224
        ref, tc, _ = build_reference(arch, cfg, path_state_dict(cfg, seed=0), llm=llm)
        ref.generate_configs = dict(do_sample=False, max_new_tokens=6)
        samp = torch.randint(3, cfg.tgb.vocab, (B, 7), generator=g)
        smask = torch.ones_like(samp)
        smask[1, 5:] = 0
        qf = torch.randint(3, cfg.qformer.vocab, (B, 6), generator=g)
        qfm = torch.ones_like(qf)
        qfm[1, 4:] = 0
        quest = torch.randint(3, tc.vocab_size, (B, 5), generator=g)
        qmask = torch.ones_like(quest)
        batch = dict(common, frames=frames, sampler_question=samp, sampler_question_attention_mask=smask, qformer_text=qf,
                     qformer_text_attention_mask=qfm, question=quest, question_attention_mask=qmask)
        if uses_of:
            batch.update(of=of, of_mask=of_mask)
        T = L if uses_of else N
        exps = [torch.empty(2 * B, T, 1).exponential_(generator=g) for _ in range(2)]
        cap = {}
        hooks = [ref.model.vision_model.register_forward_pre_hook(lambda m, a, k: cap.__setitem__("sampled", k["pixel_values"]), with_kwargs=True),
                 ref.model.language_projection.register_forward_hook(lambda m, i, o: cap.__setitem__("proj", o)),
                 ref.temporal_encoder.register_forward_hook(lambda m, i, o: cap.__setitem__("of_logits", o[1])),
                 ref.of_extractor.register_forward_hook(lambda m, i, o: cap.setdefault("raft", []).append(o)),
                 ref.model.language_model.lm_head.register_forward_hook(lambda m, i, o: cap.setdefault("lm_logits", []).append(o))]
        with NoiseQueue(exps) as nq, torch.no_grad():
            ids = getattr(mod, cls_name).eval_forward(ref, batch)
        for h in hooks:
            h.remove()
        sampler_ran = "of_logits" in cap
        assert nq.used == (2 if sampler_ran else 0)
        # frame indices: match the selected frames back to the candidates (bit-exact copies)
        sp = cap["sampled"].view(B, nframe, -1)
        pv = frames.view(B, N, -1)
        idx = torch.stack([torch.stack([(pv[b] == sp[b, i]).all(1).float().argmax() for i in range(nframe)]) for b in range(B)])
        assert all(torch.equal(pv[b, idx[b, i]], sp[b, i]) for b in range(B) for i in range(nframe))
        rec = {f"{tag}_ids": ids, f"{tag}_frame_idx": idx, f"{tag}_first_logits": cap["lm_logits"][0][:, -1].float(), f"{tag}_prefix": cap["proj"].reshape(B, -1, cap["proj"].shape[-1]),
               f"{tag}_sampler_ids": samp, f"{tag}_sampler_mask": smask, f"{tag}_qformer_ids": qf, f"{tag}_qformer_mask": qfm,
               f"{tag}_question": quest, f"{tag}_question_mask": qmask}
        if sampler_ran:
            rec[f"{tag}_of_logits"] = cap["of_logits"]
            rec[f"{tag}_noise"] = torch.stack([-e.log().squeeze(-1) for e in exps])
        if "raft" in cap:
            rec[f"{tag}_raft_flow"] = torch.stack(cap["raft"])                       # [B, N-1, 2, 224, 224] -> subsample for size
            rec[f"{tag}_raft_flow"] = rec[f"{tag}_raft_flow"][:, :, :, ::7, ::5].contiguous()
        out.update(rec)
    save("tiny_modules", **out)


def train_step_fixture():
    """C5's training forward on the reference side: ``LSTPModule.forward`` of src.models.LSTP_Vicuna_IV_module (the IVT module
    minus peft, which is absent here; same forward, :191-335) called as an unbound function on the tiny reference model in
    eval mode (dropout off), then ``loss.backward()``: the loss and the gradients of everything the reference trains on the
    prefix side (Q-Former, query_tokens, language_projection) for a ragged batch (widths 3 and 2, padded question / answer)."""
    train_stubs()
    tm_mod = sys.modules["torchmetrics"]
    if not hasattr(tm_mod, "Metric"):
        tm_mod.Metric = type("Metric", (), {})
    import src.models.LSTP_Vicuna_IV_module as iv
    from videotgb_amd.synth import path_state_dict, tiny_cfg
    cfg = tiny_cfg("instructblip")
    cfg.vit.image = 56
    ref, tc, _ = build_reference("instructblip", cfg, path_state_dict(cfg, seed=0))
    ref.processor = types.SimpleNamespace(tokenizer=types.SimpleNamespace(pad_token_id=0))
    ref.concat_text_input_output = types.MethodType(iv.LSTPModule.concat_text_input_output, ref)
    g = torch.Generator().manual_seed(61)
    widths = [3, 2]
    fq, frames = q8((sum(widths), 3, 56, 56), g, 1 / 48)
    qf = torch.randint(3, cfg.qformer.vocab, (2, 6), generator=g)
    qfm = torch.ones_like(qf)
    qfm[1, 4:] = 0
    quest = torch.randint(3, tc.vocab_size, (2, 5), generator=g)
    qm = torch.ones_like(quest)
    qm[1, 3:] = 0
    quest = quest * qm
    ans = torch.randint(3, tc.vocab_size, (2, 4), generator=g)
    ans[:, 0] = 1
    am = torch.ones_like(ans)
    am[0, 3:] = 0
    ans = ans * am
    batch = dict(frames=frames, widths=widths, nframe=3, qformer_text=qf, qformer_text_attention_mask=qfm, question=quest,
                 question_attention_mask=qm, answer=ans, answer_attention_mask=am)
    ref.zero_grad()
    loss, logits = iv.LSTPModule.forward(ref, batch)
    loss.backward()
    out = dict(frames_q8=fq, q8_scale=np.float32(1 / 48), widths=np.array(widths), qformer_ids=qf, qformer_mask=qfm, question=quest,
               question_mask=qm, answer=ans, answer_mask=am, loss=loss.detach().reshape(1), logits=logits.detach())
    for n, p in ref.model.named_parameters():
        if n.startswith("qformer.") or n in ("query_tokens", "language_projection.weight", "language_projection.bias"):
            assert p.grad is not None, n
            out["g:" + n] = p.grad.detach()
    save("tiny_train_step", **out)


def refine_answers_fixture():
    """f4 on the reference side: the per-frame answer loop of LSTPSFModule.forward (src/models/LSTP_SF_module.py:149-204,
    sliced out of the live function and executed on the tiny InstructBLIP reference model) -- ViT over all candidate
    frames, then nframe frames at a time through Q-Former + projection + language_model.generate(max_length=128) --
    on the frames / question of the tiny e2e fixture.  Records the generated ids of every frame."""
    train_stubs()
    tm_mod = sys.modules["torchmetrics"]
    if not hasattr(tm_mod, "Metric"):
        tm_mod.Metric = type("Metric", (), {})
    import src.models.LSTP_SF_module as sf
    from videotgb_amd.synth import path_state_dict, tiny_cfg
    cfg = tiny_cfg("instructblip")
    cfg.vit.image = 56
    ref, tc, _ = build_reference("instructblip", cfg, path_state_dict(cfg, seed=0))
    lines = textwrap.dedent(inspect.getsource(sf.LSTPSFModule.forward)).split("\n")
    lo = next(i for i, l in enumerate(lines) if l.strip().startswith("image_embeddings = self.model.vision_model("))
    hi = next(i for i, l in enumerate(lines) if l.strip().startswith("predict.extend(self.processor.batch_decode("))
    code = compile(textwrap.dedent("\n".join(lines[lo:hi + 1])), "<reference per-frame answers>", "exec")
    z = np.load(os.path.join(OUT, "tiny_instructblip_e2e.npz"))
    frames = torch.from_numpy(z["frames_q8"]).float() * float(z["q8_scale"])
    rows = []
    proc = types.SimpleNamespace(batch_decode=lambda ids, skip_special_tokens=True: (rows.extend(r.tolist() for r in ids), [""] * len(ids))[1])
    batch = {"frames": frames, "qformer_text": torch.from_numpy(z["qformer_ids"]), "qformer_text_attention_mask": torch.from_numpy(z["qformer_mask"]),
             "question": torch.from_numpy(z["prompt_ids"]), "question_attention_mask": torch.from_numpy(z["prompt_mask"])}
    env = dict(torch=torch, self=types.SimpleNamespace(model=ref.model, processor=proc), batch=batch, batch_size=1,
               num_frames=frames.shape[0], nframe=int(z["nframe"]))
    pref = []
    hook = ref.model.language_projection.register_forward_hook(lambda m, i, o: pref.append(o))
    with torch.no_grad():
        exec(code, env)
    hook.remove()
    assert len(rows) == frames.shape[0]
    n = max(len(r) for r in rows)
    ids = np.full((len(rows), n), -1, dtype=np.int64)
    for i, r in enumerate(rows):
        ids[i, :len(r)] = r
    save("tiny_refine_answers", ids=ids, prefix_per_frame=torch.cat(pref, 0), eos_token_id=np.int64(tc.eos_token_id), pad_token_id=np.int64(tc.pad_token_id), max_length=np.int64(128))


def raft_sensitive_fixture():
    """[raft2] The reference RAFT (src/models/components/xraft.py, RAFT-large) with the INPUT-SENSITIVE weight set
    (videotgb_amd.synth.raft_sensitive_state_dict) on three frame triples of 128 x 128:
      a: float-valued frames ~ N(0, 1), stored rounded to fp16 (the bench's / SURVEY 8d's flow frames);
      b: a moving texture, CLIP-normalised -- what eval/inference.py:68 -> eval/utils/model.py:79 hands to RAFT;
      c: the same texture as integer 0..255 frames (RAFT's own input convention).
    Recorded: the 20-iteration flows, and fnet's feature maps (every 4th channel) for a and b."""
    from videotgb_amd import synth
    from src.models.components.xraft import RAFT
    ref = RAFT().eval()
    sd = {k[len("of_extractor."):]: v for k, v in synth.raft_sensitive_state_dict(0).items()}
    ref.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(31)
    fa = torch.randn(3, 3, 128, 128, generator=g).half()
    u8 = synth.moving_texture_u8(3, 128, 5)
    fb = synth.clip_normalise(u8)
    fc = u8.float()
    cap = {}
    h = ref.fnet.register_forward_hook(lambda m, i, o: cap.__setitem__("fmap", torch.cat(list(o), 0) if isinstance(o, (list, tuple)) else o))
    out = {}
    with torch.no_grad():
        for tag, f in (("a", fa.float()), ("b", fb), ("c", fc)):
            out["flow_" + tag] = ref(f[:-1], f[1:], iters=20, test_mode=True)
            if tag != "c":
                out["fmap_" + tag] = cap["fmap"][:, ::4].contiguous()       # [4 = (img0, img1 | img1, img2), 64, 16, 16]
    h.remove()
    save("tiny_raft_sensitive", frames_a_f16=fa.numpy(), frames_u8=u8.numpy(), **out)


def raft224_fixture():
    """[raft224] The reference RAFT (RAFT-large, input-sensitive weights) at the BENCH's frame size, 224 x 224 (28 x 28 coarse pixels: the
    shape the fused GRU / conv64 / stem kernels are tuned for), 20 iterations, two frame pairs of a moving texture in the eval path's
    input convention (CLIP-normalised floats).  Recorded: the flow at every 4th fine pixel (probe elements) and the SHA-256 of the full
    fp32 flow (round-3 VERDICT: the full-size RAFT check was HIP vs oracle only)."""
    import hashlib
    from videotgb_amd import synth
    from src.models.components.xraft import RAFT
    ref = RAFT().eval()
    sd = {k[len("of_extractor."):]: v for k, v in synth.raft_sensitive_state_dict(0).items()}
    ref.load_state_dict(sd, strict=True)
    u8 = synth.moving_texture_u8(3, 224, 7)
    f = synth.clip_normalise(u8)
    with torch.no_grad():
        flow = ref(f[:-1], f[1:], iters=20, test_mode=True)                   # [2, 2, 224, 224]
    save("raft224_sensitive", frames_u8=u8.numpy(), flow_probe=flow[:, :, ::4, ::4].contiguous(),
         flow_absmax=np.float32(flow.abs().max().item()), flow_sha256=np.frombuffer(hashlib.sha256(flow.contiguous().numpy().tobytes()).digest(), dtype=np.uint8))


def sf_mrc_fixture():
    """[sfmrc] "2) optimize temporal encoder" of LSTPSFModule.forward (src/models/LSTP_SF_module.py:275-296), sliced out of the live
    function and executed on the tiny reference model (eval mode: dropout off): the TGB forward in fusion mode on batch["of"] with a
    ragged of_mask, the MRC loss against given start / end targets (one of them beyond the sequence: clamped to the ignore index),
    then ``mrc_loss.backward()``: the loss, the logits and the gradient of EVERY temporal_encoder parameter that receives one."""
    train_stubs()
    tm_mod = sys.modules["torchmetrics"]
    if not hasattr(tm_mod, "Metric"):
        tm_mod.Metric = type("Metric", (), {})
    import src.models.LSTP_SF_module as sf
    from torch.nn import CrossEntropyLoss
    from videotgb_amd.synth import path_state_dict, tiny_cfg
    cfg = tiny_cfg("instructblip")
    ref, tc, _ = build_reference("instructblip", cfg, path_state_dict(cfg, seed=0))
    lines = textwrap.dedent(inspect.getsource(sf.LSTPSFModule.forward)).split("\n")
    lo = next(i for i, l in enumerate(lines) if l.strip().startswith("of_feat, of_logits = self.temporal_encoder("))
    hi = next(i for i, l in enumerate(lines) if l.strip().startswith("mrc_loss = (start_loss + end_loss) / 2"))
    code = compile(textwrap.dedent("\n".join(lines[lo:hi + 1])), "<reference MRC step>", "exec")
    g = torch.Generator().manual_seed(71)
    B, Lf = 2, 10
    ofq = torch.randint(-127, 128, (B, Lf, 2, 56, 56), generator=g, dtype=torch.int32).to(torch.int8)
    of = up4(ofq) / 127
    of_mask = torch.ones(B, Lf + 2, dtype=torch.long)
    of_mask[1, 7 + 2:] = 0
    samp = torch.randint(3, cfg.tgb.vocab, (B, 7), generator=g)
    smask = torch.ones_like(samp)
    smask[1, 5:] = 0
    start_targets = torch.tensor([2, 1], dtype=torch.long)
    end_targets = torch.tensor([6, 12], dtype=torch.long)          # 12 > L: clamped to the ignore index
    for p_ in ref.parameters():
        p_.requires_grad_(True)
    ref.zero_grad()
    env = dict(self=ref, of=of, of_mask=of_mask, batch=dict(sampler_question=samp, sampler_question_attention_mask=smask), start_targets=start_targets,
               end_targets=end_targets, CrossEntropyLoss=CrossEntropyLoss, torch=torch)
    exec(code, env)
    env["mrc_loss"].backward()
    out = dict(of_q8=ofq, of_mask=of_mask, sampler_ids=samp, sampler_mask=smask, start_targets=start_targets, end_targets=end_targets,
               mrc_loss=env["mrc_loss"].detach().reshape(1), of_logits=env["of_logits"].detach())
    n_g = 0
    for n, p_ in ref.temporal_encoder.named_parameters():
        if p_.grad is not None and p_.grad.abs().max() > 0:
            gq = p_.grad.detach()
            if n.endswith("word_embeddings.weight"):           # the word-embedding table: only the rows that were looked up
                rows = torch.unique(samp)
                out["g_rows:" + n] = rows
                gq = gq[rows]
            out["g:" + n] = gq
            n_g += 1
    print(f"sf mrc step: loss {float(env['mrc_loss']):.6f}, {n_g} gradient tensors")
    save("tiny_sf_mrc_step", **out)


def main():
    install_shim()
    torch.manual_seed(0)
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["e2e", "int", "full", "pre", "train", "refine", "answers", "bf16", "modules", "trainstep", "raft2", "sfmrc", "raft224"]
    if "e2e" in which:
        ref_ib, cfg_ib, sd_ib = e2e_fixture("instructblip")
        ref_b2, cfg_b2, sd_b2 = e2e_fixture("blip2")
        component_fixtures(ref_ib, cfg_ib, sd_ib, ref_b2, cfg_b2, sd_b2)
    if "int" in which:
        integer_tables()
    if "full" in which:
        fullsize_probes()
    if "pre" in which:
        preprocess_fixture()
    if "train" in which:
        train_fixture()
    if "refine" in which:
        refine_fixture()
    if "answers" in which:
        refine_answers_fixture()
    if "bf16" in which:
        bf16_reference_fixtures()
    if "modules" in which:
        module_fixtures()
    if "trainstep" in which:
        train_step_fixture()
    if "raft2" in which:
        raft_sensitive_fixture()
    if "raft224" in which:
        raft224_fixture()
    if "sfmrc" in which:
        sf_mrc_fixture()


if __name__ == "__main__":
    main()
