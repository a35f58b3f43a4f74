"""-m gpu: the whole path (LSTP.generate / LSTP_blip2.generate twins) against the vectors
recorded from the reference's own generate() on a tiny configuration, RAFT inline (in libvtgb.so,
like every other stage), noise injected; HF generate and the hipGraph decoder of the bench both run.
fp32 mode: frame indices and greedy token ids bit-exact, tensors to 2e-4 of their scale (summation
order only; RAFT runs 20 recurrent iterations); bf16 mode: indices equal, tensors to the stated
bf16 tolerances."""
import pytest
import torch

from conftest import deq, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from videotgb_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def build(arch, tiny_sd, dev, dtype, raft_dtype=None):
    from videotgb_amd import llm, models
    from videotgb_amd.synth import synth_tensor
    cfg, sd = tiny_sd[arch]
    lm = llm.build_llama("tiny", torch.float32, dev)
    lsd = {k: synth_tensor("model.language_model." + k, tuple(v.shape)).to(dev) for k, v in lm.state_dict().items()}
    lm.load_state_dict(lsd, strict=True)
    cls = models.LSTP if arch == "instructblip" else models.LSTP_blip2
    m = cls(cfg, dev, language_model=lm, compute_dtype=dtype, raft_dtype=raft_dtype)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith("model.language_model.") for k in missing), (missing[:4], unexpected[:4])
    m.to(dev)
    return m, cfg


class BE(dict):
    __getattr__ = dict.__getitem__


@pytest.mark.parametrize("arch", ["instructblip", "blip2"])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("fast_decode", [False, True])
def test_generate_vs_reference(dev, tiny_sd, arch, dtype, fast_decode):
    """fast_decode=True + bf16 is the configuration bench.py times (all-HIP RAFT, graph-replayed greedy decode).  The first run of the bf16 case is the
    ALL-bf16 one (raft_dtype="bf16", the headline's opt-in); the three-way comparison below runs the module's DEFAULT: RAFT at the reference's fp32 accuracy."""
    m, cfg = build(arch, tiny_sd, dev, dtype, raft_dtype="bf16" if dtype == "bf16" else None)
    g = load_golden(f"tiny_{arch}_e2e")
    te = BE(input_ids=g["prompt_ids"].to(dev), attention_mask=g["prompt_mask"].to(dev))
    if arch == "instructblip":
        te["qformer_input_ids"] = g["qformer_ids"].to(dev)
        te["qformer_attention_mask"] = g["qformer_mask"].to(dev)
    se = BE(input_ids=g["sampler_ids"].to(dev), attention_mask=g["sampler_mask"].to(dev))
    ids, cand, st = m.generate(deq(g, "frames_q8").to(dev), deq(g, "flow_frames_q8").to(dev), int(g["nframe"]), te, se,
                               do_sample=False, temperature=None, max_new_tokens=6, use_cache=False, noise=g["noise"].to(dev),
                               return_stages=True, fast_decode=fast_decode)

    def chk(name, got, ref, tol, rms_tol=None):
        got, ref = got.float().cpu(), ref.float()
        err, scale = (got - ref).abs().max().item(), ref.abs().max().item()
        rms = float((got - ref).double().pow(2).mean().sqrt() / ref.double().pow(2).mean().sqrt())
        print(f"[e2e {arch} {dtype}] {name}: max|diff|={err:.3e} rel_rms={rms:.3e} max|ref|={scale:.3e}")
        assert err <= tol * scale, name
        assert rms_tol is None or rms <= rms_tol, name
    tol = 2e-4 if dtype == "f32" else 1.5e-2      # bf16: 2 x observed (7e-3 of the logit scale, 2.3e-3 of the prefix scale)
    # bf16 RAFT (a mode the reference does not have): 20 recurrent iterations of bf16 convolutions; observed rel-RMS 6e-3
    chk("raft flow", st["of"][0, :-1], g["raft_flow"], 2e-4 if dtype == "f32" else 6e-2, None if dtype == "f32" else 1.5e-2)
    assert torch.equal(st["of"][0, -1], st["of"][0, -2])            # last flow repeated (eval/utils/model.py:82)
    chk("tgb logits", st["tgb_logits"], g["tgb_logits"], tol)
    assert cand.cpu().tolist() == g["cand_index"].tolist()
    assert torch.equal(st["sampled"].cpu(), g["sampled"])
    chk("prefix", st["prefix"], g["prefix"], tol)
    chk("inputs_embeds", st["inputs_embeds"], g["inputs_embeds"], tol)
    if dtype == "f32":
        assert ids.cpu().tolist() == g["greedy_ids"].tolist()      # greedy token ids bit-exact at fp32
    else:
        # against the reference's OWN bf16 mode (torch.autocast(bfloat16), its Lightning `precision: bf16`; RAFT fp32 there):
        # HIP-bf16 is at least as close to the reference's fp32 numbers as the reference's bf16 run is, and the two bf16
        # runs agree to 2 x that distance (bounds = 2 x observed ratios; the numbers are printed)
        # Like for like: the reference's bf16 run keeps RAFT in fp32, so the three-way comparison runs the HIP path in the same
        # split -- which is the module's DEFAULT (raft_dtype=None -> fp32 accuracy on the matrix cores: "f16c8" since round 6, "bf16x3" in round 5; everything else bf16).  The all-bf16 run above (bf16 RAFT too: a mode the reference does not
        # have) is held to the absolute tolerances; its TGB logits additionally carry the bf16 flow's 6e-3 rel-RMS, which has no
        # counterpart in e_ref.
        from test_gpu_stages import rel_rms
        r16 = load_golden(f"tiny_{arch}_e2e_bf16ref")
        assert cand.cpu().tolist() == r16["cand_index"].tolist()
        m2, _ = build(arch, tiny_sd, dev, dtype)
        assert m2.of_extractor.code == 3                            # VTGB_F16C8
        _, cand2, st2 = m2.generate(deq(g, "frames_q8").to(dev), deq(g, "flow_frames_q8").to(dev), int(g["nframe"]), te, se,
                                    do_sample=False, temperature=None, max_new_tokens=6, use_cache=False, noise=g["noise"].to(dev),
                                    return_stages=True, fast_decode=fast_decode)
        assert cand2.cpu().tolist() == r16["cand_index"].tolist()
        for name, key in (("tgb logits", "tgb_logits"), ("prefix", "prefix")):
            hip = st2[key].float().cpu()
            e_ref, e32, e16 = rel_rms(r16[key], g[key]), rel_rms(hip, g[key]), rel_rms(hip, r16[key])
            print(f"[e2e {arch} bf16] {name}: relRMS hip16~ref32={e32:.3e} hip16~ref16={e16:.3e} ref16~ref32={e_ref:.3e}; max|diff| "
                  f"hip16~ref32={(hip - g[key]).abs().max():.3e} ref16~ref32={(r16[key] - g[key]).abs().max():.3e}")
            assert e32 <= 1.5 * e_ref and e16 <= 2.0 * e_ref, name
        if fast_decode is False:
            assert ids.cpu().tolist() == r16["greedy_ids"].tolist()


def test_precomputed_flow_and_concat_pool(dev, tiny_sd):
    """batch['of'] contract (src/models/LSTP_SF_module.py:476) + concat pooling (LSTP_module.py:477-481)
    against the oracle composed in the same order."""
    from oracle import vtgb_oracle as O
    m, cfg = build("instructblip", tiny_sd, dev, "f32")
    sd = tiny_sd["instructblip"][1]
    g = torch.Generator().manual_seed(8)
    B, T, N, nframe = 2, 10, 8, 4
    frames = torch.randn(B * N, 3, 56, 56, generator=g)
    of = torch.rand(B, T, 2, 224, 224, generator=g) * 2 - 1
    sids = torch.randint(3, cfg.tgb.vocab, (B, 7), generator=g)
    qids = torch.randint(3, cfg.qformer.vocab, (B, 5), generator=g)
    noise = O.gumbel_noise((2, 2 * B, T), g)
    ref = O.lstp_prefix(sd, arch="instructblip", frames=frames, nframe=nframe, sampler_ids=sids, sampler_mask=torch.ones_like(sids),
                        noise=noise, vit_heads=cfg.vit.heads, qf_heads=cfg.qformer.heads, tgb_heads=cfg.tgb.heads,
                        fusion_layer=cfg.tgb.fusion_layer, of=of, qformer_ids=qids, qformer_mask=torch.ones_like(qids), pool="concat")
    sampled, idx, _ = m.select_frames(frames.to(dev).view(B, N, 3, 56, 56), of.to(dev), sids.to(dev), torch.ones_like(sids).to(dev),
                                      nframe, noise.to(dev))
    assert idx.cpu().tolist() == ref["cand_index"].tolist()
    te = {"qformer_input_ids": qids.to(dev), "qformer_attention_mask": torch.ones_like(qids).to(dev)}
    prefix = m.prefix(sampled, B, nframe, te, "concat")
    assert prefix.shape == ref["prefix"].shape == (B, nframe * 32, cfg.llm_hidden)
    assert (prefix.cpu() - ref["prefix"]).abs().max().item() <= 2e-4 * ref["prefix"].abs().max().item()


class _VicunaTok:
    """What KeywordsStoppingCriteria needs of the Vicuna tokenizer (eval/utils/builder_utils.py:320-346): '</s>' -> [bos, eos]; special tokens are skipped
    by batch_decode."""
    bos_token_id, eos_token_id = 1, 2
    name_or_path = "lmsys/vicuna-7b-v1.1"

    def __call__(self, text):
        assert text == "</s>"
        return type("E", (), {"input_ids": [1, 2]})()

    def batch_decode(self, ids, skip_special_tokens=True):
        return [" ".join(str(int(t)) for t in row.tolist() if not (skip_special_tokens and int(t) <= 2)) for row in ids]


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_reference_eval_call_runs_on_the_graph_decoder(dev, tiny_sd, dtype):
    """The reference's OWN call (eval/inference.py:98-109): ``generate(frames, flow_frames, nframe, text_encoding, sampler_text_encoding, do_sample=True,
    temperature=..., max_new_tokens=128, use_cache=False, stopping_criteria=[KeywordsStoppingCriteria(['</s>'], tokenizer, input_ids)])`` with NO
    extra keyword: since round 6 it decodes on libvtgb.so (fast_decode="auto": sampling + keyword stopping inside the graph decoder) -- at
    temperature -> 0 it returns the greedy ids of HF generate under the same stopping criteria, and the profiler sees no hipBLASLt / rocBLAS kernel."""
    from torch.profiler import ProfilerActivity, profile
    from videotgb_amd.builder_utils import KeywordsStoppingCriteria
    m, cfg = build("instructblip", tiny_sd, dev, dtype)
    lm = m.model.language_model
    lm.generation_config.eos_token_id, lm.generation_config.pad_token_id = 2, 0
    g = load_golden("tiny_instructblip_e2e")
    te = BE(input_ids=g["prompt_ids"].to(dev), attention_mask=g["prompt_mask"].to(dev), qformer_input_ids=g["qformer_ids"].to(dev),
            qformer_attention_mask=g["qformer_mask"].to(dev))
    se = BE(input_ids=g["sampler_ids"].to(dev), attention_mask=g["sampler_mask"].to(dev))
    frames, flow_frames, nframe = deq(g, "frames_q8").to(dev), deq(g, "flow_frames_q8").to(dev), int(g["nframe"])
    crit = lambda: [KeywordsStoppingCriteria(["</s>"], _VicunaTok(), te.input_ids)]
    noise = g["noise"].to(dev)
    ref, cand_ref = m.generate(frames, flow_frames, nframe, te, se, do_sample=False, temperature=None, max_new_tokens=16, use_cache=False,
                               stopping_criteria=crit(), noise=noise, fast_decode=False)                       # HF generate, greedy (16 tokens: over 128
    # tokens of a random-init model a near-tie eventually separates hipBLASLt's summation order from libvtgb.so's; the 128-token comparison below is
    # against the graph decoder's own greedy path under the same criteria)
    ref128, _ = m.generate(frames, flow_frames, nframe, te, se, do_sample=False, temperature=None, max_new_tokens=128, use_cache=False,
                           stopping_criteria=crit(), noise=noise, fast_decode=True)
    assert m._graph_plan(lm, torch.zeros(1, 4, 8, device=dev), torch.ones(1, 4, device=dev), True, 0.2, crit(), {}) is not None
    kw = dict(do_sample=True, temperature=1e-9, max_new_tokens=128, use_cache=False)                             # eval/inference.py:103-107, temperature -> 0
    m.generate(frames, flow_frames, nframe, te, se, stopping_criteria=crit(), noise=noise, **kw)                  # (captures the graph)
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        ids, cand = m.generate(frames, flow_frames, nframe, te, se, stopping_criteria=crit(), noise=noise, **kw)
        torch.cuda.synchronize()
    names = {e.key for e in prof.key_averages()}
    blas = sorted(n for n in names if "Cijk_" in n or "rocblas" in n.lower() or "hipblaslt" in n.lower())
    assert not blas, blas
    assert cand.tolist() == cand_ref.tolist()
    assert ids.tolist() == ref128.tolist(), (ids.tolist(), ref128.tolist())
    if dtype == "f32":
        n = min(ids.shape[1], ref.shape[1])
        assert ids[:, :n].tolist() == ref[:, :n].tolist(), (ids.tolist(), ref.tolist())
    assert ids.shape[0] == 1 and 1 <= ids.shape[1] <= 128
    if ids.shape[1] < 128:                                          # stopped: the last token is the keyword's (EOS -> 2 survives the reference's `outputs == 0 -> 2` line)
        assert int(ids[0, -1]) == 2
    # the reference's temperature: runs, reproducible under injected noise, and is a different draw under different noise
    u = torch.rand(128, 1, generator=torch.Generator().manual_seed(0))
    a, _ = m.generate(frames, flow_frames, nframe, te, se, stopping_criteria=crit(), noise=noise, do_sample=True, temperature=0.2, max_new_tokens=128, use_cache=False, sample_noise=u)
    b, _ = m.generate(frames, flow_frames, nframe, te, se, stopping_criteria=crit(), noise=noise, do_sample=True, temperature=0.2, max_new_tokens=128, use_cache=False, sample_noise=u)
    assert a.tolist() == b.tolist()
    with pytest.raises(TypeError):
        m.generate(frames, flow_frames, nframe, te, se, noise=noise, fast_decode=True, num_beams=4, **kw)
