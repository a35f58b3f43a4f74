"""The graph-replayed greedy decoder must emit the token ids HF generate emits (the reference
calls HF generate, eval/utils/model.py:217-231).  CPU: eager step; GPU: hipGraph replay."""
import pytest
import torch


def _model(device, dtype=torch.float32):
    from videotgb_amd import llm
    return llm.build_llama("tiny", dtype, device, seed=3, num_hidden_layers=3, num_key_value_heads=1)


def _check(device, use_graph, fused=True):
    from videotgb_amd.decode import GreedyDecoder
    lm = _model(device)
    g = torch.Generator().manual_seed(0)
    emb = (torch.randn(3, 9, 32, generator=g) * 0.5).to(device)
    ref = lm.generate(inputs_embeds=emb, attention_mask=torch.ones(3, 9, dtype=torch.long, device=device), do_sample=False,
                      max_new_tokens=7, min_new_tokens=7, use_cache=True)
    dec = GreedyDecoder(lm, fused=fused)
    out = dec.generate(emb, 7, use_graph=use_graph)
    assert out.tolist() == ref.tolist()
    out2 = dec.generate(emb * 0.9, 7, use_graph=use_graph)            # state reuse / graph replay on new inputs
    ref2 = lm.generate(inputs_embeds=emb * 0.9, attention_mask=torch.ones(3, 9, dtype=torch.long, device=device), do_sample=False,
                       max_new_tokens=7, min_new_tokens=7, use_cache=True)
    assert out2.tolist() == ref2.tolist()


def _check_eos(device, use_graph):
    """No min_new_tokens on either side: rows stop at EOS, pad afterwards, and the output ends where the last row finished
    (the fixed-length decoder emitted post-EOS continuations).  EOS = a token the model does emit early."""
    from videotgb_amd.decode import GreedyDecoder
    lm = _model(device)
    g = torch.Generator().manual_seed(1)
    emb = (torch.randn(5, 6, 32, generator=g) * 0.5).to(device)
    am = torch.ones(5, 6, dtype=torch.long, device=device)
    free = lm.generate(inputs_embeds=emb, attention_mask=am, do_sample=False, max_new_tokens=9, min_new_tokens=9, use_cache=True)
    dec = GreedyDecoder(lm)
    for eos in (int(free[0, 2]), int(free[1, 0]), int(free[2, 5])):
        ref = lm.generate(inputs_embeds=emb, attention_mask=am, do_sample=False, max_new_tokens=9, eos_token_id=eos, pad_token_id=0, use_cache=True)
        out = dec.generate(emb, 9, use_graph=use_graph, eos_token_id=eos, pad_token_id=0)
        assert out.tolist() == ref.tolist(), (eos, out.tolist(), ref.tolist())
        assert (ref == eos).any()
        ref = lm.generate(inputs_embeds=emb, attention_mask=am, do_sample=False, max_new_tokens=9, min_new_tokens=4, eos_token_id=eos, pad_token_id=0,
                          use_cache=True)
        out = dec.generate(emb, 9, use_graph=use_graph, eos_token_id=eos, pad_token_id=0, min_new_tokens=4)
        assert out.tolist() == ref.tolist(), ("min_new", eos, out.tolist(), ref.tolist())
    # every row finishes at once -> shorter output, like HF
    eos = int(free[0, 0])
    e1 = emb[:1].repeat(3, 1, 1)
    ref = lm.generate(inputs_embeds=e1, attention_mask=am[:3], do_sample=False, max_new_tokens=9, eos_token_id=eos, pad_token_id=0, use_cache=True)
    out = dec.generate(e1, 9, use_graph=use_graph, eos_token_id=eos, pad_token_id=0)
    assert out.shape == ref.shape == (3, 1) and out.tolist() == ref.tolist()


def test_greedy_decoder_matches_hf_generate_cpu():
    _check("cpu", False)


def test_greedy_decoder_eos_semantics_cpu():
    _check_eos("cpu", False)


@pytest.mark.gpu
def test_greedy_decoder_hipgraph_matches_hf_generate_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    _check("cuda:0", True)
    _check("cuda:0", False)
    _check("cuda:0", True, fused=False)
    _check_eos("cuda:0", True)
    _check_eos("cuda:0", False)


@pytest.mark.gpu
def test_small_batch_decode_runs_on_the_skinny_gemm_and_matches_hipblaslt():
    """Batches <= GreedyDecoder.SKINNY_MAX_BATCH take every projection (q|k|v, o, gate|up, down, lm_head) through
    vtgb_gemm_skinny; the ids must be the ones the hipBLASLt path (F.linear) emits, under graph replay and eagerly."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from videotgb_amd import llm
    from videotgb_amd.decode import GreedyDecoder
    dev = "cuda:0"
    lm = llm.build_llama("tiny", torch.bfloat16, dev, seed=5, hidden_size=128, intermediate_size=256, num_attention_heads=4, num_key_value_heads=4,
                         num_hidden_layers=3, vocab_size=320)
    g = torch.Generator(device=dev).manual_seed(2)
    emb = (torch.randn(2, 5, 128, generator=g, device=dev) * 0.5).bfloat16()
    own, lib = GreedyDecoder(lm), GreedyDecoder(lm)
    lib.SKINNY_MAX_BATCH = 0
    for use_graph in (True, False):
        a = own.generate(emb, 6, use_graph=use_graph)
        b = lib.generate(emb, 6, use_graph=use_graph)
        assert a.tolist() == b.tolist()
    assert all("sk_ws" in st for st in own.graphs.values()) and not any("sk_ws" in st for st in lib.graphs.values())
    ref = lm.generate(inputs_embeds=emb, attention_mask=torch.ones(2, 5, dtype=torch.long, device=dev), do_sample=False, max_new_tokens=6,
                      min_new_tokens=6, use_cache=True)
    assert own.generate(emb, 6, use_graph=True, min_new_tokens=0).tolist() == ref.tolist()


@pytest.mark.gpu
def test_bf16_prefill_runs_on_libvtgb_and_matches_the_blas_path():
    """f2: at bf16 the prefill's projections go through vtgb_gemm, its causal attention through vtgb_attention
    (head_dim 128) and norms / SwiGLU through vtgb_llm_* -- no BLAS library call.  Checked against the F.linear / SDPA prefill
    (the arithmetic HF generate runs): KV caches and last hidden state within bf16 rounding, greedy ids equal."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from videotgb_amd import llm
    from videotgb_amd.decode import GreedyDecoder
    dev = "cuda:0"
    lm = llm.build_llama("tiny", torch.bfloat16, dev, seed=7, hidden_size=512, intermediate_size=1024, num_attention_heads=4, num_key_value_heads=4,
                         num_hidden_layers=3, vocab_size=320)
    g = torch.Generator(device=dev).manual_seed(4)
    for B, P in ((3, 52), (2, 131)):
        emb = (torch.randn(B, P, 512, generator=g, device=dev) * 0.5).bfloat16()
        own, blas = GreedyDecoder(lm), GreedyDecoder(lm)
        blas.PREFILL_MAX_TOKENS = 0
        assert own._use_hip_prefill(emb, P) and not blas._use_hip_prefill(emb, P)
        a = own.generate(emb, 8)
        b = blas.generate(emb, 8)
        sa, sb = next(iter(own.graphs.values())), next(iter(blas.graphs.values()))
        for li in range(3):
            for c in ("kc", "vc"):
                x, y = sa[c][li][:, :, :P].float(), sb[c][li][:, :, :P].float()
                assert (x - y).abs().max().item() <= 3e-2 * max(1.0, y.abs().max().item()), (c, li)
        # last hidden state -> first-token logits of both prefills (two bf16 arithmetics: later greedy ids of a random-weight
        # model can part at a near tie, the logits cannot differ by more than rounding)
        with torch.no_grad():
            last = own._prefill_hip(sa, emb, P)
            x = emb
            pidx = sb["ar"][:P]
            causal = torch.where(sb["ar"][None, :] <= pidx[:, None], 0.0, torch.finfo(x.dtype).min).to(x.dtype)[None, None]
            for li, w in enumerate(blas.layers):
                x = blas._layer(x, w, sb["cos"][:P], sb["sin"][:P], sb["kc"][li], sb["vc"][li], pidx, causal)
            la, lb = own._head(last).float(), blas._head(x[:, -1]).float()
        assert (la - lb).abs().max().item() <= 3e-2 * max(1.0, lb.abs().max().item())
        assert a[:, 0].tolist() == b[:, 0].tolist() == lb.argmax(-1).tolist()


def _t5(device, dtype=torch.float32):
    from transformers import T5Config, T5ForConditionalGeneration
    torch.manual_seed(0)
    cfg = T5Config(vocab_size=200, d_model=64, d_kv=16, d_ff=96, num_layers=3, num_decoder_layers=3, num_heads=4, feed_forward_proj="gated-gelu",
                   tie_word_embeddings=False, decoder_start_token_id=0, pad_token_id=0, eos_token_id=1)
    lm = T5ForConditionalGeneration(cfg).eval()
    for p in lm.parameters():
        p.data.normal_(0, 0.3)
    return lm.to(device=device, dtype=dtype)


def _check_t5(device, use_graph):
    """T5GreedyDecoder against HF generate at fp32: id for id, with and without EOS stopping (the eos id is one the model emits)."""
    from videotgb_amd.decode import T5GreedyDecoder, make_decoder
    lm = _t5(device)
    g = torch.Generator().manual_seed(1)
    emb = (torch.randn(3, 7, 64, generator=g) * 0.5).to(device)
    mask = torch.ones(3, 7, dtype=torch.long, device=device)
    dec = make_decoder(lm)
    assert isinstance(dec, T5GreedyDecoder)
    ref = lm.generate(inputs_embeds=emb, attention_mask=mask, do_sample=False, max_new_tokens=10, min_new_tokens=10)
    out = dec.generate(emb, 10, use_graph=use_graph)
    assert out.tolist() == ref.tolist()
    eos = int(ref[0, 4])                              # a token row 0 emits at step 4: rows finish at different steps
    ref = lm.generate(inputs_embeds=emb, attention_mask=mask, do_sample=False, max_new_tokens=10, eos_token_id=eos, pad_token_id=0)
    out = dec.generate(emb, 10, use_graph=use_graph, eos_token_id=eos, pad_token_id=0)
    assert out.tolist() == ref.tolist()
    ref = lm.generate(inputs_embeds=emb, attention_mask=mask, do_sample=False, max_new_tokens=10, min_new_tokens=6, eos_token_id=eos, pad_token_id=0)
    out = dec.generate(emb, 10, use_graph=use_graph, eos_token_id=eos, pad_token_id=0, min_new_tokens=6)
    assert out.tolist() == ref.tolist()


def test_t5_greedy_decoder_matches_hf_generate_cpu():
    _check_t5("cpu", False)


@pytest.mark.gpu
def test_t5_greedy_decoder_hipgraph_matches_hf_generate_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    _check_t5("cuda:0", True)
    _check_t5("cuda:0", False)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_t5_generate_runs_on_libvtgb_and_issues_no_blas_kernel(dtype):
    """SURVEY.md 8f-2, the T5 half (C1 / C2: Flan-T5 under language_model.generate, eval/utils/model.py:427-437): encoder, cross-attention
    K / V, every decoder step and the lm_head run on libvtgb.so -- vtgb_gemm / vtgb_gemm_skinny, vtgb_llm_rmsnorm (T5LayerNorm),
    vtgb_llm_attention_rows (relative position bias), vtgb_llm_gated_act (gated-gelu) -- no hipBLASLt / rocBLAS kernel in the profile,
    eager or graph-replayed.  fp32: ids equal HF generate's, id for id; bf16: the first generated token (before two bf16 arithmetics
    can part at a near tie) and the encoder output within bf16 rounding."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from torch.profiler import ProfilerActivity, profile
    from transformers import T5Config, T5ForConditionalGeneration
    from videotgb_amd.decode import T5GreedyDecoder
    dev = "cuda:0"
    torch.manual_seed(0)
    cfg = T5Config(vocab_size=320, d_model=256, d_kv=64, d_ff=512, num_layers=2, num_decoder_layers=2, num_heads=4, feed_forward_proj="gated-gelu",
                   tie_word_embeddings=False, decoder_start_token_id=0, pad_token_id=0, eos_token_id=1)
    lm = T5ForConditionalGeneration(cfg).eval()
    for p in lm.parameters():
        p.data.normal_(0, 0.05)
    for n, p in lm.named_parameters():
        if "layer_norm" in n:
            p.data.fill_(1.0)
    lm = lm.to(device=dev, dtype=dtype)
    B, P, N = 6, 21, 5
    emb = (torch.randn(B, P, 256, device=dev, generator=torch.Generator(device=dev).manual_seed(4)) * 0.5).to(dtype)
    mask = torch.ones(B, P, dtype=torch.long, device=dev)
    dec = T5GreedyDecoder(lm)
    ref = lm.generate(inputs_embeds=emb, attention_mask=mask, do_sample=False, max_new_tokens=N, min_new_tokens=N)
    out = dec.generate(emb, N, eos_token_id=1, pad_token_id=0, min_new_tokens=N)      # (captures the graph outside the profile; HF masks EOS below min_new_tokens)
    if dtype == torch.float32:
        assert out.tolist() == ref.tolist()
    else:
        assert out[:, :2].tolist() == ref[:, :2].tolist()
    if dtype == torch.float32:
        # a second call REPLAYS the captured graph on new inputs (the cross-attention K | V live in the state's buffers), and a shorter prompt
        # reuses the same state (encoder length is device data; lengths are bucketed by 64)
        for P2 in (P, P - 4):
            emb2 = (torch.randn(B, P2, 256, device=dev, generator=torch.Generator(device=dev).manual_seed(40 + P2)) * 0.5).to(dtype)
            ref2 = lm.generate(inputs_embeds=emb2, attention_mask=torch.ones(B, P2, dtype=torch.long, device=dev), do_sample=False, max_new_tokens=N,
                               min_new_tokens=N)
            assert dec.generate(emb2, N, eos_token_id=1, pad_token_id=0, min_new_tokens=N).tolist() == ref2.tolist()
        assert len(dec.graphs) == 1
    with torch.no_grad():
        enc_ref = lm.encoder(inputs_embeds=emb, attention_mask=mask).last_hidden_state.float().reshape(B * P, -1)
        enc = dec._encode_hip(emb).float()
    tol = 2e-4 if dtype == torch.float32 else 5e-2      # bf16: a few ulps of the output scale (two orders of rounding; observed 3.2e-2)
    assert (enc - enc_ref).abs().max().item() <= tol * max(1.0, enc_ref.abs().max().item())
    for use_graph in (False, True):
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            dec.generate(emb, N, use_graph=use_graph)
            torch.cuda.synchronize()
        names = {e.key for e in prof.key_averages()}
        blas = sorted(n for n in names if "Cijk_" in n or "rocblas" in n.lower() or "hipblaslt" in n.lower())
        assert not blas, blas
        if not use_graph:
            assert any("llm_attn_rows" in n for n in names) and any("llm_gated_act" in n for n in names) and any("llm_rmsnorm" in n for n in names), sorted(names)[:40]
            assert any(("gemm_skinny" in n) if dtype == torch.bfloat16 else ("gemm_f32" in n) for n in names), sorted(names)[:40]


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,kv_heads", [(torch.bfloat16, 4), (torch.float32, 4), (torch.float32, 2)])
def test_generate_issues_no_blas_kernel(dtype, kv_heads):
    """prefill, decode steps and the first-token logits run on libvtgb.so -- the profiler sees no hipBLASLt /
    rocBLAS (Tensile `Cijk_*`) kernel during GreedyDecoder.generate, eager or graph-replayed, at a batch of the throughput path's kind.
    bf16 = the throughput mode; fp32 = the exactness mode whose ids are compared token for token with HF generate (round 4: on
    vtgb_gemm's fp32 kernel, the fp32 attention and the vtgb_llm_* kernels; grouped-query heads included) -- ids checked here too."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from torch.profiler import ProfilerActivity, profile
    from videotgb_amd import llm
    from videotgb_amd.decode import GreedyDecoder
    dev = "cuda:0"
    lm = llm.build_llama("tiny", dtype, dev, seed=9, hidden_size=512, intermediate_size=1024, num_attention_heads=4, num_key_value_heads=kv_heads,
                         num_hidden_layers=2, vocab_size=320)
    emb = (torch.randn(12, 20, 512, device=dev, generator=torch.Generator(device=dev).manual_seed(4)) * 0.5).to(dtype)
    dec = GreedyDecoder(lm)
    if dtype == torch.float32:
        ref = lm.generate(inputs_embeds=emb, attention_mask=torch.ones(12, 20, dtype=torch.long, device=dev), do_sample=False, max_new_tokens=4,
                          min_new_tokens=4, use_cache=True)
        assert dec.generate(emb, 4, eos_token_id=2, pad_token_id=0, min_new_tokens=4).tolist() == ref.tolist()      # (HF masks EOS below min_new_tokens)
    dec.generate(emb, 4)                                   # capture outside the profile
    for use_graph in (False, True):
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            dec.generate(emb, 4, use_graph=use_graph)
            torch.cuda.synchronize()
        names = {e.key for e in prof.key_averages()}
        blas = sorted(n for n in names if n.startswith("Cijk_") or "Cijk_" in n or "rocblas" in n.lower() or "hipblaslt" in n.lower())
        assert not blas, blas
        if not use_graph:                                  # (graph replays show up as one launch; the eager run names the kernels)
            if dtype == torch.bfloat16:
                assert any("gemm_skinny" in n for n in names) and any("gemm_bf16" in n for n in names) and any("attn_bf16" in n for n in names), sorted(names)[:40]
            else:
                assert any("gemm_f32" in n for n in names) and any("attn_f32" in n for n in names) and any("llm_decode_attn" in n for n in names), sorted(names)[:40]


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_rope_cache_prefill_kernel_matches_hf_rotary_bit_for_bit(dtype):
    """vtgb_llm_rope_cache_prefill against transformers' apply_rotary_pos_emb arithmetic (q * cos + rotate_half(q) * sin in the model's
    dtype: each product and the sum rounded) and the cache fill of the torch prefill: identical bits."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import ctypes as C
    from videotgb_amd import _lib as L
    from videotgb_amd.decode import _rot_half
    dev = "cuda:0"
    B, S, nh, hd, tmax = 3, 21, 4, 128, 64
    g = torch.Generator(device=dev).manual_seed(0)
    qkv = torch.randn(B, S, 3 * nh, hd, generator=g, device=dev).to(dtype)
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2, device=dev, dtype=torch.float32) / hd))
    fr = torch.arange(tmax, device=dev, dtype=torch.float32)[:, None] * inv[None]
    emb = torch.cat((fr, fr), dim=-1)
    cos, sin = emb.cos().to(dtype), emb.sin().to(dtype)
    qk = qkv[:, :, : 2 * nh]
    ref_qk = qk * cos[:S, None, :] + _rot_half(qk) * sin[:S, None, :]
    kc = torch.zeros(B, nh, tmax, hd, device=dev, dtype=dtype)
    vc = torch.zeros_like(kc)
    got = qkv.clone()
    code = L.BF16 if dtype == torch.bfloat16 else L.F32
    L.check(L.lib().vtgb_llm_rope_cache_prefill(code, got.data_ptr(), kc.data_ptr(), vc.data_ptr(), cos.data_ptr(), sin.data_ptr(), B, S, nh, nh, hd, tmax,
                                                C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    assert torch.equal(got[:, :, : 2 * nh], ref_qk) and torch.equal(got[:, :, 2 * nh:], qkv[:, :, 2 * nh:])
    assert torch.equal(kc[:, :, :S], ref_qk[:, :, nh:].transpose(1, 2)) and torch.equal(vc[:, :, :S], qkv[:, :, 2 * nh:].transpose(1, 2))
    assert not kc[:, :, S:].any() and not vc[:, :, S:].any()


# ------------------------------------------------------------------------------------------------------------------------------
# round 6: the reference's own eval call (eval/inference.py:98-109: do_sample=True, temperature=0.2, stopping_criteria=[KeywordsStoppingCriteria])
# on the graph decoder
class _Tok:
    """A tokenizer of the shape KeywordsStoppingCriteria uses (eval/utils/builder_utils.py:320-346): one id per word `t<id>`, BOS = 1."""
    bos_token_id = 1
    name_or_path = "vicuna-test"

    def __call__(self, text):
        return type("E", (), {"input_ids": [1] + [int(w[1:]) for w in text.split()]})()

    def batch_decode(self, ids, skip_special_tokens=True):
        return [" ".join(f"t{int(t)}" for t in row.tolist() if not (skip_special_tokens and int(t) <= 2)) for row in ids]


def _hf_sample_reference(lm, emb, n, temperature, top_k, top_p, u):
    """HF's sampling arithmetic restated step by step on full re-forwards (GenerationMixin._sample: temperature, top-k, top-p warpers, then a draw
    from softmax) with the draw taken by inverse CDF at u[step, row] over the tokens in descending-probability order."""
    ids = torch.zeros(emb.shape[0], 0, dtype=torch.long, device=emb.device)
    x = emb
    for s in range(n):
        logits = lm(inputs_embeds=x).logits[:, -1].float() / temperature
        k = min(top_k, logits.shape[-1]) if top_k else logits.shape[-1]
        vals, idx = logits.topk(k, -1)
        p = torch.softmax(vals, -1)
        if top_p < 1.0:
            keep = (p.cumsum(-1) - p) < top_p
            p = torch.where(keep, p, torch.zeros_like(p))
            p = p / p.sum(-1, keepdim=True)
        j = (p.cumsum(-1) < u[s][:, None]).sum(-1).clamp(max=k - 1)
        nxt = idx.gather(-1, j[:, None])
        ids = torch.cat([ids, nxt], 1)
        x = torch.cat([x, lm.get_input_embeddings()(nxt)], 1)
    return ids


def _check_sampling(device, use_graph):
    from videotgb_amd.decode import GreedyDecoder
    lm = _model(device)
    g = torch.Generator().manual_seed(2)
    emb = (torch.randn(3, 8, 32, generator=g) * 0.5).to(device)
    am = torch.ones(3, 8, dtype=torch.long, device=device)
    dec = GreedyDecoder(lm)
    greedy = lm.generate(inputs_embeds=emb, attention_mask=am, do_sample=False, max_new_tokens=8, min_new_tokens=8, use_cache=True)
    # temperature -> 0: the sampler's distribution collapses onto the greedy token, whatever the noise
    cold = dec.generate(emb, 8, use_graph=use_graph, do_sample=True, temperature=1e-6, top_k=50)
    assert cold.tolist() == greedy.tolist()
    # injected noise: the ids are those of HF's warpers + an inverse-CDF draw at the same uniform numbers
    u = torch.rand(8, 3, generator=g).to(device)
    for temperature, top_k, top_p in ((0.7, 50, 1.0), (1.3, 5, 1.0), (1.0, 0, 0.8)):
        want = _hf_sample_reference(lm, emb, 8, temperature, top_k, top_p, u)
        got = dec.generate(emb, 8, use_graph=use_graph, do_sample=True, temperature=temperature, top_k=top_k, top_p=top_p, sample_noise=u)
        assert got.tolist() == want.tolist(), (temperature, top_k, top_p)
    hot = dec.generate(emb, 8, use_graph=use_graph, do_sample=True, temperature=1.5, top_k=0, sample_noise=u)
    assert hot.tolist() != greedy.tolist()                            # (a hot sampler does leave the greedy path)
    with pytest.raises(ValueError):
        dec.generate(emb, 4, use_graph=use_graph, do_sample=True, temperature=0.0)


def _check_keyword_stopping(device, use_graph):
    """Against HF generate with the SAME KeywordsStoppingCriteria object: the token-suffix test (two-token keyword met mid-sequence), the text test
    (a keyword whose own tokenisation never occurs but whose text does), both with EOS handling on, and no hit at all."""
    from videotgb_amd.builder_utils import KeywordsStoppingCriteria
    from videotgb_amd.decode import GreedyDecoder, keyword_stop_plan
    lm = _model(device)
    tok = _Tok()
    g = torch.Generator().manual_seed(5)
    emb = (torch.randn(1, 7, 32, generator=g) * 0.5).to(device)
    am = torch.ones(1, 7, dtype=torch.long, device=device)
    prompt = torch.zeros(1, 3, dtype=torch.long)                      # (start_len = 3: the text prompt's length, as in eval/inference.py:93)
    free = lm.generate(inputs_embeds=emb, attention_mask=am, do_sample=False, max_new_tokens=24, min_new_tokens=24, use_cache=True)[0].tolist()
    dec = GreedyDecoder(lm)
    cases = [f"t{free[9]} t{free[10]}",                                # token test: the ids of steps 9, 10
             f"t{free[4]}",                                            # one-token keyword
             "t4711 t4712"]                                            # never met: runs to max_new_tokens
    for kw in cases:
        crit = KeywordsStoppingCriteria([kw], tok, prompt)
        ref = lm.generate(inputs_embeds=emb, attention_mask=am, do_sample=False, max_new_tokens=24, use_cache=True, stopping_criteria=[crit],
                          eos_token_id=2, pad_token_id=0)
        out = dec.generate(emb, 24, use_graph=use_graph, eos_token_id=2, pad_token_id=0, **keyword_stop_plan([crit]))
        assert out.tolist() == ref.tolist(), (kw, out.tolist(), ref.tolist())
    assert len(lm.generate(inputs_embeds=emb, attention_mask=am, do_sample=False, max_new_tokens=24, use_cache=True,
                           stopping_criteria=[KeywordsStoppingCriteria([cases[0]], tok, prompt)], eos_token_id=None, pad_token_id=0)[0]) == 11


def test_sampling_decoder_cpu():
    _check_sampling("cpu", False)


def test_keyword_stopping_decoder_cpu():
    _check_keyword_stopping("cpu", False)


@pytest.mark.gpu
def test_sampling_and_keyword_stopping_decoder_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    for use_graph in (True, False):
        _check_sampling("cuda:0", use_graph)
        _check_keyword_stopping("cuda:0", use_graph)
