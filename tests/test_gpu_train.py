"""a14 on the GPU: HIP token splice / label masking (bit-exact), shifted cross-entropy forward + backward vs the
vectors recorded from the reference's lines, and a whole LoRA micro-step vs the same step done with plain PyTorch."""
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "the -m gpu tests need the MI355X"
    return torch.device("cuda:0")


def test_concat_text_io_and_labels_bit_exact(dev):
    from videotgb_amd import train
    g = load_golden("train_loss")
    for c in range(3):
        B, Li, Lo, prefix, V, pad = [int(x) for x in g[f"meta_{c}"]]
        toks, lens, labels = train.concat_text_input_output(g[f"q_ids_{c}"].to(dev), g[f"q_att_{c}"].to(dev), g[f"a_ids_{c}"].to(dev),
                                                            g[f"a_att_{c}"].to(dev), pad, prefix)
        assert torch.equal(toks["input_ids"].cpu(), g[f"llm_ids_{c}"])
        assert torch.equal(toks["attention_mask"].cpu(), g[f"llm_att_{c}"])
        assert torch.equal(lens.cpu(), g[f"input_len_{c}"])
        assert torch.equal(labels.cpu(), g[f"labels_{c}"])
        toks2, lens2 = train.concat_text_input_output(g[f"q_ids_{c}"].to(dev), g[f"q_att_{c}"].to(dev), g[f"a_ids_{c}"].to(dev), g[f"a_att_{c}"].to(dev))
        assert torch.equal(toks2["input_ids"], toks["input_ids"]) and torch.equal(lens2, lens)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_shifted_ce_forward_backward_vs_reference(dev, dtype):
    from videotgb_amd import train
    g = load_golden("train_loss")
    for c in range(3):
        logits = g[f"logits_{c}"].to(dev).to(dtype).requires_grad_(True)
        labels = g[f"labels_{c}"].to(dev)
        loss = train.shifted_cross_entropy(logits, labels)
        (loss * 2.0).backward()
        ref, dref = g[f"loss_{c}"], g[f"dlogits_{c}"] * 2.0
        if dtype == torch.float32:
            assert abs(loss.item() - ref.item()) <= 2e-6 * abs(ref.item())              # fp32: summation order only
            assert (logits.grad.cpu() - dref).abs().max() <= 1e-7
        else:                                                # bf16 logits: compare with fp32 math on the SAME rounded inputs
            lf = logits.detach().float().cpu().requires_grad_(True)
            V = lf.shape[-1]
            r2 = torch.nn.functional.cross_entropy(lf[:, :-1].reshape(-1, V), g[f"labels_{c}"][:, 1:].reshape(-1))
            (r2 * 2.0).backward()
            assert abs(loss.item() - r2.item()) <= 2e-6 * abs(r2.item())
            assert (logits.grad.float().cpu() - lf.grad).abs().max() <= 2.0 ** -8 * lf.grad.abs().max()   # output rounding only
            assert abs(loss.item() - ref.item()) <= 2e-2 * abs(ref.item())
    # full-vocabulary row length, every target ignored but one; all-ignored -> nan like CrossEntropyLoss(mean)
    B, S, V = 2, 5, 32000
    lg = torch.randn(B, S, V, device=dev)
    lb = torch.full((B, S), -100, dtype=torch.long, device=dev)
    assert torch.isnan(train.shifted_cross_entropy(lg, lb))
    lb[1, 3] = 31999
    want = torch.nn.functional.cross_entropy(lg[1, 2][None], lb[1, 3][None])
    assert abs(train.shifted_cross_entropy(lg, lb).item() - want.item()) <= 1e-5
    with pytest.raises(ValueError):
        train.shifted_cross_entropy(lg[0], lb)


def test_lora_micro_step_matches_plain_pytorch(dev):
    """tiny Llama + LoRA: loss and adapter gradients of LoraTrainStep (HIP splice/labels/CE) vs the oracle functions
    driving the same modules; then 4 accumulated micro-batches move only the adapter weights."""
    from oracle import vtgb_oracle as O
    from videotgb_amd import llm, train
    lm = llm.build_llama("tiny", torch.float32, dev)

    class Holder:                                            # the attribute layout LoraTrainStep expects of an LSTP twin
        pass
    m = Holder(); m.model = Holder(); m.model.language_model = lm
    step = train.LoraTrainStep(m, pad_token_id=0, lr=1e-2, accumulate_grad_batches=4, train_prefix=False)
    lm.eval()                                                # dropout off for the comparison
    with torch.no_grad():
        for n, p in lm.named_parameters():
            if "lora_B" in n:
                p.normal_(0, 0.02)
    g = torch.Generator().manual_seed(5)
    B, P, Li, Lo, H, V = 3, 4, 6, 5, lm.config.hidden_size, lm.config.vocab_size
    prefix = torch.randn(B, P, H, generator=g).to(dev)
    qlen, alen = torch.tensor([6, 3, 1]), torch.tensor([5, 2, 4])
    q_att = (torch.arange(Li)[None] < qlen[:, None]).long()
    a_att = (torch.arange(Lo)[None] < alen[:, None]).long()
    q = torch.randint(3, V, (B, Li), generator=g) * q_att
    a = torch.randint(3, V, (B, Lo), generator=g) * a_att
    loss = step.loss(prefix, q.to(dev), q_att.to(dev), a.to(dev), a_att.to(dev))
    loss.backward()
    got = {n: p.grad.clone() for n, p in lm.named_parameters() if p.requires_grad}
    lm.zero_grad()
    toks, lens = O.concat_text_input_output(q, q_att, a, a_att)
    labels = O.lm_labels(toks["input_ids"], lens, 0, P).to(dev)
    emb = lm.get_input_embeddings()(toks["input_ids"].to(dev))
    logits = lm(inputs_embeds=torch.cat([prefix, emb], 1),
                attention_mask=torch.cat([torch.ones(B, P, dtype=torch.long), toks["attention_mask"]], 1).to(dev))[0]
    ref = O.shifted_cross_entropy(logits, labels)
    ref.backward()
    assert abs(loss.item() - ref.item()) <= 1e-5 * abs(ref.item())
    assert set(got) == {n for n, p in lm.named_parameters() if "lora_" in n}
    for n, p in lm.named_parameters():
        if p.requires_grad:
            assert (got[n] - p.grad).abs().max() <= 1e-5 * p.grad.abs().max() + 1e-8, n
    lm.zero_grad()
    before = {n: p.detach().clone() for n, p in lm.named_parameters()}
    flags = [step.step(prefix, q.to(dev), q_att.to(dev), a.to(dev), a_att.to(dev))[1] for _ in range(4)]
    assert flags == [False, False, False, True]
    for n, p in lm.named_parameters():
        assert ("lora_" in n) == (not torch.equal(before[n], p.detach())), n


def test_training_forward_and_prefix_gradients_vs_reference(dev, tiny_sd):
    """C5's real trainable set (LSTP_Vicuna_IVT_module.py:682-690 freezes RAFT / ViT / TGB only): loss of one ragged
    micro-batch and the gradients of EVERY Q-Former parameter, query_tokens and language_projection against the vectors the
    reference's own ``LSTPModule.forward`` + ``loss.backward()`` produced (tests/golden/make_golden.py trainstep).
    fp32 mode: loss to 1e-5 relative, every gradient to 1e-4 of its own max (HIP forward, PyTorch-recompute backward)."""
    from test_gpu_e2e import build
    from videotgb_amd import train
    m, cfg = build("instructblip", tiny_sd, dev, "f32")
    g = load_golden("tiny_train_step")
    step = train.LoraTrainStep.__new__(train.LoraTrainStep)          # no LoRA here: the IV flavour the fixture was recorded with
    step.m, step.lm, step.pad_token_id, step.train_prefix = m, m.model.language_model, 0, True
    params = train.enable_prefix_training(m.model)
    for p in m.model.language_model.parameters():
        p.requires_grad = True
    m.model.language_model.eval()
    frames = (g["frames_q8"].float() * float(g["q8_scale"])).to(dev)
    widths = g["widths"].tolist()
    prefix = step.prefix(frames, g["qformer_ids"].to(dev), g["qformer_mask"].to(dev), widths)
    assert prefix.requires_grad and prefix.shape == (2, 32, cfg.llm_hidden)
    loss = step.loss(prefix, g["question"].to(dev), g["question_mask"].to(dev), g["answer"].to(dev), g["answer_mask"].to(dev))
    loss.backward()
    ref = float(g["loss"])
    print(f"[train step] loss {loss.item():.6f} vs reference {ref:.6f}")
    assert abs(loss.item() - ref) <= 1e-5 * abs(ref)
    names, plist = train.prefix_params(m.model)
    worst = 0.0
    for n, p in zip(names, plist):
        want = g["g:" + n]
        assert p.grad is not None, n
        err = (p.grad.float().cpu() - want).abs().max().item()
        scale = want.abs().max().item()
        if scale > 1e-7:      # (key biases have an analytically zero gradient -- softmax shift invariance -- : rounding noise on both sides)
            worst = max(worst, err / scale)
        assert err <= 1e-4 * scale + 1e-10, (n, err, scale)
    print(f"[train step] {len(names)} gradient tensors, worst max|diff| / max|ref| = {worst:.2e}")
    assert all(p.grad is None for p in m.model.vision_model.parameters())
    # the step itself: after AdamW the HIP Q-Former must see the new weights (packed tables invalidated)
    full = train.LoraTrainStep(m, pad_token_id=0, lr=1e-2, accumulate_grad_batches=1)
    before = m.model.qformer.state_dict()["encoder.layer.0.attention.attention.query.weight"].clone()
    p0 = full.prefix(frames, g["qformer_ids"].to(dev), g["qformer_mask"].to(dev), widths).detach().clone()
    _, stepped = full.step_frames(frames, g["qformer_ids"].to(dev), g["qformer_mask"].to(dev), widths, g["question"].to(dev),
                                  g["question_mask"].to(dev), g["answer"].to(dev), g["answer_mask"].to(dev))
    assert stepped and not torch.equal(before, m.model.qformer.state_dict()["encoder.layer.0.attention.attention.query.weight"])
    p1 = full.prefix(frames, g["qformer_ids"].to(dev), g["qformer_mask"].to(dev), widths).detach()
    assert (p1 - p0).abs().max() > 1e-4
    n_train = sum(p.numel() for p in full.params)
    print(f"[train step] trainable parameters incl. LoRA adapters: {n_train}")


@pytest.mark.parametrize("B,H,hd,Sq,Skv,use_mask,use_drop", [(2, 2, 24, 38, 38, True, False), (3, 12, 64, 32, 257, False, True), (2, 12, 64, 100, 100, True, True),
                                                             (1, 2, 32, 12, 9, True, True), (2, 4, 88, 70, 130, False, False)])
def test_attention_train_forward_backward_vs_torch_autograd(dev, B, H, hd, Sq, Skv, use_mask, use_drop):
    """train_attn.hip (vtgb_attn_train_forward / _backward) against torch autograd of the same expression in float64: output and
    dq / dk / dv, with an additive key mask (incl. a fully masked tail, -10000 as in the reference) and an injected dropout mask."""
    from videotgb_amd import train
    g = torch.Generator().manual_seed(B * 1000 + Sq)
    D = H * hd
    q, k, v = (torch.randn(B, s, D, generator=g).to(dev).requires_grad_(True) for s in (Sq, Skv, Skv))
    mask = None
    if use_mask:
        mask = torch.zeros(B, Skv)
        mask[-1, Skv - Skv // 3:] = -10000.0
        mask = mask.to(dev)
    drop = None
    if use_drop:
        drop = ((torch.rand(B, H, Sq, Skv, generator=g) >= 0.1).float() / 0.9).to(dev)
    scale = hd ** -0.5
    out = train._HipAttention.apply(q, k, v, H, scale, mask, drop)
    go = torch.randn(B, Sq, D, generator=g).to(dev)
    out.backward(go)
    qd, kd, vd = (t.detach().double().requires_grad_(True) for t in (q, k, v))
    s = torch.einsum("bihd,bjhd->bhij", qd.view(B, Sq, H, hd), kd.view(B, Skv, H, hd)) * scale
    if mask is not None:
        s = s + mask.double()[:, None, None, :]
    p = torch.softmax(s, -1)
    if drop is not None:
        p = p * drop.double()
    ref = torch.einsum("bhij,bjhd->bihd", p, vd.view(B, Skv, H, hd)).reshape(B, Sq, D)
    ref.backward(go.double())
    for name, got, want in (("out", out, ref), ("dq", q.grad, qd.grad), ("dk", k.grad, kd.grad), ("dv", v.grad, vd.grad)):
        err = (got.double() - want).abs().max().item()
        assert err <= 2e-5 * max(want.abs().max().item(), 1e-3), (name, err)


@pytest.mark.parametrize("code", ["f32", "bf16"])
def test_linear_forward_dgrad_wgrad_on_the_library_gemm(dev, code):
    """_HipLinear: y, dX, dW, db through vtgb_gemm (fp32 FMA kernel / bf16 MFMA kernel incl. the padded odd contraction length of
    the wgrad) against torch in float64."""
    from videotgb_amd import ops, train
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 37, 48, generator=g).to(dev).requires_grad_(True)        # M = 111 rows: the wgrad contracts over 111
    w = (torch.randn(96, 48, generator=g) * 0.2).to(dev).requires_grad_(True)
    b = torch.randn(96, generator=g).to(dev).requires_grad_(True)
    y = train._HipLinear.apply(x, w, b, ops.dtype_code(code))
    go = torch.randn(3, 37, 96, generator=g).to(dev)
    y.backward(go)
    xd, wd, bd = (t.detach().double().requires_grad_(True) for t in (x, w, b))
    yr = torch.nn.functional.linear(xd, wd, bd)
    yr.backward(go.double())
    tol = 1e-5 if code == "f32" else 1.5e-2
    for name, got, want in (("y", y, yr), ("dx", x.grad, xd.grad), ("dw", w.grad, wd.grad), ("db", b.grad, bd.grad)):
        err = (got.double() - want).abs().max().item()
        assert err <= tol * want.abs().max().item(), (name, code, err)


def test_sf_mrc_step_trains_the_temporal_encoder_vs_reference(dev, tiny_sd):
    """The self-refinement MRC step (LSTP_SF_module.py:275-296) through train.tgb_with_grad -- the TGB forward in fusion mode as an
    autograd graph on the library's own GEMM / attention kernels, forward and backward -- against the reference's own lines run on
    the same weights (tests/golden/make_golden.py sfmrc): the loss, the span logits and the gradient of every temporal_encoder
    parameter the reference's ``mrc_loss.backward()`` reaches (41 tensors).  fp32 mode; 1e-4 of each tensor's scale."""
    from videotgb_amd import models, refine, train
    cfg, sd = tiny_sd["instructblip"]
    g = load_golden("tiny_sf_mrc_step")
    te = models.TemporalEncoder(cfg.tgb, "f32")
    te.load_state_dict({k[len("temporal_encoder."):]: v for k, v in sd.items() if k.startswith("temporal_encoder.")}, strict=True)
    te.to(dev)
    for n, p in te.named_parameters():
        p.requires_grad = "embed_positions" not in n
    of = g["of_q8"].float().repeat_interleave(4, -1).repeat_interleave(4, -2).to(dev) / 127
    _, logits = train.tgb_with_grad(te, of, g["of_mask"].to(dev), g["sampler_ids"].to(dev), g["sampler_mask"].to(dev), "fusion")
    assert (logits.detach().cpu() - g["of_logits"]).abs().max() <= 1e-4 * g["of_logits"].abs().max()
    # ... and the same logits as the fused inference stage (what eval runs)
    with torch.no_grad():
        _, fused = te(encoder_embeds=of, attention_mask=g["of_mask"].to(dev), encoder_hidden_states=g["sampler_ids"].to(dev),
                      encoder_attention_mask=g["sampler_mask"].to(dev), mode="fusion")
    assert (logits.detach() - fused).abs().max() <= 1e-4 * fused.abs().max()
    loss = refine.mrc_loss(logits, g["start_targets"].to(dev), g["end_targets"].to(dev))
    loss.backward()
    assert abs(loss.item() - float(g["mrc_loss"])) <= 1e-5 * float(g["mrc_loss"])
    n_checked, worst = 0, 0.0
    for n, p in te.named_parameters():
        key = "g:" + n
        if key not in g:
            assert p.grad is None or p.grad.abs().max() <= 1e-9, n          # the reference reaches exactly these parameters
            continue
        got = p.grad.float().cpu()
        if "g_rows:" + n in g:
            got = got[g["g_rows:" + n]]
        want = g[key]
        err, scale = (got - want).abs().max().item(), want.abs().max().item()
        if scale <= 1e-6:       # analytically zero gradients (fc.bias: one scalar added to every feature of a token, removed by the LayerNorm
            assert got.abs().max().item() <= 1e-6, n      # behind it; key biases: softmax shift invariance): rounding noise on both sides
            continue
        worst = max(worst, err / scale)
        assert err <= 1e-4 * scale + 1e-10, (n, err, scale)
        n_checked += 1
    print(f"[sf mrc step] loss {loss.item():.6f}; {n_checked} gradient tensors, worst max|diff| / max|ref| = {worst:.2e}")
    assert n_checked >= 30


def test_dropout_masks_are_injectable_and_replayable(dev, tiny_sd):
    """Training-mode dropout of the Q-Former graph (xinstructblip.py:679,707,788,1045, p = 0.1): fresh masks change the prefix, the
    recorded masks replay it exactly, and the gradient with masks matches torch autograd finite differences in direction."""
    from test_gpu_e2e import build
    from videotgb_amd import train
    m, cfg = build("instructblip", tiny_sd, dev, "f32")
    train.enable_prefix_training(m.model)
    g = load_golden("tiny_train_step")
    frames = (g["frames_q8"].float() * float(g["q8_scale"])).to(dev)
    with torch.no_grad():
        img = m.model.vision_model(pixel_values=frames, return_dict=True, act_output=True).last_hidden_state
    rep = torch.as_tensor(g["widths"].tolist(), device=dev)
    ids, mask = torch.repeat_interleave(g["qformer_ids"].to(dev), rep, 0), torch.repeat_interleave(g["qformer_mask"].to(dev), rep, 0)
    widths = g["widths"].tolist()
    base = train.prefix_with_grad(m.model, img, ids, mask, widths)
    d1 = train.Dropout(0.1, generator=torch.Generator(device=dev).manual_seed(3))
    p1 = train.prefix_with_grad(m.model, img, ids, mask, widths, dropout=d1)
    assert (p1 - base).abs().max() > 1e-3 and len(d1.drawn) >= 2 * (2 * cfg.qformer.layers)          # probs + out per attention, FFNs, embeddings
    keep = float(torch.ones((), dtype=torch.float32) / (1.0 - 0.1))                                    # what train.Dropout stores for a kept element
    for name, v in d1.drawn.items():                                                                    # masks are exactly {0, 1 / (1 - p)}
        vals = set(torch.unique(v).tolist())
        assert vals <= {0.0, keep} and keep in vals, (name, sorted(vals)[:4])
    p2 = train.prefix_with_grad(m.model, img, ids, mask, widths, dropout=train.Dropout(masks=dict(d1.drawn)))
    assert torch.equal(p1, p2)                                                                          # replay: bit-identical
    frac = sum(float((v == 0).float().mean()) for v in d1.drawn.values()) / len(d1.drawn)
    assert 0.05 < frac < 0.15


# ----------------------------------------------------------------------------- train_ops.hip: operands in place, LayerNorm / GELU both ways
def _as_stored(t, dtype, pad, shift):
    """t as a view with leading dimension > width (pad) and/or a base that is not 16-byte aligned (shift): the kernel's scalar path."""
    r, c = t.shape
    buf = torch.zeros(r * (c + pad) + shift + 8, dtype=dtype, device=t.device)
    v = buf[shift: shift + r * (c + pad)].view(r, c + pad)[:, :c]
    v.copy_(t.to(dtype))
    return v


@pytest.mark.parametrize("M,N,K", [(130, 70, 50), (257, 768, 1408), (64, 192, 20), (1, 5, 3), (96, 200, 3001)])      # the last: split contraction
@pytest.mark.parametrize("compute", ["bf16", "f32"])
def test_gemm_train_reads_operands_in_place(dev, M, N, K, compute):
    """vtgb_gemm_train in all four layout pairs x storage types, aligned and not, vs float64 on the operands as the kernel rounds them
    (bf16 compute: products of bf16 values are exact in fp32, so only the summation order differs: 2e-6 of sum |a||b|; fp32: same bound)."""
    from videotgb_amd import _lib as L, train
    g = torch.Generator().manual_seed(M * 7 + N)
    a = torch.randn(M, K, generator=g).to(dev)
    b = torch.randn(N, K, generator=g).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    code = L.BF16 if compute == "bf16" else L.F32
    n = 0
    for a_km in (False, True):
        for b_km in (False, True):
            for a_dt in (torch.float32, torch.bfloat16):
                for b_dt in (torch.float32, torch.bfloat16):
                    for pad, shift in ((0, 0), (3, 1)):
                        sa = _as_stored(a.t() if a_km else a, a_dt, pad, shift)
                        sb = _as_stored(b.t() if b_km else b, b_dt, pad, shift)
                        out = train.gemm_train(sa, a_km, sb, b_km, bias if n % 2 == 0 else None, code)
                        ra = (sa.t() if a_km else sa)
                        rb = (sb.t() if b_km else sb)
                        if compute == "bf16":
                            ra, rb = ra.bfloat16(), rb.bfloat16()
                        ra, rb = ra.double(), rb.double()
                        want = ra @ rb.t() + (bias.double() if n % 2 == 0 else 0.0)
                        bound = 2e-6 * (ra.abs() @ rb.abs().t() + 1.0)
                        assert out.shape == (M, N) and out.dtype == torch.float32
                        assert ((out.double() - want).abs() <= bound).all(), (a_km, b_km, a_dt, b_dt, pad, shift, (out.double() - want).abs().max().item())
                        n += 1
    with pytest.raises(ValueError):
        train.gemm_train(a, False, b[:, :-1], False, None, code)


def test_col_sum_is_deterministic_and_exact_to_rounding(dev):
    from videotgb_amd import train
    x = torch.randn(1031, 777, device=dev)
    s1, s2 = train.col_sum(x), train.col_sum(x)
    assert torch.equal(s1, s2)
    assert (s1.double() - x.double().sum(0)).abs().max() <= 1e-5 * x.abs().sum(0).max()
    v = x[:, 5:300]                                              # a column slice: leading dimension 777
    assert (train.col_sum(v).double() - v.double().sum(0)).abs().max() <= 1e-5 * x.abs().sum(0).max()


@pytest.mark.parametrize("D", [48, 768, 1408, 2048])
@pytest.mark.parametrize("use_resid,use_mask", [(False, False), (True, False), (True, True), (False, True)])
def test_layernorm_train_forward_backward_vs_torch_autograd(dev, D, use_resid, use_mask):
    """LayerNorm(x * mask + resid) both ways vs the torch graph the reference runs (dropout -> add -> nn.LayerNorm) in float64."""
    from videotgb_amd import train
    g = torch.Generator().manual_seed(D)
    rows = (3, 37)
    x = (torch.randn(*rows, D, generator=g) * 2 + 0.5).to(dev).requires_grad_(True)
    r = torch.randn(*rows, D, generator=g).to(dev).requires_grad_(True) if use_resid else None
    m = ((torch.rand(*rows, D, generator=g) >= 0.1).float() / 0.9).to(dev) if use_mask else None
    gamma = (1 + 0.1 * torch.randn(D, generator=g)).to(dev).requires_grad_(True)
    beta = (0.1 * torch.randn(D, generator=g)).to(dev).requires_grad_(True)
    w = torch.randn(*rows, D, generator=g).to(dev)
    y = train.layer_norm(x, gamma, beta, 1e-12, r, m)
    (y * w).sum().backward()
    leaves = [x, gamma, beta] + ([r] if use_resid else [])
    got = [t.grad.clone() for t in leaves]
    x64, g64, b64 = (t.detach().double().requires_grad_(True) for t in (x, gamma, beta))
    r64 = r.detach().double().requires_grad_(True) if use_resid else None
    s = x64 * (m.double() if use_mask else 1.0) + (r64 if use_resid else 0.0)
    y64 = torch.nn.functional.layer_norm(s, (D,), g64, b64, 1e-12)
    (y64 * w.double()).sum().backward()
    assert (y.double() - y64).abs().max() <= 2e-6 * y64.abs().max()
    for t, ref in zip(got, [x64, g64, b64] + ([r64] if use_resid else [])):
        assert (t.double() - ref.grad).abs().max() <= 5e-6 * ref.grad.abs().max(), (t.shape, (t.double() - ref.grad).abs().max().item())
    # deterministic parameter gradients: a second backward gives the same bits
    for t in leaves:
        t.grad = None
    (train.layer_norm(x, gamma, beta, 1e-12, r, m) * w).sum().backward()
    assert torch.equal(gamma.grad, got[1]) and torch.equal(beta.grad, got[2])


def test_gelu_train_forward_backward_vs_torch(dev):
    from videotgb_amd import train
    for n in (7, 1024, 4099 * 3):
        x = (torch.randn(n, device=dev) * 3).requires_grad_(True)
        w = torch.randn(n, device=dev)
        y = train.gelu(x)
        (y * w).sum().backward()
        x64 = x.detach().double().requires_grad_(True)
        y64 = torch.nn.functional.gelu(x64)
        (y64 * w.double()).sum().backward()
        assert (y.double() - y64).abs().max() <= 1e-6 * (1 + y64.abs().max())
        assert (x.grad.double() - x64.grad).abs().max() <= 2e-6 * (1 + x64.grad.abs().max())


def test_training_graph_issues_no_blas_and_no_torch_layernorm_kernel(dev, tiny_sd):
    """The Q-Former training graph forward + backward: every GEMM, attention, LayerNorm and GELU launch is the library's -- no
    hipBLASLt / rocBLAS kernel (Cijk_*), no torch layer_norm / GELU kernel, and no operand copy kernels for the GEMMs' sake."""
    from torch.profiler import ProfilerActivity, profile
    from test_gpu_e2e import build
    from videotgb_amd import train
    m, cfg = build("instructblip", tiny_sd, dev, "f32")
    pm = m.model
    train.enable_prefix_training(pm)
    g = torch.Generator().manual_seed(3)
    img = torch.randn(8, (cfg.vit.image // cfg.vit.patch) ** 2 + 1, cfg.vit.hidden, generator=g).to(dev)
    ids = torch.randint(3, cfg.qformer.vocab, (8, 6), generator=g).to(dev)

    def step(code):
        out = train.prefix_with_grad(pm, img, ids, torch.ones_like(ids), [4, 4], "mean", compute_dtype=code, dropout=train.Dropout(0.1, generator=torch.Generator(device=dev).manual_seed(1)))
        out.square().sum().backward()

    for code in ("f32", "bf16"):
        step(code)
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            step(code)
            torch.cuda.synchronize()
        names = [e.key for e in prof.key_averages() if e.device_time_total > 0]
        assert any("tg_mfma_kernel" in n or "tg_f32_kernel" in n for n in names), names
        assert any("ln_train_fwd_kernel" in n for n in names) and any("ln_train_bwd_kernel" in n for n in names) and any("gelu_bwd_kernel" in n for n in names)
        bad = [n for n in names if "Cijk_" in n or "layer_norm" in n.lower() or "GeluCUDAKernel" in n or "GeluBackward" in n]
        assert not bad, bad


def test_training_ops_accept_what_the_torch_ops_accepted(dev):
    """a row-broadcast gradient (strides (0, 1): a batch-1 linear output that was mean-reduced), empty inputs (a rank whose clips
    all have width 0: dist.py) -- F.linear / F.layer_norm / torch.sum took them, so do the library-backed ops."""
    from videotgb_amd import train
    from videotgb_amd._lib import BF16, F32
    g = torch.Generator(device=dev).manual_seed(3)
    row = torch.randn(1, 24, generator=g, device=dev)
    bc = row.expand(7, 24)                                           # strides (0, 1)
    assert torch.allclose(train.col_sum(bc), bc.sum(0), rtol=1e-6, atol=1e-6)
    assert torch.equal(train.col_sum(torch.empty(0, 24, device=dev)), torch.zeros(24, device=dev))
    w = torch.randn(16, 24, generator=g, device=dev, requires_grad=True)
    b = torch.randn(16, generator=g, device=dev, requires_grad=True)
    for code in (F32, BF16):
        x = torch.randn(1, 24, generator=g, device=dev, requires_grad=True)
        y = train._HipLinear.apply(x, w, b, code)
        y.mean().backward()                                          # dY is a broadcast of one value: strides (0, 0)
        assert torch.isfinite(x.grad).all() and torch.allclose(b.grad, torch.full_like(b, 1.0 / 16), rtol=1e-5, atol=1e-6)
        w.grad = b.grad = None
        e = train._HipLinear.apply(torch.empty(0, 24, device=dev, requires_grad=True), w, b, code)      # widths = [0, 0]: no rows at all
        assert tuple(e.shape) == (0, 16)
        e.sum().backward()
        assert torch.equal(w.grad, torch.zeros_like(w)) and torch.equal(b.grad, torch.zeros_like(b))
        w.grad = b.grad = None
    gam = torch.ones(24, device=dev, requires_grad=True)
    bet = torch.zeros(24, device=dev, requires_grad=True)
    z = train.layer_norm(torch.empty(0, 24, device=dev), gam, bet, 1e-12)
    assert tuple(z.shape) == (0, 24)
    z.sum().backward()
    assert torch.equal(gam.grad, torch.zeros_like(gam))
