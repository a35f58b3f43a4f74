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
