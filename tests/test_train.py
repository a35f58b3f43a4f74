"""a14 (config C5 training step), CPU side: the oracle vs the vectors recorded from the reference's own lines; host
logic of the product (LoRA wrapper, schedule, freezing); 2-rank gloo gradient exchange is in test_dist.py."""
import math

import pytest
import torch

from conftest import load_golden
from oracle import vtgb_oracle as O


def cases():
    g = load_golden("train_loss")
    for c in range(3):
        B, Li, Lo, prefix, V, pad = [int(x) for x in g[f"meta_{c}"]]
        yield c, g, B, Li, Lo, prefix, V, pad


def test_oracle_concat_labels_and_loss_match_reference():
    for c, g, B, Li, Lo, prefix, V, pad in cases():
        toks, lens = O.concat_text_input_output(g[f"q_ids_{c}"], g[f"q_att_{c}"], g[f"a_ids_{c}"], g[f"a_att_{c}"])
        assert torch.equal(toks["input_ids"], g[f"llm_ids_{c}"]) and torch.equal(toks["attention_mask"], g[f"llm_att_{c}"])
        assert lens == g[f"input_len_{c}"].tolist()
        labels = O.lm_labels(toks["input_ids"], lens, pad, prefix)
        assert torch.equal(labels, g[f"labels_{c}"])
        logits = g[f"logits_{c}"].clone().requires_grad_(True)
        loss = O.shifted_cross_entropy(logits, labels)
        assert torch.equal(loss.reshape(1), g[f"loss_{c}"])                       # same ATen kernels
        loss.backward()
        assert torch.allclose(logits.grad, g[f"dlogits_{c}"], rtol=0, atol=1e-9)


def test_lora_linear_matches_formula_and_freezes_base():
    from videotgb_amd import train
    torch.manual_seed(0)
    base = torch.nn.Linear(24, 40, bias=False)
    w0 = base.weight.detach().clone()
    lin = train.LoraLinear(base, r=8, lora_alpha=32, lora_dropout=0.1).eval()
    x = torch.randn(5, 24)
    assert torch.equal(lin(x), base(x))                                            # B = 0 at init: identity update
    with torch.no_grad():
        lin.lora_B["default"].weight.normal_()
    want = O.lora_linear(x, w0, None, lin.lora_A["default"].weight, lin.lora_B["default"].weight, 32, 8)
    assert torch.allclose(lin(x), want, atol=1e-6)
    names = dict(lin.named_parameters())
    assert set(names) == {"weight", "lora_A.default.weight", "lora_B.default.weight"}          # peft 0.4.0 key layout
    assert not names["weight"].requires_grad and names["lora_A.default.weight"].requires_grad


def test_apply_lora_trainable_set_and_schedule():
    from videotgb_amd import llm, train
    lm = llm.build_llama("tiny", torch.float32, "cpu")
    params = train.apply_lora(lm)
    cfg = lm.config
    per_layer = 2 * 8 * cfg.hidden_size + 8 * (cfg.num_attention_heads * (cfg.hidden_size // cfg.num_attention_heads)) \
        + 8 * (cfg.num_key_value_heads * (cfg.hidden_size // cfg.num_attention_heads))
    assert sum(p.numel() for p in params) == cfg.num_hidden_layers * per_layer
    assert all(("lora_" in n) == p.requires_grad for n, p in lm.named_parameters())
    # Vicuna-7B geometry: 32 x 2 x (4096*8 + 8*4096) = 4,194,304 (16.8 MB of fp32 gradients)
    assert 32 * 2 * (4096 * 8 + 8 * 4096) == 4_194_304
    out = train.configure_optimizers(params, lr=1e-4, max_steps=-1, warmup_ratio=0.1)
    f = train.cosine_schedule_lambda(int(-1 * 0.1), -1)
    for step in range(6):
        assert f(step) == O.cosine_schedule_lambda(step, 0, -1)
    assert out["lr_scheduler"]["interval"] == "epoch" and isinstance(out["optimizer"], torch.optim.AdamW)
    with pytest.raises(NotImplementedError):
        train.configure_optimizers(params, scheduler="linear")
    f2 = train.cosine_schedule_lambda(10, 100)
    assert f2(5) == 0.5 and abs(f2(55) - 0.5 * (1 + math.cos(math.pi * 0.5))) < 1e-12 and f2(100) == 0.0
