"""The LLM at the end of the path is third-party on both sides (HF transformers):
the reference calls ``language_model.generate(inputs_embeds=...)`` and so do we.  This file only
builds random-init HF models of the named shapes (there is no network for checkpoints)."""
from __future__ import annotations

import torch

LLM_SHAPES = {
    # Vicuna-7B == Llama-7B geometry (InstructBLIP-Vicuna-7B text_config)
    "vicuna-7b": dict(hidden_size=4096, intermediate_size=11008, num_hidden_layers=32, num_attention_heads=32,
                      num_key_value_heads=32, vocab_size=32000, max_position_embeddings=2048),
    # the tiny model of the golden fixtures (tests/golden/make_golden.py)
    "tiny": dict(hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=2,
                 vocab_size=120),
}


def build_llama(name: str = "vicuna-7b", dtype=torch.bfloat16, device="cuda", seed: int = 0, **overrides):
    """Random-init LlamaForCausalLM created directly on ``device`` in ``dtype`` (N(0, 0.02) weights)."""
    from transformers import LlamaConfig, LlamaForCausalLM
    kw = dict(LLM_SHAPES[name])
    kw.update(overrides)
    cfg = LlamaConfig(architectures=["LlamaForCausalLM"], bos_token_id=1, eos_token_id=2, pad_token_id=0, **kw)
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        with torch.device(device):
            model = LlamaForCausalLM(cfg)
    finally:
        torch.set_default_dtype(prev)
    g = torch.Generator(device=device).manual_seed(seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() >= 2:
                p.normal_(0.0, 0.02, generator=g)
            elif "norm" in n:
                p.fill_(1.0)
            else:
                p.zero_()
    model.eval()
    model.generation_config.pad_token_id = 0
    return model
