import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: (torch.from_numpy(z[k]) if z[k].ndim else z[k].item()) for k in z.files}


def deq(g, key, scale_key="q8_scale"):
    return g[key].float() * float(g[scale_key])


@pytest.fixture(scope="session")
def tiny_sd():
    from videotgb_amd.synth import path_state_dict, tiny_cfg
    out = {}
    for arch in ("instructblip", "blip2"):
        cfg = tiny_cfg(arch)
        cfg.vit.image = 56
        out[arch] = (cfg, path_state_dict(cfg, seed=0))
    return out


def write_hf_config(dirname, arch, cfg, llm="llama"):
    """Save the HF config.json the reference's ``*Config.from_pretrained(base_model_path)`` reads (eval/utils/model.py:33,252),
    at the dims of ``cfg`` with a tiny Llama / T5 text model -- the same configs tests/golden/make_golden.py builds."""
    from transformers import (Blip2Config, Blip2QFormerConfig, Blip2VisionConfig, InstructBlipConfig, InstructBlipQFormerConfig,
                              InstructBlipVisionConfig, LlamaConfig, T5Config)
    v, q = cfg.vit, cfg.qformer
    vkw = dict(hidden_size=v.hidden, intermediate_size=v.mlp, num_hidden_layers=v.layers, num_attention_heads=v.heads, image_size=v.image,
               patch_size=v.patch, layer_norm_eps=v.eps)
    qkw = dict(hidden_size=q.hidden, num_hidden_layers=q.layers, num_attention_heads=q.heads, intermediate_size=q.ffn,
               encoder_hidden_size=q.enc_hidden, vocab_size=q.vocab, max_position_embeddings=q.max_pos, cross_attention_frequency=q.cross_freq)
    if llm == "t5":
        tc = T5Config(vocab_size=120, d_model=cfg.llm_hidden, d_kv=16, d_ff=64, num_layers=2, num_decoder_layers=2, num_heads=2,
                      feed_forward_proj="gated-gelu", tie_word_embeddings=False, decoder_start_token_id=0, pad_token_id=0, eos_token_id=1,
                      architectures=["T5ForConditionalGeneration"])
    else:
        tc = LlamaConfig(hidden_size=cfg.llm_hidden, intermediate_size=64, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=2,
                         vocab_size=120, architectures=["LlamaForCausalLM"], bos_token_id=1, eos_token_id=2, pad_token_id=0)
    if arch == "instructblip":
        c = InstructBlipConfig(vision_config=InstructBlipVisionConfig(**vkw).to_dict(), qformer_config=InstructBlipQFormerConfig(**qkw).to_dict(),
                               text_config=tc.to_dict(), num_query_tokens=q.n_query)
    else:
        c = Blip2Config(vision_config=Blip2VisionConfig(**vkw).to_dict(), qformer_config=Blip2QFormerConfig(**qkw).to_dict(),
                        text_config=tc.to_dict(), num_query_tokens=q.n_query)
    os.makedirs(dirname, exist_ok=True)
    c.save_pretrained(dirname)
    return dirname


def full_state_dict(cfg, language_model):
    """The Lightning checkpoint's ``state_dict`` of the fixtures: seeded hot-path weights + seeded LLM weights (T5's tied
    token embeddings follow ``shared``), exactly as tests/golden/make_golden.py loads them into the reference."""
    from videotgb_amd.synth import path_state_dict, synth_tensor
    sd = path_state_dict(cfg, seed=0)
    for k, p in language_model.state_dict().items():
        sd["model.language_model." + k] = synth_tensor("model.language_model." + k, tuple(p.shape))
    for k in list(sd):
        if k.endswith("encoder.embed_tokens.weight") or k.endswith("decoder.embed_tokens.weight"):
            sd[k] = sd[k.rsplit(".", 3)[0] + ".shared.weight"]
    return sd
