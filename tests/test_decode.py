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


def test_greedy_decoder_matches_hf_generate_cpu():
    _check("cpu", False)


@pytest.mark.gpu
def test_greedy_decoder_hipgraph_matches_hf_generate_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    _check("cuda:0", True)
    _check("cuda:0", False)
    _check("cuda:0", True, fused=False)
