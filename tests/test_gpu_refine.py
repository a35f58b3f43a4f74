"""f4 on the GPU: per-frame answers produced in ONE batch (HIP prefix path + graph-replayed greedy decode) equal
the reference's double loop -- nframe frames at a time through Q-Former + projection + HF generate
(src/models/LSTP_SF_module.py:157-204) -- token for token at fp32."""
import pytest
import torch

from conftest import deq, load_golden
from test_gpu_e2e import build, dev  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def test_frame_answers_match_reference_double_loop(dev, tiny_sd):
    from videotgb_amd import refine
    m, cfg = build("instructblip", tiny_sd, dev, "f32")
    g = load_golden("tiny_instructblip_e2e")
    frames = deq(g, "frames_q8").to(dev)                       # [B*N, 3, H, W] with B = 1
    B, N, nframe = 1, frames.shape[0], int(g["nframe"])
    q, qm = g["prompt_ids"].to(dev), g["prompt_mask"].to(dev)
    qt, qtm = g["qformer_ids"].to(dev), g["qformer_mask"].to(dev)
    max_length = 32 + q.shape[1] + 5
    got = refine.frame_answers(m, frames, B, qt, qtm, q, qm, max_length=max_length)
    assert got.shape == (B * N, 5)
    # the reference's loop: minibatches of nframe frames, one sequence per frame, HF generate
    lm = m.model.language_model
    ref = []
    for jj in range(N // nframe):
        mb = frames[jj * nframe:(jj + 1) * nframe]
        enc = {"qformer_input_ids": torch.repeat_interleave(qt, nframe, 0), "qformer_attention_mask": torch.repeat_interleave(qtm, nframe, 0)}
        lmi = m.prefix(mb, nframe, 1, enc, "mean")
        emb = torch.cat([lmi, m.model.get_input_embeddings()(torch.repeat_interleave(q, nframe, 0))], 1)
        am = torch.ones(emb.shape[:2], dtype=torch.long, device=dev)
        out = lm.generate(inputs_embeds=emb, attention_mask=am, do_sample=False, max_new_tokens=5, min_new_tokens=5)
        out[out == 0] = 2
        ref.append(out)
    assert torch.equal(got, torch.cat(ref, 0))
    with pytest.raises(ValueError):
        refine.frame_answers(m, frames, B, qt, qtm, q, qm, max_length=8)
