"""f4 on the GPU: per-frame answers produced in ONE batch (HIP prefix path + graph-replayed greedy decode with HF's EOS
semantics) against the ids and per-frame prefixes recorded from the reference's own loop -- LSTPSFModule.forward,
src/models/LSTP_SF_module.py:149-204, executed on the tiny reference model by tests/golden/make_golden.py -- at fp32:
prefix to 2e-4 of its scale, token ids identical."""
import pytest
import torch

from conftest import deq, load_golden
from test_gpu_e2e import build, dev  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def test_frame_answers_match_reference_loop(dev, tiny_sd):
    from videotgb_amd import refine
    m, cfg = build("instructblip", tiny_sd, dev, "f32")
    g = load_golden("tiny_instructblip_e2e")
    r = load_golden("tiny_refine_answers")
    frames = deq(g, "frames_q8").to(dev)                       # [B*N, 3, H, W] with B = 1
    B, N = 1, frames.shape[0]
    q, qm = g["prompt_ids"].to(dev), g["prompt_mask"].to(dev)
    qt, qtm = g["qformer_ids"].to(dev), g["qformer_mask"].to(dev)
    enc = {"qformer_input_ids": torch.repeat_interleave(qt, N, 0), "qformer_attention_mask": torch.repeat_interleave(qtm, N, 0)}
    pref = m.prefix(frames, N, 1, enc, "mean").cpu()           # every frame its own prefix (nframe = 1)
    ref = r["prefix_per_frame"]
    err = (pref - ref).abs().max().item()
    print(f"[refine] per-frame prefix max|diff|={err:.3e} max|ref|={ref.abs().max():.3e}")
    assert err <= 2e-4 * ref.abs().max().item()
    got = refine.frame_answers(m, frames, B, qt, qtm, q, qm, max_length=int(r["max_length"])).cpu()
    want = r["ids"]
    assert got.shape[0] == want.shape[0] == B * N
    eos = int(r["eos_token_id"])
    for i in range(B * N):                                      # rows end at their first EOS; what follows is padding on both sides
        a, b = got[i].tolist(), [t for t in want[i].tolist() if t >= 0]
        a = a[: a.index(eos) + 1] if eos in a else a
        b = b[: b.index(eos) + 1] if eos in b else b
        assert a == b, (i, a[:12], b[:12])
    with pytest.raises(ValueError):
        refine.frame_answers(m, frames, B, qt, qtm, q, qm, max_length=8)
