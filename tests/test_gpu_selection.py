"""-m gpu: does running RAFT in the bf16 MFMA mode (what bench.py times) change WHICH frames the Temporal Grounding Bridge
selects, relative to the fp32 exactness mode (the reference's arithmetic)?  argmax over (logit + Gumbel noise) is
discontinuous, so a small flow error could in principle flip a span.  64 synthetic clips per length at the real sizes
(224 x 224 frames, RAFT-large, BERT-base TGB, 32 candidate frames -> 8), T = 96 (configs C3 / C5) and T = 256 (C4):
the same clips, the same injected noise, flows from both RAFT modes, everything downstream identical.

Reported: flow rel-RMS, TGB logit max|diff|, the fraction of the 4 span endpoints per clip that move, and the fraction
of clips whose final ``cand_index`` differs.  Bound (stated): at most 2 of 64 clips per length may differ in
``cand_index`` (observed: 0 of 64 at T = 96, 1 of 64 at T = 256: 2 of 256 span endpoints moved), and the TGB logits must
agree to 2e-2 of their range (observed 9e-3).

Round 3: run for two RAFT weight sets -- the default N(0, 0.02) one, whose flow hardly depends on the images, and the
INPUT-SENSITIVE one (synth.raft_sensitive_state_dict: fan-in-scaled, the flow follows the correlation features), on which an
encoder-level error does reach the flow, the TGB logits and the selection."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from videotgb_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def parts(dev):
    from videotgb_amd import models, synth
    cfg = synth.full_cfg("instructblip")
    sd = synth.synth_state_dict({**synth.tgb_shapes(cfg.tgb), **synth.raft_shapes()}, 0)
    for k in list(sd):
        if ".downsample.1." in k:
            sd[k] = sd[k.replace(".downsample.1.", ".norm3.")]
    tgb = models.TemporalEncoder(cfg.tgb, "bf16")
    tgb.load_state_dict({k[len("temporal_encoder."):]: v for k, v in sd.items() if k.startswith("temporal_encoder.")}, strict=True)
    rafts = {}
    for wset, rsd in (("default", sd), ("sensitive", synth.raft_sensitive_state_dict(0))):
        for dt in ("bf16", "f32"):
            r = models.Raft(dt)
            r.load_state_dict({k[len("of_extractor."):]: v for k, v in rsd.items() if k.startswith("of_extractor.")}, strict=True)
            rafts[wset, dt] = r.to(dev)
    return cfg, tgb.to(dev), rafts


def make_clips(kind, n, T, gen, dev):
    """randn: the bench's frames.  moving: a smooth random texture translated by a per-clip velocity that
    changes twice along the clip (CLIP-normalised range), so the flow field has temporal structure for the TGB."""
    if kind == "randn":
        return torch.randn(n, T, 3, 224, 224, generator=gen, device=dev)
    base = torch.nn.functional.interpolate(torch.randn(n, 3, 40, 40, generator=gen, device=dev), size=(448, 448), mode="bicubic", align_corners=False)
    v = torch.randn(n, 3, 2, generator=gen, device=dev) * 1.5
    out = torch.empty(n, T, 3, 224, 224, device=dev)
    pos = torch.zeros(n, 2, device=dev)
    for t in range(T):
        pos = pos + v[:, min(3 * t // T, 2)]
        for i in range(n):
            oy, ox = int(pos[i, 0].round().item()) % 224, int(pos[i, 1].round().item()) % 224
            out[i, t] = base[i, :, oy:oy + 224, ox:ox + 224]
    return out


@pytest.mark.parametrize("weights", ["default", "sensitive"])
@pytest.mark.parametrize("T,per_call", [(96, 8), (256, 4)])
def test_cand_index_stable_under_bf16_raft(dev, parts, T, per_call, weights):
    from videotgb_amd import ops
    cfg, tgb, rafts = parts
    gen = torch.Generator(device=dev).manual_seed(1000 + T)
    n_clips, N, nframe = 64, 32, 8
    moved, differ, total_ep = 0, 0, 0
    flow_rms, logit_err, logit_scale = 0.0, 0.0, 0.0
    for c0 in range(0, n_clips, per_call):
        kind = "randn" if (c0 // per_call) % 2 == 0 else "moving"
        frames = make_clips(kind, per_call, T, gen, dev)
        sids = torch.cat([torch.full((per_call, 1), 101, device=dev), torch.randint(1000, 30000, (per_call, 12), generator=gen, device=dev),
                          torch.full((per_call, 1), 102, device=dev)], 1)
        noise = -torch.empty(2, 2 * per_call, T, device=dev).exponential_(generator=gen).log()
        res = {}
        for dt in ("bf16", "f32"):
            fl = rafts[weights, dt].forward_clips(frames)
            of = torch.cat([fl, fl[:, -1:]], dim=1)                         # last flow repeated (eval/utils/model.py:81-82)
            _, logits = tgb(encoder_embeds=of, attention_mask=torch.ones(per_call, T + 2, dtype=torch.long, device=dev),
                            encoder_hidden_states=sids, encoder_attention_mask=torch.ones_like(sids), mode="multi_modal")
            sel = ops.span_select(logits, noise, 0.5)
            idx = ops.span_to_frames(sel, T, N, nframe, "A")
            res[dt] = (of, logits, sel, idx)
            del fl
        a, b = res["bf16"], res["f32"]
        flow_rms = max(flow_rms, float(((a[0] - b[0]).double().pow(2).mean().sqrt() / b[0].double().pow(2).mean().sqrt()).item()))
        logit_err = max(logit_err, float((a[1] - b[1]).abs().max().item()))
        logit_scale = max(logit_scale, float((b[1].max() - b[1].min()).item()))
        moved += int((a[2] != b[2]).sum().item())
        total_ep += a[2].numel()
        differ += int((a[3] != b[3]).any(dim=1).sum().item())
        del res, a, b, frames
    print(f"[selection T={T} raft weights={weights}] {n_clips} clips: flow rel-RMS (bf16 vs fp32 RAFT) <= {flow_rms:.3e}; TGB logits max|diff| {logit_err:.3e} of range "
          f"{logit_scale:.3e}; span endpoints moved {moved}/{total_ep}; clips with a different cand_index {differ}/{n_clips}")
    # logits: default weights 2e-2 of their range (observed 9e-3); sensitive weights 1e-1 (observed 4.5e-2 ... 5.3e-2: the bf16 flow is
    # 1.45e-2 off there and the flow reaches the logits) -- the assertion that matters is the selection itself
    assert logit_err <= (2e-2 if weights == "default" else 1e-1) * logit_scale
    assert differ <= 2
