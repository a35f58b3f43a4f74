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
    tgbs = {}
    for dt in ("bf16", "f32"):       # bf16: what bench.py runs; f32: isolates the flow's contribution to the logit difference (see below)
        tgbs[dt] = models.TemporalEncoder(cfg.tgb, dt)
        tgbs[dt].load_state_dict({k[len("temporal_encoder."):]: v for k, v in sd.items() if k.startswith("temporal_encoder.")}, strict=True)
        tgbs[dt].to(dev)
    rafts = {}
    for wset, rsd in (("default", sd), ("sensitive", synth.raft_sensitive_state_dict(0))):
        for dt in ("bf16", "bf16x3", "f16c8", "f32"):
            r = models.Raft(dt)
            r.load_state_dict({k[len("of_extractor."):]: v for k, v in rsd.items() if k.startswith("of_extractor.")}, strict=True)
            rafts[wset, dt] = r.to(dev)
    return cfg, tgbs, rafts


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
    cfg, tgbs, rafts = parts
    gen = torch.Generator(device=dev).manual_seed(1000 + T)
    n_clips, N, nframe = 64, 32, 8
    modes = ("bf16", "bf16x3", "f16c8")
    keys = [(m, t) for m in modes for t in ("bf16", "f32")]
    moved, differ, total_ep = {k: 0 for k in keys}, {k: 0 for k in keys}, 0
    flow_rms, logit_err, logit_scale = {m: 0.0 for m in modes}, {k: 0.0 for k in keys}, {"bf16": 0.0, "f32": 0.0}
    floor16, floor_moved = 0.0, 0
    for c0 in range(0, n_clips, per_call):
        kind = "randn" if (c0 // per_call) % 2 == 0 else "moving"
        frames = make_clips(kind, per_call, T, gen, dev)
        sids = torch.cat([torch.full((per_call, 1), 101, device=dev), torch.randint(1000, 30000, (per_call, 12), generator=gen, device=dev),
                          torch.full((per_call, 1), 102, device=dev)], 1)
        noise = -torch.empty(2, 2 * per_call, T, device=dev).exponential_(generator=gen).log()
        res = {}
        for dt in modes + ("f32",):
            fl = rafts[weights, dt].forward_clips(frames)
            of = torch.cat([fl, fl[:, -1:]], dim=1)                         # last flow repeated (eval/utils/model.py:81-82)
            res[dt] = {"of": of}
            for tdt, tgb in tgbs.items():
                _, logits = tgb(encoder_embeds=of, attention_mask=torch.ones(per_call, T + 2, dtype=torch.long, device=dev),
                                encoder_hidden_states=sids, encoder_attention_mask=torch.ones_like(sids), mode="multi_modal")
                sel = ops.span_select(logits, noise, 0.5)
                res[dt][tdt] = (logits, sel, ops.span_to_frames(sel, T, N, nframe, "A"))
            del fl
        b = res["f32"]
        # control: the bf16 TGB on the fp32-RAFT flow perturbed by 1e-5 relative noise (about the bf16x3 mode's distance from fp32)
        pert = b["of"] * (1.0 + 1e-5 * torch.randn(b["of"].shape, generator=gen, device=dev))
        _, lp = tgbs["bf16"](encoder_embeds=pert, attention_mask=torch.ones(per_call, T + 2, dtype=torch.long, device=dev),
                             encoder_hidden_states=sids, encoder_attention_mask=torch.ones_like(sids), mode="multi_modal")
        floor16 = max(floor16, float((lp - b["bf16"][0]).abs().max().item()))
        floor_moved += int((ops.span_select(lp, noise, 0.5) != b["bf16"][1]).sum().item())
        del pert, lp
        total_ep += b["f32"][1].numel()
        for tdt in tgbs:
            logit_scale[tdt] = max(logit_scale[tdt], float((b[tdt][0].max() - b[tdt][0].min()).item()))
        for m in modes:
            a = res[m]
            flow_rms[m] = max(flow_rms[m], float(((a["of"] - b["of"]).double().pow(2).mean().sqrt() / b["of"].double().pow(2).mean().sqrt()).item()))
            for tdt in tgbs:
                logit_err[m, tdt] = max(logit_err[m, tdt], float((a[tdt][0] - b[tdt][0]).abs().max().item()))
                moved[m, tdt] += int((a[tdt][1] != b[tdt][1]).sum().item())
                differ[m, tdt] += int((a[tdt][2] != b[tdt][2]).any(dim=1).sum().item())
        del res, a, b, frames
    for m, tdt in keys:
        print(f"[selection T={T} raft weights={weights} RAFT {m} vs fp32, TGB {tdt}] {n_clips} clips: flow rel-RMS <= {flow_rms[m]:.3e}; TGB logits max|diff| "
              f"{logit_err[m, tdt]:.3e} = {logit_err[m, tdt] / logit_scale[tdt]:.2e} of their range {logit_scale[tdt]:.3e}; span endpoints moved "
              f"{moved[m, tdt]}/{total_ep}; clips with a different cand_index {differ[m, tdt]}/{n_clips}")
    print(f"[selection T={T} raft weights={weights} control] the bf16 TGB on the fp32-RAFT flow x (1 + 1e-5 randn): logits max|diff| {floor16:.3e} = "
          f"{floor16 / logit_scale['bf16']:.2e} of their range; span endpoints moved {floor_moved}/{total_ep}")
    # bf16 RAFT (a REDUCED-PRECISION opt-in the reference does not have; round 6: no longer what bench.py's headline runs), bf16 TGB.  logits: default weights 2e-2 of their range (observed 9e-3);
    # sensitive weights 1e-1 (observed 4.5e-2 ... 5.3e-2: the bf16 flow is 1.45e-2 off there and the flow reaches the logits); at most 2 of
    # 64 clips may differ
    assert logit_err["bf16", "bf16"] <= (2e-2 if weights == "default" else 1e-1) * logit_scale["bf16"]
    assert differ["bf16", "bf16"] <= 2
    # bf16x3 RAFT (split-bf16 operands: the reference's fp32 accuracy on the matrix cores).  Through the fp32 TGB -- which isolates what the
    # flow difference does -- the north star's 1e-3 on the logit tensor holds and EVERY clip selects the same frames.  Through the bf16 TGB any
    # perturbation of the flow, however small, flips some of the bf16 roundings inside the TGB and re-draws part of ITS quantisation noise
    # (~sqrt(perturbation x ulp) per rounding site): the control above -- 1e-5 relative noise on the fp32 flow -- moves the bf16 logits by the same
    # 5e-3 ... 9e-3 of their range, the size of the reference's own bf16-vs-fp32 TGB difference.  That number is a property of
    # the bf16 TGB, not of RAFT: printed, and bounded only by the selection (at most 2 of 64 clips, as for any flow source).
    # f16c8 (round 6: the update block on fp16 + fp8-correction operands, encoders / correlation at bf16x3 -- the module's default and bench.py's
    # headline mode) is held to exactly the same hard bounds.
    for m in ("bf16x3", "f16c8"):
        assert logit_err[m, "f32"] <= 1e-3 * logit_scale["f32"], m
        assert differ[m, "f32"] == 0 and moved[m, "f32"] == 0, m
        assert differ[m, "bf16"] <= 2, m
