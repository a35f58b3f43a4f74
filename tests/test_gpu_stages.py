"""-m gpu: each stage of the hot path (C ABI -> HIP kernels) against the CPU oracle on the
same seeded inputs, and against the vectors recorded from the reference (tests/golden).

Tolerances (stated here, used below):
  fp32 mode : |diff| <= 2e-4 * max|ref|   (summation order only)
  bf16 mode : tiny configurations: relative RMS error <= 6e-3 and |diff| <= 8e-3 * max|ref| against the reference's fp32
              vectors (2 x the largest observed: 2.9e-3 / 3.3e-3; bf16 operands, fp32 accumulate / residual / LayerNorm /
              softmax); full size: stated against the reference's own bf16-autocast run, see below
  integer / index outputs: bit-exact."""
import pytest
import torch

from conftest import deq, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from videotgb_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def to_dev(sd, dev):
    return {k: v.to(dev) for k, v in sd.items()}


def check(name, got, ref, dtype):
    got, ref = got.detach().float().cpu(), ref.float()
    scale = ref.abs().max().item()
    err = (got - ref).abs().max().item()
    rms = float(((got - ref).double().pow(2).mean().sqrt()) / ref.double().pow(2).mean().sqrt())
    print(f"[{name} {dtype}] max|diff|={err:.3e} rel_rms={rms:.3e} max|ref|={scale:.3e}")
    if dtype == "f32":
        assert err <= 2e-4 * scale, name
    else:
        assert rms <= 6e-3 and err <= 8e-3 * scale, name


DTYPES = ["f32", "bf16"]


@pytest.mark.parametrize("dtype", DTYPES)
def test_vit_tiny_vs_reference_vectors(dev, tiny_sd, dtype):
    from videotgb_amd import ops
    cfg, sd = tiny_sd["instructblip"]
    g = load_golden("tiny_vit")
    w = ops.VitWeights(to_dev(sd, dev), "model.vision_model.", ops.dtype_code(dtype), cfg.vit.heads, cfg.vit.eps)
    out32, outa = ops.vit_forward(w, deq(g, "pixel_q8").to(dev), True, True)
    check("vit tiny", out32, g["last_hidden_state"], dtype)
    check("vit tiny act", outa, g["last_hidden_state"], dtype)
    with pytest.raises(ValueError, match="pixel_values"):
        ops.vit_forward(w, None)
    if dtype == "bf16":
        #  the folded-LayerNorm table: the unfolded qkv / fc1 weights are not uploaded, the unfolded table gives the same result class, and
        # a PARTIALLY filled folded block is refused instead of silently falling back
        import ctypes as C
        assert w.tensors[6 + 2] is None and w.tensors[6 + 8] is None and w.tensors[6 + 12] is not None
        wu = ops.VitWeights(to_dev(sd, dev), "model.vision_model.", ops.BF16, cfg.vit.heads, cfg.vit.eps, fold_ln=False)
        assert wu.tensors[6 + 2] is not None and wu.tensors[6 + 12] is None
        check("vit tiny unfolded", ops.vit_forward(wu, deq(g, "pixel_q8").to(dev), True, False)[0], g["last_hidden_state"], dtype)
        keep = w.array[6 + 13]
        w.array[6 + 13] = None
        with pytest.raises(ValueError, match="folded-LayerNorm"):
            ops.vit_forward(w, deq(g, "pixel_q8").to(dev))
        w.array[6 + 13] = keep


@pytest.mark.parametrize("dtype", DTYPES)
def test_qformer_tiny_vs_reference_vectors(dev, tiny_sd, dtype):
    from videotgb_amd import ops
    g = load_golden("tiny_qformer")
    cfg, sd = tiny_sd["instructblip"]
    sdd = to_dev(sd, dev)
    code = ops.dtype_code(dtype)
    w = ops.QFormerWeights(sdd, "model.qformer.", code, cfg.qformer.heads)
    assert w.has_text
    q = ops.qformer_forward(w, sdd["model.query_tokens"], g["image_embeds"].to(dev), g["qformer_ids"].to(dev),
                            g["qformer_mask"].to(dev), torch.ones(3, g["image_embeds"].shape[1], dtype=torch.long, device=dev))
    check("qformer instructblip", q, g["seq_instructblip"][:, :32], dtype)
    pw = ops.pack_weight(sdd["model.language_projection.weight"], code)
    pb = sdd["model.language_projection.bias"]
    check("pool mean", ops.pool_project(q, [3], pw, pb, "mean", code), g["prefix_mean"], dtype)
    check("pool concat", ops.pool_project(q, [3], pw, pb, "concat", code), g["prefix_concat"], dtype)
    cfg2, sd2 = tiny_sd["blip2"]
    sdd2 = to_dev(sd2, dev)
    w2 = ops.QFormerWeights(sdd2, "model.qformer.", code, cfg2.qformer.heads)
    assert not w2.has_text
    q2 = ops.qformer_forward(w2, sdd2["model.query_tokens"], g["image_embeds_blip2"].to(dev))
    check("qformer blip2", q2, g["seq_blip2"], dtype)


def test_pool_ragged_widths_and_empty_clip(dev, tiny_sd):
    from oracle import vtgb_oracle as O
    from videotgb_amd import ops
    cfg, sd = tiny_sd["instructblip"]
    sdd = to_dev(sd, dev)
    q = torch.randn(5, 32, cfg.qformer.hidden)
    for dtype in DTYPES:
        code = ops.dtype_code(dtype)
        pw = ops.pack_weight(sdd["model.language_projection.weight"], code)
        out = ops.pool_project(q.to(dev), [2, 0, 3], pw, sdd["model.language_projection.bias"], "mean", code)
        check("pool ragged", out, O.pool_project(sd, "model.language_projection", q, [2, 0, 3], "mean"), dtype)
    with pytest.raises(ValueError, match="INVALID POOL MODE"):
        ops.pool_project(q.to(dev), [5], pw, sdd["model.language_projection.bias"], "max", code)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("mode", ["multi_modal", "fusion", "vision"])
def test_tgb_tiny_vs_reference_vectors(dev, tiny_sd, dtype, mode):
    from videotgb_amd import ops
    cfg, sd = tiny_sd["instructblip"]
    g = load_golden("tiny_tgb")
    w = ops.TgbWeights(to_dev(sd, dev), "temporal_encoder.", ops.dtype_code(dtype), cfg.tgb.heads, cfg.tgb.fusion_layer)
    seq, logits = ops.tgb_forward(w, deq(g, "of_q8").to(dev), g["of_mask"].to(dev), g["text_ids"].to(dev),
                                  g["text_mask"].to(dev), mode)
    check(f"tgb seq {mode}", seq, g[f"seq_{mode}"], dtype)
    check(f"tgb logits {mode}", logits, g[f"logits_{mode}"], dtype)


def test_tgb_invalid_mode(dev, tiny_sd):
    from videotgb_amd import ops
    cfg, sd = tiny_sd["instructblip"]
    w = ops.TgbWeights(to_dev(sd, dev), "temporal_encoder.", ops.F32, cfg.tgb.heads, cfg.tgb.fusion_layer)
    with pytest.raises(ValueError, match="INVALID MODE"):
        ops.tgb_forward(w, torch.zeros(1, 4, 2, 224, 224, device=dev), torch.ones(1, 6, device=dev),
                        torch.ones(1, 3, device=dev), torch.ones(1, 3, device=dev), "bogus")


# ----------------------------------------------------------------------------- full size
# bf16 bounds at full size are stated against BOTH of the reference's own runs (tests/golden/make_golden.py):
#   ref32 = the reference in fp32, ref16 = the reference under torch.autocast(bfloat16) (its Lightning `precision: bf16`).
#   e_ref = relRMS(ref16, ref32) is what the reference's own bf16 mode costs; the HIP bf16 mode must satisfy
#     relRMS(hip16, ref32) <= 1.25 * e_ref     (at least as close to the fp32 truth as the reference's bf16 mode;
#                                               observed 0.55 ... 0.9 x e_ref: fp32 accumulation AND fp32 residual stream)
#     relRMS(hip16, ref16) <= 1.6 * e_ref      (two bf16 roundings of the same computation; observed 1.1 ... 1.3 x e_ref)
#   and max|diff| to ref32 may not exceed 2 x the reference's own bf16 max|diff|.
#   The north-star figure "1e-3 at bf16" is NOT met by the reference's own bf16 mode at these depths
#   (ViT-g: 7.6e-3 relRMS, 2.9e-2 abs on a 3.9 scale), so it cannot be the bound for ours; DESIGN.md section 2 has the table.
def rel_rms(a, b):
    return float(((a.double() - b.double()).pow(2).mean().sqrt()) / b.double().pow(2).mean().sqrt())


def check_bf16_three_way(name, hip16, ref32, ref16):
    hip16, ref32, ref16 = hip16.detach().float().cpu(), ref32.float(), ref16.float()
    e_ref, e32, e16 = rel_rms(ref16, ref32), rel_rms(hip16, ref32), rel_rms(hip16, ref16)
    m_ref, m32 = (ref16 - ref32).abs().max().item(), (hip16 - ref32).abs().max().item()
    print(f"[{name} bf16] relRMS hip16~ref32={e32:.3e} hip16~ref16={e16:.3e} ref16~ref32={e_ref:.3e} | max|diff| hip16~ref32={m32:.3e} "
          f"ref16~ref32={m_ref:.3e} (max|ref|={ref32.abs().max():.3e})")
    assert e32 <= 1.25 * e_ref, name
    assert e16 <= 1.6 * e_ref, name
    assert m32 <= 2.0 * m_ref, name


@pytest.fixture(scope="module")
def full_probes():
    p = load_golden("full_probes")
    p.update(load_golden("full_probes_bf16ref"))
    return p


@pytest.fixture(scope="module")
def vit_g(dev):
    """EVA-ViT-g weight tables (both modes share the fp32 state_dict on the device)."""
    from videotgb_amd.synth import VitCfg, synth_state_dict, vit_shapes
    return to_dev(synth_state_dict(vit_shapes(VitCfg(), ""), 0), dev)


@pytest.mark.parametrize("dtype", DTYPES)
def test_vit_g_full_size_vs_reference_probes(dev, full_probes, vit_g, dtype):
    """EVA-ViT-g (39 layers, 1408-d, 16x88 heads), 2 frames: probe elements recorded from the
    reference's InstructBlipVisionModel with the same seeded weights (fp32 and bf16-autocast runs)."""
    from videotgb_amd import ops
    p = full_probes
    w = ops.VitWeights(vit_g, "", ops.dtype_code(dtype), 16, 1e-6)
    pix = p["vit_pixel_q8"].float() / 48
    out32, _ = ops.vit_forward(w, pix.to(dev))
    assert list(out32.shape) == p["vit_shape"].tolist()
    got = out32.flatten()[p["vit_probe_idx"].to(dev)]
    if dtype == "f32":
        check("vit-g probes", got, p["vit_probe_val"], dtype)
    else:
        check_bf16_three_way("vit-g probes", got, p["vit_probe_val"], p["vit_probe_val_bf16ref"])


@pytest.mark.parametrize("dtype", DTYPES)
def test_qformer_full_size_vs_reference_probes(dev, full_probes, dtype):
    """Q-Former at full size (12 layers, 12 x 64 heads, 1408-wide cross-attention K/V GEMMs, text branch with padding),
    2 frames, on a seeded image-token tensor: probes from the reference's InstructBlipQFormerModel."""
    from videotgb_amd import ops
    from videotgb_amd.synth import QFormerCfg, qformer_shapes, synth_state_dict, synth_tensor
    p = full_probes
    sd = to_dev(synth_state_dict(qformer_shapes(QFormerCfg(), ""), 0), dev)
    w = ops.QFormerWeights(sd, "", ops.dtype_code(dtype), 12)
    img = (p["qf2_image_q8"].float() * float(p["qf2_q8_scale"])).to(dev)
    qtok = synth_tensor("model.query_tokens", (1, 32, 768)).to(dev)
    q = ops.qformer_forward(w, qtok, img, p["qf2_ids"].to(dev), p["qf2_mask"].to(dev))
    got = q.flatten()[p["qf2_probe_idx"].to(dev)]
    if dtype == "f32":
        check("qformer full", got, p["qf2_probe_val"], dtype)
    else:
        check_bf16_three_way("qformer full", got, p["qf2_probe_val"], p["qf2_probe_val_bf16ref"])


def test_vit_g_into_qformer_full_size_chain_fp32(dev, full_probes, vit_g):
    """The two full-size stages chained (our ViT-g output feeds our Q-Former) against the probes the reference recorded
    from its own chain (fixture `qf_probe_*`), fp32 mode."""
    from videotgb_amd import ops
    from videotgb_amd.synth import QFormerCfg, qformer_shapes, synth_state_dict, synth_tensor
    p = full_probes
    vw = ops.VitWeights(vit_g, "", ops.F32, 16, 1e-6)
    img, _ = ops.vit_forward(vw, (p["vit_pixel_q8"].float() / 48).to(dev))
    sd = to_dev(synth_state_dict(qformer_shapes(QFormerCfg(), ""), 0), dev)
    w = ops.QFormerWeights(sd, "", ops.F32, 12)
    q = ops.qformer_forward(w, synth_tensor("model.query_tokens", (1, 32, 768)).to(dev), img, p["qf_ids"].to(dev),
                            torch.ones_like(p["qf_ids"]).to(dev))
    check("vit-g -> qformer chain", q.flatten()[p["qf_probe_idx"].to(dev)], p["qf_probe_val"], "f32")
    assert abs(q.abs().mean().item() - float(p["qf_absmean"])) < 1e-4


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("mode", ["multi_modal", "fusion"])
def test_tgb_full_size_vs_reference(dev, full_probes, dtype, mode):
    """BERT-base TGB (12 layers, 12 x 64 rotary heads, fusion_layer 6), T = 24: logits of the reference's RopeBertModel."""
    from videotgb_amd import ops
    from videotgb_amd.synth import TgbCfg, synth_state_dict, tgb_shapes
    p = full_probes
    sd = to_dev(synth_state_dict(tgb_shapes(TgbCfg(), ""), 0), dev)
    w = ops.TgbWeights(sd, "", ops.dtype_code(dtype), 12, 6)
    of = (p["tgb_of_q8"].float() / 127).to(dev)
    tids = p["tgb_text_ids"].to(dev)
    seq, logits = ops.tgb_forward(w, of, torch.ones(1, 26, dtype=torch.long, device=dev), tids, torch.ones_like(tids), mode)
    if dtype == "f32":
        check(f"tgb full {mode}", logits, p[f"tgb_logits_{mode}"], dtype)
        assert abs(seq.abs().mean().item() - float(p[f"tgb_seq_absmean_{mode}"])) < 1e-4
    else:
        check_bf16_three_way(f"tgb full {mode}", logits, p[f"tgb_logits_{mode}"], p[f"tgb_logits_{mode}_bf16ref"])


# ------------------------------------------------------------------------------------------ f3 preprocessing
@pytest.mark.gpu
def test_preprocess_frames_vs_reference_and_oracle(dev):
    """get_frames on the device: HIP resize / truncate / normalise vs the vectors recorded from the reference's
    functional_video calls (probe elements) and vs the oracle on a full-HD-like clip.  The chain ends in a
    truncation to uint8, so a 1-ulp difference of the fp32 bilinear value can move a pixel by one level
    (1/255/std = 0.0146..0.0150 after normalisation) when it sits on an integer: at most 1e-4 of the
    elements may differ, each by exactly one level; everything else must be bit-identical."""
    from oracle import vtgb_oracle as O
    from videotgb_amd import ops, video
    g = load_golden("preprocess")
    level = 1.0 / 255.0 / min(O.CLIP_STD) * 1.001
    for name in "abc":
        raw = g[f"raw_{name}"]
        got = ops.preprocess_frames(raw.to(dev)).cpu()
        ref = g[f"probe_{name}"]
        d = (got[:, :, ::7, ::5] - ref).abs()
        assert d.max() <= level and (d > 0).float().mean() <= 1e-4, (name, d.max(), (d > 0).float().mean())
        assert torch.allclose(got.double().sum(dim=(2, 3)), g[f"sum_{name}"], rtol=0, atol=1.0)
    raw = torch.randint(0, 256, (40, 180, 320, 3), generator=torch.Generator().manual_seed(3), dtype=torch.uint8)
    frames, flow_frames = video.get_frames(raw.to(dev))
    rf, rff = O.get_frames(raw)
    for a, b in ((frames.cpu(), rf), (flow_frames.cpu(), rff)):
        d = (a - b).abs()
        assert a.shape == b.shape and d.max() <= level and (d > 0).float().mean() <= 1e-4
    with pytest.raises(TypeError):
        ops.preprocess_frames(raw.float().to(dev))


@pytest.mark.gpu
def test_frame_stager_overlapped_uploads_equal_the_direct_path(dev):
    """row f3's decode feed (video.FrameStager): clips written by an external decoder into pinned slots and uploaded on a copy stream give the
    same tensors as preprocessing the clip directly; slots are reused across more clips than there are slots."""
    import numpy as np
    from videotgb_amd import builder_utils, video
    rng = np.random.default_rng(1)
    st = video.FrameStager(dev, max_frames=40, height=90, width=160, slots=2)
    clips = [rng.integers(0, 256, (t, 90, 160, 3), dtype=np.uint8) for t in (40, 17, 33, 40, 9)]
    tickets = [st.stage(clips[0]), st.stage(iter(clips[1]))]
    for i, c in enumerate(clips):
        frames, flow = st.frames(tickets[i])
        rf, rff = video.get_frames(torch.from_numpy(c).to(dev))
        assert torch.equal(frames, rf) and torch.equal(flow, rff) and tuple(flow.shape) == (c.shape[0], 3, 224, 224)
        bf, bff = builder_utils.get_frames(c, device=dev)                      # the reference-shaped entry with a host-side clip
        assert torch.equal(bf, rf) and torch.equal(bff, rff)
        if i + 2 < len(clips):
            tickets.append(st.stage(clips[i + 2]))                             # reuses the slot of clip i
    #  a slot whose clip has not been consumed is not overwritten, and a ticket whose clip is gone is refused
    st2 = video.FrameStager(dev, max_frames=40, height=90, width=160, slots=2)
    ta, tb = st2.stage(clips[0]), st2.stage(clips[1])
    with pytest.raises(RuntimeError, match="still holds a staged clip"):
        st2.stage(clips[2])
    st2.frames(ta)
    tc = st2.stage(clips[2])                                                   # slot of clip 0, now free
    with pytest.raises(RuntimeError, match="stale ticket"):
        st2.frames(ta)
    assert torch.equal(st2.frames(tc)[1], video.get_frames(torch.from_numpy(clips[2]).to(dev))[1]) and st2.frames(tb) is not None
