"""-m gpu: RAFT in libvtgb.so (encoders, all-pairs correlation + pyramid, refinement loop, convex upsample; row a2 / f1)
against the vectors recorded from the reference RAFT and against the fp32 oracle, in both arithmetic modes.

Tolerances (relative RMS error of the compared tensor):
  VTGB_F32  (the reference's arithmetic, fp32 FMAs): flow after 5 / 20 iterations <= 1e-4 (observed ~1e-6: summation
            order only, through 20 recurrent iterations); encoders <= 1e-5; correlation levels <= 1e-5 of the level's max.
  VTGB_BF16 (bf16 MFMA convolutions, half-precision correlation -- a mode the reference does not have): final flow
            <= 1e-2 (observed 2.6e-3 ... 3.6e-3), one iteration <= 2e-2, encoders <= 2e-2, correlation levels within
            half-precision rounding of inputs and outputs."""
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


def rel_rms(a, b):
    return float(((a.double() - b.double()).pow(2).mean().sqrt()) / b.double().pow(2).mean().sqrt())


def make(dev, tiny_sd, dtype):
    from videotgb_amd import models
    sd = {k[len("of_extractor."):]: v for k, v in tiny_sd["instructblip"][1].items() if k.startswith("of_extractor.")}
    r = models.Raft(dtype)
    r.load_state_dict(sd, strict=True)
    return r.to(dev)


FLOW_TOL = {"f32": 1e-4, "bf16": 1e-2, "bf16x3": 5e-4, "f16c8": 5e-4}      # (f16c8, round 6: held to the bf16x3 mode's bounds everywhere)
ENC_TOL = {"f32": 1e-5, "bf16": 2e-2, "bf16x3": 2e-4, "f16c8": 2e-4}


@pytest.mark.parametrize("dtype", ["f32", "bf16", "bf16x3", "f16c8"])
@pytest.mark.parametrize("iters", [5, 20])
def test_raft_pairs_vs_reference(dev, tiny_sd, dtype, iters):
    """of_extractor(image1, image2) -- the reference-shaped entry (xraft.py:102) -- vs the reference's flows."""
    g = load_golden("tiny_raft")
    fr = deq(g, "frames_q8").to(dev)
    ref = g[f"flow_iters{iters}"]
    got = make(dev, tiny_sd, dtype)(fr[:-1], fr[1:], iters=iters).cpu()
    e = rel_rms(got, ref)
    print(f"[raft {dtype} iters={iters}] rel_rms={e:.3e} max|diff|={(got - ref).abs().max():.3e} max|ref|={ref.abs().max():.3e}")
    assert e <= FLOW_TOL[dtype]


@pytest.mark.parametrize("dtype", ["f32", "bf16", "bf16x3", "f16c8"])
def test_raft_clip_path_vs_reference(dev, tiny_sd, dtype):
    """forward_clips (what LSTP.flow uses): fnet once per distinct frame; same flows as the pair entry."""
    g = load_golden("tiny_raft")
    fr = deq(g, "frames_q8").to(dev)
    r = make(dev, tiny_sd, dtype)
    flow = r.forward_clips(fr[None], iters=20)[0].cpu()          # [2, 2, 128, 128]
    e = rel_rms(flow, g["flow_iters20"])
    print(f"[raft clips {dtype}] rel_rms={e:.3e}")
    assert e <= FLOW_TOL[dtype]
    if dtype == "f32":   # encoding each frame once is the same arithmetic as encoding cat(image1, image2)
        pair = r(fr[:-1], fr[1:], iters=20).cpu()
        assert (pair - flow).abs().max() <= 1e-5 * flow.abs().max()


@pytest.mark.parametrize("dtype", ["f32", "bf16", "bf16x3", "f16c8"])
def test_raft_single_iteration_and_flow_init(dev, tiny_sd, dtype):
    """One iteration isolates the kernels from the recurrence; flow_init = coords1 - coords0 offset (xraft.py:131-132)."""
    from oracle import vtgb_oracle as O
    g = load_golden("tiny_raft")
    fr = deq(g, "frames_q8")
    sd = tiny_sd["instructblip"][1]
    r = make(dev, tiny_sd, dtype)
    ref = O.raft_forward(sd, "of_extractor.", fr[:-1], fr[1:], iters=1)
    got = r(fr[:-1].to(dev), fr[1:].to(dev), iters=1).cpu()
    e = rel_rms(got, ref)
    print(f"[raft {dtype} 1 iteration] rel_rms={e:.3e} max|ref|={ref.abs().max():.3e}")
    assert e <= {"f32": 1e-5, "bf16": 2e-2, "bf16x3": 5e-4, "f16c8": 5e-4}[dtype]
    fi = torch.randn(2, 2, 16, 16, generator=torch.Generator().manual_seed(3)) * 0.5
    ref = O.raft_forward(sd, "of_extractor.", fr[:-1], fr[1:], iters=2, flow_init=fi)
    got = r(fr[:-1].to(dev), fr[1:].to(dev), iters=2, flow_init=fi.to(dev)).cpu()
    e = rel_rms(got, ref)
    print(f"[raft {dtype} flow_init] rel_rms={e:.3e}")
    assert e <= {"f32": 1e-5, "bf16": 2e-2, "bf16x3": 5e-4, "f16c8": 5e-4}[dtype]
    flows = r(fr[:-1].to(dev), fr[1:].to(dev), iters=2, test_mode=False)      # (round 4: the all-iteration form exists; test_raft_all_iteration_flows checks it)
    assert isinstance(flows, list) and len(flows) == 2


@pytest.mark.parametrize("dtype", ["f32", "bf16", "bf16x3", "f16c8"])
@pytest.mark.parametrize("net,kind", [("fnet.", "instance"), ("cnet.", "batch")])
def test_raft_encoder_vs_oracle(dev, tiny_sd, net, kind, dtype):
    """BasicEncoder in HIP vs the fp32 oracle."""
    from oracle import vtgb_oracle as O
    from videotgb_amd import ops
    sd = tiny_sd["instructblip"][1]
    g = load_golden("tiny_raft")
    fr = deq(g, "frames_q8")                                   # [3, 3, 128, 128]
    ref = O.raft_encoder(sd, "of_extractor." + net, 2 * (fr / 255.0) - 1.0, kind)       # [3, 256, 16, 16]
    rsd = {k[len("of_extractor."):]: v.to(dev) for k, v in sd.items() if k.startswith("of_extractor.")}
    w = ops.RaftEncoderWeights(rsd, net, kind == "batch", ops.raft_dtype_code(dtype))
    out = ops.raft_encoder(w, fr.to(dev)).cpu().view(3, 16, 16, 256).permute(0, 3, 1, 2)
    e = rel_rms(out, ref)
    print(f"[raft encoder {net} {dtype}] rel_rms={e:.3e} max|ref|={ref.abs().max():.3e}")
    assert e <= ENC_TOL[dtype]


@pytest.mark.parametrize("dtype", ["f32", "bf16", "bf16x3", "f16c8"])
@pytest.mark.parametrize("size,n", [(224, 3), (96, 2)])
def test_raft_context_encoder_image_sizes(dev, tiny_sd, size, n, dtype):
    """cnet (BatchNorm folded): ReLU, the skip connection and the cast live in the convolution epilogues (two
    workgroups per CU on the 64-wide tiles, padded 96 -> 128 output rows in stage 2): other sizes than the golden 128."""
    from oracle import vtgb_oracle as O
    from videotgb_amd import ops
    sd = tiny_sd["instructblip"][1]
    fr = torch.randint(0, 256, (n, 3, size, size), generator=torch.Generator().manual_seed(size + 1)).float()
    ref = O.raft_encoder(sd, "of_extractor.cnet.", 2 * (fr / 255.0) - 1.0, "batch")
    rsd = {k[len("of_extractor."):]: v.to(dev) for k, v in sd.items() if k.startswith("of_extractor.")}
    w = ops.RaftEncoderWeights(rsd, "cnet.", True, ops.raft_dtype_code(dtype))
    out = ops.raft_encoder(w, fr.to(dev)).cpu().view(n, size // 8, size // 8, 256).permute(0, 3, 1, 2)
    e = rel_rms(out, ref)
    print(f"[raft cnet {size}x{size} {dtype}] rel_rms={e:.3e}")
    assert e <= {"f32": 1e-5, "bf16": 1e-2, "bf16x3": 2e-4, "f16c8": 2e-4}[dtype]


@pytest.mark.parametrize("dtype", ["f32", "bf16", "bf16x3", "f16c8"])
@pytest.mark.parametrize("size,n", [(224, 5), (224, 1), (224, 2), (224, 3), (160, 1), (160, 2), (192, 1), (64, 3), (96, 2)])
def test_raft_encoder_image_sizes(dev, tiny_sd, size, n, dtype):
    """InstanceNorm moments come from the convolution epilogue (bf16 mode): 224 -> 28x28 = 784-row images straddle the
    256-row GEMM tiles, 64 -> 8x8 images are below the fused path's minimum (separate statistics pass).
    (224, 1..3), (160, 1..2), (192, 1): the LAST tile of a stage holds <= 64 valid rows, so the waves that fold the column
    statistics are row-inactive -- rounds 1-2 dropped that tile from the last image's moments (3-10 % feature error on the last
    image; found in round 3).  Checked per image."""
    from oracle import vtgb_oracle as O
    from videotgb_amd import ops
    sd = tiny_sd["instructblip"][1]
    fr = torch.randint(0, 256, (n, 3, size, size), generator=torch.Generator().manual_seed(size)).float()
    ref = O.raft_encoder(sd, "of_extractor.fnet.", 2 * (fr / 255.0) - 1.0, "instance")
    rsd = {k[len("of_extractor."):]: v.to(dev) for k, v in sd.items() if k.startswith("of_extractor.")}
    w = ops.RaftEncoderWeights(rsd, "fnet.", False, ops.raft_dtype_code(dtype))
    out = ops.raft_encoder(w, fr.to(dev)).cpu().view(n, size // 8, size // 8, 256).permute(0, 3, 1, 2)
    e = rel_rms(out, ref)
    per = [rel_rms(out[i], ref[i]) for i in range(n)]
    print(f"[raft encoder {size}x{size} n={n} {dtype}] rel_rms={e:.3e} per image {['%.2e' % x for x in per]}")
    assert max(per) <= ENC_TOL[dtype]


@pytest.mark.parametrize("enc", ["fnet", "cnet"])
def test_raft_encoder_large_batch_matches_chunks(dev, tiny_sd, enc):
    """conv64.hip / the stem cut images into runs of rows until the persistent grid is balanced; with >= 4 x 256 images per call a
    workgroup owns whole images and STORES the InstanceNorm moments (no memset, no atomics).  1040 images in one call take that
    path; the same images in chunks of 65 take the atomic one: same features up to the bf16 mode's run-to-run floor (a changed
    summation order flips bf16 roundings by 1 ulp, which fnet's InstanceNorms amplify to ~5e-3 rel-RMS: DESIGN.md section 2; cnet has
    no moments and agrees to rounding)."""
    from videotgb_amd import ops
    sd = tiny_sd["instructblip"][1]
    rsd = {k[len("of_extractor."):]: v.to(dev) for k, v in sd.items() if k.startswith("of_extractor.")}
    w = ops.RaftEncoderWeights(rsd, enc + ".", enc == "cnet", ops.dtype_code("bf16"))
    fr = torch.randint(0, 256, (1040, 3, 64, 64), generator=torch.Generator().manual_seed(5)).float().to(dev)
    full = ops.raft_encoder(w, fr, max_images=2048).float()          # one C call: 1040 >= 4 x 256 images
    parts = torch.cat([ops.raft_encoder(w, fr[i:i + 65]).float() for i in range(0, 1040, 65)], dim=0)
    assert torch.isfinite(full).all()
    e = ((full - parts).pow(2).mean().sqrt() / parts.pow(2).mean().sqrt()).item()
    print(f"[raft encoder {enc} 1040 images, one call vs 16 chunks] rel_rms={e:.3e}")
    assert e <= (1.5e-2 if enc == "fnet" else 2e-3)
    # and both against the oracle on the first images (the stored-moments path must be as close as the atomic one)
    from oracle import vtgb_oracle as O
    k = 6
    ref = O.raft_encoder(sd, f"of_extractor.{enc}.", 2 * (fr[:k].cpu() / 255.0) - 1.0, "instance" if enc == "fnet" else "batch")
    shape = lambda t: t[:k].cpu().view(k, 8, 8, 256).permute(0, 3, 1, 2)
    e_full, e_parts = rel_rms(shape(full), ref), rel_rms(shape(parts), ref)
    print(f"[... vs oracle on {k} images] one call {e_full:.3e}, chunks {e_parts:.3e}")
    assert e_full <= ENC_TOL["bf16"] and e_parts <= ENC_TOL["bf16"]


@pytest.mark.parametrize("dtype", ["f32", "bf16", "bf16x3", "f16c8"])
@pytest.mark.parametrize("size,n", [(224, 5), (64, 3), (224, 300)])
def test_raft_encoder_is_bit_reproducible(dev, tiny_sd, dtype, size, n):
    """fnet (InstanceNorm) twice on the same frames: the same bits.  Every producer of the moments stores per-tile / per-run partial sums
    that a second pass adds in a fixed order (r5; rounds 1-4 used atomics).  (224, 5): GEMM epilogue slots + conv64 runs of rows;
    (64, 3): the separate statistics pass (images below 256 coarse pixels); (224, 300): conv64 workgroups that own whole images."""
    from videotgb_amd import ops
    sd = tiny_sd["instructblip"][1]
    fr = torch.randn(n, 3, size, size, generator=torch.Generator().manual_seed(size + n)).to(dev)
    rsd = {k[len("of_extractor."):]: v.to(dev) for k, v in sd.items() if k.startswith("of_extractor.")}
    w = ops.RaftEncoderWeights(rsd, "fnet.", False, ops.raft_dtype_code(dtype))
    outs = [ops.raft_encoder(w, fr).clone() for _ in range(3)]
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def float_frames(kind, n, size, seed):
    """Float-valued RAFT inputs: "randn" = SURVEY 8d's / bench.py's flow frames; "clip" = a moving texture after CLIP normalisation,
    what the eval path hands to RAFT (eval/inference.py:68 -> eval/utils/model.py:79): after 2*(x/255)-1 both are -1 +- 0.02."""
    from videotgb_amd import synth
    if kind == "randn":
        return torch.randn(n, 3, size, size, generator=torch.Generator().manual_seed(seed))
    return synth.clip_normalise(synth.moving_texture_u8(n, size, seed))


# fp32 mode on float-valued frames: the kernel packs the reference's own 2*(x/255)-1 (same roundings); what is left is the
# summation order of the stem convolution, whose ~1e-7 relative noise sits on a -1 +- 0.008 image and is amplified ~128 x by
# InstanceNorm (observed 1.1e-5 on fnet, cnet 1e-7): bound 1e-4 instead of the integer-frame 1e-5
FLOAT_ENC_TOL = {"f32": 1e-4, "bf16": 2e-2, "bf16x3": 5e-4, "f16c8": 5e-4}      # (f16c8: the bf16x3 encoders with layer1 on f16c8 operands)
@pytest.mark.parametrize("dtype", ["f32", "bf16", "bf16x3"])
@pytest.mark.parametrize("weights", ["default", "sensitive"])
@pytest.mark.parametrize("kind,size,n", [("randn", 128, 3), ("clip", 128, 3), ("randn", 224, 2), ("clip", 224, 2)])
def test_raft_encoders_float_valued_frames(dev, tiny_sd, dtype, weights, kind, size, n):
    """Both encoders on FLOAT-VALUED frames, at the encoder level, vs the fp32 oracle.  Round 2 fed the stem bf16(x - 127.5):
    exact for integer 0..255 frames (all the encoder tests had), a 0.5 .. 1 quantisation step on these -- fnet was 7 % off and no
    flow test saw it (insensitive weights).  The bf16 stem now carries x - 127.5 as a hi | lo bf16 pair (raft_enc.hip)."""
    from oracle import vtgb_oracle as O
    from videotgb_amd import ops, synth
    sd = synth.raft_sensitive_state_dict(0) if weights == "sensitive" else tiny_sd["instructblip"][1]
    fr = float_frames(kind, n, size, 40 + size)
    rsd = {k[len("of_extractor."):]: v.to(dev) for k, v in sd.items() if k.startswith("of_extractor.")}
    for net, okind in (("fnet.", "instance"), ("cnet.", "batch")):
        ref = O.raft_encoder(sd, "of_extractor." + net, 2 * (fr / 255.0) - 1.0, okind)
        w = ops.RaftEncoderWeights(rsd, net, okind == "batch", ops.raft_dtype_code(dtype))
        out = ops.raft_encoder(w, fr.to(dev)).cpu().view(n, size // 8, size // 8, 256).permute(0, 3, 1, 2)
        e = rel_rms(out, ref)
        print(f"[raft {net} float frames {kind} {size} {weights} {dtype}] rel_rms={e:.3e} max|ref|={ref.abs().max():.3e}")
        assert e <= FLOAT_ENC_TOL[dtype], (net, e)


SENS_FLOW_TOL = {"f32": 2e-4, "bf16": 2e-2, "bf16x3": 5e-4, "f16c8": 5e-4}


@pytest.mark.parametrize("dtype", ["f32", "bf16", "bf16x3", "f16c8"])
def test_raft_sensitive_weights_vs_reference(dev, dtype):
    """The INPUT-SENSITIVE weight set (synth.raft_sensitive_state_dict: fan-in-scaled, the flow depends on the correlation
    features -- tests/test_oracle.py shows a 7 % fnet error moves it by > 1e-2) against the reference RAFT's own flows and fnet
    feature maps (tests/golden/tiny_raft_sensitive.npz) for float-valued, CLIP-normalised and integer frames."""
    from test_oracle import sensitive_inputs
    from videotgb_amd import models, ops, synth
    g = load_golden("tiny_raft_sensitive")
    sd = {k[len("of_extractor."):]: v for k, v in synth.raft_sensitive_state_dict(0).items()}
    r = models.Raft(dtype)
    r.load_state_dict(sd, strict=True)
    r.to(dev)
    for tag, f in sensitive_inputs(g).items():
        got = r(f[:-1].to(dev), f[1:].to(dev), iters=20).cpu()
        e = rel_rms(got, g["flow_" + tag])
        print(f"[raft sensitive {tag} {dtype}] flow rel_rms={e:.3e} max|ref|={g['flow_' + tag].abs().max():.3e}")
        assert e <= SENS_FLOW_TOL[dtype], (tag, e)
        if tag != "c":
            w = ops.RaftEncoderWeights({k: v.to(dev) for k, v in sd.items()}, "fnet.", False, ops.raft_dtype_code(dtype))
            fm = ops.raft_encoder(w, torch.cat([f[:-1], f[1:]], 0).to(dev)).cpu().view(4, 16, 16, 256).permute(0, 3, 1, 2)[:, ::4]
            ef = rel_rms(fm, g["fmap_" + tag])
            print(f"[raft sensitive {tag} {dtype}] fnet rel_rms vs the reference's feature maps={ef:.3e}")
            assert ef <= FLOAT_ENC_TOL[dtype], (tag, ef)


@pytest.mark.parametrize("dtype", ["f32", "bf16", "bf16x3", "f16c8"])
def test_raft_224_sensitive_weights_vs_reference(dev, dtype):
    """Full frame size (224 x 224 -> 28 x 28 coarse pixels: the geometry the fused GRU half-step, conv64 and the stem kernel run at in
    the bench), input-sensitive weights, 20 iterations, against the REFERENCE's own flow (tests/golden/raft224_sensitive.npz,
    make_golden.py [raft224]: every 4th fine pixel) -- round-3 VERDICT: the full-size check had been HIP vs oracle only."""
    from videotgb_amd import models, synth
    g = load_golden("raft224_sensitive")
    assert torch.equal(synth.moving_texture_u8(3, 224, 7), g["frames_u8"])
    sd = {k[len("of_extractor."):]: v for k, v in synth.raft_sensitive_state_dict(0).items()}
    r = models.Raft(dtype)
    r.load_state_dict(sd, strict=True)
    r.to(dev)
    f = synth.clip_normalise(g["frames_u8"])
    got = r(f[:-1].to(dev), f[1:].to(dev), iters=20).cpu()
    probe, ref = got[:, :, ::4, ::4], g["flow_probe"]
    e, mx = rel_rms(probe, ref), float((probe - ref).abs().max())
    print(f"[raft 224 sensitive {dtype}] flow rel_rms={e:.3e} max|diff|={mx:.3e} max|ref|={float(g['flow_absmax']):.3e}")
    assert e <= SENS_FLOW_TOL[dtype] and mx <= 5 * SENS_FLOW_TOL[dtype] * float(g["flow_absmax"])
    if dtype == "f32":      # the clip path (forward_clips: consecutive frames of one clip) computes the same two pairs
        clip = r.forward_clips(f[None].to(dev), iters=20).cpu()
        assert rel_rms(clip.reshape(got.shape)[:, :, ::4, ::4], ref) <= SENS_FLOW_TOL[dtype]


def test_raft_all_iteration_flows(dev, tiny_sd):
    """RAFT.forward(test_mode=False) (xraft.py:146-156): every iteration's upsampled flow, against the reference's flows after 5 and 20
    iterations (tests/golden/tiny_raft.npz) and the test-mode call."""
    g = load_golden("tiny_raft")
    fr = deq(g, "frames_q8").to(dev)
    r = make(dev, tiny_sd, "f32")
    flows = r(fr[:-1], fr[1:], iters=20, test_mode=False)
    assert isinstance(flows, list) and len(flows) == 20
    assert rel_rms(flows[4].cpu(), g["flow_iters5"]) <= FLOW_TOL["f32"] and rel_rms(flows[19].cpu(), g["flow_iters20"]) <= FLOW_TOL["f32"]
    assert torch.equal(flows[19], r(fr[:-1], fr[1:], iters=20))      # (r5: bit-equal -- the InstanceNorm moments are added in a fixed order)


@pytest.mark.parametrize("n,h8,w8,iters", [(300, 28, 28, 6), (7, 16, 16, 20), (5, 9, 13, 12)])
def test_raft_update_is_bit_reproducible(dev, n, h8, w8, iters):
    """vtgb_raft_update has no atomics: calls on the same inputs agree BIT FOR BIT -- a guard against races in the hand-synchronised kernels
    (the fused GRU half-step's in-place r * h, its LDS-DMA images and register rings; the persistent GEMM's epilogue staging)."""
    from videotgb_amd import ops, synth
    sd = {k[len("of_extractor."):]: v.to(dev) for k, v in synth.synth_state_dict(synth.raft_shapes("of_extractor."), 0).items()}
    w = ops.RaftWeights(sd, "update_block.", ops.BF16)
    g = torch.Generator(device=dev).manual_seed(n)
    cnet = torch.randn(n, h8 * w8, 256, generator=g, device=dev)
    pyr = [torch.randn(n * h8 * w8, 1, max(h8 >> l, 1), max(w8 >> l, 1), generator=g, device=dev).half() for l in range(4)]
    outs = [ops.raft_update(w, None, None, pyr, iters=iters, cnet_nhwc=cnet, hw=(h8, w8)).clone() for _ in range(3)]
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("mode", ["bf16x3", "f16c8"])
@pytest.mark.parametrize("n,h8,w8,iters", [(300, 28, 28, 4), (7, 16, 16, 12), (5, 9, 13, 8)])
def test_raft_update_bf16x3_is_bit_reproducible(dev, n, h8, w8, iters, mode):
    """The bf16x3 refinement loop (pair-store, gate and GRU-update epilogues with hand-counted waits, in-place h update) on the same inputs three
    times: the same bits -- a guard against races; (300, 28, 28) = 918 m-tiles on the persistent grid, (5, 9, 13): tiles that straddle images."""
    from videotgb_amd import ops, synth
    sd = {k[len("of_extractor."):]: v.to(dev) for k, v in synth.raft_sensitive_state_dict(0).items()}
    w = ops.RaftWeights(sd, "update_block.", ops.raft_dtype_code(mode))      # (f16c8: gemm_h8.hip's four-phase k-loop and its pair epilogues)
    g = torch.Generator(device=dev).manual_seed(n + 1)
    cnet = torch.randn(n, h8 * w8, 256, generator=g, device=dev)
    pyr = [torch.randn(n * h8 * w8, 1, max(h8 >> l, 1), max(w8 >> l, 1), generator=g, device=dev) for l in range(4)]
    outs = [ops.raft_update(w, None, None, pyr, iters=iters, cnet_nhwc=cnet, hw=(h8, w8)).clone() for _ in range(3)]
    assert torch.isfinite(outs[0]).all() and outs[0].abs().max() > 0
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("n,h8,w8", [(3, 28, 28), (2, 9, 13)])
def test_raft_update_fp32_pyramid_equals_half_pyramid_at_bf16(dev, n, h8, w8):
    """The bf16 update block reads its correlation pyramid as IEEE half (what vtgb_raft_corr writes at bf16) or as fp32: both instantiations of
    the fused lookup + convc1 kernel on the SAME values (the half pyramid widened to fp32) must agree bit for bit."""
    from videotgb_amd import ops, synth
    sd = {k[len("of_extractor."):]: v.to(dev) for k, v in synth.synth_state_dict(synth.raft_shapes("of_extractor."), 0).items()}
    w = ops.RaftWeights(sd, "update_block.", ops.BF16)
    g = torch.Generator(device=dev).manual_seed(n * 31 + h8)
    cnet = torch.randn(n, h8 * w8, 256, generator=g, device=dev)
    pyr16 = [torch.randn(n * h8 * w8, 1, max(h8 >> l, 1), max(w8 >> l, 1), generator=g, device=dev).half() for l in range(4)]
    pyr32 = [t.float() for t in pyr16]
    a = ops.raft_update(w, None, None, pyr16, iters=4, cnet_nhwc=cnet, hw=(h8, w8)).clone()
    b = ops.raft_update(w, None, None, pyr32, iters=4, cnet_nhwc=cnet, hw=(h8, w8)).clone()
    assert torch.isfinite(a).all() and torch.equal(a, b)


def test_raft_float_valued_frames(dev, tiny_sd):
    """The eval path feeds RAFT CLIP-normalised floats (eval/inference.py:68 -> eval/utils/model.py:79), not 0..255
    integers: after 2*(x/255)-1 the image is -1 +- 0.02.  Flow level, default weights, both modes (the encoder-level and
    sensitive-weight versions are above)."""
    from oracle import vtgb_oracle as O
    sd = tiny_sd["instructblip"][1]
    fr = torch.randn(3, 3, 128, 128, generator=torch.Generator().manual_seed(12))
    ref = O.raft_forward(sd, "of_extractor.", fr[:-1], fr[1:], iters=6)
    for dtype, tol in (("f32", 1e-4), ("bf16", 2e-2), ("bf16x3", 5e-4), ("f16c8", 5e-4)):
        got = make(dev, tiny_sd, dtype)(fr[:-1].to(dev), fr[1:].to(dev), iters=6).cpu()
        e = rel_rms(got, ref)
        print(f"[raft normalised frames {dtype}] rel_rms={e:.3e} max|ref|={ref.abs().max():.3e}")
        assert e <= tol


@pytest.mark.parametrize("dtype", ["f32", "bf16", "bf16x3"])
@pytest.mark.parametrize("h8,w8", [(28, 28), (16, 16), (9, 13), (36, 40), (33, 21)])      # (36 x 40: the 16-row slice of the MFMA modes; 33 x 21: HW % 8 != 0)
def test_corr_pyramid_vs_oracle(dev, h8, w8, dtype):
    """CorrBlock.__init__ (corr.py:12-27, :52-60) in one kernel: all-pairs product / sqrt(dim) + three avg_pool2d, odd sizes
    floored like avg_pool2d; both pair -> image maps (consecutive frames of clips; cat(image1, image2))."""
    from oracle import vtgb_oracle as O
    from videotgb_amd import ops
    g = torch.Generator().manual_seed(h8 * 31 + w8)
    b, t = 2, 3
    fm = torch.randn(b * t, h8 * w8, 256, generator=g) * 1.5
    code = ops.raft_dtype_code(dtype)
    got = ops.raft_corr(fm.to(dev), b * (t - 1), h8, w8, t - 1, t, 0, 1, code)
    nchw = fm.view(b, t, h8, w8, 256).permute(0, 1, 4, 2, 3)
    f1, f2 = nchw[:, :-1].reshape(-1, 256, h8, w8), nchw[:, 1:].reshape(-1, 256, h8, w8)
    ref = O.raft_corr_pyramid(f1, f2)
    tol = 3 * 2.0 ** -11 if dtype == "bf16" else 1e-5
    for l in range(4):
        assert got[l].shape == ref[l].shape and got[l].dtype == (torch.float16 if dtype == "bf16" else torch.float32)
        err = (got[l].float().cpu() - ref[l]).abs().max().item()
        print(f"[corr {dtype} {h8}x{w8}] level {l}: max|diff|={err:.3e} max|ref|={ref[l].abs().max():.3e}")
        assert err <= tol * ref[l].abs().max().item()
    # cat(image1, image2) map: pair n = images n and N + n
    n = 3
    got2 = ops.raft_corr(fm.to(dev), n, h8, w8, n, n, 0, n, code)
    ref2 = O.raft_corr_pyramid(nchw.reshape(-1, 256, h8, w8)[:n], nchw.reshape(-1, 256, h8, w8)[n:2 * n])
    for l in range(4):
        assert (got2[l].float().cpu() - ref2[l]).abs().max().item() <= tol * ref2[l].abs().max().item()
    with pytest.raises(ValueError):
        ops.raft_corr(fm.to(dev), b * t, h8, w8, t, t, 0, 1, code)          # the last pair would read past the feature maps


def test_corr_large_features_do_not_overflow(dev):
    """The 1/sqrt(dim) scale is applied to the fp32 accumulator: raw dot products beyond the half range (65504) still give
    finite half-precision levels (trained RAFT features are not bounded by the synthetic N(0, 0.02) weights)."""
    from videotgb_amd import ops
    fm = torch.full((2, 64, 256), 20.0)                     # raw dot product 256 * 400 = 102400 > 65504; scaled 6400
    got = ops.raft_corr(fm.to(dev), 1, 8, 8, 1, 2, 0, 1, ops.BF16)
    assert torch.isfinite(got[0].float()).all() and abs(got[0].float().mean().item() - 6400.0) < 4.0


@pytest.mark.parametrize("dtype", ["f32", "bf16", "bf16x3", "f16c8"])
def test_raft_partial_last_tile(dev, tiny_sd, dtype):
    """2 frame pairs of 144 x 144: 2 x 324 = 648 coarse pixels, so the tiles of every update-block launch straddle the two images
    and the third one is partial -- fragment-order start maps, the fused flow-head tail, the gated and GRU epilogues and the
    correlation lookup all meet rows beyond M.  (Coarse grids below 16 x 16 are not valid RAFT inputs: the 1 x 1 top pyramid level
    divides by zero in the reference's own sampler.)"""
    from oracle import vtgb_oracle as O
    sd = tiny_sd["instructblip"][1]
    fr = torch.randn(3, 3, 144, 144, generator=torch.Generator().manual_seed(21))
    ref = O.raft_forward(sd, "of_extractor.", fr[:-1], fr[1:], iters=4)
    got = make(dev, tiny_sd, dtype)(fr[:-1].to(dev), fr[1:].to(dev), iters=4).cpu()
    e = rel_rms(got, ref)
    print(f"[raft 144x144 x 2 pairs {dtype}] rel_rms={e:.3e} max|ref|={ref.abs().max():.3e}")
    assert torch.isfinite(got).all() and e <= {"f32": 1e-4, "bf16": 2e-2, "bf16x3": 5e-4, "f16c8": 5e-4}[dtype]


@pytest.mark.parametrize("dtype", ["f32", "bf16", "bf16x3", "f16c8"])
def test_raft_non_square_frames(dev, tiny_sd, dtype):
    """128 x 208 frames (16 x 26 coarse pixels, 416-pixel images: rows of a 256-row tile straddle image lines and images; the vertical GRU half has
    lines of 16, the horizontal one of 26) vs the fp32 oracle, 6 iterations, 3 pairs."""
    from oracle import vtgb_oracle as O
    sd = tiny_sd["instructblip"][1]
    fr = torch.randn(4, 3, 128, 208, generator=torch.Generator().manual_seed(33))
    ref = O.raft_forward(sd, "of_extractor.", fr[:-1], fr[1:], iters=6)
    got = make(dev, tiny_sd, dtype)(fr[:-1].to(dev), fr[1:].to(dev), iters=6).cpu()
    e = rel_rms(got, ref)
    print(f"[raft 128x208 {dtype}] rel_rms={e:.3e} max|ref|={ref.abs().max():.3e}")
    assert tuple(got.shape) == (3, 2, 128, 208) and torch.isfinite(got).all() and e <= {"f32": 1e-4, "bf16": 2e-2, "bf16x3": 5e-4, "f16c8": 5e-4}[dtype]
