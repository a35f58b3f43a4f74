"""-m gpu: every HIP kernel against an independent reference (torch fp32 on the device for
the floating-point blocks, the CPU oracle for the integer stages), through the C ABI."""
import math

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from videotgb_amd import _lib
    _lib.lib()           # fail loudly if the extension is missing
    return torch.device("cuda:0")


def rel_rms(a, b):
    a, b = a.double(), b.double()
    return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp_min(1e-30))


def report(name, got, ref):
    err = (got.double() - ref.double()).abs().max().item()
    print(f"[{name}] max|diff|={err:.3e} rel_rms={rel_rms(got, ref):.3e} max|ref|={ref.abs().max().item():.3e}")
    return err


# ----------------------------------------------------------------------------- integer stages
def test_span_select_bit_exact(dev):
    from oracle import vtgb_oracle as O
    from videotgb_amd import ops
    g = torch.Generator().manual_seed(3)
    for B, L in ((1, 96), (3, 256), (2, 5), (1, 1)):
        logits = torch.randn(B, L, 2, generator=g)
        noise = O.gumbel_noise((2, 2 * B, L), g)
        exp = O.span_select(logits, noise)
        got = ops.span_select(logits.to(dev), noise.to(dev)).cpu()
        assert torch.equal(got, exp), (B, L)
    # ties -> lowest index; a single spike wins
    z = torch.zeros(1, 70, 2)
    n = torch.zeros(2, 2, 70)
    n[1, 1, 69] = 1.0
    assert ops.span_select(z.to(dev), n.to(dev)).cpu().tolist() == [[0, 0], [0, 69]]


def test_span_to_frames_golden_table_bit_exact(dev):
    from videotgb_amd import ops
    rows = load_golden("integer_tables")["span_map"].numpy()
    # group rows by (variant, V, N, nframe): one launch per group, clips as the batch
    keys = {}
    for r in rows:
        keys.setdefault(tuple(int(x) for x in r[:4]), []).append(r)
    for (variant, V, N, nframe), rs in keys.items():
        rs = np.stack(rs)
        B = len(rs)
        sel = torch.tensor(np.stack([np.concatenate([rs[:, 4], rs[:, 5]]), np.concatenate([rs[:, 6], rs[:, 7]])]), dtype=torch.int64)
        got = ops.span_to_frames(sel.to(dev), V, N, nframe, "AB"[variant]).cpu().numpy()
        np.testing.assert_array_equal(got, rs[:, 8:8 + nframe], err_msg=str((variant, V, N, nframe)))


def test_span_to_frames_per_clip_lengths_and_oracle(dev):
    from oracle import vtgb_oracle as O
    from videotgb_amd import ops
    rng = np.random.default_rng(9)
    for variant in "AB":
        V = rng.integers(2, 300, size=64)
        sel = torch.tensor(rng.integers(0, 320, size=(2, 128)), dtype=torch.int64)
        got = ops.span_to_frames(sel.to(dev), torch.tensor(V), 32, 8, variant).cpu()
        for j in range(64):
            exp = O.span_to_frames(sel[:, j].tolist(), sel[:, 64 + j].tolist(), int(V[j]), 32, 8, variant)
            assert got[j].tolist() == exp, (variant, j, int(V[j]), sel[:, j].tolist(), sel[:, 64 + j].tolist())


def test_gather_frames_bit_exact(dev):
    from oracle import vtgb_oracle as O
    from videotgb_amd import ops
    g = torch.Generator().manual_seed(4)
    pix = torch.randn(2, 32, 3, 224, 224, generator=g)
    idx = torch.randint(0, 32, (2, 8), generator=g)
    assert torch.equal(ops.gather_frames(pix.to(dev), idx.to(dev)).cpu(), O.gather_frames(pix, idx))
    small = torch.randn(1, 4, 3, 2, 2, generator=g)
    idx = torch.tensor([[3, 3, 0]])
    assert torch.equal(ops.gather_frames(small.to(dev), idx.to(dev)).cpu(), small[:, [3, 3, 0]])


# ----------------------------------------------------------------------------- GEMM
GEMM_SHAPES = [(257, 384, 128), (2056, 4224, 1408), (300, 1408, 6144), (32, 768, 768), (17, 48, 96), (98, 2, 64),
               (513, 132, 72), (128, 128, 64)]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_gemm_epilogues(dev, M, N, K, dtype):
    from videotgb_amd import _lib as L, ops
    g = torch.Generator().manual_seed(M * 7 + N)
    td = torch.bfloat16 if dtype == "bf16" else torch.float32
    A = (torch.randn(M, K, generator=g)).to(dev).to(td)
    W = (torch.randn(N, K, generator=g) * 0.05).to(dev).to(td)
    bias = torch.randn(N, generator=g).to(dev)
    resid = torch.randn(M, N, generator=g).to(dev)
    ref = A.float() @ W.float().t() + bias
    tol = 2e-2 if dtype == "bf16" else 1e-4
    scale = ref.abs().max().item()
    out = ops.gemm(A, W, bias, L.EPI_STORE_F32)
    assert report(f"gemm {dtype} {M}x{N}x{K} store_f32", out, ref) <= (1e-4 if dtype == "f32" else 2e-3) * scale
    out = ops.gemm(A, W, bias, L.EPI_RESID_F32, resid)
    assert report("resid", out, ref + resid) <= (1e-4 if dtype == "f32" else 2e-3) * scale
    out = ops.gemm(A, W, bias, L.EPI_STORE)
    assert out.dtype == td and report("store", out.float(), ref) <= tol * scale
    out = ops.gemm(A, W, bias, L.EPI_GELU)
    assert report("gelu", out.float(), torch.nn.functional.gelu(ref)) <= tol * scale
    out = ops.gemm(A, W, None, L.EPI_STORE_F32)
    assert report("nobias", out, ref - bias) <= (1e-4 if dtype == "f32" else 2e-3) * scale


def test_gemm_bf16_exact_integers_catch_layout_bugs(dev):
    """Small-integer operands make the MFMA result exact: any fragment/row/column mix-up shows."""
    from videotgb_amd import _lib as L, ops
    g = torch.Generator().manual_seed(1)
    M, N, K = 200, 264, 192
    A = torch.randint(-3, 4, (M, K), generator=g).float()
    W = torch.randint(-3, 4, (N, K), generator=g).float()     # asymmetric
    out = ops.gemm(A.to(dev).bfloat16(), W.to(dev).bfloat16(), None, L.EPI_STORE_F32).cpu()
    assert torch.equal(out, A @ W.t())


# ----------------------------------------------------------------------------- attention
def ref_attention(q, k, v, heads, scale, mask=None, rope_q=None, rope_k=None):
    from oracle import vtgb_oracle as O
    B, Sq, D = q.shape
    hd = D // heads
    qh = q.float().view(B, Sq, heads, hd).permute(0, 2, 1, 3)
    kh = k.float().view(B, -1, heads, hd).permute(0, 2, 1, 3)
    vh = v.float().view(B, -1, heads, hd).permute(0, 2, 1, 3)
    if rope_q is not None:
        qh = O.apply_rope(rope_q[:Sq].float()[None, None], qh)
    if rope_k is not None:
        kh = O.apply_rope(rope_k[: kh.shape[2]].float()[None, None], kh)
    s = qh @ kh.transpose(-1, -2) * scale
    if mask is not None:
        s = s + mask[:, None, None, :]
    return (torch.softmax(s, -1) @ vh).permute(0, 2, 1, 3).reshape(B, Sq, D)


ATTN_CASES = [  # B, heads, hd, Sq, Skv, mask, rope
    (2, 16, 88, 257, 257, False, False),   # ViT
    (3, 12, 64, 44, 44, True, False),      # Q-Former self
    (3, 12, 64, 32, 257, False, False),    # Q-Former cross
    (2, 12, 64, 98, 98, True, True),       # TGB self, T = 96
    (2, 12, 64, 98, 14, True, True),       # TGB cross
    (1, 12, 64, 258, 258, True, True),     # TGB self, T = 256
    (1, 12, 64, 402, 402, False, True),    # 512-key instantiation
    (2, 2, 32, 17, 17, False, False),      # tiny ViT
    (2, 2, 24, 38, 17, True, False),       # tiny Q-Former cross
]


@pytest.mark.parametrize("B,H,hd,Sq,Skv,use_mask,use_rope", ATTN_CASES)
@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_attention(dev, B, H, hd, Sq, Skv, use_mask, use_rope, dtype):
    from videotgb_amd import ops
    from videotgb_amd.synth import rope_table
    g = torch.Generator().manual_seed(Sq * 13 + Skv)
    td = torch.bfloat16 if dtype == "bf16" else torch.float32
    D = H * hd
    # q/k/v as column slices of one fused buffer (the layout the stages use)
    qkv = torch.randn(B, max(Sq, Skv), 3 * D, generator=g).to(dev).to(td)
    q, k, v = qkv[:, :Sq, :D], qkv[:, :Skv, D:2 * D], qkv[:, :Skv, 2 * D:]
    mask = None
    if use_mask:
        m = (torch.rand(B, Skv, generator=g) > 0.3).float()
        m[:, 0] = 1
        mask = ((1 - m) * -10000.0).to(dev)
    tab = rope_table(512, hd).to(dev) if use_rope else None
    scale = hd ** -0.5
    out = ops.attention(q, k, v, H, scale, mask, tab, tab)
    ref = ref_attention(q, k, v, H, scale, mask, tab, tab)
    err = report(f"attn {dtype} B{B} H{H} hd{hd} {Sq}x{Skv}", out.float(), ref)
    assert err <= (3e-2 if dtype == "bf16" else 2e-5) * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("B,H,hd,Sq,Skv", [(3, 4, 128, 52, 52), (2, 4, 128, 130, 130), (2, 3, 128, 7, 40), (2, 12, 64, 98, 98), (1, 2, 96, 33, 64)])
@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_attention_causal(dev, B, H, hd, Sq, Skv, dtype):
    """The LLM prefill's attention (transformers LlamaAttention under language_model.generate, eval/utils/model.py:223-233):
    head_dim 128, query q attends keys <= q + (Skv - Sq)."""
    from videotgb_amd import ops
    g = torch.Generator().manual_seed(Sq * 7 + Skv + hd)
    td = torch.bfloat16 if dtype == "bf16" else torch.float32
    D = H * hd
    qkv = torch.randn(B, Skv, 3 * D, generator=g).to(dev).to(td)
    q, k, v = qkv[:, :Sq, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:]
    out = ops.attention(q, k, v, H, hd ** -0.5, causal=True)
    qh, kh, vh = (t.float().reshape(B, -1, H, hd).transpose(1, 2) for t in (q, k, v))
    allowed = torch.arange(Skv, device=dev)[None, :] <= torch.arange(Sq, device=dev)[:, None] + (Skv - Sq)
    sc = (qh @ kh.transpose(-1, -2)) * hd ** -0.5
    ref = (torch.softmax(sc.masked_fill(~allowed, float("-inf")), -1) @ vh).transpose(1, 2).reshape(B, Sq, D)
    err = report(f"causal attn {dtype} hd{hd} {Sq}x{Skv}", out.float(), ref)
    assert err <= (3e-2 if dtype == "bf16" else 2e-5) * max(1.0, ref.abs().max().item())


def test_attention_bf16_finfo_min_mask(dev):
    """HF invert_attention_mask uses finfo.min for cross-attention pads (xropebert.py:1127)."""
    from videotgb_amd import ops
    g = torch.Generator().manual_seed(0)
    q = torch.randn(1, 20, 128, generator=g).to(dev).bfloat16()
    kv = torch.randn(1, 9, 256, generator=g).to(dev).bfloat16()
    mask = torch.zeros(1, 9)
    mask[0, 5:] = torch.finfo(torch.float32).min
    out = ops.attention(q, kv[:, :, :128], kv[:, :, 128:], 2, 0.125, mask.to(dev))
    ref = ref_attention(q, kv[:, :5, :128], kv[:, :5, 128:], 2, 0.125)
    assert report("attn finfo.min", out.float(), ref) < 3e-2


# ----------------------------------------------------------------------------- LayerNorm
@pytest.mark.parametrize("M,D", [(2056, 1408), (264, 768), (5, 48), (1, 64), (37, 2048)])
def test_layernorm(dev, M, D):
    from videotgb_amd import ops
    g = torch.Generator().manual_seed(D)
    x = (torch.randn(M, D, generator=g) * 3 + 1).to(dev)
    w, b = torch.randn(D, generator=g).to(dev), torch.randn(D, generator=g).to(dev)
    for eps in (1e-12, 1e-5):
        ref = torch.nn.functional.layer_norm(x, (D,), w, b, eps)
        assert report(f"ln {M}x{D}", ops.layernorm(x, w, b, eps), ref) < 2e-5
        assert report("ln bf16", ops.layernorm(x, w, b, eps, torch.bfloat16).float(), ref) < 5e-2


def test_unsupported_shapes_fail_loudly(dev):
    from videotgb_amd import ops
    with pytest.raises(NotImplementedError):
        ops.layernorm(torch.zeros(2, 4096, device=dev), torch.ones(4096, device=dev), torch.zeros(4096, device=dev), 1e-5)
    with pytest.raises(NotImplementedError):
        q = torch.zeros(1, 600, 64, device=dev, dtype=torch.bfloat16)
        ops.attention(q, q, q, 1, 1.0)
    with pytest.raises(ValueError):
        ops.tgb_forward(None, None, None, None, None, "bogus")
