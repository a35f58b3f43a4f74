"""-m gpu: vtgb_gemm_skinny (the decode step's projections: M <= 128 rows against a weight matrix streamed once, equal runs of (tile, k-tile) steps per workgroup,
fragments added in a fixed order) against torch's fp32 matmul of the same bf16 operands, on the Vicuna-7B shapes and on ragged ones."""
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


@pytest.mark.parametrize("M,N,K,S", [(124, 12288, 4096, 0), (124, 4096, 4096, 0), (124, 22016, 4096, 0), (124, 4096, 11008, 0), (124, 32000, 4096, 0),
                                     (1, 4096, 4096, 0), (32, 4096, 11008, 8), (128, 256, 64, 0), (5, 100, 128, 2), (77, 1000, 320, 3), (124, 4096, 4096, 17), (3, 640, 448, 1)])
def test_skinny_matches_fp32_matmul(dev, M, N, K, S):
    from videotgb_amd import ops
    g = torch.Generator(device=dev).manual_seed(M * 7 + N)
    x = torch.randn(M, K, generator=g, device=dev).bfloat16()
    w = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).bfloat16()
    ref = x.float() @ w.float().t()
    out32 = ops.gemm_skinny(x, w, n_splits=S, out_dtype=torch.float32)
    # fp32 accumulation of exact bf16 products: only the summation order differs from torch
    assert (out32 - ref).abs().max().item() <= 2e-5 * ref.abs().max().item() * max(1.0, (K / 4096) ** 0.5)
    out = ops.gemm_skinny(x, w, n_splits=S)
    assert out.dtype == torch.bfloat16 and torch.equal(out, out32.bfloat16())        # one rounding, of the reduced sum
    assert torch.equal(ops.gemm_skinny(x, w, n_splits=S), out)                         # deterministic (no atomics)
    assert torch.equal(ops.gemm_skinny(x, ops.SkinnyWeight(w), n_splits=S), out)       # the tiled weight layout: same products, same order


def test_skinny_strided_operands_and_errors(dev):
    from videotgb_amd import ops
    g = torch.Generator(device=dev).manual_seed(3)
    xb = torch.randn(16, 512, generator=g, device=dev).bfloat16()
    wb = torch.randn(200, 512, generator=g, device=dev).bfloat16()
    x, w = xb[:, :256], wb[:, 128:384]                    # row pitches 512, K = 256
    out = torch.zeros(16, 208, dtype=torch.bfloat16, device=dev)[:, :200]
    ops.gemm_skinny(x, w, out=out)
    ref = (x.float() @ w.float().t()).bfloat16()
    assert torch.equal(out, ref) or (out.float() - ref.float()).abs().max() <= 2 ** -7 * ref.float().abs().max()
    with pytest.raises((ValueError, NotImplementedError, RuntimeError)):
        ops.gemm_skinny(torch.zeros(129, 64, dtype=torch.bfloat16, device=dev), w[:, :64].contiguous())     # M > 128
    with pytest.raises((ValueError, NotImplementedError, RuntimeError)):
        ops.gemm_skinny(x[:, :100].contiguous(), w[:, :100].contiguous())                                  # K not a multiple of 64


def test_skinny_shared_workspace(dev):
    """One workspace serves calls of different shapes back to back (the decode step's five projections); repeated results are
    equal bit for bit."""
    from videotgb_amd import ops
    g = torch.Generator(device=dev).manual_seed(11)
    shapes = [(124, 4096, 4096, 0), (124, 12288, 4096, 2), (124, 4096, 11008, 0), (7, 1000, 320, 3)]
    ops_in = [(torch.randn(M, K, generator=g, device=dev).bfloat16(), (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).bfloat16(), S)
              for M, N, K, S in shapes]
    ws = torch.empty(max(1, max(ops.gemm_skinny_workspace_bytes(M, N, K, S) for M, N, K, S in shapes)), dtype=torch.uint8, device=dev)
    first = [ops.gemm_skinny(x, w, n_splits=S, workspace=ws) for x, w, S in ops_in]
    for _ in range(3):
        for (x, w, S), ref in zip(ops_in, first):
            assert torch.equal(ops.gemm_skinny(x, w, n_splits=S, workspace=ws), ref)
    for (x, w, S), ref in zip(ops_in, first):
        assert (ref.float() - x.float() @ w.float().t()).abs().max().item() <= 2 ** -7 * ref.float().abs().max().item()


@pytest.mark.parametrize("M", [1, 124])
def test_deferred_split_consumers_equal_the_two_launch_form_bit_for_bit(dev, M):
    """r5: the decode step's split-K projections leave their fp32 fragments to the consumer (vtgb_llm_rmsnorm_parts: residual add + RMSNorm;
    vtgb_llm_rope_cache_parts: rotary + cache append), which adds them in split order and rounds once -- exactly what the separate reduce
    launch stored.  Vicuna-7B shapes: o / down -> RMSNorm, qkv -> rotary."""
    import ctypes as C
    from videotgb_amd import _lib as L, ops
    lib = L.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device=dev).manual_seed(M)
    H, nq, hd, tmax = 4096, 32, 128, 64
    for K in (4096, 11008):                                          # o (S = 4) and down (S = 7)
        a = torch.randn(M, K, generator=g, device=dev).bfloat16()
        w = ops.SkinnyWeight((torch.randn(H, K, generator=g, device=dev) * K ** -0.5).bfloat16())
        gam = torch.randn(H, generator=g, device=dev).bfloat16()
        x0 = torch.randn(M, H, generator=g, device=dev).bfloat16()
        delta = ops.gemm_skinny(a, w)
        xa, ha = x0.clone(), torch.empty_like(x0)
        L.check(lib.vtgb_llm_rmsnorm(L.BF16, xa.data_ptr(), delta.data_ptr(), gam.data_ptr(), ha.data_ptr(), M, H, 1e-6, st))
        out, S, ws = ops.gemm_skinny(a, w, defer_reduce=True)
        assert S > 1
        xb, hb = x0.clone(), torch.empty_like(x0)
        L.check(lib.vtgb_llm_rmsnorm_parts(L.BF16, xb.data_ptr(), ws.data_ptr(), S, gam.data_ptr(), hb.data_ptr(), M, H, 1e-6, st))
        assert torch.equal(xa, xb) and torch.equal(ha, hb), K
    a = torch.randn(M, H, generator=g, device=dev).bfloat16()
    w = ops.SkinnyWeight((torch.randn(3 * H, H, generator=g, device=dev) * H ** -0.5).bfloat16())
    cos, sin = torch.randn(tmax, hd, generator=g, device=dev).bfloat16(), torch.randn(tmax, hd, generator=g, device=dev).bfloat16()
    pos = torch.tensor([5], device=dev)
    res = []
    for deferred in (False, True):
        q = torch.zeros(M, nq * hd, dtype=torch.bfloat16, device=dev)
        kc = torch.zeros(M, nq, tmax, hd, dtype=torch.bfloat16, device=dev)
        vc = torch.zeros_like(kc)
        if deferred:
            _, S, ws = ops.gemm_skinny(a, w, defer_reduce=True)
            assert S > 1
            L.check(lib.vtgb_llm_rope_cache_parts(L.BF16, ws.data_ptr(), S, q.data_ptr(), kc.data_ptr(), vc.data_ptr(), cos.data_ptr(), sin.data_ptr(),
                                                  pos.data_ptr(), M, nq, nq, hd, tmax, st))
        else:
            qkv = ops.gemm_skinny(a, w)
            L.check(lib.vtgb_llm_rope_cache(L.BF16, qkv.data_ptr(), q.data_ptr(), kc.data_ptr(), vc.data_ptr(), cos.data_ptr(), sin.data_ptr(),
                                            pos.data_ptr(), M, nq, nq, hd, tmax, st))
        res.append((q, kc, vc))
    for t0, t1 in zip(*res):
        assert torch.equal(t0, t1) and t0.abs().sum() > 0
