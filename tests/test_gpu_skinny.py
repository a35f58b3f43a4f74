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
