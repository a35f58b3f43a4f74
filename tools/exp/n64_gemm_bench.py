"""Is the 64-wide tile bound by its shape or by the convolution's tap re-reads?  Plain GEMM [M, 576] x [64, 576]^T vs the 3x3 64->64 convolution."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from videotgb_amd import ops
dev = "cuda:0"
def bench(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
M = 376 * 12544
for N, K in ((64, 576), (128, 576), (256, 576), (64, 1152)):
    A = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
    W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    t = bench(lambda: ops.gemm(A, W))
    print(f"plain GEMM M={M} N={N} K={K}: {t * 1e3:.0f} us, {2.0 * M * N * K / t / 1e9:.0f} TFLOP/s, A stream {M * K * 2 / t / 1e9:.2f} TB/s", flush=True)
    del A, W
