"""vtgb_gemm vs hipBLASLt (F.linear) on the LLM prefill shapes, with the natural and a padded leading dimension."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from videotgb_amd import ops

dev = "cuda:0"


def bench(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


M = int(sys.argv[1]) if len(sys.argv) > 1 else 6448
for (N, K) in ((12288, 4096), (4096, 4096), (22016, 4096), (4096, 11008), (4224, 1408), (6144, 1408), (1408, 6144)):
    for pad in (0, 64):
        A = (torch.randn(M, K + pad, device=dev) * 0.5).bfloat16()[:, :K]
        W = (torch.randn(N, K + pad, device=dev) * 0.02).bfloat16()[:, :K]
        t = bench(lambda: ops.gemm(A, W))
        fl = 2.0 * M * N * K
        line = f"M={M} N={N} K={K} ld={K + pad}: own {t * 1e3:.0f} us {fl / t / 1e9:.0f} TF/s"
        if pad == 0:
            t2 = bench(lambda: F.linear(A, W))
            line += f" | hipBLASLt {t2 * 1e3:.0f} us {fl / t2 / 1e9:.0f} TF/s"
        print(line, flush=True)
