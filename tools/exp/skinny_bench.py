"""Decode-step projections at B = 124 and B = 1: vtgb_gemm_skinny against F.linear (hipBLASLt), us per call and TB/s of weights."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from videotgb_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)


def t_us(fn, n=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for M in (124, 32, 1):
    tot_a = tot_b = 0.0
    for name, N, K in (("qkv", 12288, 4096), ("o", 4096, 4096), ("gate_up", 22016, 4096), ("down", 4096, 11008), ("lm_head", 32000, 4096)):
        x = torch.randn(M, K, generator=g, device=dev).bfloat16()
        # a fresh weight per layer in the real loop: rotate over 8 copies so that nothing stays cached
        ws = [(torch.randn(N, K, generator=g, device=dev) * 0.02).bfloat16() for _ in range(8)]
        wt = [ops.SkinnyWeight(w_) for w_ in ws]
        i = [0]
        def own():
            i[0] = (i[0] + 1) % 8; return ops.gemm_skinny(x, wt[i[0]])
        def lib():
            i[0] = (i[0] + 1) % 8; return F.linear(x, ws[i[0]])
        a, b = t_us(own), t_us(lib)
        rep = 1 if name == "lm_head" else 32
        tot_a += a * rep; tot_b += b * rep
        print(f"M={M:3d} {name:8s} N={N:5d} K={K:5d}: skinny {a:7.1f} us ({N * K * 2 / a / 1e6:5.2f} TB/s)   F.linear {b:7.1f} us ({N * K * 2 / b / 1e6:5.2f} TB/s)")
    print(f"M={M}: 32 layers + lm_head: skinny {tot_a / 1e3:.2f} ms, F.linear {tot_b / 1e3:.2f} ms per token")
