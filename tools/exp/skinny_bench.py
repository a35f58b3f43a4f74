"""Decode-step projections at B = 124 / 32 / 1: vtgb_gemm_skinny against F.linear (hipBLASLt).  Every variant is captured into a hipGraph of
16 calls over 8 rotating weight copies (nothing stays cached; no host launch overhead, as in the decode loop) and the replay is timed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from videotgb_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
CALLS = 16


def graph_us(fn):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for i in range(8):
            fn(i)
    torch.cuda.current_stream().wait_stream(s)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for i in range(CALLS):
            fn(i % 8)
    for _ in range(3):
        gr.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (10 * CALLS) * 1e3


Ms = [int(a) for a in sys.argv[1:]] or [124, 32, 1]
for M in Ms:
    tot_a = tot_b = 0.0
    for name, N, K in (("qkv", 12288, 4096), ("o", 4096, 4096), ("gate_up", 22016, 4096), ("down", 4096, 11008), ("lm_head", 32000, 4096)):
        x = torch.randn(M, K, generator=g, device=dev).bfloat16()
        ws = [(torch.randn(N, K, generator=g, device=dev) * 0.02).bfloat16() for _ in range(8)]
        wt = [ops.SkinnyWeight(w_) for w_ in ws]
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        wsb = torch.empty(max(1, max(ops.gemm_skinny_workspace_bytes(M, N, K, S) for S in (0, 1, 2, 3, 4, 6, 8))), dtype=torch.uint8, device=dev)
        a = graph_us(lambda i: ops.gemm_skinny(x, wt[i], out=out, workspace=wsb))
        b = graph_us(lambda i: F.linear(x, ws[i]))
        rep = 1 if name == "lm_head" else 32
        tot_a += a * rep; tot_b += b * rep
        print(f"M={M:3d} {name:8s} N={N:5d} K={K:5d}: skinny {a:7.1f} us ({N * K * 2 / a / 1e6:5.2f} TB/s)   F.linear {b:7.1f} us ({N * K * 2 / b / 1e6:5.2f} TB/s)")
        sw = [f"S={S_}: {graph_us(lambda i: ops.gemm_skinny(x, wt[i], out=out, workspace=wsb, n_splits=S_)):.1f}" for S_ in (1, 2, 3, 4, 6, 8)]
        print("      splits sweep (us): " + ", ".join(sw) + f"   row-major S=0: {graph_us(lambda i: ops.gemm_skinny(x, ws[i], out=out, workspace=wsb)):.1f}", flush=True)
    print(f"M={M}: 32 layers + lm_head: skinny {tot_a / 1e3:.2f} ms, F.linear {tot_b / 1e3:.2f} ms per token")
