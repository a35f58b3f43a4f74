"""Sweep a device-side experiment selector of the large GEMM kernel on the ViT-g layer shapes (debug-hook build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from videotgb_amd import _lib as L, ops
dev = torch.device("cuda:0"); lib = L.lib()
M = 257 * 248
g = torch.Generator(device=dev).manual_seed(0)
vals = [int(x) for x in (sys.argv[1:] or ["0", "1", "2", "3", "0"])]
for name, n, k, epi in (("qkv", 4224, 1408, L.EPI_STORE), ("proj", 1408, 1408, L.EPI_RESID_F32), ("fc1", 6144, 1408, L.EPI_GELU), ("fc2", 1408, 6144, L.EPI_RESID_F32)):
    A = torch.randn(M, k, generator=g, device=dev).bfloat16(); W = (torch.randn(n, k, generator=g, device=dev) * 0.05).bfloat16()
    bias = torch.randn(n, generator=g, device=dev); resid = torch.randn(M, n, generator=g, device=dev) if epi == L.EPI_RESID_F32 else None
    for _ in range(5): ops.gemm(A, W, bias, epi, resid)
    row = []
    for v in vals:
        lib.vtgb_debug_set_exp(v)
        for _ in range(3): ops.gemm(A, W, bias, epi, resid)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.gemm(A, W, bias, epi, resid)
        e1.record(); torch.cuda.synchronize()
        row.append(f"exp={v}: {2.0 * M * n * k / (e0.elapsed_time(e1) / 20) / 1e9:6.0f}")
    print(name, " | ".join(row))
lib.vtgb_debug_set_exp(0)
