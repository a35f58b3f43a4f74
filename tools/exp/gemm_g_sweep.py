"""Sweep the XCD super-tile height G (m-tiles per super-tile) of the large GEMM kernel on the ViT-g layer shapes (debug-hook build)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from videotgb_amd import _lib as L, ops
dev = torch.device("cuda:0"); lib = L.lib()
M = 257 * 248
g = torch.Generator(device=dev).manual_seed(0)
for name, n, k, epi in (("qkv", 4224, 1408, L.EPI_STORE), ("proj", 1408, 1408, L.EPI_RESID_F32), ("fc1", 6144, 1408, L.EPI_GELU), ("fc2", 1408, 6144, L.EPI_RESID_F32)):
    A = torch.randn(M, k, generator=g, device=dev).bfloat16(); W = (torch.randn(n, k, generator=g, device=dev) * 0.05).bfloat16()
    bias = torch.randn(n, generator=g, device=dev); resid = torch.randn(M, n, generator=g, device=dev) if epi == L.EPI_RESID_F32 else None
    row = []
    for G in (0, 1, 2, 4, 8, 16, 32):
        lib.vtgb_debug_set_gemm_large_variant(G)
        for _ in range(3): ops.gemm(A, W, bias, epi, resid)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.gemm(A, W, bias, epi, resid)
        e1.record(); torch.cuda.synchronize()
        row.append(f"G={G}: {2.0 * M * n * k / (e0.elapsed_time(e1) / 10) / 1e9:6.0f}")
    print(name, " | ".join(row))
lib.vtgb_debug_set_gemm_large_variant(0)
