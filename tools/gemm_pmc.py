#!/usr/bin/env python3
"""Few launches of the large GEMM kernel on the four ViT-g layer shapes (256 frames) for PMC collection."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from videotgb_amd import _lib as L, ops
dev = torch.device("cuda:0"); L.lib()
M = 257 * 248   # 31 clips x 8 frames
g = torch.Generator(device=dev).manual_seed(0)
for name, n, k, epi in (("qkv", 4224, 1408, L.EPI_STORE), ("proj", 1408, 1408, L.EPI_RESID_F32), ("fc1", 6144, 1408, L.EPI_GELU), ("fc2", 1408, 6144, L.EPI_RESID_F32)):
    A = torch.randn(M, k, generator=g, device=dev).bfloat16(); W = (torch.randn(n, k, generator=g, device=dev) * 0.05).bfloat16()
    bias = torch.randn(n, generator=g, device=dev); resid = torch.randn(M, n, generator=g, device=dev) if epi == L.EPI_RESID_F32 else None
    for _ in range(3):
        ops.gemm(A, W, bias, epi, resid)
    torch.cuda.synchronize()
    print(name, "algorithmic bytes", (M * k + n * k) * 2 + M * n * (8 if resid is not None else 2), "flops", 2.0 * M * n * k)
