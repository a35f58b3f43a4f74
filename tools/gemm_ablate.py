#!/usr/bin/env python3
"""Timing-only ablations of the large GEMM kernel (results are wrong when ablate != 0)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
# (the vtgb_debug_* knobs exist only in a library built with VTGB_DEBUG_HOOKS=1 python -m videotgb_amd.build --force)
from videotgb_amd import _lib as L, ops
dev = torch.device("cuda:0"); lib = L.lib()
lib.vtgb_debug_set_gemm_large_min_tiles(0); lib.vtgb_debug_set_gemm_large_variant(0)
g = torch.Generator(device=dev).manual_seed(0)
for name, M, n, k, epi in (("fc2", 257 * 248, 1408, 6144, L.EPI_RESID_F32), ("proj", 257 * 248, 1408, 1408, L.EPI_RESID_F32), ("qkv", 257 * 248, 4224, 1408, L.EPI_STORE),
                           ("raft-gru-shape", 595840, 256, 1920, L.EPI_STORE)):
    A = torch.randn(M, k, generator=g, device=dev).bfloat16(); W = (torch.randn(n, k, generator=g, device=dev) * 0.05).bfloat16()
    bias = torch.randn(n, generator=g, device=dev); resid = torch.randn(M, n, generator=g, device=dev) if epi == L.EPI_RESID_F32 else None
    for ab, label in ((0, "full"), (32, "full, clocked"), (1, "no DMA in loop"), (8, "no barrier/wait (race)"), (16, "no epilogue"), (17, "no epilogue, no DMA"), (25, "no epilogue, no DMA, no barrier")):
        lib.vtgb_debug_set_gemm_ablate(ab)
        for _ in range(3): ops.gemm(A, W, bias, epi, resid)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.gemm(A, W, bias, epi, resid)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        clk = (C.c_ulonglong * 2)(); lib.vtgb_debug_read_clk(clk, 1)
        mhz = f"{clk[0] / clk[1] * 100:6.0f} MHz in the k-loop" if clk[1] else ""
        print(f"{name} ablate={ab:2d} {label:45s}: {ms:7.3f} ms  {2.0*M*n*k/ms/1e9:7.1f} TF/s-equivalent  {mhz}")
lib.vtgb_debug_set_gemm_ablate(0)
