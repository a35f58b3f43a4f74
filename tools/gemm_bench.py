#!/usr/bin/env python3
"""GEMM micro-benchmark on the ViT/Q-Former shapes: checks vtgb_gemm against torch and reports TFLOP/s
for the 128x128 register-staged kernel (v1) and the 256x256 LDS-DMA kernel (v2)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
# (the vtgb_debug_* knobs exist only in a library built with VTGB_DEBUG_HOOKS=1 python -m videotgb_amd.build --force)

from videotgb_amd import _lib as L, ops

dev = torch.device("cuda:0")
lib = L.lib()
HOOKS = hasattr(lib, "vtgb_debug_set_gemm_large_min_tiles")
if HOOKS:
    lib.vtgb_debug_set_gemm_large_min_tiles.argtypes = [C.c_int]
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 256
M = 257 * frames
shapes = [("qkv", M, 4224, 1408, L.EPI_STORE), ("proj", M, 1408, 1408, L.EPI_RESID_F32), ("fc1", M, 6144, 1408, L.EPI_GELU),
          ("fc2", M, 1408, 6144, L.EPI_RESID_F32), ("qf_kv", M, 768, 1408, L.EPI_STORE)]
g = torch.Generator(device=dev).manual_seed(0)
tot = {}
for name, m, n, k, epi in shapes:
    A = torch.randn(m, k, generator=g, device=dev).bfloat16()
    W = (torch.randn(n, k, generator=g, device=dev) * 0.05).bfloat16()
    bias = torch.randn(n, generator=g, device=dev)
    resid = torch.randn(m, n, generator=g, device=dev) if epi == L.EPI_RESID_F32 else None
    ref = A[:4096].float() @ W.float().t() + bias
    if epi == L.EPI_GELU:
        ref = torch.nn.functional.gelu(ref)
    if resid is not None:
        ref = ref + resid[:4096]
    for variant, thr, var in ((("v1", 1 << 30, 0), ("L", 0, 0)) if HOOKS else (("L", 0, 0),)):
        if HOOKS:
            lib.vtgb_debug_set_gemm_large_min_tiles(thr)
            lib.vtgb_debug_set_gemm_large_variant(var)
        out = ops.gemm(A, W, bias, epi, resid)
        err = (out[:4096].float() - ref).abs().max().item() / ref.abs().max().item()
        tail = (out[-300:].float() - ((A[-300:].float() @ W.float().t() + bias) if epi != L.EPI_GELU else torch.nn.functional.gelu(A[-300:].float() @ W.float().t() + bias)) - (resid[-300:] if resid is not None else 0)).abs().max().item()
        for _ in range(3):
            ops.gemm(A, W, bias, epi, resid)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 10
        e0.record()
        for _ in range(iters):
            ops.gemm(A, W, bias, epi, resid)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        tf = 2.0 * m * n * k / ms / 1e9
        tot.setdefault(variant, [0.0, 0.0])
        if name != "qf_kv":
            tot[variant][0] += 2.0 * m * n * k
            tot[variant][1] += ms
        print(f"{name:6s} {variant} M={m} N={n} K={k}: {ms:8.3f} ms {tf:7.1f} TF/s  rel_err={err:.2e} tail_abs_err={tail:.2e}")
for v, (fl, ms) in tot.items():
    print(f"ViT layer GEMMs {v}: {fl / ms / 1e9:.1f} TF/s ({ms:.3f} ms per layer at {frames} frames)")
