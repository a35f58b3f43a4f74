"""How long is a tile's prologue (entry -> first operands in LDS) against its whole life?  Debug-hook build; exp 10."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from videotgb_amd import _lib as L, ops
dev = torch.device("cuda:0"); lib = L.lib()
g = torch.Generator(device=dev).manual_seed(0)
clk = (C.c_ulonglong * 2)()
M = 257 * 992
for name, n, k, epi in (("qkv", 4224, 1408, L.EPI_STORE), ("proj", 1408, 1408, L.EPI_RESID_F32), ("fc2", 1408, 6144, L.EPI_RESID_F32)):
    A = torch.randn(M, k, generator=g, device=dev).bfloat16(); W = (torch.randn(n, k, generator=g, device=dev) * 0.05).bfloat16()
    bias = torch.randn(n, generator=g, device=dev); resid = torch.randn(M, n, generator=g, device=dev) if epi == L.EPI_RESID_F32 else None
    for _ in range(3): ops.gemm(A, W, bias, epi, resid)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.gemm(A, W, bias, epi, resid); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    lib.vtgb_debug_read_clk(clk, 1); lib.vtgb_debug_read_stamps((C.c_ulonglong * 8)(), 1); lib.vtgb_debug_set_exp(10)
    ops.gemm(A, W, bias, epi, resid); torch.cuda.synchronize()
    lib.vtgb_debug_read_clk(clk, 1); lib.vtgb_debug_set_exp(0)
    tiles = clk[1]
    st = (C.c_ulonglong * 8)(); lib.vtgb_debug_read_stamps(st, 1)
    ph = [st[i] / max(st[5], 1) * 10 for i in range(5)]
    print(f"{name}: kernel {us:.0f} us, {tiles} tiles = {tiles / 256:.1f} per CU -> {us / (tiles / 256):.1f} us per tile; prologue {clk[0] / tiles * 10:.0f} ns per tile"
          f" = setup {ph[0]:.0f} + acc-init issue {ph[1]:.0f} + DMA issue {ph[2]:.0f} + wait {ph[3]:.0f} + barrier {ph[4]:.0f}")

# the RAFT stage: average prologue over every large-kernel tile of one pass (all convolutions together)
if len(sys.argv) > 1 and sys.argv[1] == "raft":
    from videotgb_amd import models, synth
    r = models.Raft("bf16")
    sd = {k[len("of_extractor."):]: v for k, v in synth.synth_state_dict(synth.raft_shapes("of_extractor."), 0).items()}
    for k in list(sd):
        if ".downsample.1." in k: sd[k] = sd[k.replace(".downsample.1.", ".norm3.")]
    r.load_state_dict(sd, strict=True); r.to(dev)
    frames = torch.randn(31, 96, 3, 224, 224, generator=g, device=dev)
    r.forward_clips(frames); torch.cuda.synchronize()
    import time
    t0 = time.time(); r.forward_clips(frames); torch.cuda.synchronize(); ms = (time.time() - t0) * 1e3
    lib.vtgb_debug_read_clk(clk, 1); lib.vtgb_debug_read_stamps((C.c_ulonglong * 8)(), 1); lib.vtgb_debug_set_exp(10)
    r.forward_clips(frames); torch.cuda.synchronize()
    lib.vtgb_debug_read_clk(clk, 1); lib.vtgb_debug_set_exp(0)
    st = (C.c_ulonglong * 8)(); lib.vtgb_debug_read_stamps(st, 1)
    print("  phases (ns): setup %.0f | acc-init issue %.0f | DMA issue %.0f | wait %.0f | barrier %.0f" % tuple(st[i] / max(st[5], 1) * 10 for i in range(5)))
    lib.vtgb_debug_set_exp(11); r.forward_clips(frames); torch.cuda.synchronize(); lib.vtgb_debug_set_exp(0)
    lib.vtgb_debug_read_stamps(st, 1)
    print("  per-wave entry -> own first operands landed (ns):", [round(st[i] / clk[1] * 10) for i in range(8)])
    print(f"RAFT pass {ms:.0f} ms: {clk[1]} tiles, mean prologue {clk[0] / clk[1] * 10:.0f} ns; sum of prologues / 256 CUs = {clk[0] * 10 / 256 / 1e6:.1f} ms")
