"""Where does the large GEMM kernel's epilogue time go?  (debug-hook build: VTGB_DEBUG_HOOKS=1 python -m videotgb_amd.build)

ViT-g qkv shape (N = 4224, K = 1408, bf16 staged store), grids of 1/8 .. 62 waves of 256 workgroups:
  full            the production kernel
  no-store        exp 5: the whole epilogue except the global store instructions
  no-epilogue     ablation 16
If the store cost per tile is the same on a 32-CU grid as on a full chip, the per-CU store path bounds it; if it vanishes on
small grids, the chip-wide write burst of 256 synchronised workgroups does -- and exp 4 (first-wave phase stagger) should help."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from videotgb_amd import _lib as L, ops  # noqa: E402

dev = torch.device("cuda:0")
lib = L.lib()
lib.vtgb_debug_set_gemm_large_min_tiles(1)
g = torch.Generator(device=dev).manual_seed(0)
N, K = 4224, 1408
W = (torch.randn(N, K, generator=g, device=dev) * 0.05).bfloat16()
bias = torch.randn(N, generator=g, device=dev)


def t_us(A, reps=20):
    for _ in range(3):
        ops.gemm(A, W, bias, L.EPI_STORE, None)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.gemm(A, W, bias, L.EPI_STORE, None)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for m_tiles in (2, 8, 15, 60, 240, 996):      # x 16.5 n-tiles: 34 .. 16932 workgroups
    M = 256 * m_tiles
    A = torch.randn(M, K, generator=g, device=dev).bfloat16()
    row = []
    for name, exp, abl in (("full", 0, 0), ("no-store", 5, 0), ("no-epilogue", 0, 16)):
        lib.vtgb_debug_set_exp(exp)
        lib.vtgb_debug_set_gemm_ablate(abl)
        us = t_us(A)
        row.append(f"{name} {us:8.1f} us ({2.0 * M * N * K / us / 1e6:5.0f} TF/s)")
    lib.vtgb_debug_set_gemm_ablate(0)
    print(f"m_tiles {m_tiles:4d} ({m_tiles * 17:5d} workgroups): " + " | ".join(row), flush=True)

M = 257 * 992
A = torch.randn(M, K, generator=g, device=dev).bfloat16()
row = []
for spread in (0, 6, 12, 24, 48, 0):
    lib.vtgb_debug_set_exp(4 | (spread << 8) if spread else 0)
    us = t_us(A)
    row.append(f"stagger {spread:2d} us: {2.0 * M * N * K / us / 1e6:5.0f}")
print("qkv at the bench's M = 254944: " + " | ".join(row))
lib.vtgb_debug_set_exp(0)

# exp 6: in-kernel stamps around the staged store loop of every wave (10 ns ticks)
import ctypes as C  # noqa: E402
clk = (C.c_ulonglong * 2)()
lib.vtgb_debug_read_clk(clk, 1)
for m_tiles in (15, 240, 996):
    M = 256 * m_tiles
    A = torch.randn(M, K, generator=g, device=dev).bfloat16()
    lib.vtgb_debug_set_exp(0)
    ops.gemm(A, W, bias, L.EPI_STORE, None)
    torch.cuda.synchronize()
    lib.vtgb_debug_read_clk(clk, 1)
    lib.vtgb_debug_set_exp(6)
    ops.gemm(A, W, bias, L.EPI_STORE, None)
    torch.cuda.synchronize()
    lib.vtgb_debug_read_clk(clk, 1)
    waves = m_tiles * 16.5 * 8
    print(f"m_tiles {m_tiles}: store loop per wave: issue {clk[0] / waves * 10:.0f} ns, issue + completion {clk[1] / waves * 10:.0f} ns")
lib.vtgb_debug_set_exp(0)
