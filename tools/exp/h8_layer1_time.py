"""Timing experiment: a layer1-shaped convolution (3x3, 64 -> 64 channels, 112 x 112, 192 images) over f16c8 pairs on the 256 x 64 tile of gemm_h8.hip
(vtgb_pair_conv), against the 0.71-0.89 ms the bf16x3 form takes on the 64-wide tile of gemm.hip (profiles/r06_encoder_trace_bf16x3.log)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from videotgb_amd import ops
dev = torch.device("cuda:0")
n, H, W, C = 192, 112, 112, 64
g = torch.Generator(device=dev).manual_seed(0)
a = ops.pair_pack(torch.randn(n * H * W, C, generator=g, device=dev).abs())
wk = torch.randn(64, 3, 3, C, generator=g, device=dev) * 0.05
sw, _ = ops.h8_weight_scale(wk)
for _ in range(3):
    ops.pair_conv(a, wk, sw, H, W, relu=True)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(10):
    ops.pair_conv(a, wk, sw, H, W, relu=True)
torch.cuda.synchronize()
print(f"f16c8 256 x 64 tile: {(time.time() - t0) / 10 * 1e3:.3f} ms per launch (incl. the wrapper's packing; see the profiler line)")
from torch.profiler import ProfilerActivity, profile
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    ops.pair_conv(a, wk, sw, H, W, relu=True); torch.cuda.synchronize()
for e in prof.events():
    if "conv_h8" in e.name:
        print(f"   {e.device_time_total:8.1f} us  {e.name[:90]}")
