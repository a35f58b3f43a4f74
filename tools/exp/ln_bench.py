import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from videotgb_amd import ops
dev = torch.device("cuda:0")
for M in (257 * 248, 257 * 992):
    x = torch.randn(M, 1408, device=dev); g = torch.randn(1408, device=dev); b = torch.randn(1408, device=dev)
    ref = torch.nn.functional.layer_norm(x[:1000], (1408,), g, b, 1e-6)
    out = ops.layernorm(x, g, b, 1e-6, torch.bfloat16)
    print("max err", (out[:1000].float() - ref).abs().max().item())
    for _ in range(3): ops.layernorm(x, g, b, 1e-6, torch.bfloat16)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.layernorm(x, g, b, 1e-6, torch.bfloat16)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"M={M}: {ms * 1e3:.1f} us, {M * 1408 * 6 / ms / 1e9:.2f} TB/s")
