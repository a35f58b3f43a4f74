"""ViT-g attention at the bench's shape (992 frames x 257 tokens, 16 heads x 88) through vtgb_attention: us per call, TFLOP/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from videotgb_amd import ops
dev = torch.device("cuda:0")
B, S, H, hd = int(sys.argv[1]) if len(sys.argv) > 1 else 992, 257, 16, 88
g = torch.Generator(device=dev).manual_seed(0)
qkv = (torch.randn(B, S, 3 * H * hd, generator=g, device=dev) * 0.5).bfloat16()
q, k, v = qkv[..., :H * hd], qkv[..., H * hd:2 * H * hd], qkv[..., 2 * H * hd:]
out = ops.attention(q, k, v, H, hd ** -0.5)
ref = torch.nn.functional.scaled_dot_product_attention(*(t[:4].float().view(4, S, H, hd).transpose(1, 2) for t in (q, k, v))).transpose(1, 2).reshape(4, S, H * hd)
print("max err", (out[:4].float() - ref).abs().max().item())
for _ in range(3): ops.attention(q, k, v, H, hd ** -0.5)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): ops.attention(q, k, v, H, hd ** -0.5)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 10 * 1e3
print(f"attention B={B}: {us:.1f} us, {4.0 * B * H * S * S * hd / us / 1e6:.1f} TFLOP/s, {B * S * 4 * H * hd * 2 / us / 1e6:.2f} TB/s of q,k,v,o")
