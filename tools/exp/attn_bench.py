"""ViT-g attention at the bench's batch (496 frames x 16 heads x 257 tokens x 88): us per call and TFLOP/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from videotgb_amd import ops
dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 496
H, hd, S = 16, 88, 257
qkv = torch.randn(B, S, 3 * H * hd, device=dev).bfloat16()
q, k, v = qkv[:, :, :H * hd], qkv[:, :, H * hd:2 * H * hd], qkv[:, :, 2 * H * hd:]
for _ in range(3): ops.attention(q, k, v, H, hd ** -0.5)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ops.attention(q, k, v, H, hd ** -0.5)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 20
print(f"{os.path.basename(os.environ.get('VTGB_LIB', 'default')):36s} B={B}: {t * 1e3:.0f} us, {4.0 * B * H * S * S * hd / t / 1e9:.0f} TFLOP/s, {t * 1e3 * 256 / (B * H):.1f} us per (frame, head) per CU")
