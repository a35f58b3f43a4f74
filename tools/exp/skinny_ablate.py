"""One shape of vtgb_gemm_skinny under hipGraph replay (16 calls, 8 rotating weights): us per call.  args: M N K S"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from videotgb_amd import ops
dev = torch.device("cuda:0")
M, N, K, S = (int(a) for a in sys.argv[1:5])
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(M, K, generator=g, device=dev).bfloat16()
wt = [ops.SkinnyWeight((torch.randn(N, K, generator=g, device=dev) * 0.02).bfloat16()) for _ in range(8)]
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
wsb = torch.empty(max(1, ops.gemm_skinny_workspace_bytes(M, N, K, S)), dtype=torch.uint8, device=dev)
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for i in range(8): ops.gemm_skinny(x, wt[i], out=out, workspace=wsb, n_splits=S)
torch.cuda.current_stream().wait_stream(s)
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    for i in range(16): ops.gemm_skinny(x, wt[i % 8], out=out, workspace=wsb, n_splits=S)
for _ in range(3): gr.replay()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): gr.replay()
e1.record(); torch.cuda.synchronize()
print(f"{os.environ.get('VTGB_LIB', 'default'):50s} M={M} N={N} K={K} S={S}: {e0.elapsed_time(e1) / 160 * 1e3:.1f} us")
