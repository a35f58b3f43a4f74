"""f2 prefill at the bench's shapes (Vicuna-7B, 124 clips x P prefix+prompt tokens): libvtgb (vtgb_gemm + vtgb_attention +
vtgb_llm_*) against F.linear (hipBLASLt) + SDPA, same weights.  Usage: python tools/exp/prefill_bench.py [B] [P] [N]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from videotgb_amd import llm
from videotgb_amd.decode import GreedyDecoder

B = int(sys.argv[1]) if len(sys.argv) > 1 else 124
P = int(sys.argv[2]) if len(sys.argv) > 2 else 52
N = int(sys.argv[3]) if len(sys.argv) > 3 else 16
dev = "cuda:0"
lm = llm.build_llama("vicuna-7b", torch.bfloat16, dev, seed=0)
emb = (torch.randn(B, P, lm.config.hidden_size, device=dev) * 0.02).bfloat16()
for name, maxtok in (("libvtgb", 288), ("blas", 0), ("libvtgb", 288), ("blas", 0)):
    dec = GreedyDecoder(lm)
    dec.PREFILL_MAX_TOKENS = maxtok
    dec.generate(emb, N)
    for n_new in (1, N):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            ids = dec.generate(emb, n_new)
        torch.cuda.synchronize()
        print(f"{name:8s} B={B} P={P} new={n_new}: {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms", flush=True)
    del dec
