"""LLM stage of the bench (Vicuna-7B geometry, B=124, 52 prompt positions, 16 new tokens): prefill vs decode time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from videotgb_amd import llm
from videotgb_amd.decode import GreedyDecoder
dev = torch.device("cuda:0")
lm = llm.build_llama("vicuna-7b", torch.bfloat16, dev)
dec = GreedyDecoder(lm)
for B in (124, 32):
    emb = torch.randn(B, 52, 4096, device=dev, dtype=torch.bfloat16) * 0.02
    for N in (1, 16):
        for _ in range(2): dec.generate(emb, N)
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(3): dec.generate(emb, N)
        torch.cuda.synchronize(); print(f"B={B} new tokens={N}: {(time.time() - t0) / 3 * 1e3:.1f} ms")
