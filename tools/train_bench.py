#!/usr/bin/env python3
"""Config C5 micro-step at full size (row a14): Vicuna-7B geometry + LoRA (r=8 on q_proj/v_proj, 4.19 M trainable),
32 visual prefix tokens + question + answer, HIP token splice / label masking / shifted cross-entropy, AdamW every 4
micro-batches.  The frozen prefix path is not part of this measurement (its cost is bench.py's); the prefix is a random
[B, 32, 4096] tensor.  Prints ms per micro-batch and sequences/s on one MI355X."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from videotgb_amd import llm, train
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4          # per-GPU micro-batch of the reference experiment config
lm = llm.build_llama("vicuna-7b", torch.bfloat16, dev)
class H: pass
m = H(); m.model = H(); m.model.language_model = lm
step = train.LoraTrainStep(m, pad_token_id=0, lr=1e-4, accumulate_grad_batches=4)
lm.train()
print("trainable parameters:", sum(p.numel() for p in step.params))
g = torch.Generator(device=dev).manual_seed(0)
P, Li, Lo = 32, 48, 32
prefix = torch.randn(B, P, 4096, generator=g, device=dev, dtype=torch.bfloat16)
q = torch.randint(3, 32000, (B, Li), generator=g, device=dev); qm = torch.ones_like(q)
a = torch.randint(3, 32000, (B, Lo), generator=g, device=dev); am = torch.ones_like(a)
for _ in range(4): loss, _ = step.step(prefix, q, qm, a, am)
torch.cuda.synchronize(); t0 = time.time()
n = 8
for _ in range(n): loss, stepped = step.step(prefix, q, qm, a, am)
torch.cuda.synchronize(); dt = (time.time() - t0) / n
print(f"C5 LoRA micro-batch B={B}, S={P + Li + Lo - 1}: {dt * 1e3:.1f} ms ({B / dt:.1f} sequences/s), loss {loss.item():.3f}, "
      f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
