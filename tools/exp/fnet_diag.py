#!/usr/bin/env python3
"""Diagnostic (round 3): fnet bf16 error vs the fp32 oracle over image sizes / image counts / frame kinds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import vtgb_oracle as O
from videotgb_amd import ops, synth
dev = torch.device("cuda:0")
sd = synth.path_state_dict(synth.tiny_cfg(), 0)
rsd = {k[len("of_extractor."):]: v.to(dev) for k, v in sd.items() if k.startswith("of_extractor.")}
def rr(a, b): return float((a - b).double().pow(2).mean().sqrt() / b.double().pow(2).mean().sqrt())
w = ops.RaftEncoderWeights(rsd, "fnet.", False, ops.BF16)
for size in (128, 160, 192, 224, 256):
    for n in (1, 2, 3):
        for kind in ("int", "randn", "randn_x20"):
            g = torch.Generator().manual_seed(size * 7 + n)
            if kind == "int":
                fr = torch.randint(0, 256, (n, 3, size, size), generator=g).float()
            else:
                fr = torch.randn(n, 3, size, size, generator=g) * (20.0 if kind == "randn_x20" else 1.0)
            ref = O.raft_encoder(sd, "of_extractor.fnet.", 2 * (fr / 255.0) - 1.0, "instance")
            out = ops.raft_encoder(w, fr.to(dev)).cpu().view(n, size // 8, size // 8, 256).permute(0, 3, 1, 2)
            per = [rr(out[i], ref[i]) for i in range(n)]
            print(f"size {size} n {n} {kind:9s}: rel_rms {rr(out, ref):.3e}  per image {['%.2e' % e for e in per]}", flush=True)
