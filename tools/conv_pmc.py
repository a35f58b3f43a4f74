#!/usr/bin/env python3
"""One RAFT pass over B clips (the bench's per-call batch) for PMC collection on the implicit-GEMM convolutions."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from videotgb_amd import models, synth
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 31
r = models.Raft(os.environ.get('RAFT_DTYPE', 'bf16'))
sd = {k[len("of_extractor."):]: v for k, v in synth.synth_state_dict(synth.raft_shapes("of_extractor."), 0).items()}
for k in list(sd):
    if ".downsample.1." in k:
        sd[k] = sd[k.replace(".downsample.1.", ".norm3.")]
r.load_state_dict(sd, strict=True); r.to(dev)
g = torch.Generator(device=dev).manual_seed(0)
frames = torch.randint(0, 256, (B, 96, 3, 224, 224), generator=g, device=dev).float()
out = r.forward_clips(frames)
torch.cuda.synchronize()
print("flow absmax", out.abs().max().item())
