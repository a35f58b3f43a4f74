#!/usr/bin/env python3
"""One encoder call per mode and net at a small size (n frames of S x SW), f16c8 vs bf16x3 outputs: usage enc_small.py [n] [S] [SW]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from videotgb_amd import ops, synth
dev = torch.device("cuda:0")
sd = {k[len("of_extractor."):]: v.to(dev) for k, v in synth.synth_state_dict(synth.raft_shapes("of_extractor."), 0).items()}
for k in list(sd):
    if ".downsample.1." in k:
        sd[k] = sd[k.replace(".downsample.1.", ".norm3.")]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
SW = int(sys.argv[3]) if len(sys.argv) > 3 else S
fr = torch.randint(0, 256, (n, 3, S, SW), device=dev).float()
for name, bn in (("fnet.", False), ("cnet.", True)):
    outs = {}
    for mode in ("bf16x3", "f16c8"):
        print(name, mode, "...", flush=True)
        w = ops.RaftEncoderWeights(sd, name, bn, ops.raft_dtype_code(mode))
        outs[mode] = ops.raft_encoder(w, fr)
        torch.cuda.synchronize()
        print("   ok", float(outs[mode].abs().max()), flush=True)
    d = (outs["f16c8"] - outs["bf16x3"]).double()
    print(f"== {name} f16c8 vs bf16x3 rel rms {float(d.pow(2).mean().sqrt() / outs['bf16x3'].double().pow(2).mean().sqrt()):.3e}", flush=True)
