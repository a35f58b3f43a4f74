#!/usr/bin/env python3
"""Encoder time per frame against the chunk size (does a chunk whose fp32 convolution output fits the 256 MB Infinity Cache pay less for the
normalisation passes?): 384 frames of 224 x 224, fnet and cnet, RAFT_DTYPE mode."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from videotgb_amd import ops, synth
dev = torch.device("cuda:0")
sd = {k[len("of_extractor."):]: v.to(dev) for k, v in synth.synth_state_dict(synth.raft_shapes("of_extractor."), 0).items()}
for k in list(sd):
    if ".downsample.1." in k:
        sd[k] = sd[k.replace(".downsample.1.", ".norm3.")]
fr = torch.randint(0, 256, (768, 3, 224, 224), device=dev).float()
mode = os.environ.get("RAFT_DTYPE", "f16c8")
for name, bn in (("fnet.", False), ("cnet.", True)):
    w = ops.RaftEncoderWeights(sd, name, bn, ops.raft_dtype_code(mode))
    for mi in (1536, 1024, 768, 512, 384):      # (halved inside raft_encoder for the pair modes)
        for _ in range(2): ops.raft_encoder(w, fr, max_images=mi)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): ops.raft_encoder(w, fr, max_images=mi)
        e1.record(); torch.cuda.synchronize()
        print(f"{name} {mode} chunk {mi // 2 if mode in ('f16c8', 'bf16x3', 'f32') else mi:4d} frames: {e0.elapsed_time(e1) / 3 / 768 * 1e3:7.2f} us per frame", flush=True)
