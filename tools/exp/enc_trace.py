#!/usr/bin/env python3
"""Every launch of one RAFT encoder call (384 frames of 224 x 224) in issue order with its duration -- fnet (InstanceNorm) and cnet
(BatchNorm folded)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import ProfilerActivity, profile
from videotgb_amd import ops, synth
dev = torch.device("cuda:0")
sd = {k[len("of_extractor."):]: v.to(dev) for k, v in synth.synth_state_dict(synth.raft_shapes("of_extractor."), 0).items()}
for k in list(sd):
    if ".downsample.1." in k:
        sd[k] = sd[k.replace(".downsample.1.", ".norm3.")]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 384
fr = torch.randint(0, 256, (n, 3, 224, 224), device=dev).float()
for name, bn in (("fnet.", False), ("cnet.", True)):
    w = ops.RaftEncoderWeights(sd, name, bn, ops.raft_dtype_code(os.environ.get("RAFT_DTYPE", "bf16")))
    for _ in range(2): ops.raft_encoder(w, fr)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        ops.raft_encoder(w, fr); torch.cuda.synchronize()
    ev = sorted([e for e in prof.events() if e.device_time_total > 0], key=lambda e: e.time_range.start)
    tot = sum(e.device_time_total for e in ev)
    print(f"== {name} {n} frames: {tot / 1e3:.2f} ms in {len(ev)} launches")
    for e in ev:
        print(f"   {e.device_time_total:8.1f} us  {e.name[:110]}")
