#!/usr/bin/env python3
"""vtgb_raft_update has no atomics: two calls on the same inputs must agree BIT FOR BIT (a difference = a race).  Bench-sized and small batches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from videotgb_amd import ops, synth
dev = torch.device("cuda:0")
sd = {k[len("of_extractor."):]: v.to(dev) for k, v in synth.synth_state_dict(synth.raft_shapes("of_extractor."), 0).items()}
w = ops.RaftWeights(sd, "update_block.", ops.BF16)
for n, H8, W8, iters in ((510, 28, 28, 20), (2945, 28, 28, 4), (7, 16, 16, 20), (5, 9, 13, 12)):
    g = torch.Generator(device=dev).manual_seed(n)
    cnet = torch.randn(n, H8 * W8, 256, generator=g, device=dev)
    pyr = [torch.randn(n * H8 * W8, 1, max(H8 >> l, 1), max(W8 >> l, 1), generator=g, device=dev).half() for l in range(4)]
    outs = [ops.raft_update(w, None, None, pyr, iters=iters, cnet_nhwc=cnet, hw=(H8, W8)).clone() for _ in range(4)]
    torch.cuda.synchronize()
    diffs = [int((outs[0] != o).sum()) for o in outs[1:]]
    print(f"n={n} {H8}x{W8} iters={iters}: elements differing from run 0: {diffs}  (absmax {outs[0].abs().max().item():.3f}, nan {int(torch.isnan(outs[0]).sum())})")
