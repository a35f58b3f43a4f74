#!/usr/bin/env python3
"""RAFT stage alone (all-HIP clip path): ms per clip at T=96, and a target for rocprofv3 --stats."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
# (the vtgb_debug_* knobs exist only in a library built with VTGB_DEBUG_HOOKS=1 python -m videotgb_amd.build --force)
from videotgb_amd import models, synth
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
NWN = int(sys.argv[2]) if len(sys.argv) > 2 else 0   # > 0 forces the conv tile width (4 = 256 wide for every conv)
from videotgb_amd import _lib
if hasattr(_lib.lib(), 'vtgb_debug_set_conv_nwn'): _lib.lib().vtgb_debug_set_conv_nwn(NWN)
r = models.Raft(os.environ.get('RAFT_DTYPE', 'bf16'))
sd = {k[len("of_extractor."):]: v for k, v in synth.synth_state_dict(synth.raft_shapes("of_extractor."), 0).items()}
for k in list(sd):
    if ".downsample.1." in k:
        sd[k] = sd[k.replace(".downsample.1.", ".norm3.")]
r.load_state_dict(sd, strict=True); r.to(dev)
g = torch.Generator(device=dev).manual_seed(0)
frames = torch.randn(B, 96, 3, 224, 224, generator=g, device=dev)
for _ in range(2): r.forward_clips(frames)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(3): out = r.forward_clips(frames)
torch.cuda.synchronize(); dt = (time.time() - t0) / 3
print(f"nwn={NWN} RAFT all-HIP: {dt * 1e3 / B:.2f} ms per clip (B={B}, T=96), flow absmax {out.abs().max().item():.3f}")
