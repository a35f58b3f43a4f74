#!/usr/bin/env python3
"""Robustness: RAFT on non-finite / huge frames must return (non-finite) flows, never fault.  usage: raft_nan.py [mode] [kind]; no args: every combination,
each in its own process."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) < 3:
    bad = 0
    for mode in ("f16c8", "bf16x3", "bf16", "f32"):
        for kind in ("nan", "inf", "huge", "scaled"):
            r = subprocess.run([sys.executable, __file__, mode, kind], capture_output=True, text=True)
            tail = (r.stdout.strip().splitlines() or ["<no output>"])[-1]
            print(f"{mode:7s} {kind:6s} rc={r.returncode:4d}  {tail}", flush=True)
            bad += r.returncode != 0
            if r.returncode != 0:
                print("   stderr:", "\n           ".join(r.stderr.strip().splitlines()[-4:]), flush=True)
    sys.exit(1 if bad else 0)
import torch
from videotgb_amd import models, synth
mode, kind = sys.argv[1], sys.argv[2]
dev = torch.device("cuda:0")
sd = {k[len("of_extractor."):]: v for k, v in synth.synth_state_dict(synth.raft_shapes("of_extractor."), 0).items()}
for k in list(sd):
    if ".downsample.1." in k:
        sd[k.replace(".downsample.1.", ".norm3.")] = sd[k]
r = models.Raft(mode)
if kind == "scaled":      # weights that blow the features up (what a wrongly packed table did in round 6)
    for k in sd:
        if k.startswith("cnet.") and k.endswith("weight") and sd[k].dim() == 4:
            sd[k] = sd[k] * 50.0
r.load_state_dict(sd, strict=False)
r = r.to(dev)
fr = torch.randint(0, 256, (3, 3, 128, 128), device=dev).float()
if kind == "nan": fr[1, :, 40:60, 40:60] = float("nan")
if kind == "inf": fr[1, :, 40:60, 40:60] = float("inf")
if kind == "huge": fr = fr * 1e30
out = r(fr[:-1], fr[1:], iters=6)
torch.cuda.synchronize()
print(f"ok finite={bool(torch.isfinite(out).all())} absmax={float(out.nan_to_num(0, 0, 0).abs().max()):.3g}")
