#!/usr/bin/env python3
"""A/B the persistent kernel against the one-tile-per-workgroup kernel on the sensitive RAFT set: run twice (VTGB_GEMM_OLD=1 / unset), each
dumps the flow; the second run prints where they differ."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from videotgb_amd import models, synth
dev = torch.device("cuda:0")
z = np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "tiny_raft_sensitive.npz"))
fa = torch.from_numpy(z["frames_a_f16"]).float()
sd = {k[len("of_extractor."):]: v for k, v in synth.raft_sensitive_state_dict(0).items()}
r = models.Raft("bf16"); r.load_state_dict(sd, strict=True); r.to(dev)
outs = {}
for it in (1, 2, 20):
    outs[it] = r(fa[:-1].to(dev), fa[1:].to(dev), iters=it).cpu()
tag = "old" if os.environ.get("VTGB_GEMM_OLD") == "1" else "new"
torch.save(outs, f"/tmp/flow_{tag}.pt")
ref = torch.from_numpy(z["flow_a"])
print(tag, "iters20 rel_rms vs reference", float((outs[20] - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()))
if tag == "new" and os.path.exists("/tmp/flow_old.pt"):
    old = torch.load("/tmp/flow_old.pt")
    for it in (1, 2, 20):
        d = (outs[it] - old[it]).abs()
        print(f"iters {it}: max|new-old| {d.max():.3e} (max|old| {old[it].abs().max():.3e}); fraction of pixels differing by > 1e-3: {(d > 1e-3).float().mean():.4f}")
        if it == 1:
            bad = (d > 1e-3).any(1)          # [n, H, W]
            for n in range(bad.shape[0]):
                ys, xs = torch.nonzero(bad[n], as_tuple=True)
                if len(ys): print(f"  pair {n}: differing pixel rows {ys.min().item()}..{ys.max().item()} cols {xs.min().item()}..{xs.max().item()} count {len(ys)}")
