#!/usr/bin/env python3
"""Run-to-run determinism of the RAFT stages (same process, same inputs), sensitive weight set, bf16."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from videotgb_amd import models, synth, ops
dev = torch.device("cuda:0")
z = np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "tiny_raft_sensitive.npz"))
fa = torch.from_numpy(z["frames_a_f16"]).float().to(dev)
sd = {k[len("of_extractor."):]: v for k, v in synth.raft_sensitive_state_dict(0).items()}
r = models.Raft("bf16"); r.load_state_dict(sd, strict=True); r.to(dev)
upd, fw, cw = r._hip_tables()
def d(a, b): return f"max|diff| {(a.float() - b.float()).abs().max().item():.3e} (max {a.float().abs().max().item():.3e})"
img = torch.cat([fa[:-1], fa[1:]], 0)
f1, f2 = ops.raft_encoder(fw, img), ops.raft_encoder(fw, img)
print("fnet twice:", d(f1, f2))
c1, c2 = ops.raft_encoder(cw, fa[:-1]), ops.raft_encoder(cw, fa[:-1])
print("cnet twice:", d(c1, c2))
p1, p2 = ops.raft_corr(f1, 2, 16, 16, 2, 2, 0, 2, r.code), ops.raft_corr(f1, 2, 16, 16, 2, 2, 0, 2, r.code)
print("corr twice:", [d(a, b) for a, b in zip(p1, p2)])
for it in (1, 2, 20):
    u1 = ops.raft_update(upd, None, None, p1, it, cnet_nhwc=c1, hw=(16, 16))
    u2 = ops.raft_update(upd, None, None, p1, it, cnet_nhwc=c1, hw=(16, 16))
    print(f"update x{it} twice (same inputs):", d(u1, u2))
u1 = ops.raft_update(upd, None, None, p1, 1, cnet_nhwc=c1, hw=(16, 16))
for k in range(4):
    u2 = ops.raft_update(upd, None, None, p1, 1, cnet_nhwc=c1, hw=(16, 16))
    dd = (u1 - u2).abs().amax(1)              # [n, 128, 128]
    blk = dd.view(2, 16, 8, 16, 8).amax((2, 4))   # per coarse pixel
    bad = (blk > 1e-4)
    print(f"run {k}: coarse pixels differing: pair0 {int(bad[0].sum())} pair1 {int(bad[1].sum())}; rows with diffs pair0 {sorted(set(torch.nonzero(bad[0])[:,0].tolist()))} pair1 {sorted(set(torch.nonzero(bad[1])[:,0].tolist()))}")
