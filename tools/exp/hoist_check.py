"""inp-hoist (GRU start maps in bf16): flow accuracy vs the fp32 exactness mode and speed, with and without."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from videotgb_amd import models, ops, synth
dev = torch.device("cuda:0")
sd = {k[len("of_extractor."):]: v for k, v in synth.synth_state_dict(synth.raft_shapes("of_extractor."), 0).items()}
for k in list(sd):
    if ".downsample.1." in k:
        sd[k] = sd[k.replace(".downsample.1.", ".norm3.")]
def mk(dt):
    r = models.Raft(dt); r.load_state_dict(sd, strict=True); return r.to(dev)
r32, r16 = mk("f32"), mk("bf16")
g = torch.Generator(device=dev).manual_seed(0)
def rr(a, b): return float(((a.double() - b.double()).pow(2).mean().sqrt()) / b.double().pow(2).mean().sqrt()), float((a - b).abs().max() / b.abs().max())
fr = torch.randn(2, 24, 3, 224, 224, generator=g, device=dev)
ref = r32.forward_clips(fr)
for hoist in (True, False):
    r16._table = None; tabs = r16._hip_tables()
    dsd = {k: v for k, v in r16.state_dict().items()}
    r16._table = (ops.RaftWeights(dsd, "update_block.", ops.BF16, hoist_inp=hoist), tabs[1], tabs[2])
    print("hoist", hoist, "bf16 vs f32 (rel-RMS, max/max):", rr(r16.forward_clips(fr), ref))
    big = torch.randn(16, 96, 3, 224, 224, generator=g, device=dev)
    for _ in range(2): r16.forward_clips(big)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(3): r16.forward_clips(big)
    torch.cuda.synchronize(); print("   ms per clip (B=16):", (time.time() - t0) / 3 / 16 * 1e3)
