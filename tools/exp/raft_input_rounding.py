"""How much of the bf16-RAFT flow error on float-valued frames is the bf16 rounding of the raw pixels at the stem?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from videotgb_amd import models, synth
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
for name, fr in (("randn", torch.randn(1, 6, 3, 224, 224, generator=g, device=dev)), ("uint8", torch.randint(0, 256, (1, 6, 3, 224, 224), generator=g, device=dev).float())):
    ref = r32.forward_clips(fr)
    print(name, "bf16 vs f32:", rr(r16.forward_clips(fr), ref))
    print(name, "f32 on bf16-rounded pixels vs f32:", rr(r32.forward_clips(fr.bfloat16().float()), ref))
    print(name, "bf16 on bf16-rounded pixels vs f32 on same:", rr(r16.forward_clips(fr.bfloat16().float()), r32.forward_clips(fr.bfloat16().float())))
