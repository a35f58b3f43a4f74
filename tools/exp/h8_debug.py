"""Where the f16c8 update block first departs from the bf16x3 one: the same vtgb_raft_update call in both modes, then the workspace buffers
(caller-owned scratch: raft_x3.hip's take() order) decoded and compared.  usage: python tools/exp/h8_debug.py [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from videotgb_amd import ops, synth

dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n, h8, w8 = 2, 16, 16
sd = {k[len("of_extractor."):]: v.to(dev) for k, v in synth.raft_sensitive_state_dict(0).items()}
g = torch.Generator(device=dev).manual_seed(3)
cnet = torch.randn(n, h8 * w8, 256, generator=g, device=dev)
pyr = [torch.randn(n * h8 * w8, 1, max(h8 >> l, 1), max(w8 >> l, 1), generator=g, device=dev) for l in range(4)]
M = n * h8 * w8


def al(x):
    return (x + 255) // 256 * 256


layout = [("hb", 512, "pair", 128), ("X", 512, "pair", 128), ("INP", 512, "bfpair", 128), ("ZRI0", 1024, "f32", 256), ("ZRI1", 1024, "f32", 256), ("QI0", 512, "f32", 128),
          ("QI1", 512, "f32", 128), ("corrf", 1536, "pair", 384), ("c1", 1024, "pair", 256), ("CF", 1024, "pair", 256), ("f1", 512, "bfpair", 128), ("RH", 512, "pair", 128),
          ("FH", 1024, "bfpair", 256), ("ZR", 1024, "f32z", 128), ("Q", 512, "f32q", 64), ("flow", 8, "f32", 2)]
res = {}
for mode in ("bf16x3", "f16c8"):
    code = ops.raft_dtype_code(mode)
    w = ops.RaftWeights(sd, "update_block.", code)
    out = ops.raft_update(w, None, None, pyr, iters=iters, cnet_nhwc=cnet, hw=(h8, w8))
    torch.cuda.synchronize()
    ws = ops._ws.get(1, dev)
    off, bufs = 0, {}
    for name, rb, kind, C_ in layout:
        if name == "corrf" and mode == "f16c8":      # (not allocated: the taps stay inside the fused lookup + convc1 launch)
            continue
        off = al(off)
        raw = ws[off:off + M * rb].clone()
        off += M * rb
        if kind == "f32":
            v = raw.view(torch.float32).view(M, -1)
        elif kind == "f32z":
            v = raw.view(torch.float32).view(M, 256)[:, :128] if False else raw.view(torch.float32).view(-1)[:M * 128].view(M, 128)
        elif kind == "f32q":
            v = raw.view(torch.float32).view(-1)[:M * 64].view(M, 64)
        else:
            fmt = ops.F16C8 if (kind == "pair" and mode == "f16c8") else ops.BF16X3
            v = ops.pair_unpack(raw.view(torch.int16).view(M, -1), C_, fmt)
        bufs[name] = v.float().cpu()
    bufs["flow_up"] = out.float().cpu()
    res[mode] = bufs
for name in list(res["f16c8"]):
    a, b = res["bf16x3"][name], res["f16c8"][name]
    d = (a - b).abs()
    rel = float(d.pow(2).mean().sqrt() / a.pow(2).mean().sqrt().clamp_min(1e-30))
    worst = int(d.max(1).values.argmax()) if d.dim() == 2 else 0
    print(f"{name:8s} rel_rms(f16c8 vs bf16x3) = {rel:.3e}   max|diff| = {float(d.max()):.3e}  max|ref| = {float(a.abs().max()):.3e}  worst row {worst}, finite {bool(torch.isfinite(b).all())}")
