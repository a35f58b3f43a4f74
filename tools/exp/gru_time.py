#!/usr/bin/env python3
"""Average duration of the RAFT update block's launches at the bench batch (31 clips, T = 96 -> 2945 pairs, 28 x 28 coarse pixels), by kernel,
from torch.profiler -- library selected through VTGB_LIB (tools/exp/gru_abl.sh builds the ablation variants)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from videotgb_amd import ops, synth
dev = torch.device("cuda:0")
n, H8, W8 = int(os.environ.get("GRU_PAIRS", 2945)), 28, 28
sd = {k[len("of_extractor."):]: v.to(dev) for k, v in synth.synth_state_dict(synth.raft_shapes("of_extractor."), 0).items()}
w = ops.RaftWeights(sd, "update_block.", ops.BF16)
g = torch.Generator(device=dev).manual_seed(0)
cnet = torch.randn(n, H8 * W8, 256, generator=g, device=dev)
pyr = [torch.randn(n * H8 * W8, 1, H8 >> l, W8 >> l, generator=g, device=dev).half() for l in range(4)]
iters = int(os.environ.get("GRU_ITERS", 4))
for _ in range(2):
    ops.raft_update(w, None, None, pyr, iters=iters, cnet_nhwc=cnet, hw=(H8, W8))
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    out = ops.raft_update(w, None, None, pyr, iters=iters, cnet_nhwc=cnet, hw=(H8, W8))
    torch.cuda.synchronize()
rows = [(e.key, e.count, e.device_time_total / max(e.count, 1)) for e in prof.key_averages() if e.device_time_total > 0]
tag = os.environ.get("VTGB_LIB", "product")
for k, c, t in sorted(rows, key=lambda r: -r[1] * r[2]):
    if "gru_half" in k or os.environ.get("GRU_ALL"):
        print(f"{tag:40s} {t:9.1f} us x {c:3d}  {k[:90]}")
print(f"{tag:40s} flow absmax {out.abs().max().item():.3f}")
from videotgb_amd import _lib
import ctypes
lib = ctypes.CDLL(_lib.LIB_PATH)
if hasattr(lib, "vtgb_debug_set_gru_dbg"):
    n_tiles = (n * 28 + 3) // 4
    dbg = torch.zeros(n_tiles, 8, dtype=torch.int64, device=dev)
    lib.vtgb_debug_set_gru_dbg(ctypes.c_void_p(dbg.data_ptr()))
    ops.raft_update(w, None, None, pyr, iters=1, cnet_nhwc=cnet, hw=(H8, W8))
    torch.cuda.synchronize()
    lib.vtgb_debug_set_gru_dbg(ctypes.c_void_p(0))
    raw = dbg.cpu()
    hw = (raw[:, 0] >> 48) & 0xFFFF
    xcc = (raw[:, 0] >> 44) & 0xF
    raw = raw & 0xFFFFFFFFFFF
    import collections
    cu = ((xcc << 16) | (hw & 0xFF00)).tolist()      # (xcc, se, sh, cu)
    slot = (hw & 0xF).tolist()
    t1 = raw[:, 1].tolist()
    by = collections.defaultdict(list)
    for i in range(n_tiles):
        by[cu[i]].append((t1[i], slot[i], i))
    print(f"   CUs seen {len(by)}, wave-slot histogram {collections.Counter(slot).most_common(6)}")
    k = sorted(by)[3]
    ev = sorted(by[k])[:12]
    print("   one CU, S1 start stamps (cycles rel.), slot:", [(e[0] - ev[0][0], e[1]) for e in ev])
    d = raw.double()
    ph = (d[:, 1:] - d[:, :-1])
    ok = (d[:, 7] > 0)
    names = ["load+wait", "S1 k-loop", "drain+bar", "E1+bar", "S2 k-loop", "lo+bar", "E2"]
    print("phase cycles (mean over tiles, last launch = vertical half; 100 MHz memtime ticks x ? -> cycles are s_memtime units):")
    for i, nm in enumerate(names):
        print(f"   {nm:12s} {ph[ok, i].mean().item():10.0f}  (median {ph[ok, i].median().item():8.0f})")
    print(f"   tile total   {(d[ok, 7] - d[ok, 0]).mean().item():10.0f}")
