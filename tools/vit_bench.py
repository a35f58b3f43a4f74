#!/usr/bin/env python3
"""EVA-ViT-g stage alone at the bench's 992 frames: ms per call and MFMA utilisation, LayerNorms folded into the GEMMs (default) or as their own
passes (FOLD=0); a target for rocprofv3 --kernel-trace --stats."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from videotgb_amd import ops
from videotgb_amd.synth import VitCfg, synth_state_dict, vit_shapes
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 992
fold = os.environ.get("FOLD", "1") != "0"
sd = {k: v.to(dev) for k, v in synth_state_dict(vit_shapes(VitCfg(), ""), 0).items()}
w = ops.VitWeights(sd, "", ops.BF16, 16, 1e-6, fold_ln=fold)
pix = torch.randn(n, 3, 224, 224, generator=torch.Generator(device=dev).manual_seed(0), device=dev)
for _ in range(2):
    ops.vit_forward(w, pix, want_f32=False, want_act=True)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(5):
    ops.vit_forward(w, pix, want_f32=False, want_act=True)
torch.cuda.synchronize(); dt = (time.time() - t0) / 5
print(f"ViT-g {n} frames fold_ln={fold}: {dt * 1e3:.1f} ms per call, {520.72e9 * n / dt / 1e12:.1f} TFLOP/s = {520.72e9 * n / dt / 2.5e15:.4f} of the bf16 MFMA peak")
