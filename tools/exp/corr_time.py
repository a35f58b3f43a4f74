#!/usr/bin/env python3
"""The correlation stage alone at the bench shape (31 clips of 96 frames, 28 x 28 x 256 features): ms per call, RAFT_DTYPE mode."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from videotgb_amd import ops
dev = torch.device("cuda:0")
clips, T = int(sys.argv[1]) if len(sys.argv) > 1 else 31, 96
fmap = torch.randn(clips * T, 784, 256, device=dev)
code = ops.raft_stage_code(ops.raft_dtype_code(os.environ.get("RAFT_DTYPE", "f16c8")))
for _ in range(2): pyr = ops.raft_corr(fmap, clips * (T - 1), 28, 28, T - 1, T, 0, 1, code)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3): pyr = ops.raft_corr(fmap, clips * (T - 1), 28, 28, T - 1, T, 0, 1, code)
e1.record(); torch.cuda.synchronize()
print(f"corr {os.environ.get('VTGB_LIB', 'default').split('_')[-1]}: {e0.elapsed_time(e1) / 3:.2f} ms per call")
