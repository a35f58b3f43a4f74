#!/bin/bash
# Same-box A/B of two library builds on the ViT-g GEMM shapes, the RAFT stage and the kernel parity tests.
cd $GRAFT_REPO_ROOT
cp videotgb_amd/libvtgb.so /tmp/new.so
run() {
  echo "== $1"; python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids | tail -6
  python tools/raft_bench.py 31 2>&1 | tail -1
}
run new
cp videotgb_amd/libvtgb_old.so videotgb_amd/libvtgb.so; run old
cp /tmp/new.so videotgb_amd/libvtgb.so; run new
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_raft.py -q -m gpu -x 2>&1 | tail -3
