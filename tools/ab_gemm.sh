#!/bin/bash
# Same-box A/B on the ViT-g GEMM shapes and the RAFT stage; see tools/ab_bench.sh for the two modes (nothing is overwritten).
cd "${GRAFT_REPO_ROOT:-.}"
OTHER="$1"
run() {
  echo "== $1"; python tools/gemm_bench.py 992 2>&1 | grep -v amdgpu.ids | tail -6
  python tools/raft_bench.py 31 2>&1 | tail -n 1
}
for i in 1 2; do
  run new
  if [ -n "$OTHER" ]; then VTGB_LIB="$(realpath "$OTHER")" run other; else VTGB_GEMM_OLD=1 run old-kernel; fi
done
