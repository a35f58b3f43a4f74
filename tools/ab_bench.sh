#!/bin/bash
# Same-box A/B of the headline (boxes differ by 5-12 %).  The product library is never overwritten:
#   tools/ab_bench.sh                 -> persistent kernel (default) vs the one-tile-per-workgroup kernel (VTGB_GEMM_OLD=1)
#   tools/ab_bench.sh path/to/old.so  -> the current build vs another build of libvtgb.so (selected through VTGB_LIB)
# Run through gpurun from the repo root; prints the headline and the roofline families for A / B / A / B.
cd "${GRAFT_REPO_ROOT:-.}"
OTHER="$1"
run() {
  python bench.py --no-secondary --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1', 'clips/s', d['value'], 'ms/step', d['ms_per_step'], '| conv', r['achieved'], 'TF/s', r['avg_launch_us'], 'us | gemm', r['other'][0]['achieved'], 'TF/s', r['other'][0]['ms_per_step'], 'ms')"
}
for i in 1 2; do
  run new
  if [ -n "$OTHER" ]; then VTGB_LIB="$(realpath "$OTHER")" run other; else VTGB_GEMM_OLD=1 run old-kernel; fi
done
