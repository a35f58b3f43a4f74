#!/bin/bash
# Same-box A/B of two library builds (boxes differ by 5-12 %): videotgb_amd/libvtgb.so (new) against videotgb_amd/libvtgb_old.so.
# Run through gpurun from the repo root; prints the headline for new / old / new (and, with an argument, the ms of the kernels
# whose name matches it, from the bench's own profiling pass).
cd $GRAFT_REPO_ROOT
cp videotgb_amd/libvtgb.so /tmp/new.so
run() {
  python bench.py --no-secondary --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1', 'clips/s', d['value'], 'ms/step', d['ms_per_step'], '| conv', r['achieved'], 'TF/s', r['avg_launch_us'], 'us | gemm', r['other'][0]['achieved'], 'TF/s', r['other'][0]['ms_per_step'], 'ms')"
}
run new
cp videotgb_amd/libvtgb_old.so videotgb_amd/libvtgb.so; run old
cp /tmp/new.so videotgb_amd/libvtgb.so; run new
cp videotgb_amd/libvtgb_old.so videotgb_amd/libvtgb.so; run old
