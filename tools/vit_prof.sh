#!/bin/bash
# ViT-g stage under rocprofv3, folded and unfolded LayerNorms.  usage: tools/vit_prof.sh [tag]
TAG=${1:-vit}
for F in 1 0; do
  cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/vp_$TAG$F
  FOLD=$F rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vp_$TAG$F -- python3 $GRAFT_REPO_ROOT/tools/vit_bench.py 992 2>/dev/null | grep ViT-g
  cd $GRAFT_REPO_ROOT
  python3 tools/summarize_rocprof.py /tmp/vp_$TAG$F gpurun_out/${TAG}_fold${F}_stats vit_bench 992 FOLD=$F > /dev/null
  head -16 gpurun_out/${TAG}_fold${F}_stats.md | cut -c1-150
done
