#!/bin/bash
# GPU box: everything under profiles/ for round 6 (run from the repo root through gpurun; outputs in gpurun_out/, copies in profiles/).
# usage: tools/collect_r06.sh [part ...]   parts: pmc bench trace mfma x3 single tests   (default: all)
R=r06
O=$GRAFT_REPO_ROOT/gpurun_out
P=$GRAFT_REPO_ROOT/profiles
mkdir -p $O $P; export TMPDIR=/tmp
PARTS=${@:-pmc bench trace mfma x3 single tests}
has() { [[ " $PARTS " == *" $1 "* ]]; }
if has pmc; then
  cd /tmp
  # HBM traffic (FETCH_SIZE / WRITE_SIZE, one counter per pass) of the conv family in the f16c8 (headline), bf16x3 and bf16 RAFT modes
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/ph_$c /tmp/px_$c /tmp/pc_$c
    export RAFT_DTYPE=f16c8;  timeout 900 rocprofv3 --pmc $c --output-format csv -d /tmp/ph_$c -- python3 $GRAFT_REPO_ROOT/tools/conv_pmc.py 62 > /tmp/ph_$c.log 2>&1
    export RAFT_DTYPE=bf16x3; timeout 900 rocprofv3 --pmc $c --output-format csv -d /tmp/px_$c -- python3 $GRAFT_REPO_ROOT/tools/conv_pmc.py 62 > /tmp/px_$c.log 2>&1
    export RAFT_DTYPE=bf16;   timeout 900 rocprofv3 --pmc $c --output-format csv -d /tmp/pc_$c -- python3 $GRAFT_REPO_ROOT/tools/conv_pmc.py 62 > /tmp/pc_$c.log 2>&1
    unset RAFT_DTYPE
  done
  python3 $GRAFT_REPO_ROOT/tools/pmc_traffic.py /tmp/ph_FETCH_SIZE /tmp/ph_WRITE_SIZE $O/${R}_pmc_traffic_convh8.json conv > /dev/null
  python3 $GRAFT_REPO_ROOT/tools/pmc_traffic.py /tmp/px_FETCH_SIZE /tmp/px_WRITE_SIZE $O/${R}_pmc_traffic_convx3.json conv > /dev/null
  python3 $GRAFT_REPO_ROOT/tools/pmc_traffic.py /tmp/pc_FETCH_SIZE /tmp/pc_WRITE_SIZE $O/${R}_pmc_traffic_conv.json conv > /dev/null
  cp $O/${R}_pmc_traffic_*.json $P/
fi
if has bench; then
  cd $GRAFT_REPO_ROOT; python3 bench.py > $O/${R}_bench.json 2> $O/${R}_bench.err; tail -c 300 $O/${R}_bench.json; cp $O/${R}_bench.json $P/
fi
if has trace; then
  cd /tmp; rm -rf /tmp/pb
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary > /tmp/pb.log 2>&1
  cd $GRAFT_REPO_ROOT; python3 tools/summarize_rocprof.py /tmp/pb $O/${R}_bench_kernel_stats python bench.py --no-cpu-baseline --no-secondary > /dev/null
  head -24 $O/${R}_bench_kernel_stats.md | cut -c1-160; cp $O/${R}_bench_kernel_stats.* $P/
fi
if has mfma; then
  cd $GRAFT_REPO_ROOT
  export RAFT_DTYPE=f16c8; bash tools/pmc_mfma_conv.sh ${R}h8 62 > /dev/null 2>&1; unset RAFT_DTYPE
  cp $O/${R}h8_pmc_mfma_conv_a.txt $P/${R}_pmc_mfma_convh8_a.txt; cp $O/${R}h8_pmc_mfma_conv_b.txt $P/${R}_pmc_mfma_convh8_b.txt; rm -f $P/${R}h8_pmc_mfma_conv_?.txt
  tail -14 $P/${R}_pmc_mfma_convh8_a.txt
fi
if has x3; then
  cd $GRAFT_REPO_ROOT
  for m in f16c8 bf16x3 bf16; do
    bash tools/x3_prof.sh $m ${R}_$m | head -20
    cp $O/${R}_${m}_raft_stats.md $P/${R}_raft_${m}_kernel_stats.md; cp $O/${R}_${m}_iter_trace.log $P/${R}_raft_${m}_iter_trace.log
  done
fi
if has single; then
  cd $GRAFT_REPO_ROOT; bash tools/single_clip_prof.sh ${R}_single_clip | head -30
  cp $O/${R}_single_clip_kernel_stats.md $O/${R}_single_clip_kernel_stats.json $P/ 2>/dev/null
fi
if has tests; then
  cd $GRAFT_REPO_ROOT
  python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" > $O/${R}_gpu_tests.log; cat $O/${R}_gpu_tests.log; cp $O/${R}_gpu_tests.log $P/
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
fi
