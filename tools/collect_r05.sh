#!/bin/bash
# GPU box: everything under profiles/ for round 5 (run from the repo root through gpurun; outputs in gpurun_out/, copies in profiles/).
# usage: tools/collect_r05.sh [part ...]   parts: pmc bench trace mfma x3 single vit tests   (default: all)
R=r05
O=$GRAFT_REPO_ROOT/gpurun_out
P=$GRAFT_REPO_ROOT/profiles
mkdir -p $O $P; export TMPDIR=/tmp
PARTS=${@:-pmc bench trace mfma x3 single vit tests}
has() { [[ " $PARTS " == *" $1 "* ]]; }
if has pmc; then
  cd /tmp
  # HBM traffic (FETCH_SIZE / WRITE_SIZE, one counter per pass) of the conv family in the bf16 and the bf16x3 RAFT mode and of the plain GEMMs
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pc_$c /tmp/px_$c /tmp/pg_$c
    export RAFT_DTYPE=bf16;   timeout 900 rocprofv3 --pmc $c --output-format csv -d /tmp/pc_$c -- python3 $GRAFT_REPO_ROOT/tools/conv_pmc.py 31 > /tmp/pc_$c.log 2>&1
    export RAFT_DTYPE=bf16x3; timeout 900 rocprofv3 --pmc $c --output-format csv -d /tmp/px_$c -- python3 $GRAFT_REPO_ROOT/tools/conv_pmc.py 31 > /tmp/px_$c.log 2>&1
    unset RAFT_DTYPE;         timeout 600 rocprofv3 --pmc $c --output-format csv -d /tmp/pg_$c -- python3 $GRAFT_REPO_ROOT/tools/gemm_pmc.py > /tmp/pg_$c.log 2>&1
  done
  python3 $GRAFT_REPO_ROOT/tools/pmc_traffic.py /tmp/pc_FETCH_SIZE /tmp/pc_WRITE_SIZE $O/${R}_pmc_traffic_conv.json conv > /dev/null
  python3 $GRAFT_REPO_ROOT/tools/pmc_traffic.py /tmp/px_FETCH_SIZE /tmp/px_WRITE_SIZE $O/${R}_pmc_traffic_convx3.json conv > /dev/null
  python3 $GRAFT_REPO_ROOT/tools/pmc_traffic.py /tmp/pg_FETCH_SIZE /tmp/pg_WRITE_SIZE $O/${R}_pmc_traffic_gemm.json > /dev/null
  cp $O/${R}_pmc_traffic_*.json $P/
fi
if has bench; then
  cd $GRAFT_REPO_ROOT; python3 bench.py > $O/${R}_bench.json 2> $O/${R}_bench.err; tail -c 400 $O/${R}_bench.json; cp $O/${R}_bench.json $P/
fi
if has trace; then
  cd /tmp; rm -rf /tmp/pb
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary > /tmp/pb.log 2>&1
  cd $GRAFT_REPO_ROOT; python3 tools/summarize_rocprof.py /tmp/pb $O/${R}_bench_kernel_stats python bench.py --no-cpu-baseline --no-secondary > /dev/null
  head -24 $O/${R}_bench_kernel_stats.md | cut -c1-160; cp $O/${R}_bench_kernel_stats.* $P/
fi
if has mfma; then
  cd $GRAFT_REPO_ROOT; bash tools/pmc_mfma.sh $R > /dev/null 2>&1; cp $O/${R}_pmc_mfma_a.txt $O/${R}_pmc_mfma_b.txt $P/ 2>/dev/null
  export RAFT_DTYPE=bf16; bash tools/pmc_mfma_conv.sh $R > /dev/null 2>&1
  export RAFT_DTYPE=bf16x3; bash tools/pmc_mfma_conv.sh ${R}x3 > /dev/null 2>&1; unset RAFT_DTYPE
  tail -14 $O/${R}_pmc_mfma_conv_a.txt $O/${R}x3_pmc_mfma_conv_a.txt
fi
if has x3; then
  cd $GRAFT_REPO_ROOT; bash tools/x3_prof.sh bf16x3 ${R}_x3 | head -22
  cp $O/${R}_x3_raft_stats.md $P/${R}_raft_bf16x3_kernel_stats.md; cp $O/${R}_x3_iter_trace.log $P/${R}_raft_bf16x3_iter_trace.log
  bash tools/x3_prof.sh bf16 ${R}_b16 | head -16
  cp $O/${R}_b16_raft_stats.md $P/${R}_raft_bf16_kernel_stats.md; cp $O/${R}_b16_iter_trace.log $P/${R}_raft_bf16_iter_trace.log
fi
if has single; then
  cd $GRAFT_REPO_ROOT; bash tools/single_clip_prof.sh ${R}_single_clip | head -30
  cp $O/${R}_single_clip_kernel_stats.md $O/${R}_single_clip_kernel_stats.json $P/
fi
if has vit; then
  cd $GRAFT_REPO_ROOT; bash tools/vit_prof.sh ${R}_vit
  cp $O/${R}_vit_fold1_stats.md $P/${R}_vit_stage_kernel_stats.md; cp $O/${R}_vit_fold0_stats.md $P/${R}_vit_stage_kernel_stats_unfolded_ln.md
fi
if has tests; then
  cd $GRAFT_REPO_ROOT
  python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" > $O/${R}_gpu_tests.log; cat $O/${R}_gpu_tests.log; cp $O/${R}_gpu_tests.log $P/
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
fi
