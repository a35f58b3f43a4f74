#!/bin/bash
# GPU box: everything under profiles/ for one round (run from the repo root through gpurun; outputs in gpurun_out/)
R=${1:-r02}
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
# 1. PMC: HBM traffic of the conv family (one RAFT pass, 31 clips = the bench batch) and of the plain GEMMs; one counter per pass
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pc_$c /tmp/pg_$c
  timeout 900 rocprofv3 --pmc $c --output-format csv -d /tmp/pc_$c -- python3 $GRAFT_REPO_ROOT/tools/conv_pmc.py 31 > /tmp/pc_$c.log 2>&1
  timeout 600 rocprofv3 --pmc $c --output-format csv -d /tmp/pg_$c -- python3 $GRAFT_REPO_ROOT/tools/gemm_pmc.py > /tmp/pg_$c.log 2>&1
done
python3 $GRAFT_REPO_ROOT/tools/pmc_traffic.py /tmp/pc_FETCH_SIZE /tmp/pc_WRITE_SIZE $O/${R}_pmc_traffic_conv.json conv > /dev/null
python3 $GRAFT_REPO_ROOT/tools/pmc_traffic.py /tmp/pg_FETCH_SIZE /tmp/pg_WRITE_SIZE $O/${R}_pmc_traffic_gemm.json > /dev/null
mkdir -p $GRAFT_REPO_ROOT/profiles; cp $O/${R}_pmc_traffic_*.json $GRAFT_REPO_ROOT/profiles/ 2>/dev/null
# 2. the bench line (reads the PMC files just written)
cd $GRAFT_REPO_ROOT; python3 bench.py > $O/${R}_bench.json 2> $O/${R}_bench.err; tail -c 600 $O/${R}_bench.json
# 3. kernel trace of the same command
cd /tmp; rm -rf /tmp/pb
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary > /tmp/pb.log 2>&1
cd $GRAFT_REPO_ROOT; python3 tools/summarize_rocprof.py /tmp/pb $O/${R}_bench_kernel_stats python bench.py --no-cpu-baseline --no-secondary > /dev/null
head -20 $O/${R}_bench_kernel_stats.md
# 4. MFMA utilisation counters of the plain GEMM kernel on the ViT-g layer shapes
cd $GRAFT_REPO_ROOT; bash tools/pmc_mfma.sh $R > /dev/null 2>&1; cp $O/${R}_pmc_mfma_a.txt $O/${R}_pmc_mfma_b.txt $GRAFT_REPO_ROOT/profiles/ 2>/dev/null
