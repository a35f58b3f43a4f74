#!/bin/bash
# RAFT stage in one mode under rocprofv3: per-clip time, one refinement iteration's launches, kernel stats.  usage: tools/x3_prof.sh [dtype] [tag]
DT=${1:-bf16x3}; TAG=${2:-x3}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/rt_$TAG
RAFT_DTYPE=$DT rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rt_$TAG -- python3 $GRAFT_REPO_ROOT/tools/raft_bench.py 31 > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_raft_bench.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/raft_iter_trace.py /tmp/rt_$TAG > gpurun_out/${TAG}_iter_trace.log
python3 tools/summarize_rocprof.py /tmp/rt_$TAG gpurun_out/${TAG}_raft_stats raft_bench 31 > /dev/null
grep "RAFT all" gpurun_out/${TAG}_raft_bench.log; cat gpurun_out/${TAG}_iter_trace.log; head -26 gpurun_out/${TAG}_raft_stats.md
