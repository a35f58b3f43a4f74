#!/bin/bash
# GPU box, end of a round: the whole -m gpu suite, smoke(), all five configs, one RAFT refinement iteration per launch, one encoder call
# per launch -> gpurun_out/ (copy what is to be judged into profiles/)
R=${1:-r04}
O=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" > $O/gpu_tests_$R.log; cat $O/gpu_tests_$R.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python tools/configs_check.py > $O/${R}_configs_check.log 2>&1
grep -E "^C[1-5]|device time|prefix-graph" $O/${R}_configs_check.log | cut -c1-300
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/rt
rocprofv3 --kernel-trace --output-format csv -d /tmp/rt -- python3 $GRAFT_REPO_ROOT/tools/raft_bench.py 31 > /tmp/rt.log 2>&1; tail -1 /tmp/rt.log
python3 $GRAFT_REPO_ROOT/tools/raft_iter_trace.py /tmp/rt > $O/${R}_raft_iter_trace.log; cat $O/${R}_raft_iter_trace.log
python3 $GRAFT_REPO_ROOT/tools/exp/enc_trace.py > $O/${R}_encoder_trace.log 2>&1
