#!/bin/bash
# GPU box: the round-3 same-box A/B logs and traces quoted in DESIGN.md (outputs in gpurun_out/, copied to profiles/)
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; cd $GRAFT_REPO_ROOT
{ echo "# persistent kernel (gemm_pp.hip) vs one tile per workgroup (VTGB_GEMM_OLD=1), conv64 on in both: bench A / B / A / B"; bash tools/ab_bench.sh; } > $O/r03_exp_ab_persistent.log 2>&1
{ echo "# conv64.hip on / off (VTGB_CONV64=1 / 0): tools/raft_bench.py 31, A / B / A / B"; for v in 1 0 1 0; do echo -n "VTGB_CONV64=$v  "; VTGB_CONV64=$v python tools/raft_bench.py 31 2>&1 | tail -n 1; done; } > $O/r03_exp_ab_conv64.log 2>&1
{ echo "# bench.py --overlap (decode on a side stream under the next batch) vs default"; for f in "" "--overlap"; do python bench.py $f --no-cpu-baseline --no-secondary --no-prof 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$f', d['value'], 'clips/s', d['ms_per_step'], 'ms/step')"; done; python bench.py --stage-times --no-cpu-baseline --no-secondary --no-prof 2>&1 | grep "ms/step by stage"; } > $O/r03_exp_overlap_stages.log 2>&1
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/rt
rocprofv3 --kernel-trace --output-format csv -d /tmp/rt -- python3 $GRAFT_REPO_ROOT/tools/raft_bench.py 31 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; { echo "# one RAFT refinement iteration at the bench batch (31 clips, 2.31 M coarse pixels): tools/raft_iter_trace.py"; python3 tools/raft_iter_trace.py /tmp/rt; } > $O/r03_raft_iter_trace.log 2>&1
{ python tools/exp/llm_gemm_bench.py; python tools/exp/llm_gemm_bench.py 31868; } 2>&1 | grep "M=" > $O/r03_exp_llm_gemm_vs_hipblaslt.log
python tools/exp/prefill_bench.py 2>&1 | grep "B=" > $O/r03_exp_prefill.log
python tools/exp/skinny_bench.py 124 32 1 2>&1 | grep "M=\|sweep" > $O/r03_exp_skinny.log
python tools/exp/attn_bench.py 2>&1 | tail -n 1 > $O/r03_exp_attn.log
python tools/configs_check.py > $O/r03_configs_check.log 2>&1
cp $O/r03_exp_*.log $O/r03_raft_iter_trace.log $O/r03_configs_check.log profiles/ 2>/dev/null
tail -n 3 $O/r03_exp_ab_persistent.log $O/r03_exp_ab_conv64.log $O/r03_configs_check.log
