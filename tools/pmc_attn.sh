#!/bin/bash
# GPU box: SQ counters of the ViT-g attention kernel at the bench's batch (tools/exp/attn_bench.py 992): MFMA pipe busy, wave / wait cycles,
# LDS activity and bank conflicts.  Three SQ passes -> gpurun_out/<R>_pmc_attn.txt
R=${1:-r04}
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pa_a /tmp/pa_b /tmp/pa_c
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU --output-format csv -d /tmp/pa_a -- python3 $GRAFT_REPO_ROOT/tools/exp/attn_bench.py 992 > /tmp/pa_a.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS --output-format csv -d /tmp/pa_b -- python3 $GRAFT_REPO_ROOT/tools/exp/attn_bench.py 992 > /tmp/pa_b.log 2>&1
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA --output-format csv -d /tmp/pa_c -- python3 $GRAFT_REPO_ROOT/tools/exp/attn_bench.py 992 > /tmp/pa_c.log 2>&1
for d in a b c; do python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pa_$d; done > $O/${R}_pmc_attn.txt 2>&1
grep -i "attn_bf16\|^#" $O/${R}_pmc_attn.txt
