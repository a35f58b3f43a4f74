#!/bin/bash
# GPU box: MFMA-utilisation counters of the plain GEMM kernel on the four ViT-g layer shapes (tools/gemm_pmc.py).
# Two SQ passes (8 SQ slots per pass); outputs a per-kernel mean table under gpurun_out/.
R=${1:-r02}
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pm_a /tmp/pm_b
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d /tmp/pm_a -- python3 $GRAFT_REPO_ROOT/tools/gemm_pmc.py > /tmp/pm_a.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_MFMA --output-format csv -d /tmp/pm_b -- python3 $GRAFT_REPO_ROOT/tools/gemm_pmc.py > /tmp/pm_b.log 2>&1
tail -3 /tmp/pm_a.log /tmp/pm_b.log
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pm_a > $O/${R}_pmc_mfma_a.txt
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pm_b > $O/${R}_pmc_mfma_b.txt
cat $O/${R}_pmc_mfma_a.txt $O/${R}_pmc_mfma_b.txt
