#!/bin/bash
# GPU box: MFMA-pipe and wait counters of the CONVOLUTION kernels (implicit-GEMM instantiations, conv64, the fused GRU half-step) over one RAFT
# pass at the bench's batch (tools/conv_pmc.py, 31 clips).  Two SQ passes (8 SQ slots per pass), program straight after `--` (no env / shell hop).
# MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8); SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512 = executed FLOPs.
R=${1:-r04}
B=${2:-31}
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pmc_a /tmp/pmc_b
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_a -- python3 $GRAFT_REPO_ROOT/tools/conv_pmc.py $B > /tmp/pmc_a.log 2>&1
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/pmc_b -- python3 $GRAFT_REPO_ROOT/tools/conv_pmc.py $B > /tmp/pmc_b.log 2>&1
tail -2 /tmp/pmc_a.log /tmp/pmc_b.log
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pmc_a > $O/${R}_pmc_mfma_conv_a.txt
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pmc_b > $O/${R}_pmc_mfma_conv_b.txt
python3 - <<PY >> $O/${R}_pmc_mfma_conv_a.txt
import re, ast
print("# MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 * GRBM_GUI_ACTIVE / 8), per kernel (means per dispatch)")
for line in open("$O/${R}_pmc_mfma_conv_a.txt").read().splitlines():
    if "{" not in line or line.startswith("#"): continue
    name, d = line[:line.index("{")].strip(), ast.literal_eval(line[line.index("{"):])
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d and d["GRBM_GUI_ACTIVE"][0] > 0:
        busy = d["SQ_VALU_MFMA_BUSY_CYCLES"][0] / (1024.0 * d["GRBM_GUI_ACTIVE"][0] / 8.0)
        print(f"#   {busy * 100:5.1f} %  x{d['GRBM_GUI_ACTIVE'][1]:5d}  {name}")
PY
cat $O/${R}_pmc_mfma_conv_a.txt | tail -20
cp $O/${R}_pmc_mfma_conv_a.txt $O/${R}_pmc_mfma_conv_b.txt $GRAFT_REPO_ROOT/profiles/ 2>/dev/null
