#!/bin/bash
# usage: tools/kernel_resources.sh <file.hip> [name filter]   -- VGPRs / spills / scratch per kernel (gfx950)
f=$1; pat=${2:-.}
root="$(cd "$(dirname "$0")/.." && pwd)"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I"$root/include" -I"$root/videotgb_amd/csrc" -c "$f" -o /dev/null \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|VGPRs Spill|ScratchSize" | sed 's/.*remark: *//; s/ \[-Rpass.*//' | \
  awk '/Function Name/{if(n)print n, v, s, sc; n=$3} /^VGPRs:/{v="vgpr="$2} /VGPRs Spill/{s="spill="$3} /ScratchSize/{sc="scratch="$NF} END{print n, v, s, sc}' | grep -E "$pat"
