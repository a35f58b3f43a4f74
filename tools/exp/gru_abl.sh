#!/bin/bash
# Timing-only ablations of gru_half_kernel (gru_fused.hip, GF_ABL bit set): builds one library per variant next to the product one
# (tools/exp/_build/libvtgb_abl_<n>.so, selected through VTGB_LIB) -- run on the build host; then on the GPU box:
#   for n in 0 1 2 4 8 ...; do VTGB_LIB=tools/exp/_build/libvtgb_abl_$n.so python3 tools/exp/gru_time.py; done
set -e
cd "$(dirname "$0")/../.."
mkdir -p tools/exp/_build
OBJS=$(ls videotgb_amd/build/*.o | grep -v gru_fused.o)
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -DGF_ABL=$n ${GF_EXTRA} -I include -I videotgb_amd/csrc -c videotgb_amd/csrc/gru_fused.hip -o tools/exp/_build/gru_fused_$n.o &
done
wait
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/_build/libvtgb_abl_$n.so $OBJS tools/exp/_build/gru_fused_$n.o -ldl
done
ls -la tools/exp/_build/*.so
