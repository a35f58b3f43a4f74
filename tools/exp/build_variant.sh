#!/bin/bash
# build_variant.sh NAME SRC.hip [-DFLAG ...]: libvtgb with ONE source rebuilt under extra flags -> tools/exp/variants/libvtgb_NAME.so
# (select it with VTGB_LIB=...; the other objects are the ones of the last regular build)
set -e
cd "$(dirname "$0")/../.."
name="$1"; src="$2"; shift 2
mkdir -p tools/exp/variants
obj="tools/exp/variants/${name}_${src%.hip}.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-function -I include -I videotgb_amd/csrc "$@" -c "videotgb_amd/csrc/$src" -o "$obj"
others=$(ls videotgb_amd/build/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "tools/exp/variants/libvtgb_${name}.so" $obj $others -ldl
rm -f "$obj"
echo "tools/exp/variants/libvtgb_${name}.so"
