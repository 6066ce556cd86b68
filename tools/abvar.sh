#!/bin/bash
# Development tool: build variants of one kernel translation unit into build/var/lib<TAG>.so so that one gpurun call can
# A/B them on the same box:   tools/abvar.sh <unit> <tag> [-Dflags...]      e.g.  tools/abvar.sh ds256_r16 st -DWEKWS_R16_STAMPS
#   WEKWS_HIP_LIB=build/var/libst.so python tools/time_ds.py
set -e
cd "$(dirname "$0")/.."
unit=$1; tag=$2; shift 2
mkdir -p build/var
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 "$@" -c wekws_amd/csrc/$unit.hip -o build/var/${unit}_$tag.o
objs=$(ls wekws_amd/lib/obj/*.o | grep -v "/$unit.o" | grep -v "_hooks.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/var/lib$tag.so build/var/${unit}_$tag.o $objs
ls -la build/var/lib$tag.so
