#!/bin/bash
# Development tool: build variants of the 16-wave DS-TCN kernel (-DW16_VAR=n; add `#if W16_VAR == n` blocks to
# ds256_w16.hip.h while experimenting -- none are kept in the shipped source) into build/var/libN.so
# so that one gpurun call can A/B them on the same box:  WEKWS_HIP_LIB=build/var/lib1.so python tools/time_ds.py
set -e
cd "$(dirname "$0")/.."
mkdir -p build/var
for v in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DW16_VAR=$v -c wekws_amd/csrc/ds256_w16.hip -o build/var/w16_$v.o
  objs=$(ls wekws_amd/lib/obj/*.o | grep -v ds256_w16.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/var/lib$v.so build/var/w16_$v.o $objs
done
ls -la build/var/*.so
