#!/bin/bash
# Development tool: build ablated variants of the library (WEKWS_ABLATE=1..4, results are wrong by design) into
# build/ablate/ and print the bench time of each.  Usage (GPU box): bash tools/ablate.sh
set -e
cd "$(dirname "$0")/.."
mkdir -p build/ablate
for v in ${ABLATE_SET:-0 1 2 3 4}; do
  if [ ! -f build/ablate/lib$v.so ]; then
    d=$(mktemp -d)
    for f in wekws_hip conv_stack_ds conv_stack_f16_ds; do
      /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DWEKWS_ABLATE=${v%%p*} $( [[ $v == *p* ]] && echo -DWEKWS_SETPRIO=${v##*p} ) -c wekws_amd/csrc/$f.hip -o $d/$f.o &
    done
    wait
    # tcn / mdtc launchers are not needed for the DS-TCN bench but the library must link: reuse the product objects
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/ablate/lib$v.so $d/wekws_hip.o $d/conv_stack_ds.o $d/conv_stack_f16_ds.o \
        wekws_amd/lib/obj/conv_stack_tcn.o wekws_amd/lib/obj/conv_stack_mdtc.o wekws_amd/lib/obj/conv_stack_f16_tcn.o wekws_amd/lib/obj/conv_stack_f16_mdtc.o wekws_amd/lib/obj/dense_stack_f16_tcn.o
  fi
done
