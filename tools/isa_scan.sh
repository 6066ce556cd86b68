#!/bin/bash
# Development tool: compile every translation unit of the library to gfx950 assembly (build/isa/) and list the packed-f32 instructions
# with the operand-select pattern of wekws_amd/csrc/pk_safe.hip.h (the test of record works on the built library: tests/test_isa_hazard.py).
cd "$(dirname "$0")/.."
mkdir -p build/isa && rm -f build/isa/*.s
for f in wekws_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 --cuda-device-only -S -o build/isa/$(basename ${f%.hip}).s $f 2> /dev/null &
done
wait
python3 - <<'PY'
import glob, re, sys
sys.path.insert(0, "tests")
from test_isa_hazard import hazardous_line
tot = 0
for path in sorted(glob.glob('build/isa/*.s')):
    kern = None
    for line in open(path):
        m = re.match(r'^(_Z\w+):', line)
        if m:
            kern = m[1]
        elif 'v_pk_' in line and hazardous_line(line):
            tot += 1
            print(path, kern[:80], line.strip())
print("hazardous packed-f32 instructions:", tot)
PY
