#!/bin/bash
# usage: tools/kernel_regs.sh <object.o> [name-filter]   -- VGPR / AGPR / SGPR / scratch / LDS of every gfx950 kernel in an object
set -e
obj=$1; filt=${2:-.}
tmp=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$tmp/fb.bin $obj
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$tmp/fb.bin --output=$tmp/dev.co --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $tmp/dev.co | python3 -c "
import sys,re
txt=sys.stdin.read()
for blk in txt.split('- .agpr_count')[1:]:
    g=lambda k:(re.search(r'\.'+k+r':\s+(\S+)',blk) or [None,'?'])[1]
    name=g('name')
    if re.search(sys.argv[1],name): print(f\"{name[:90]:90s} vgpr={g('vgpr_count'):>4s} agpr={blk.split()[0]:>3s} sgpr={g('sgpr_count'):>4s} scratch={g('private_segment_fixed_size'):>5s} lds={g('group_segment_fixed_size')}\")
" "$filt"
rm -rf $tmp
