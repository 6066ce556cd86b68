#!/bin/bash
# Which kernels serve the C++ runtime's `kws_main 40 80 <model> <wav>` (the Android caller's 80-frame chunks,
# runtime/android/app/src/main/cpp/wekws.cc:84-97) on the reference's shipped trained model (converted once by tests/tools/make_ref_asset.py
# into build/ref_asset/kws.wekwship)?  rocprofv3 kernel trace -> gpurun_out/<tag>_kws_main_chunk80_kernels.txt
set -u
tag=${1:-r05h}
root=${GRAFT_REPO_ROOT:-$PWD}; out=$root/gpurun_out; mkdir -p $out; cd $root
python - <<'PY'
import os, struct, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from wekws_amd.utils import synth
pcm = np.clip(np.round(synth.synth_pcm(1, 160000, seed=11, kind="noise")[0] * 0.5), -32768, 32767).astype(np.int16)
with open("gpurun_out/_trace.wav", "wb") as f:
    f.write(b"RIFF" + struct.pack("<I", 36 + pcm.nbytes) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 1, 16000, 32000, 2, 16) + b"data" + struct.pack("<I", pcm.nbytes))
    f.write(pcm.tobytes())
PY
model=build/ref_asset/kws.wekwship
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/prof_kwsmain -o t -- $root/runtime/build/kws_main 40 80 $root/$model $root/gpurun_out/_trace.wav > $out/prof_kwsmain.log 2>&1
cd $root
python tools/prof_summary.py $(find $out/prof_kwsmain -name "*_results.db" | sort) > $out/${tag}_kws_main_chunk80_kernels.txt 2>&1
rm -rf $out/prof_kwsmain $out/_trace.wav
cat $out/${tag}_kws_main_chunk80_kernels.txt | cut -c1-150
