#!/bin/bash
# dev: build/abl/libwekws_fb.so = library whose fbank kernel accumulates clock64() per phase (wave 0 of block 0) and
# writes the eight sums over the first eight output values
set -e
cd /root/repo
mkdir -p /tmp/ablfb build/abl
python3 - <<'PY'
s=open('/root/repo/wekws_amd/csrc/fbank.hip.h').read()
s=s.replace('  const int64_t total = int64_t(B) * nframes;','  long long tph[8] = {0,0,0,0,0,0,0,0}; long long tlast = clock64();\n#define PH(id) do { long long now_ = clock64(); tph[id] += now_ - tlast; tlast = now_; } while (0)\n  const int64_t total = int64_t(B) * nframes;',1)
s=s.replace('    // ---- DC removal (fbank.h:155-160)','    PH(7);\n    // ---- DC removal (fbank.h:155-160)')
s=s.replace('    // ---- pre-emphasis 0.97','    PH(0);\n    // ---- pre-emphasis 0.97')
s=s.replace('    // ---- 256-point complex FFT','    PH(1);\n    // ---- 256-point complex FFT')
s=s.replace('    // ---- real-FFT untangle + power','    PH(2);\n    // ---- real-FFT untangle + power')
s=s.replace('    // ---- mel (fbank.h:179-186)','    PH(3);\n    // ---- mel (fbank.h:179-186)')
s=s.replace('    // ---- log (fbank.h:187-190), store.','    PH(5);\n    // ---- log (fbank.h:187-190), store.')
s=s.replace('        feats[(b * nframes + fr) * P.num_bins + sbin[r]] = logf(fmaxf(e, FLT_EPSILON));\n      }\n    }\n    wave_sync();\n  }\n}','        feats[(b * nframes + fr) * P.num_bins + sbin[r]] = logf(fmaxf(e, FLT_EPSILON));\n      }\n    }\n    wave_sync();\n    PH(4);\n  }\n  if (threadIdx.x == 0 && blockIdx.x == 0) for (int i = 0; i < 8; ++i) feats[i] = float(tph[i]);\n}')
open('/tmp/ablfb/fbank.hip.h','w').write(s)
w=open('/root/repo/wekws_amd/csrc/wekws_hip.hip').read().replace('#include "fbank.hip.h"','#include "/tmp/ablfb/fbank.hip.h"')
for h in ["conv_stack.hip.h","conv_stack_f16.hip.h","dense_stack_f16.hip.h","ds256_w16.hip.h","ds256_mm.hip.h","mdtc64_w16.hip.h","fsmn_f16.hip.h","gru.hip.h","gru_f16.hip.h","splice.hip.h","topk.hip.h"]:
    w=w.replace('#include "%s"'%h,'#include "/root/repo/wekws_amd/csrc/%s"'%h)
w=w.replace('#include "../../include/wekws_hip.h"','#include "/root/repo/include/wekws_hip.h"')
open('/tmp/ablfb/wekws_hip.hip','w').write(w)
PY
(cd /tmp/ablfb && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -c wekws_hip.hip -o /root/repo/build/abl/wh_fb.o 2>&1 | grep -E "rror" -A3 || true)
rm -f build/abl/*.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/abl/libwekws_fb.so $(ls wekws_amd/lib/obj/*.o | grep -v "obj/wekws_hip.o") build/abl/wh_fb.o
ls -la build/abl/libwekws_fb.so
