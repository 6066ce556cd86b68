#!/bin/bash
# dev: build/abl/libwekws_m.so = product library with conv_stack_f16 (MDTC) instrumented with per-phase clock64 sums
set -e
cd /root/repo
mkdir -p /tmp/ablm build/abl
python3 - <<'PY'
s=open('/root/repo/wekws_amd/csrc/conv_stack_f16.hip.h').read()
s=s.replace('#include "conv_stack.hip.h"','#include "/root/repo/wekws_amd/csrc/conv_stack.hip.h"')
# accumulators
s=s.replace('''  f32x4 acc[OW][NT];
  f32x4 zsum''','''  long long tph[8] = {0,0,0,0,0,0,0,0}; long long tlast = clock64();
#define PH(id) do { long long now_ = clock64(); tph[id] += now_ - tlast; tlast = now_; } while (0)
  f32x4 acc[OW][NT];
  f32x4 zsum''',1)
i0=s.index('  // ======================================= residual blocks')
head=s[:i0]; body=s[i0:]
head=head.replace('    __syncthreads();\n  }\n\n  // =====','    __syncthreads();\n  }\n  PH(5);\n\n  // =====')
# in body: label barriers in order
parts=body.split('__syncthreads();')
ids=[0,1,2,2,3,4,6,7]  # produce0 ; (n loop) sync a ; sync b ; (NBUF==1 path) ; mid ; block end ; zsum ; (none)
out=parts[0]
for k,p in enumerate(parts[1:]):
    out+='__syncthreads(); PH(%d);'%ids[min(k,len(ids)-1)]+p
body=out
body=body.replace('  conv_stack_head<KIND, C, NT>(P, A, hbuf, reinterpret_cast<float*>(slab), b0);\n}','''  PH(5);
  conv_stack_head<KIND, C, NT>(P, A, hbuf, reinterpret_cast<float*>(slab), b0);
  __syncthreads(); PH(7);
  if (tid == 0 && blockIdx.x == 0 && A.out_cache) { for (int i = 0; i < 8; ++i) A.out_cache[i] = float(tph[i]); }
}''')
open('/tmp/ablm/conv_stack_f16.hip.h','w').write(head+body)
PY
cp wekws_amd/csrc/conv_stack_f16_mdtc.hip /tmp/ablm/
(cd /tmp/ablm && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -c conv_stack_f16_mdtc.hip -o /root/repo/build/abl/csf16_mdtc_st.o 2>&1 | grep -E "rror" -A3 || true)
rm -f build/abl/*.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/abl/libwekws_m.so $(ls wekws_amd/lib/obj/*.o | grep -v conv_stack_f16_mdtc.o) build/abl/csf16_mdtc_st.o
ls -la build/abl/libwekws_m.so
