#!/bin/bash
set -e
cd /root/repo
mkdir -p /tmp/ablm16 build/abl
python3 - <<'PY'
s=open('/root/repo/wekws_amd/csrc/mdtc64_w16.hip.h').read()
s=s.replace('#include "conv_stack_f16.hip.h"','#include "/root/repo/wekws_amd/csrc/conv_stack_f16.hip.h"').replace('#include "ds256_w16.hip.h"','#include "/root/repo/wekws_amd/csrc/ds256_w16.hip.h"')
s=s.replace('  f32x4 acc[NTW], zsum[NTW];\n','''  f32x4 acc[NTW], zsum[NTW];
  long long tph[10] = {0,0,0,0,0,0,0,0,0,0}; long long tlast = clock64();
#define PH(id) do { long long now_ = clock64(); tph[id] += now_ - tlast; tlast = now_; } while (0)
''',1)
i0=s.index('  // ======================================= residual blocks')
head=s[:i0]+'  PH(8);\n'; body=s[i0:]
parts=body.split('__syncthreads();')
# barriers in body: after producer, after gemm1, after mid, after gemm2/epi, after zsum
out=parts[0]
ids=[(0,1),(2,3),(4,5),(6,7),(9,9)]
for k,p in enumerate(parts[1:]):
    a,b=ids[min(k,len(ids)-1)]
    out+='PH(%d); __syncthreads(); PH(%d);'%(a,b)+p
body=out
body=body.replace('  conv_stack_head<KIND_MDTC, 64, NT, kW16Threads>(P, A, hbuf, reinterpret_cast<float*>(slab), b0);\n}','''  conv_stack_head<KIND_MDTC, 64, NT, kW16Threads>(P, A, hbuf, reinterpret_cast<float*>(slab), b0);
  __syncthreads(); PH(9);
  if (tid == 0 && blockIdx.x == 0 && A.out_cache) { for (int i = 0; i < 10; ++i) A.out_cache[i] = float(tph[i]); }
}''')
open('/tmp/ablm16/mdtc64_w16.hip.h','w').write(head+body)
PY
cp wekws_amd/csrc/mdtc64_w16.hip /tmp/ablm16/
(cd /tmp/ablm16 && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -c mdtc64_w16.hip -o /root/repo/build/abl/m16_st.o 2>&1 | grep -E "rror" -A3 || true)
rm -f build/abl/*.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/abl/libwekws_m16.so $(ls wekws_amd/lib/obj/*.o | grep -v mdtc64_w16.o) build/abl/m16_st.o
ls -la build/abl/libwekws_m16.so
