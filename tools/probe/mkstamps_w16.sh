#!/bin/bash
set -e
cd /root/repo
mkdir -p /tmp/ablw build/abl
python3 - <<'PY'
s=open('/root/repo/wekws_amd/csrc/ds256_w16.hip.h').read()
s=s.replace('#include "conv_stack_f16.hip.h"','#include "/root/repo/wekws_amd/csrc/conv_stack_f16.hip.h"')
s=s.replace('  f32x4 acc[1][NT];\n','''  f32x4 acc[1][NT];
  long long tph[8] = {0,0,0,0,0,0,0,0}; long long tlast = clock64();
#define PH(id) do { long long now_ = clock64(); tph[id] += now_ - tlast; tlast = now_; } while (0)
''',1)
s=s.replace('''  // ======================================= residual blocks''','''  PH(5);
  // ======================================= residual blocks''')
s=s.replace('''      // row r -> K step r>>5, k-octet (r&31)>>3, half (r&7) of the [k-octet][frame][8] planes''','''      PH(0);
      // row r -> K step r>>5, k-octet (r&31)>>3, half (r&7) of the [k-octet][frame][8] planes''')
s=s.replace('''      produce_iv(iv);
      load_dw(nx);
      __syncthreads();''','''      PH(6);
      produce_iv(iv);
      load_dw(nx);
      PH(1);
      __syncthreads();
      PH(2);''')
s=s.replace('''      load_a16<1>(a1, ap1 + (2 * nx + 1) * 128, 0);
      __syncthreads();
    }''','''      load_a16<1>(a1, ap1 + (2 * nx + 1) * 128, 0);
      PH(3);
      __syncthreads();
      PH(4);
    }''')
s=s.replace('''  conv_stack_head<KIND_DS, 256, NT, kW16Threads>(P, A, hbuf, reinterpret_cast<float*>(slab), b);
}''','''  PH(7);
  conv_stack_head<KIND_DS, 256, NT, kW16Threads>(P, A, hbuf, reinterpret_cast<float*>(slab), b);
  __syncthreads();
  if (tid == 0 && b == 0 && A.out_cache) { for (int i = 0; i < 8; ++i) A.out_cache[i] = float(tph[i]); }
}''')
open('/tmp/ablw/ds256_w16.hip.h','w').write(s)
PY
cp wekws_amd/csrc/ds256_w16.hip /tmp/ablw/
(cd /tmp/ablw && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -c ds256_w16.hip -o /root/repo/build/abl/w16_st.o 2>&1 | grep -E "rror" -A3 || true)
rm -f build/abl/*.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/abl/libwekws_w.so $(ls wekws_amd/lib/obj/*.o | grep -v ds256_w16.o) build/abl/w16_st.o
ls -la build/abl/libwekws_w.so
