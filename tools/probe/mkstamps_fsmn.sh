#!/bin/bash
# dev: build build/abl/libwekws_a9.so = product library with the FSMN kernel instrumented with clock64() stamps
set -e
cd /root/repo
mkdir -p /tmp/abl build/abl
python3 - <<'PY'
s=open('/root/repo/wekws_amd/csrc/fsmn_f16.hip.h').read()
s=s.replace('#include "conv_stack_f16.hip.h"','#include "/root/repo/wekws_amd/csrc/conv_stack_f16.hip.h"')
s=s.replace('''  const int frag_off = (lq * TT + l15) * 16;
''','''  const int frag_off = (lq * TT + l15) * 16;
  long long stamp[24]; int ns = 0;
#define STAMP() do { stamp[ns++] = clock64(); } while (0)
  STAMP();
''',1)
s=s.replace('  __syncthreads();\n','  __syncthreads(); STAMP();\n')
s=s.replace('    __syncthreads();\n','    __syncthreads(); STAMP();\n')
s=s.replace('''      // new cache = last P valid columns of x_pad''','''      STAMP();
      // new cache = last P valid columns of x_pad''')
s=s.replace("""lane, wave, store_y);
  }
}
""","""lane, wave, store_y);
  }
  __syncthreads(); STAMP();
  if (tid == 0 && (blockIdx.x == 0 || blockIdx.x == 300)) {
    float* dbg = A.out_cache + int64_t(b0) * P.proj * Pc * L;
    for (int i = 1; i < ns; ++i) dbg[i - 1] = float(stamp[i] - stamp[i - 1]);
    dbg[ns - 1] = float(stamp[ns - 1] - stamp[0]);
  }
}
""")
open('/tmp/abl/fsmn_f16.hip.h','w').write(s)
PY
cp wekws_amd/csrc/fsmn_f16.hip /tmp/abl/
(cd /tmp/abl && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -c fsmn_f16.hip -o /root/repo/build/abl/fsmn_f16_a9.o 2>&1 | grep -E "rror:" || true)
rm -f build/abl/*.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/abl/libwekws_a9.so $(ls wekws_amd/lib/obj/*.o | grep -v fsmn_f16.o) build/abl/fsmn_f16_a9.o
ls -la build/abl/libwekws_a9.so
