#!/bin/bash
set -e
cd /root/repo
mkdir -p /tmp/ablmm build/abl
python3 - <<'PY'
s=open('/root/repo/wekws_amd/csrc/ds256_mm.hip.h').read()
s=s.replace('#include "ds256_w16.hip.h"','#include "/root/repo/wekws_amd/csrc/ds256_w16.hip.h"')
s=s.replace('  f32x4 acc[1][NT];\n','''  f32x4 acc[1][NT];
  long long tph[8] = {0,0,0,0,0,0,0,0}; long long tlast = clock64();
#define PH(id) do { long long now_ = clock64(); tph[id] += now_ - tlast; tlast = now_; } while (0)
''',1)
s=s.replace('''    if (P.nblocks > 0) stage_halo(P.blocks[0], 0);
    __syncthreads();
  }
''','''    if (P.nblocks > 0) stage_halo(P.blocks[0], 0);
    __syncthreads();
  }
  PH(6);
''')
s=s.replace("      // ---- the interval's slice of the new streaming cache","      PH(0);\n      // ---- the interval's slice of the new streaming cache")
s=s.replace("      // ---- depthwise on the matrix cores: tiles (ct, ft), ft = fq + 4 rd.","      PH(1);\n      // ---- depthwise on the matrix cores: tiles (ct, ft), ft = fq + 4 rd.")
s=s.replace("      __syncthreads();\n      // ---- pointwise: two K steps","      PH(2);\n      __syncthreads();\n      PH(3);\n      // ---- pointwise: two K steps")
s=s.replace("      else if (bi + 1 < P.nblocks) { load_taps(bdn, 0); stage_halo(bdn, 0); }\n      __syncthreads();\n    }\n","      else if (bi + 1 < P.nblocks) { load_taps(bdn, 0); stage_halo(bdn, 0); }\n      PH(4);\n      __syncthreads();\n      PH(5);\n    }\n")
s=s.replace('''      *reinterpret_cast<f16x4*>(hp + HP) = vl;
    }
    __syncthreads();
  }
''','''      *reinterpret_cast<f16x4*>(hp + HP) = vl;
    }
    __syncthreads();
    PH(7);
  }
''')
s=s.replace('''          A.y[int64_t(b) * A.ys_b + int64_t(t) * K + k] = v;
        }
      }
    }
  }
}''','''          A.y[int64_t(b) * A.ys_b + int64_t(t) * K + k] = v;
        }
      }
    }
  }
  if (tid == 0 && b == 0 && A.out_cache) { for (int i = 0; i < 8; ++i) A.out_cache[i] = float(tph[i]); }
}''')
open('/tmp/ablmm/ds256_mm.hip.h','w').write(s)
PY
cp wekws_amd/csrc/ds256_mm.hip /tmp/ablmm/
(cd /tmp/ablmm && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -c ds256_mm.hip -o /root/repo/build/abl/mm_st.o 2>&1 | grep -E "rror" -A3 || true)
rm -f build/abl/*.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/abl/libwekws_mm.so $(ls wekws_amd/lib/obj/*.o | grep -v ds256_mm.o) build/abl/mm_st.o
ls -la build/abl/libwekws_mm.so
