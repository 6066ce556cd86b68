// Instantiations of the all-matrix-core DS-TCN h256 kernel.  See ds256_mm.hip.h.
#include "ds256_mm.hip.h"
namespace wekws {
int launch_ds256_mm(int nt, const StackParams& P, const CallArgs& A, uint32_t head_a16, hipStream_t stream) {
  if (P.ksize != 8) return -4;
  switch (nt) {
    case 1: return launch_ds256_mm_nt<1>(P, A, head_a16, stream);
    case 2: return launch_ds256_mm_nt<2>(P, A, head_a16, stream);
    case 4: return launch_ds256_mm_nt<4>(P, A, head_a16, stream);
    case 7: return launch_ds256_mm_nt<7>(P, A, head_a16, stream);
    default: return -1;
  }
}
}  // namespace wekws
