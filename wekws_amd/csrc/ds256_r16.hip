// Instantiations of the role-split 16-wave DS-TCN h256 kernel.  See ds256_r16.hip.h.
#include "ds256_r16.hip.h"
namespace wekws {
int launch_ds256_r16(int nt, bool split, const StackParams& P, const CallArgs& A, hipStream_t stream) {
  if (P.ksize != 8) return -4;
  switch (nt) {
    case 1: return launch_ds256_r16_nt<1>(split, P, A, stream);
    case 2: return launch_ds256_r16_nt<2>(split, P, A, stream);
    case 4: return launch_ds256_r16_nt<4>(split, P, A, stream);
    case 7: return launch_ds256_r16_nt<7>(split, P, A, stream);
    default: return -1;
  }
}
}  // namespace wekws
