// Instantiations of the register-resident DS-TCN h256 kernel.  See ds256_g16.hip.h.
#include "ds256_g16.hip.h"
namespace wekws {
template <int NT>
static int launch_nt(bool split, const StackParams& P, const CallArgs& A, hipStream_t stream, int cus) {
  return split ? launch_ds256_g16_nts<NT, true>(P, A, stream, cus) : launch_ds256_g16_nts<NT, false>(P, A, stream, cus);
}
int launch_ds256_g16(int nt, bool split, const StackParams& P, const CallArgs& A, hipStream_t stream, int cus) {
  if (P.ksize != 8) return -4;
  switch (nt) {
    case 1: return launch_nt<1>(split, P, A, stream, cus);
    case 2: return launch_nt<2>(split, P, A, stream, cus);
    case 4: return launch_nt<4>(split, P, A, stream, cus);
    case 7: return launch_nt<7>(split, P, A, stream, cus);
    default: return -1;
  }
}
}  // namespace wekws
