// Instantiations of the pipelined register-resident DS-TCN h256 kernel.  See ds256_p16.hip.h.
#include "ds256_p16.hip.h"
namespace wekws {
template <int NT>
static int launch_nt(bool split, const StackParams& P, const CallArgs& A, hipStream_t stream) {
  return split ? launch_ds256_p16_nts<NT, true>(P, A, stream) : launch_ds256_p16_nts<NT, false>(P, A, stream);
}
int launch_ds256_p16(int nt, bool split, const StackParams& P, const CallArgs& A, hipStream_t stream) {
  if (P.ksize != 8 || A.in_cache || 2 * P.nblocks + 4 > kAmaxCells) return -4;
  switch (nt) {
    case 4: return launch_nt<4>(split, P, A, stream);
    case 7: return launch_nt<7>(split, P, A, stream);
    default: return launch_ds256_g16(nt, split, P, A, stream);
  }
}
}  // namespace wekws
