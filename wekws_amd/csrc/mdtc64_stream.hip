// Instantiations of the MDTC h64 streaming-step kernel.  See mdtc64_stream.hip.h.
#include "mdtc64_stream.hip.h"
namespace wekws {
int launch_mdtc64_stream(bool split, const StackParams& P, const CallArgs& A, hipStream_t stream) {
  if (P.ksize != 5 || A.T > 16 || P.kpre16 > 128 || !(P.idim % 8 == 0 && (reinterpret_cast<uintptr_t>(A.x) & 15) == 0 && A.xs_b % 4 == 0) || (64 * P.cache_len) % 4 != 0 || mdtc64_stream_lds_bytes(P.cache_len) > 158 * 1024)
    return -4;
  return split ? launch_mdtc64_stream_s<true>(P, A, stream) : launch_mdtc64_stream_s<false>(P, A, stream);
}
}  // namespace wekws
