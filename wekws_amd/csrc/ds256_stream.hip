// Instantiations of the streaming (LDS-resident cache) 16-wave DS-TCN h256 kernel.  See ds256_stream.hip.h.
#include "ds256_stream.hip.h"
namespace wekws {
int launch_ds256_stream(bool split, const StackParams& P, const CallArgs& A, hipStream_t stream) {
  if (P.ksize != 8 || A.T > 16 || (256 * P.cache_len) % 4 != 0 || 256 * P.cache_len > 7 * 4 * 1024 || ds256_stream_lds_bytes(P.cache_len) > 160 * 1024) return -4;
  return split ? launch_ds256_stream_s<true>(P, A, stream) : launch_ds256_stream_s<false>(P, A, stream);
}
}  // namespace wekws
