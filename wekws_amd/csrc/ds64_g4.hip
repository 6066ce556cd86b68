// Instantiations of the register-resident DS-TCN h64 kernel (one utterance per 4-wave workgroup).  See ds64_g4.hip.h.
#include "ds64_g4.hip.h"
namespace wekws {
template <int NT, bool SPLIT, bool ALIGNED>
static int launch_d4(const StackParams& P, const CallArgs& A, hipStream_t stream) {
  constexpr int LDS = 2 * Plane<64, 16 * NT>::BYTES;
  hipLaunchKernelGGL((ds64_g4_kernel<NT, SPLIT, ALIGNED>), dim3(A.B), dim3(kG4Threads), LDS, stream, P, A);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
template <int NT, bool SPLIT, bool ALIGNED>
static int launch_d4_ctx(const StackParams& P, const CallArgs& A, hipStream_t stream) {
  constexpr int LDS = 2 * Plane<64, 16 * NT>::BYTES;
  hipLaunchKernelGGL((ds64_g4_kernel<NT, SPLIT, ALIGNED, true>), dim3(A.B), dim3(kG4Threads), LDS, stream, P, A);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
template <int NT>
static int launch_d4_ctx_nt(bool split, const StackParams& P, const CallArgs& A, hipStream_t stream) {
  const bool al = A.T % NT == 0;
  if (split) return al ? launch_d4_ctx<NT, true, true>(P, A, stream) : launch_d4_ctx<NT, true, false>(P, A, stream);
  return al ? launch_d4_ctx<NT, false, true>(P, A, stream) : launch_d4_ctx<NT, false, false>(P, A, stream);
}
template <int NT>
static int launch_d4_nt(bool split, const StackParams& P, const CallArgs& A, hipStream_t stream) {
  const bool al = NT == 1 || A.T % NT == 0;
  if (split) return al ? launch_d4<NT, true, true>(P, A, stream) : launch_d4<NT, true, false>(P, A, stream);
  return al ? launch_d4<NT, false, true>(P, A, stream) : launch_d4<NT, false, false>(P, A, stream);
}
int launch_ds64_g4(int nt, bool split, const StackParams& P, const CallArgs& A, hipStream_t stream) {
  const bool ok = P.ksize == 8 && P.head == HEAD_LINEAR && P.odim <= 2 && P.kpre16 <= 96 && P.idim % 8 == 0 &&
                  (reinterpret_cast<uintptr_t>(A.x) & 15) == 0 && A.xs_b % 4 == 0;
  if (!ok) return -4;
  if (A.in_cache)                                            // a later chunk of a stream: the context variant
    return nt <= 4 ? launch_d4_ctx_nt<4>(split, P, A, stream) : nt == 7 ? launch_d4_ctx_nt<7>(split, P, A, stream) : -4;
  switch (nt) {
    case 1: return launch_d4_nt<1>(split, P, A, stream);
    case 2: return launch_d4_nt<2>(split, P, A, stream);
    case 4: return launch_d4_nt<4>(split, P, A, stream);
    case 7: return launch_d4_nt<7>(split, P, A, stream);
    default: return -4;
  }
}
}  // namespace wekws
