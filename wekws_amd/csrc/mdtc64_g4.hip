// Instantiations of the register-resident MDTC h64 kernel (one utterance per 4-wave workgroup).  See mdtc64_g4.hip.h.
#include "mdtc64_g4.hip.h"
namespace wekws {
template <int NT, bool SPLIT, bool POOLED, bool ALIGNED>
static int launch_g4(const StackParams& P, const CallArgs& A, hipStream_t stream) {
  constexpr int LDS = 2 * Plane<64, 16 * NT>::BYTES;
  hipLaunchKernelGGL((mdtc64_g4_kernel<NT, SPLIT, POOLED, ALIGNED>), dim3(A.B), dim3(kG4Threads), LDS, stream, P, A);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
template <int NT, bool SPLIT, bool POOLED>
static int launch_g4_a(const StackParams& P, const CallArgs& A, hipStream_t stream) {
  if constexpr (NT == 1) return launch_g4<NT, SPLIT, POOLED, true>(P, A, stream);     // (NT = 1 divides every T)
  else return A.T % NT == 0 ? launch_g4<NT, SPLIT, POOLED, true>(P, A, stream) : launch_g4<NT, SPLIT, POOLED, false>(P, A, stream);
}
template <int NT>
static int launch_g4_nt(bool split, bool pooled, const StackParams& P, const CallArgs& A, hipStream_t stream) {
  if (pooled) return split ? launch_g4_a<NT, true, true>(P, A, stream) : launch_g4_a<NT, false, true>(P, A, stream);
  return split ? launch_g4_a<NT, true, false>(P, A, stream) : launch_g4_a<NT, false, false>(P, A, stream);
}
int launch_mdtc64_g4(int nt, bool split, const StackParams& P, const CallArgs& A, hipStream_t stream) {
  const bool linear = P.head == HEAD_LINEAR && P.odim <= 2;
  const bool pooled = (P.head == HEAD_GLOBAL || P.head == HEAD_LAST) && P.head_hidden <= 448;   // (64 + hidden floats of LDS)
  const bool ok = P.ksize == 5 && !A.in_cache && (linear || pooled) && P.kpre16 <= 96 && P.idim % 8 == 0 &&
                  (reinterpret_cast<uintptr_t>(A.x) & 15) == 0 && A.xs_b % 4 == 0;
  if (!ok) return -4;
  switch (nt) {
    case 1: return launch_g4_nt<1>(split, pooled, P, A, stream);
    case 2: return launch_g4_nt<2>(split, pooled, P, A, stream);
    case 4: return launch_g4_nt<4>(split, pooled, P, A, stream);
    case 7: return launch_g4_nt<7>(split, pooled, P, A, stream);
    default: return -4;
  }
}
}  // namespace wekws
