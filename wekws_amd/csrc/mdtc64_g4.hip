// Instantiations of the register-resident MDTC kernels (one utterance per workgroup of C / 16 waves).  See mdtc64_g4.hip.h.
#include "mdtc64_g4.hip.h"
namespace wekws {
template <int C, int NT, bool SPLIT, bool POOLED, bool ALIGNED>
static int launch_g4(const StackParams& P, const CallArgs& A, hipStream_t stream) {
  constexpr int LDS = 2 * Plane<C, 16 * NT>::BYTES;
  hipLaunchKernelGGL((mdtc_g4_kernel<C, NT, SPLIT, POOLED, ALIGNED>), dim3(A.B), dim3(C * 4), LDS, stream, P, A);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
// with an incoming cache: the context variant (keyword head, NT >= 4)
template <int C, int NT, bool SPLIT, bool ALIGNED>
static int launch_g4_ctx(const StackParams& P, const CallArgs& A, hipStream_t stream) {
  constexpr int LDS = 2 * Plane<C, 16 * NT>::BYTES;
  hipLaunchKernelGGL((mdtc_g4_kernel<C, NT, SPLIT, false, ALIGNED, true>), dim3(A.B), dim3(C * 4), LDS, stream, P, A);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
template <int C, int NT>
static int launch_g4_ctx_nt(bool split, const StackParams& P, const CallArgs& A, hipStream_t stream) {
  if (A.T % NT == 0) return split ? launch_g4_ctx<C, NT, true, true>(P, A, stream) : launch_g4_ctx<C, NT, false, true>(P, A, stream);
  return split ? launch_g4_ctx<C, NT, true, false>(P, A, stream) : launch_g4_ctx<C, NT, false, false>(P, A, stream);
}
template <int C, int NT, bool SPLIT, bool POOLED>
static int launch_g4_a(const StackParams& P, const CallArgs& A, hipStream_t stream) {
  if constexpr (NT == 1) return launch_g4<C, NT, SPLIT, POOLED, true>(P, A, stream);     // (NT = 1 divides every T)
  else return A.T % NT == 0 ? launch_g4<C, NT, SPLIT, POOLED, true>(P, A, stream) : launch_g4<C, NT, SPLIT, POOLED, false>(P, A, stream);
}
template <int C, int NT>
static int launch_g4_nt(bool split, bool pooled, const StackParams& P, const CallArgs& A, hipStream_t stream) {
  if (pooled) return split ? launch_g4_a<C, NT, true, true>(P, A, stream) : launch_g4_a<C, NT, false, true>(P, A, stream);
  return split ? launch_g4_a<C, NT, true, false>(P, A, stream) : launch_g4_a<C, NT, false, false>(P, A, stream);
}
template <int C>
static int launch_g4_c(int nt, bool split, const StackParams& P, const CallArgs& A, hipStream_t stream) {
  const bool linear = P.head == HEAD_LINEAR && P.odim <= 2;
  const bool pooled = (P.head == HEAD_GLOBAL || P.head == HEAD_LAST) && P.head_hidden <= 448;   // (C + hidden floats of LDS)
  const bool ok = P.ksize == 5 && (linear || pooled) && P.kpre16 <= (C == 64 ? 96 : 64) && P.idim % 8 == 0 &&
                  (reinterpret_cast<uintptr_t>(A.x) & 15) == 0 && A.xs_b % 4 == 0;
  if (!ok) return -4;
  if (A.in_cache) {                                          // a later chunk of a stream
    if (linear) return nt <= 4 ? launch_g4_ctx_nt<C, 4>(split, P, A, stream) : nt == 7 ? launch_g4_ctx_nt<C, 7>(split, P, A, stream) : -4;
    return -4;
  }
  switch (nt) {
    case 1: return launch_g4_nt<C, 1>(split, pooled, P, A, stream);
    case 2: return launch_g4_nt<C, 2>(split, pooled, P, A, stream);
    case 4: return launch_g4_nt<C, 4>(split, pooled, P, A, stream);
    case 7: return launch_g4_nt<C, 7>(split, pooled, P, A, stream);
    default: return -4;
  }
}
int launch_mdtc64_g4(int nt, bool split, const StackParams& P, const CallArgs& A, hipStream_t stream) { return launch_g4_c<64>(nt, split, P, A, stream); }
int launch_mdtc32_g4(int nt, bool split, const StackParams& P, const CallArgs& A, hipStream_t stream) { return launch_g4_c<32>(nt, split, P, A, stream); }
}  // namespace wekws
