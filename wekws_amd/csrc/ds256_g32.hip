// Instantiations of the register-resident exact-f32 DS-TCN h256 kernel.  See ds256_g32.hip.h.
#include "ds256_g32.hip.h"
namespace wekws {
template <int NT>
static int launch_nt32(const StackParams& P, const CallArgs& A, hipStream_t stream, int cus) {
  using G = W16Geom<NT>;
  static DynLdsGrant grant;
  auto kern = ds256_g32_kernel<NT>;
  if (grant_dynamic_lds(kern, int(G::LDS_BYTES), grant)) return -3;
  hipLaunchKernelGGL(kern, dim3(A.B < cus ? A.B : cus), dim3(kW16Threads), G::LDS_BYTES, stream, P, A);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
int launch_ds256_g32(int nt, const StackParams& P, const CallArgs& A, hipStream_t stream, int cus) {
  // the keyword configuration only (FAST in ds256_g16.hip.h): whole aligned 8-float feature items, <= 64 dims, a per-frame
  // linear head with one or two outputs
  const bool fast = P.head == HEAD_LINEAR && P.odim <= 2 && P.kpre16 <= 64 && P.idim % 8 == 0 &&
                    (reinterpret_cast<uintptr_t>(A.x) & 15) == 0 && A.xs_b % 4 == 0;
  if (P.ksize != 8 || A.in_cache || !fast) return -4;
  switch (nt) {
    case 1: return launch_nt32<1>(P, A, stream, cus);
    case 2: return launch_nt32<2>(P, A, stream, cus);
    case 4: return launch_nt32<4>(P, A, stream, cus);
    case 7: return launch_nt32<7>(P, A, stream, cus);
    default: return -4;
  }
}
}  // namespace wekws
