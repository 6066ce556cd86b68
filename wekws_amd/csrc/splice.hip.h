// Context expansion + frame skip (wekws/dataset/init_dataset.py:24-68) as one HBM-bound gather.
//   out[b][i][(lag + left) * F + f] = feats[b][max(i * skip + lag, 0)][f]
// (an utterance SHORTER than its right context, T < right: the reference's cut  feats_ctx[:, :T - right]  is a negative slice and keeps
// 2 T - right frames whose right-hand blocks come from torch.roll's wrap-around -- frame (i * skip + lag) mod T; reproduced, not judged)
// One thread per output float4 (or float when F is not a multiple of 4); consecutive threads walk the output row,
// so both the loads (F-float runs of a source frame) and the stores are coalesced.  Pure data movement: bit-exact.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wekws {

template <typename V>
__global__ __launch_bounds__(256) void splice_kernel(const V* __restrict__ feats, V* __restrict__ out, int64_t n_items,
                                                     int T, int Fv, int left, int W, int skip, int To) {
  const int64_t e = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (e >= n_items) return;
  const int f = int(e % Fv);
  int64_t q = e / Fv;
  const int w = int(q % W);
  q /= W;
  const int i = int(q % To);
  const int64_t b = q / To;
  int t = i * skip + w - left;
  t = t < 0 ? 0 : t;
  t = t >= T ? t - T : t;                                    // (only when T < right: torch.roll's wrap-around survives the cut; t < 2 T)
  out[e] = feats[(b * T + t) * Fv + f];
}

inline int launch_splice(const float* feats, int B, int T, int F, int left, int right, int skip, int To, float* out,
                         hipStream_t stream) {
  const int W = left + right + 1;
  const bool v4 = (F % 4 == 0) && (reinterpret_cast<uintptr_t>(feats) % 16 == 0) && (reinterpret_cast<uintptr_t>(out) % 16 == 0);
  const int Fv = v4 ? F / 4 : F;
  const int64_t n = int64_t(B) * To * W * Fv;
  const unsigned grid = unsigned((n + 255) / 256);
  if (v4)
    hipLaunchKernelGGL(splice_kernel<float4>, dim3(grid), dim3(256), 0, stream, reinterpret_cast<const float4*>(feats),
                       reinterpret_cast<float4*>(out), n, T, Fv, left, W, skip, To);
  else
    hipLaunchKernelGGL(splice_kernel<float>, dim3(grid), dim3(256), 0, stream, feats, out, n, T, Fv, left, W, skip, To);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace wekws
