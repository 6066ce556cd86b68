// Fused row softmax + top-k: the first beam prune of the CTC prefix beam search (wekws/model/loss.py:236-238,
// `probs.topk(score_beam_size)` on `logits.softmax(2)`, wekws/bin/stream_kws_ctc.py:487-488) without materialising or
// copying the (frames x vocabulary) posterior matrix: softmax is monotonic, so the k best posteriors are the k best
// logits, exp(l - max) / sum.  One wave per row: every lane keeps the running maximum, the rescaled sum of exponentials
// and its own k best (value, index) over a strided slice, then k rounds of a wave-wide arg-max (ties -> lower index)
// pop the winners.  HBM-bound: the logits are read exactly once, 8 k bytes per row are written.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wekws {

constexpr int kTopkMax = 8;

template <int KK>
__global__ __launch_bounds__(256) void softmax_topk_kernel(const float* __restrict__ logits, int64_t rows, int K,
                                                           float* __restrict__ probs, int32_t* __restrict__ idx) {
  const int64_t row = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* p = logits + row * K;
  float bv[KK];
  int bi[KK];
#pragma unroll
  for (int j = 0; j < KK; ++j) { bv[j] = -INFINITY; bi[j] = 0x7fffffff; }
  float mx = -INFINITY, sum = 0.f;
  auto take = [&](float v, int k) __attribute__((always_inline)) {
    if (v > mx) { sum *= __expf(mx - v); mx = v; }           // online softmax denominator
    sum += __expf(v - mx);
    if (v > bv[KK - 1]) {                                    // insert into the lane's sorted k best (indices ascend
      bv[KK - 1] = v; bi[KK - 1] = k;                        // within a lane, so strict > keeps the lower index first)
#pragma unroll
      for (int j = KK - 1; j > 0; --j)
        if (bv[j] > bv[j - 1]) {
          const float tv = bv[j]; bv[j] = bv[j - 1]; bv[j - 1] = tv;
          const int ti = bi[j]; bi[j] = bi[j - 1]; bi[j - 1] = ti;
        }
    }
  };
  // 16-byte loads through a 4-byte-aligned type (rows of an odd-length matrix are only dword aligned): a wave reads
  // 1 KiB contiguous per step
  struct __attribute__((packed, aligned(4))) V4 { float v[4]; };
  const int K4 = K & ~3;
  for (int k = lane * 4; k < K4; k += 256) {
    const V4 q = *reinterpret_cast<const V4*>(p + k);
#pragma unroll
    for (int j = 0; j < 4; ++j) take(q.v[j], k + j);
  }
  if (K4 + lane < K) take(p[K4 + lane], K4 + lane);
  // wave-wide maximum and denominator
  float gm = mx;
  for (int off = 32; off > 0; off >>= 1) gm = fmaxf(gm, __shfl_xor(gm, off));
  float gs = (mx == -INFINITY) ? 0.f : sum * __expf(mx - gm);
  for (int off = 32; off > 0; off >>= 1) gs += __shfl_xor(gs, off);
  const float inv = 1.0f / gs;
  // k rounds: arg-max over the lanes' current heads, the winning lane pops its head
#pragma unroll
  for (int r = 0; r < KK; ++r) {
    float v = bv[0];
    int i = bi[0];
    for (int off = 32; off > 0; off >>= 1) {
      const float ov = __shfl_xor(v, off);
      const int oi = __shfl_xor(i, off);
      if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
    }
    if (bi[0] == i && bv[0] == v) {                          // this lane held the winner: pop
#pragma unroll
      for (int j = 0; j < KK - 1; ++j) { bv[j] = bv[j + 1]; bi[j] = bi[j + 1]; }
      bv[KK - 1] = -INFINITY; bi[KK - 1] = 0x7fffffff;
    }
    if (lane == 0) {
      probs[row * KK + r] = (i == 0x7fffffff) ? 0.f : __expf(v - gm) * inv;
      idx[row * KK + r] = (i == 0x7fffffff) ? -1 : i;
    }
  }
}

inline int launch_softmax_topk(const float* logits, int64_t rows, int K, int k, float* probs, int32_t* idx,
                               hipStream_t stream) {
  const unsigned grid = unsigned((rows + 3) / 4);
#define WEKWS_TOPK_CASE(KK) \
  case KK: hipLaunchKernelGGL(softmax_topk_kernel<KK>, dim3(grid), dim3(256), 0, stream, logits, rows, K, probs, idx); break;
  switch (k) {
    WEKWS_TOPK_CASE(1) WEKWS_TOPK_CASE(2) WEKWS_TOPK_CASE(3) WEKWS_TOPK_CASE(4)
    WEKWS_TOPK_CASE(5) WEKWS_TOPK_CASE(6) WEKWS_TOPK_CASE(7) WEKWS_TOPK_CASE(8)
    default: return -1;
  }
#undef WEKWS_TOPK_CASE
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace wekws
