// Detection-error-tradeoff scoring on the device: what wekws/bin/compute_det.py:79-106 does with the per-frame keyword
// posteriors of wekws/bin/score.py:128-137, without shipping the (B, T, K) score matrix to the host.
//
//   max pool      score = max(score_list)                       (compute_det.py:82-85: a keyword utterance is a false
//                                                                reject at threshold th iff its maximum is < th)
//   alarm count   i = 0; while i < len: if s[i] >= th: n += 1, i += window_shift  else i += 1
//                                                               (compute_det.py:88-96, filler utterances)
//
// Both are pure comparisons / data movement: bit-exact against the reference's Python (scores widened to double, the
// threshold list passed as doubles exactly as `threshold += step` accumulates them).  One wave per (utterance, keyword)
// for the max (wave arg-max, ties -> lowest frame, like list.index(max(list))); one thread per (utterance, threshold)
// for the sequential alarm scan (T is ~100 frames for a 1-s utterance; the scan of one threshold is inherently serial,
// the 101 thresholds and the utterances are the parallelism).
//
// "Bit-exact" has one more link in the reference's chain: score.py:134-135 writes every score as '{:.6f}' text and
// compute_det.py compares the PARSED values, so a score within 5e-7 of a threshold (0.4999997 -> "0.500000") counts on the
// other side of it than the float32 it came from.  The plain entry points compare the float32 scores (what a caller that
// skips the text file means); the `_text` alarm scan (template R6) first rounds each score to six decimals the way the
// text round trip does -- rint(s * 1e6) / 1e6 in double, which is float(format(s, '.6f')) except on exact decimal ties
// of s * 1e6 that a float32 cannot produce -- and the host mirror rounds the maxima the same way.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wekws {

__global__ __launch_bounds__(256) void det_maxpool_kernel(const float* __restrict__ scores, int64_t rows, int T, int K,
                                                          const int32_t* __restrict__ lengths, float* __restrict__ mx,
                                                          int32_t* __restrict__ amx) {
  const int64_t row = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);      // row = utterance * K + keyword
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const int64_t b = row / K;
  const int k = int(row - b * K);
  const int len = lengths ? min(max(lengths[b], 0), T) : T;
  const float* p = scores + b * int64_t(T) * K + k;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int t = lane; t < len; t += 64) {
    const float v = p[int64_t(t) * K];
    if (v > best || (v == best && t < bi)) { best = v; bi = t; }
  }
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_xor(best, off);
    const int oi = __shfl_xor(bi, off);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if (lane == 0) {
    mx[row] = best;                                          // -inf for an empty utterance (max([]) raises in Python)
    if (amx) amx[row] = len > 0 ? bi : -1;
  }
}

template <bool R6>
__global__ __launch_bounds__(256) void det_alarm_kernel(const float* __restrict__ scores, int B, int T, int K, int keyword,
                                                        const int32_t* __restrict__ lengths,
                                                        const double* __restrict__ thresholds, int n_thr, int window_shift,
                                                        int32_t* __restrict__ alarms) {
  const int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;     // e = utterance * n_thr + threshold
  if (e >= int64_t(B) * n_thr) return;
  const int b = int(e / n_thr), j = int(e - int64_t(b) * n_thr);
  const int len = lengths ? min(max(lengths[b], 0), T) : T;
  const double th = thresholds[j];
  const float* p = scores + int64_t(b) * T * K + keyword;
  int n = 0, i = 0;
  while (i < len) {
    double v = double(p[int64_t(i) * K]);
    if constexpr (R6) v = rint(v * 1e6) / 1e6;                // score.py:134-135 '{:.6f}' -> float(): six decimals
    if (v >= th) { ++n; i += window_shift; }
    else ++i;
  }
  alarms[e] = n;
}

inline int launch_det_maxpool(const float* scores, int B, int T, int K, const int32_t* lengths, float* mx, int32_t* amx,
                              hipStream_t stream) {
  const int64_t rows = int64_t(B) * K;
  hipLaunchKernelGGL(det_maxpool_kernel, dim3(unsigned((rows + 3) / 4)), dim3(256), 0, stream, scores, rows, T, K, lengths,
                     mx, amx);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

inline int launch_det_alarms(bool text6, const float* scores, int B, int T, int K, int keyword, const int32_t* lengths,
                             const double* thresholds, int n_thr, int window_shift, int32_t* alarms, hipStream_t stream) {
  const int64_t n = int64_t(B) * n_thr;
  if (text6)
    hipLaunchKernelGGL(det_alarm_kernel<true>, dim3(unsigned((n + 255) / 256)), dim3(256), 0, stream, scores, B, T, K, keyword,
                       lengths, thresholds, n_thr, window_shift, alarms);
  else
    hipLaunchKernelGGL(det_alarm_kernel<false>, dim3(unsigned((n + 255) / 256)), dim3(256), 0, stream, scores, B, T, K, keyword,
                       lengths, thresholds, n_thr, window_shift, alarms);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace wekws
