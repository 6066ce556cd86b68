// MFCC tail of the training-side features: log-mel rows -> cepstra.  torchaudio.compliance.kaldi.mfcc after its fbank
// call (wekws/dataset/processor.py:160-169): feature @ DCT-II('ortho', column 0 = sqrt(1/N))[:, :num_ceps], then the
// cepstral lifter 1 + 0.5 Q sin(pi i / Q).  PARITY UNPINNED on the reference side (torchaudio is not installed here);
// checked against oracle/kaldi_feats_oracle.py.
//
// One workgroup builds the (N x num_ceps) matrix with the lifter folded in (float64 cosines, once, in LDS: 25.6 KB at
// 80 x 80) and then walks row tiles: 16 rows staged in LDS, thread = (row, cepstrum) pairs; the row element is an LDS
// broadcast and the matrix element is read with consecutive lanes on consecutive words -- no bank conflicts.  The work
// is 2 N num_ceps flop per row against 4 (N + num_ceps) bytes: vector-ALU-bound, ~1 % of the fbank kernel's time.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wekws {

constexpr int kMfccMaxBins = 128;
constexpr int kMfccRows = 16;

__global__ __launch_bounds__(256) void dct_lifter_kernel(const float* __restrict__ logmel, float* __restrict__ out,
                                                         int64_t rows, int N, int NC, float lifter) {
  extern __shared__ float mfcc_lds[];
  float* D = mfcc_lds;                 // [N][NC]
  float* X = mfcc_lds + N * NC;        // [kMfccRows][N]
  const int tid = threadIdx.x;
  const double pi = 3.14159265358979323846;
  for (int e = tid; e < N * NC; e += 256) {
    const int n = e / NC, k = e - n * NC;
    double v = k == 0 ? sqrt(1.0 / double(N)) : cos(pi / double(N) * (double(n) + 0.5) * double(k)) * sqrt(2.0 / double(N));
    if (lifter != 0.f) v *= 1.0 + 0.5 * double(lifter) * sin(pi * double(k) / double(lifter));
    D[e] = float(v);
  }
  const int64_t ntiles = (rows + kMfccRows - 1) / kMfccRows;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t r0 = tile * kMfccRows;
    const int nr = int(rows - r0 < kMfccRows ? rows - r0 : kMfccRows);
    __syncthreads();                   // D ready (first trip) / previous tile consumed
    for (int e = tid; e < nr * N; e += 256) X[e] = logmel[r0 * N + e];
    __syncthreads();
    for (int e = tid; e < nr * NC; e += 256) {
      const int r = e / NC, k = e - r * NC;
      const float* x = X + r * N;
      float s0 = 0.f, s1 = 0.f;
      int n = 0;
      for (; n + 1 < N; n += 2) {
        s0 = fmaf(x[n], D[n * NC + k], s0);
        s1 = fmaf(x[n + 1], D[(n + 1) * NC + k], s1);
      }
      if (n < N) s0 = fmaf(x[n], D[n * NC + k], s0);
      out[r0 * NC + e] = s0 + s1;
    }
  }
}

inline int launch_dct_lifter(const float* logmel, int64_t rows, int N, int NC, float lifter, float* out, int cus,
                             hipStream_t stream) {
  const size_t lds = size_t(N) * NC * 4 + size_t(kMfccRows) * N * 4;
  const int64_t ntiles = (rows + kMfccRows - 1) / kMfccRows;
  const int64_t cap = int64_t(cus > 0 ? cus : 256) * 4;      // each workgroup pays for the matrix once
  const unsigned grid = unsigned(ntiles < cap ? ntiles : cap);
  hipLaunchKernelGGL(dct_lifter_kernel, dim3(grid), dim3(256), lds, stream, logmel, out, rows, N, NC, lifter);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace wekws
