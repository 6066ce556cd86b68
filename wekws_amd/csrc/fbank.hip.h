// Log-mel filterbank front-end for gfx950.
//
// Reference semantics (paths relative to the reference tree):
//   wenet::Fbank::Compute                       runtime/core/frontend/fbank.h:138-198
//   mel bank / window construction              runtime/core/frontend/fbank.h:33-97
//   framing (snip edges, no centering)          runtime/core/frontend/fbank.h:141-142
//   int16-scale float input, no normalisation   runtime/core/frontend/feature_pipeline.cc:49-55
//
// GPU design (not the reference's scalar table-driven radix-2): one wave64 per 25 ms frame.
//   1. the 400 samples of the frame are read coalesced, the DC mean is a wave-level shuffle reduction,
//      pre-emphasis + window are applied while packing the real signal as a 256-point COMPLEX sequence
//      z[n] = y[2n] + i*y[2n+1] (a real 512-point FFT costs one 256-point complex FFT + a twiddle pass);
//   2. 256 = 4^4: four radix-4 DIF stages, exactly one butterfly per lane per stage, data exchanged
//      between stages through a 2 KiB LDS strip owned by the wave; output is base-4 digit-reversed;
//   3. lane l rebuilds X[k], k = l + 64m, from Z[k] and conj(Z[256-k]), squares it, and the 256 power
//      bins go back to LDS;
//   4. lanes < num_bins apply the triangular mel weights (sparse: first index + run of weights, summed in
//      ascending bin order like the reference), floor at FLT_EPSILON, logf, and store (T, num_bins)
//      rows -- 160 contiguous bytes per frame at 40 bins.
// Twiddles / window / mel weights are built on the host in double precision and staged into LDS once per
// workgroup.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cfloat>
#include <cmath>
#include <vector>

namespace wekws {

constexpr int kFbankMaxBins = 128;
constexpr int kFbankMaxFft = 512;   // frame_length in (256, 512] -> 512-point FFT, the only size built
constexpr int kFbankWaves = 4;      // frames in flight per workgroup

struct FbankParams {
  const float* tables;  // device: [tw256: 256 x (cos,sin)] [tw512: 256 x (cos,sin)] [window: 512] [mel...]
  int32_t num_bins;
  int32_t frame_length;
  int32_t frame_shift;
  int32_t mel_first_off;  // float offset of int-as-float first index per bin [num_bins]
  int32_t mel_size_off;   // sizes per bin [num_bins]
  int32_t mel_start_off;  // start offset of each bin's weights inside the weight run [num_bins]
  int32_t mel_w_off;      // concatenated weights
  int32_t mel_w_count;
  int32_t table_floats;
};

// Host: tables for the kernel.  Mel bank follows fbank.h:51-88 in float32 like the reference; the
// window follows fbank.h:90-96 (double, stored as float); twiddles are exact-rounded from double.
inline void fbank_build_tables(int num_bins, int sample_rate, int frame_length, int frame_shift, int window,
                               FbankParams* fp, std::vector<float>* out) {
  const int N = 512, NB = N / 2;
  std::vector<float>& t = *out;
  t.clear();
  const double two_pi = 6.283185307179586476925286766559;
  for (int j = 0; j < 256; ++j) {  // e^{-2 pi i j / 256}
    t.push_back(float(std::cos(two_pi * j / 256.0)));
    t.push_back(float(-std::sin(two_pi * j / 256.0)));
  }
  for (int k = 0; k < 256; ++k) {  // e^{-2 pi i k / 512}
    t.push_back(float(std::cos(two_pi * k / 512.0)));
    t.push_back(float(-std::sin(two_pi * k / 512.0)));
  }
  const double a = two_pi / (frame_length - 1);
  for (int i = 0; i < N; ++i) {
    double w = 0.0;
    if (i < frame_length) {
      if (window == 0) w = 0.54 - 0.46 * std::cos(a * double(i));
      else w = std::pow(0.5 - 0.5 * std::cos(a * double(i)), 0.85);
    }
    t.push_back(float(w));
  }
  auto mel = [](float f) { return 1127.0f * logf(1.0f + f / 700.0f); };
  const float bin_width = float(sample_rate) / N;
  const float mel_lo = mel(20.0f), mel_hi = mel(float(sample_rate / 2));
  const float delta = (mel_hi - mel_lo) / (num_bins + 1);
  std::vector<float> first(num_bins), size(num_bins), start(num_bins), weights;
  for (int b = 0; b < num_bins; ++b) {
    const float left = mel_lo + b * delta, center = mel_lo + (b + 1) * delta, right = mel_lo + (b + 2) * delta;
    int fi = -1, li = -1;
    std::vector<float> w(NB, 0.f);
    for (int i = 0; i < NB; ++i) {
      const float m = mel(bin_width * i);
      if (m > left && m < right) {
        w[i] = (m <= center) ? (m - left) / (center - left) : (right - m) / (right - center);
        if (fi < 0) fi = i;
        li = i;
      }
    }
    if (fi < 0) { fi = 0; li = -1; }  // empty filter (reference CHECK-fails; here it yields the floor)
    first[b] = float(fi);
    size[b] = float(li + 1 - fi);
    start[b] = float(weights.size());
    for (int i = fi; i <= li; ++i) weights.push_back(w[i]);
  }
  fp->num_bins = num_bins;
  fp->frame_length = frame_length;
  fp->frame_shift = frame_shift;
  fp->mel_first_off = int(t.size()); t.insert(t.end(), first.begin(), first.end());
  fp->mel_size_off = int(t.size()); t.insert(t.end(), size.begin(), size.end());
  fp->mel_start_off = int(t.size()); t.insert(t.end(), start.begin(), start.end());
  fp->mel_w_off = int(t.size()); t.insert(t.end(), weights.begin(), weights.end());
  fp->mel_w_count = int(weights.size());
  fp->table_floats = int(t.size());
}

// The LDS strip of a frame is private to one wave and a wave's LDS operations execute in program order, so
// hand-offs between lanes of the wave need no workgroup barrier: a wave-scope fence (orders the compiler and
// drains the wave's outstanding LDS traffic) is enough, and the four waves of a workgroup run unsynchronised.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

__device__ __forceinline__ int rev4_256(int k) {  // reverse the four base-4 digits of k
  return ((k & 3) << 6) | (((k >> 2) & 3) << 4) | (((k >> 4) & 3) << 2) | ((k >> 6) & 3);
}

__global__ __launch_bounds__(64 * kFbankWaves) void fbank_kernel(const FbankParams P, const float* __restrict__ pcm,
                                                                 int B, int nsamp, int nframes,
                                                                 float* __restrict__ feats) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const tab = lds;                                  // [table_floats]
  const int tab_pad = (P.table_floats + 3) & ~3;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* const strip = lds + tab_pad + wave * 1024;        // per wave: 512 samples / 256 complex / 256 power

  for (int i = threadIdx.x; i < P.table_floats; i += blockDim.x) tab[i] = P.tables[i];
  __syncthreads();
  const float2* tw256 = reinterpret_cast<const float2*>(tab);
  const float2* tw512 = reinterpret_cast<const float2*>(tab + 512);
  const float* win = tab + 1024;
  float2* z = reinterpret_cast<float2*>(strip);

  const int64_t total = int64_t(B) * nframes;
  const int64_t stride = int64_t(gridDim.x) * kFbankWaves;
  const int FL = P.frame_length;
  // the samples of a wave's NEXT frame are requested before the current frame is processed (a frame is one HBM round
  // trip followed by ~20 dependent LDS hand-offs; without this the round trip is fully exposed)
  float vn[8];
  auto fetch = [&](int64_t f) __attribute__((always_inline)) {
    const int64_t b = f / nframes;
    const int fr = int(f - b * nframes);
    const float* src = pcm + b * nsamp + int64_t(fr) * P.frame_shift;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int i = lane + 64 * m;
      vn[m] = (f < total && i < FL) ? src[i] : 0.f;
    }
  };
  fetch(int64_t(blockIdx.x) * kFbankWaves + wave);
  for (int64_t f = int64_t(blockIdx.x) * kFbankWaves + wave; f < total; f += stride) {
    const bool live = true;
    const int64_t b = f / nframes;
    const int fr = int(f - b * nframes);

    // ---- load, DC removal (fbank.h:155-160)
    float v[8];
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      v[m] = vn[m];
      s += v[m];
    }
    fetch(f + stride);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    const float mean = s / float(FL);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int i = lane + 64 * m;
      strip[i] = (i < FL) ? v[m] - mean : 0.f;
    }
    wave_sync();
    // ---- pre-emphasis 0.97 (fbank.h:122-127), window (fbank.h:130-135), pack as complex
    float y[8];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int n = lane + 64 * m;  // complex index: samples 2n, 2n+1
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int i = 2 * n + h;
        const float cur = strip[i];
        const float prev = strip[i > 0 ? i - 1 : 0];
        y[2 * m + h] = (i < FL) ? (cur - 0.97f * prev) * win[i] : 0.f;
      }
    }
    wave_sync();
    // ---- 256-point complex FFT, radix-4 DIF, 4 stages
    float2 a0 = make_float2(y[0], y[1]), a1 = make_float2(y[2], y[3]), a2 = make_float2(y[4], y[5]),
           a3 = make_float2(y[6], y[7]);
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const int q = 64 >> (2 * st);        // quarter size: 64, 16, 4, 1
      const int L = 4 * q;                 // block size
      const int blk = lane / q, j = lane - blk * q;
      const int base = blk * L + j;
      if (st > 0) {
        a0 = z[base]; a1 = z[base + q]; a2 = z[base + 2 * q]; a3 = z[base + 3 * q];
      }
      const float2 t0 = make_float2(a0.x + a2.x, a0.y + a2.y);
      const float2 t1 = make_float2(a0.x - a2.x, a0.y - a2.y);
      const float2 t2 = make_float2(a1.x + a3.x, a1.y + a3.y);
      const float2 d13 = make_float2(a1.x - a3.x, a1.y - a3.y);
      const float2 t3 = make_float2(d13.y, -d13.x);  // -i * (a1 - a3)
      float2 o0 = make_float2(t0.x + t2.x, t0.y + t2.y);
      float2 o1 = make_float2(t1.x + t3.x, t1.y + t3.y);
      float2 o2 = make_float2(t0.x - t2.x, t0.y - t2.y);
      float2 o3 = make_float2(t1.x - t3.x, t1.y - t3.y);
      if (st < 3) {
        const int tstep = 256 / L;         // w_L^j = w_256^(j*256/L)
        o1 = cmul(o1, tw256[(j * tstep) & 255]);
        o2 = cmul(o2, tw256[(2 * j * tstep) & 255]);
        o3 = cmul(o3, tw256[(3 * j * tstep) & 255]);
      }
      z[base] = o0; z[base + q] = o1; z[base + 2 * q] = o2; z[base + 3 * q] = o3;
      wave_sync();
    }
    // ---- real-FFT untangle + power (fbank.h:173-175): X[k] = (Zk + conj(Zn))/2 - i w^k (Zk - conj(Zn))/2
    float pw[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int k = lane + 64 * m;
      const float2 zk = z[rev4_256(k)];
      const float2 zn = z[rev4_256((256 - k) & 255)];
      const float2 e = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));
      const float2 o = make_float2(0.5f * (zk.x - zn.x), 0.5f * (zk.y + zn.y));
      const float2 wo = cmul(tw512[k], o);
      const float xr = e.x + wo.y, xi = e.y - wo.x;  // e - i*wo
      pw[m] = xr * xr + xi * xi;
    }
    wave_sync();
#pragma unroll
    for (int m = 0; m < 4; ++m) strip[lane + 64 * m] = pw[m];
    wave_sync();
    // ---- mel + log (fbank.h:179-190)
    for (int bin = lane; bin < P.num_bins; bin += 64) {
      const int first = int(tab[P.mel_first_off + bin]);
      const int size = int(tab[P.mel_size_off + bin]);
      const float* w = tab + P.mel_w_off + int(tab[P.mel_start_off + bin]);
      const float* pwr = strip + first;
      // the reference's summation order (fbank.h:181-186), eight taps requested at a time: taps past the filter's
      // width re-read its last one with weight 0
      float e = 0.f;
      for (int k0 = 0; k0 < size; k0 += 8) {
        float wv[8], pv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int kk = min(k0 + u, size - 1);
          wv[u] = (k0 + u < size) ? w[kk] : 0.f;
          pv[u] = pwr[kk];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) e += wv[u] * pv[u];
      }
      e = logf(fmaxf(e, FLT_EPSILON));
      if (live) feats[(b * nframes + fr) * P.num_bins + bin] = e;
    }
    wave_sync();
  }
}

inline int launch_fbank(const FbankParams& P, const float* pcm, int B, int nsamp, int nframes, float* feats,
                        hipStream_t stream) {
  const int tab_pad = (P.table_floats + 3) & ~3;
  const size_t lds = size_t(tab_pad + kFbankWaves * 1024) * sizeof(float);
  const int64_t total = int64_t(B) * nframes;
  int64_t grid = (total + kFbankWaves - 1) / kFbankWaves;
  if (grid > 256 * 8) grid = 256 * 8;
  hipLaunchKernelGGL(fbank_kernel, dim3(unsigned(grid)), dim3(64 * kFbankWaves), lds, stream, P, pcm, B, nsamp,
                     nframes, feats);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace wekws
