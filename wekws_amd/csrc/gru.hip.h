// GRU backbone forward for gfx950: preprocessing Linear(+ReLU) -> stacked GRU layers -> per-frame linear
// head (+sigmoid), ONE persistent kernel over all T steps of a tile of streams.
//
// Reference semantics (paths relative to the reference tree):
//   torch.nn.GRU(hdim, hdim, num_layers, batch_first=True)   wekws/model/kws_model.py:128-133, called :73
//   gate equations (PyTorch, gate order r,z,n in the stacked weights):
//     r = s(W_ir x + b_ir + W_hr h + b_hr)      z = s(W_iz x + b_iz + W_hz h + b_hz)
//     n = tanh(W_in x + b_in + r * (W_hn h + b_hn))            h' = (1 - z) n + z h
//   in_cache = h0 (L, B, H), out_cache = h_n (L, B, H)
//   LinearSubsampling1 / LinearClassifier                      subsampling.py:53-57, classifier.py:63-67
//
// Decomposition: a workgroup (8 waves) owns 16*NN independent streams for the whole call; hidden state of
// every layer stays in LDS as [unit][stream] (row stride == 16 mod 32), which is directly the MFMA B operand
// (k = unit, column = stream) of the next step.  Wave w owns hidden units [16w, 16w+16) of ALL THREE gates
// (o-tiles w, 8+w, 16+w of the stacked 3H x H matrices), so after the MFMAs each lane holds r, z, n
// pre-activations of the same (unit, stream) and the gate math needs no cross-lane traffic.  The input and
// recurrent products of r and z share one accumulator (K = 2H).  Weights are streamed from L2 each step as
// pre-packed A fragments (16-byte loads); x_{t+1} is prefetched into registers while step t computes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "conv_stack.hip.h"

namespace wekws {

constexpr int kGruMaxLayers = 4;
constexpr int kGruH = 128;

struct GruLayer {
  uint32_t a_ih, a_hh;  // packed A of W_ih, W_hh: [3H/16][H/16][64][4]
  uint32_t b_ih, b_hh;  // [3H]
};

struct GruParams {
  const float* w;
  int32_t idim, kpre, odim, pre_relu;
  uint32_t pre_a, pre_b;
  int32_t nlayers;
  GruLayer layer[kGruMaxLayers];
  uint32_t head_w, head_b;
  int32_t sigmoid;
};

template <int NN>
struct GruGeom {
  static constexpr int MB = 16 * NN;                              // streams per workgroup
  static constexpr int SS = (NN % 2) ? 16 * NN : 16 * NN + 16;    // row stride, == 16 mod 32
  static constexpr int XPT = (MB * 128 + kThreads - 1) / kThreads;  // x prefetch registers (kpre <= 128)
  static size_t lds_bytes(int kpre, int nlayers) { return size_t(kpre + (1 + nlayers) * kGruH) * SS * 4; }
};

template <int NN>
__global__ __launch_bounds__(kThreads, 2) void gru_kernel(const GruParams P, const float* __restrict__ x, int B, int T,
                                                          const float* __restrict__ h0, float* __restrict__ y,
                                                          float* __restrict__ hn) {
  using G = GruGeom<NN>;
  constexpr int MB = G::MB, SS = G::SS, H = kGruH;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const xin = lds;                       // [kpre][SS]
  float* const in0 = xin + P.kpre * SS;         // [H][SS]   preprocessing output of the current step
  float* const hst = in0 + H * SS;              // [L][H][SS] hidden state of every layer

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  const int b0 = blockIdx.x * MB;
  const float* __restrict__ W = P.w;
  const int L = P.nlayers, K = P.odim, idim = P.idim, kpre = P.kpre;
  const int u0 = wave * 16 + lq * 4;            // first of this lane's 4 hidden units

  // ---- h <- h0 (or zeros: the empty-cache sentinel; the reference itself raises here)
  for (int e = tid; e < L * H * MB; e += kThreads) {
    const int s = e % MB, lu = e / MB;
    const int l = lu / H, u = lu - l * H;
    float v = 0.f;
    if (h0 && b0 + s < B) v = h0[(int64_t(l) * B + b0 + s) * H + u];
    hst[(l * H + u) * SS + s] = v;
  }

  // x prefetch: element e = s * idim + k  (k fastest -> each stream reads one contiguous feature row)
  const int xelems = MB * idim;
  float xr[G::XPT];
  auto prefetch = [&](int t) {
#pragma unroll
    for (int i = 0; i < G::XPT; ++i) {
      const int e = tid + i * kThreads;
      float v = 0.f;
      if (e < xelems && t < T) {
        const int s = e / idim, k = e - s * idim;
        if (b0 + s < B) v = x[(int64_t(b0 + s) * T + t) * idim + k];
      }
      xr[i] = v;
    }
  };
  // zero the K-padding rows of xin once
  for (int e = tid; e < (kpre - idim) * SS; e += kThreads) xin[idim * SS + e] = 0.f;
  prefetch(0);
  __syncthreads();

  const int ot_pre = (kpre / 16) * 64;
  const float4* ap_pre = reinterpret_cast<const float4*>(W + P.pre_a) + wave * ot_pre + lane;
  const float4 bpre = *reinterpret_cast<const float4*>(W + P.pre_b + u0);

  for (int t = 0; t < T; ++t) {
    // ---- stage x_t, prefetch x_{t+1}
#pragma unroll
    for (int i = 0; i < G::XPT; ++i) {
      const int e = tid + i * kThreads;
      if (e < xelems) {
        const int s = e / idim, k = e - s * idim;
        xin[k * SS + s] = xr[i];
      }
    }
    prefetch(t + 1);
    __syncthreads();
    // ---- in0 = [ReLU](Wpre x_t + b)      (subsampling.py:53-57)
    {
      f32x4 acc[1][NN];
      zero_acc(acc);
      mfma_rows<1, NN, SS>(acc, ap_pre, ot_pre, xin + lq * SS + l15, kpre / 16);
#pragma unroll
      for (int nn = 0; nn < NN; ++nn)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[0][nn][r] + (r == 0 ? bpre.x : r == 1 ? bpre.y : r == 2 ? bpre.z : bpre.w);
          if (P.pre_relu) v = fmaxf(v, 0.f);
          in0[(u0 + r) * SS + nn * 16 + l15] = v;
        }
    }
    __syncthreads();
    // ---- GRU layers
    for (int l = 0; l < L; ++l) {
      const GruLayer gl = P.layer[l];
      const float* bx = (l == 0 ? in0 : hst + (l - 1) * H * SS) + lq * SS + l15;  // layer input, this step
      float* const hl = hst + l * H * SS;                                         // own state, previous step
      const float* bh = hl + lq * SS + l15;
      const float4* aih = reinterpret_cast<const float4*>(W + gl.a_ih) + lane;
      const float4* ahh = reinterpret_cast<const float4*>(W + gl.a_hh) + lane;
      constexpr int OTS = (H / 16) * 64;  // float4 per o-tile
      f32x4 ar[NN], az[NN], ain[NN], ahn[NN];
#pragma unroll
      for (int nn = 0; nn < NN; ++nn) ar[nn] = az[nn] = ain[nn] = ahn[nn] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int g = 0; g < H / 16; ++g) {
        const float4 wir = aih[(wave) * OTS + g * 64], wiz = aih[(8 + wave) * OTS + g * 64],
                     win = aih[(16 + wave) * OTS + g * 64];
        const float4 whr = ahh[(wave) * OTS + g * 64], whz = ahh[(8 + wave) * OTS + g * 64],
                     whn = ahh[(16 + wave) * OTS + g * 64];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          float vx[NN], vh[NN];
#pragma unroll
          for (int nn = 0; nn < NN; ++nn) {
            vx[nn] = bx[(g * 16 + s * 4) * SS + nn * 16];
            vh[nn] = bh[(g * 16 + s * 4) * SS + nn * 16];
          }
          auto pick = [&](const float4& q) { return s == 0 ? q.x : s == 1 ? q.y : s == 2 ? q.z : q.w; };
#pragma unroll
          for (int nn = 0; nn < NN; ++nn) {
            ar[nn] = __builtin_amdgcn_mfma_f32_16x16x4f32(pick(wir), vx[nn], ar[nn], 0, 0, 0);
            az[nn] = __builtin_amdgcn_mfma_f32_16x16x4f32(pick(wiz), vx[nn], az[nn], 0, 0, 0);
            ain[nn] = __builtin_amdgcn_mfma_f32_16x16x4f32(pick(win), vx[nn], ain[nn], 0, 0, 0);
            ar[nn] = __builtin_amdgcn_mfma_f32_16x16x4f32(pick(whr), vh[nn], ar[nn], 0, 0, 0);
            az[nn] = __builtin_amdgcn_mfma_f32_16x16x4f32(pick(whz), vh[nn], az[nn], 0, 0, 0);
            ahn[nn] = __builtin_amdgcn_mfma_f32_16x16x4f32(pick(whn), vh[nn], ahn[nn], 0, 0, 0);
          }
        }
      }
      // gate math, register-local
      const float4 b_ir = *reinterpret_cast<const float4*>(W + gl.b_ih + u0);
      const float4 b_iz = *reinterpret_cast<const float4*>(W + gl.b_ih + H + u0);
      const float4 b_in = *reinterpret_cast<const float4*>(W + gl.b_ih + 2 * H + u0);
      const float4 b_hr = *reinterpret_cast<const float4*>(W + gl.b_hh + u0);
      const float4 b_hz = *reinterpret_cast<const float4*>(W + gl.b_hh + H + u0);
      const float4 b_hn = *reinterpret_cast<const float4*>(W + gl.b_hh + 2 * H + u0);
      float hnew[NN][4];
#pragma unroll
      for (int nn = 0; nn < NN; ++nn)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          auto comp = [&](const float4& q) { return r == 0 ? q.x : r == 1 ? q.y : r == 2 ? q.z : q.w; };
          const float hold = hl[(u0 + r) * SS + nn * 16 + l15];
          const float rg = 1.0f / (1.0f + expf(-(ar[nn][r] + comp(b_ir) + comp(b_hr))));
          const float zg = 1.0f / (1.0f + expf(-(az[nn][r] + comp(b_iz) + comp(b_hz))));
          const float ng = tanhf(ain[nn][r] + comp(b_in) + rg * (ahn[nn][r] + comp(b_hn)));
          hnew[nn][r] = (1.0f - zg) * ng + zg * hold;
        }
      __syncthreads();  // every wave has finished reading h_l(t-1) and the layer input
#pragma unroll
      for (int nn = 0; nn < NN; ++nn)
#pragma unroll
        for (int r = 0; r < 4; ++r) hl[(u0 + r) * SS + nn * 16 + l15] = hnew[nn][r];
      __syncthreads();
    }
    // ---- head on the top layer's output of this step      (classifier.py:63-67, kws_model.py:196-199)
    const float* htop = hst + (L - 1) * H * SS;
    for (int e = tid; e < K * MB; e += kThreads) {
      const int s = e % MB, k = e / MB;
      if (b0 + s >= B) continue;
      const float* wk = W + P.head_w + k * H;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      for (int c = 0; c < H; c += 4) {
        const float4 w4 = *reinterpret_cast<const float4*>(wk + c);
        s0 = fmaf(w4.x, htop[(c + 0) * SS + s], s0);
        s1 = fmaf(w4.y, htop[(c + 1) * SS + s], s1);
        s2 = fmaf(w4.z, htop[(c + 2) * SS + s], s2);
        s3 = fmaf(w4.w, htop[(c + 3) * SS + s], s3);
      }
      float v = (s0 + s1) + (s2 + s3) + W[P.head_b + k];
      if (P.sigmoid) v = sigmoidf_(v);
      y[(int64_t(b0 + s) * T + t) * K + k] = v;
    }
    // (the next step's first barrier orders these reads before the next state update)
  }
  // ---- h_n
  if (hn) {
    __syncthreads();
    for (int e = tid; e < L * MB * H; e += kThreads) {
      const int u = e % H, ls = e / H;
      const int s = ls % MB, l = ls / MB;
      if (b0 + s < B) hn[(int64_t(l) * B + b0 + s) * H + u] = hst[(l * H + u) * SS + s];
    }
  }
}

template <int NN>
inline int launch_gru_nn(const GruParams& P, const float* x, int B, int T, const float* h0, float* y, float* hn,
                         hipStream_t stream) {
  using G = GruGeom<NN>;
  const size_t lds = G::lds_bytes(P.kpre, P.nlayers);
  if (lds > 160 * 1024) return -4;
  auto kern = gru_kernel<NN>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)) !=
      hipSuccess)
    return -3;
  const int grid = (B + G::MB - 1) / G::MB;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), lds, stream, P, x, B, T, h0, y, hn);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

inline int launch_gru(const GruParams& P, const float* x, int B, int T, const float* h0, float* y, float* hn,
                      hipStream_t stream) {
  if (P.kpre > 128) return -4;
  // 64-stream tiles amortise the per-step weight stream 4x better but need B large enough to fill the chip
  const bool big = B >= 64 * 256 && GruGeom<4>::lds_bytes(P.kpre, P.nlayers) <= 160 * 1024;
  return big ? launch_gru_nn<4>(P, x, B, T, h0, y, hn, stream) : launch_gru_nn<1>(P, x, B, T, h0, y, hn, stream);
}

// GRU: one workgroup per stream, behind the GRU kernels of the call (their loads sanitise, see nf_clean): a stream whose
// features or incoming states hold a NaN / Inf is re-computed; the others cost one pass over their features.
__global__ __launch_bounds__(256) void gru_nf_fix_kernel(const NfCtx* R, const float* x, int B, int T, const float* h0, float* hn,
                                                                float* y) {
  __shared__ unsigned cell;
  const int b = blockIdx.x;
  const int idim = R->d.idim, H = R->d.hdim, L = R->d.num_layers;
  bool bad = nf_scan(x + int64_t(b) * T * idim, int64_t(T) * idim, &cell);
  if (!bad && h0) bad = nf_scan_rows(h0 + int64_t(b) * H, L, H, int64_t(B) * H, &cell);
  if (bad) nf_repair_gru(R, x, int64_t(T) * idim, h0, hn, y, int64_t(T) * R->d.odim, B, T, b);
}

}  // namespace wekws
