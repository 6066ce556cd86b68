// Split-precision (3 x fp16 MFMA, fp32 accumulate) GRU forward.  Same structure, reference semantics and wave
// ownership as gru.hip.h (read that header first); differences:
//   - every operand of the matrix cores is an fp16 hi/lo pair (conv_stack_f16.hip.h explains the arithmetic):
//     the step input, the preprocessing output and the hidden state of every layer live in LDS as operand planes
//     [k-octet][stream][8 halves], a B fragment is one conflict-free ds_read_b128;
//   - the f32 hidden state itself lives in REGISTERS: wave w / lane l owns (units 16w+4(l>>4)..+3, stream l&15 of each
//     stream tile) of every layer for the whole call, which is exactly the MFMA D fragment it computes each step, so
//     the gate math h' = (1-z) n + z h never leaves the lane and only the fp16 image is written back for the next
//     step's products;
//   - the per-frame linear head is a seventh (padded) o-tile on the matrix cores instead of a VALU reduction.
// Per layer and step a wave issues 72*NN MFMAs of 17 cycles instead of 192*NN of 32.
#pragma once
#include "conv_stack_f16.hip.h"
#include "gru.hip.h"

namespace wekws {

struct GruF16Params {
  GruParams base;           // sizes, f32 biases (b_ih, b_hh, pre_b, head_b offsets)
  int32_t kpre16;           // idim rounded up to 32
  uint32_t pre_a16;         // packed fp16 hi/lo A fragments ([o-tile][k32][hi|lo][lane][8])
  uint32_t a_ih16[kGruMaxLayers], a_hh16[kGruMaxLayers];
  uint32_t head_a16;        // classifier rows padded to a multiple of 16
};

template <int NN>
struct GruF16Geom {
  static constexpr int MB = 16 * NN;                       // streams per workgroup
  static constexpr int PLANE_H = (kGruH / 8) * MB * 16;     // bytes of one hi (or lo) plane of an H-wide operand
  static size_t lds_bytes(int kpre16, int nlayers) {
    return size_t(2 * (kpre16 / 8) * MB * 16) + size_t(2 * PLANE_H) * (1 + nlayers);
  }
};

// acc[nn] += A x B over one 32-deep K step for NN stream tiles; b: this lane's item of tile 0 in the hi plane,
// lo plane `plane` bytes further, tiles 256 B apart.
template <int NN>
__device__ __forceinline__ void gru_mfma(f32x4 (&acc)[NN], const F16Frag& a, const f16x8 (&bh)[NN], const f16x8 (&bl)[NN]) {
#pragma unroll
  for (int nn = 0; nn < NN; ++nn) {
    acc[nn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.h, bh[nn], acc[nn], 0, 0, 0);
    acc[nn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.h, bl[nn], acc[nn], 0, 0, 0);
    acc[nn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.l, bh[nn], acc[nn], 0, 0, 0);
  }
}

__device__ __forceinline__ void gru_mfma1(f32x4& acc, const F16Frag& a, const f16x8& bh, const f16x8& bl) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.h, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.h, bl, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.l, bh, acc, 0, 0, 0);
}

__device__ __forceinline__ F16Frag load_frag(const uint4* __restrict__ p) {
  F16Frag f;
  f.h = __builtin_bit_cast(f16x8, p[0]);
  f.l = __builtin_bit_cast(f16x8, p[64]);
  return f;
}

template <int NN, int LT>
__global__ __launch_bounds__(kThreads, 2) void gru_f16_kernel(const GruF16Params Q, const float* __restrict__ x, int B,
                                                              int T, const float* __restrict__ h0,
                                                              float* __restrict__ y, float* __restrict__ hn) {
  using G = GruF16Geom<NN>;
  constexpr int MB = G::MB, H = kGruH, PH = G::PLANE_H;
  const GruParams& P = Q.base;
  extern __shared__ __attribute__((aligned(16))) char gru16_lds[];
  const int PX = (Q.kpre16 / 8) * MB * 16;                  // one plane of the step input
  char* const xin = gru16_lds;                                // [hi | lo] planes of x_t
  char* const in0 = xin + 2 * PX;                          // planes of the preprocessing output
  char* const hpl = in0 + 2 * PH;                          // [layer][hi | lo] planes of the hidden state

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  const int b0 = blockIdx.x * MB;
  const float* __restrict__ W = P.w;
  constexpr int L = LT;                                     // layers (compile time: the state is register-resident)
  const int K = P.odim, idim = P.idim;
  const int u0 = wave * 16 + lq * 4;                        // first of this lane's 4 hidden units
  const int frag = (lq * MB + l15) * 16;                    // this lane's B-fragment item of stream tile 0, K step 0
  // where this lane's 4 consecutive units of (stream tile nn) go inside an H-wide plane (8-byte store)
  const int wr_off = (((u0 >> 3) * MB + l15) * 8 + (u0 & 7)) * 2;

  // ---- f32 hidden state in registers; fp16 image in LDS
  float hreg[LT][NN][4];
#pragma unroll
  for (int l = 0; l < LT; ++l) {
    {
#pragma unroll
      for (int nn = 0; nn < NN; ++nn) {
        const int s = nn * 16 + l15;
        f16x4 vh, vl;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = 0.f;
          if (h0 && b0 + s < B) v = h0[(int64_t(l) * B + b0 + s) * H + u0 + r];
          hreg[l][nn][r] = v;
          _Float16 a, b;
          split16(v, a, b);
          vh[r] = a; vl[r] = b;
        }
        char* dst = hpl + l * 2 * PH + wr_off + nn * 256;
        *reinterpret_cast<f16x4*>(dst) = vh;
        *reinterpret_cast<f16x4*>(dst + PH) = vl;
      }
    }
  }

  // x staging: item = (k-octet, stream): 8 consecutive features of one stream at step t
  constexpr int XI = (16 * MB + kThreads - 1) / kThreads;      // items per thread for kpre16 <= 128 (16 octets)
  const int nitems = (Q.kpre16 / 8) * MB;
  float xr[XI][8];
  auto prefetch = [&](int t) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int e = tid + i * kThreads;
      const int s = e % MB, oct = e / MB;
      const bool ok = e < nitems && t < T && (b0 + s) < B;
      const float* src = x + (int64_t(b0 + s) * T + t) * idim + oct * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) xr[i][j] = (ok && oct * 8 + j < idim) ? src[j] : 0.f;
    }
  };
  prefetch(0);

  const int nkp = Q.kpre16 / 32;
  const uint4* ap_pre = reinterpret_cast<const uint4*>(W + Q.pre_a16) + size_t(wave) * nkp * 128 + lane;
  const float4 bpre = *reinterpret_cast<const float4*>(W + P.pre_b + u0);
  const int head_tiles = (K + 15) / 16;
  constexpr int OTS = (H / 32) * 128;                         // uint4 per o-tile (4 K steps)
  const uint4* aihp[LT];
  const uint4* ahhp[LT];
#pragma unroll
  for (int l = 0; l < LT; ++l) {
    aihp[l] = reinterpret_cast<const uint4*>(W + Q.a_ih16[l]) + lane;
    ahhp[l] = reinterpret_cast<const uint4*>(W + Q.a_hh16[l]) + lane;
  }
  // six weight fragments (W_ir, W_iz, W_in, W_hr, W_hz, W_hn rows of this wave) of one K step, double-buffered:
  // step ks+1 is requested while step ks is multiplied (unrolling all four steps keeps 4 x 48 registers live and
  // spills; requesting the next layer's first step across the barriers measured slower as well)
  F16Frag wq[2][6];
  auto load_w = [&](F16Frag (&w)[6], const uint4* aih, const uint4* ahh, int ks) __attribute__((always_inline)) {
    w[0] = load_frag(aih + (wave)*OTS + ks * 128);
    w[1] = load_frag(aih + (8 + wave) * OTS + ks * 128);
    w[2] = load_frag(aih + (16 + wave) * OTS + ks * 128);
    w[3] = load_frag(ahh + (wave)*OTS + ks * 128);
    w[4] = load_frag(ahh + (8 + wave) * OTS + ks * 128);
    w[5] = load_frag(ahh + (16 + wave) * OTS + ks * 128);
  };
  __syncthreads();

  for (int t = 0; t < T; ++t) {
    // ---- stage x_t as operand planes, prefetch x_{t+1}
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int e = tid + i * kThreads;
      if (e < nitems) {
        const int s = e % MB, oct = e / MB;
        f16x8 vh, vl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          _Float16 a, b;
          split16(xr[i][j], a, b);
          vh[j] = a; vl[j] = b;
        }
        char* dst = xin + (oct * MB + s) * 16;
        *reinterpret_cast<f16x8*>(dst) = vh;
        *reinterpret_cast<f16x8*>(dst + PX) = vl;
      }
    }
    prefetch(t + 1);
    __syncthreads();
    // ---- in0 = [ReLU](Wpre x_t + b)      (subsampling.py:53-57); o-tile = wave
    {
      f32x4 acc[NN];
#pragma unroll
      for (int nn = 0; nn < NN; ++nn) acc[nn] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int ks = 0; ks < nkp; ++ks) {
        const F16Frag a = load_frag(ap_pre + ks * 128);
        f16x8 bh[NN], bl[NN];
#pragma unroll
        for (int nn = 0; nn < NN; ++nn) {
          bh[nn] = *reinterpret_cast<const f16x8*>(xin + ks * 4 * MB * 16 + frag + nn * 256);
          bl[nn] = *reinterpret_cast<const f16x8*>(xin + PX + ks * 4 * MB * 16 + frag + nn * 256);
        }
        gru_mfma<NN>(acc, a, bh, bl);
      }
#pragma unroll
      for (int nn = 0; nn < NN; ++nn) {
        f16x4 vh, vl;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[nn][r] + f4c(bpre, r);
          if (P.pre_relu) v = fmaxf(v, 0.f);
          _Float16 a, b;
          split16(v, a, b);
          vh[r] = a; vl[r] = b;
        }
        char* dst = in0 + wr_off + nn * 256;
        *reinterpret_cast<f16x4*>(dst) = vh;
        *reinterpret_cast<f16x4*>(dst + PH) = vl;
      }
    }
    __syncthreads();
    // ---- GRU layers (statically unrolled so the register-resident state is indexed at compile time)
#pragma unroll
    for (int l = 0; l < LT; ++l) {
      {
        const GruLayer gl = P.layer[l];
        const char* bx = (l == 0 ? in0 : hpl + (l - 1) * 2 * PH) + frag;   // layer input of this step
        char* const hl = hpl + l * 2 * PH;                                  // own state image (previous step)
        const char* bhp = hl + frag;
        f32x4 ar[NN], az[NN], ain[NN], ahn[NN];
#pragma unroll
        for (int nn = 0; nn < NN; ++nn) ar[nn] = az[nn] = ain[nn] = ahn[nn] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto step = [&](const F16Frag (&w)[6], int ks) __attribute__((always_inline)) {
#pragma unroll
          for (int nn = 0; nn < NN; ++nn) {   // B fragments one stream tile at a time: 16 live registers, not 16*NN
            const f16x8 xh = *reinterpret_cast<const f16x8*>(bx + ks * 4 * MB * 16 + nn * 256);
            const f16x8 xl = *reinterpret_cast<const f16x8*>(bx + PH + ks * 4 * MB * 16 + nn * 256);
            const f16x8 hh = *reinterpret_cast<const f16x8*>(bhp + ks * 4 * MB * 16 + nn * 256);
            const f16x8 hlo = *reinterpret_cast<const f16x8*>(bhp + PH + ks * 4 * MB * 16 + nn * 256);
            gru_mfma1(ar[nn], w[0], xh, xl);
            gru_mfma1(az[nn], w[1], xh, xl);
            gru_mfma1(ain[nn], w[2], xh, xl);
            gru_mfma1(ar[nn], w[3], hh, hlo);
            gru_mfma1(az[nn], w[4], hh, hlo);
            gru_mfma1(ahn[nn], w[5], hh, hlo);
          }
        };
        load_w(wq[0], aihp[l], ahhp[l], 0);
#pragma unroll 1
        for (int ks = 0; ks < H / 32; ks += 2) {       // K steps rolled in pairs: 2 x 48 fragment registers live
          load_w(wq[1], aihp[l], ahhp[l], ks + 1);
          __builtin_amdgcn_sched_barrier(0);
          step(wq[0], ks);
          load_w(wq[0], aihp[l], ahhp[l], min(ks + 2, H / 32 - 1));
          __builtin_amdgcn_sched_barrier(0);
          step(wq[1], ks + 1);
        }
        // gate math, register-local (PyTorch formulation, gate order r, z, n)
        const float4 b_ir = *reinterpret_cast<const float4*>(W + gl.b_ih + u0);
        const float4 b_iz = *reinterpret_cast<const float4*>(W + gl.b_ih + H + u0);
        const float4 b_in = *reinterpret_cast<const float4*>(W + gl.b_ih + 2 * H + u0);
        const float4 b_hr = *reinterpret_cast<const float4*>(W + gl.b_hh + u0);
        const float4 b_hz = *reinterpret_cast<const float4*>(W + gl.b_hh + H + u0);
        const float4 b_hn = *reinterpret_cast<const float4*>(W + gl.b_hh + 2 * H + u0);
        f16x4 vh[NN], vl[NN];
#pragma unroll
        for (int nn = 0; nn < NN; ++nn)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float rg = 1.0f / (1.0f + expf(-(ar[nn][r] + f4c(b_ir, r) + f4c(b_hr, r))));
            const float zg = 1.0f / (1.0f + expf(-(az[nn][r] + f4c(b_iz, r) + f4c(b_hz, r))));
            const float ng = tanhf(ain[nn][r] + f4c(b_in, r) + rg * (ahn[nn][r] + f4c(b_hn, r)));
            const float hv = (1.0f - zg) * ng + zg * hreg[l][nn][r];
            hreg[l][nn][r] = hv;
            _Float16 a, b;
            split16(hv, a, b);
            vh[nn][r] = a; vl[nn][r] = b;
          }
        __syncthreads();  // every wave has finished reading h_l(t-1) and the layer input
#pragma unroll
        for (int nn = 0; nn < NN; ++nn) {
          char* dst = hl + wr_off + nn * 256;
          *reinterpret_cast<f16x4*>(dst) = vh[nn];
          *reinterpret_cast<f16x4*>(dst + PH) = vl[nn];
        }
        __syncthreads();
      }
    }
    // ---- head on the top layer's output of this step: o-tile = wave (classifier rows padded to 16)
    if (wave < head_tiles) {
      const char* bt = hpl + (L - 1) * 2 * PH + frag;
      const uint4* ahd = reinterpret_cast<const uint4*>(W + Q.head_a16) + size_t(wave) * (H / 32) * 128 + lane;
      f32x4 acc[NN];
#pragma unroll
      for (int nn = 0; nn < NN; ++nn) acc[nn] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < H / 32; ++ks) {
        const F16Frag a = load_frag(ahd + ks * 128);
        f16x8 bh[NN], bl[NN];
#pragma unroll
        for (int nn = 0; nn < NN; ++nn) {
          bh[nn] = *reinterpret_cast<const f16x8*>(bt + ks * 4 * MB * 16 + nn * 256);
          bl[nn] = *reinterpret_cast<const f16x8*>(bt + PH + ks * 4 * MB * 16 + nn * 256);
        }
        gru_mfma<NN>(acc, a, bh, bl);
      }
#pragma unroll
      for (int nn = 0; nn < NN; ++nn) {
        const int s = nn * 16 + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int k = wave * 16 + lq * 4 + r;
          if (k < K && b0 + s < B) {
            float v = acc[nn][r] + W[P.head_b + k];
            if (P.sigmoid) v = sigmoidf_(v);
            y[(int64_t(b0 + s) * T + t) * K + k] = v;
          }
        }
      }
    }
    // (the barriers of the next step order these plane reads before the next state update)
  }
  // ---- h_n from the register-resident state
  if (hn) {
#pragma unroll
    for (int l = 0; l < LT; ++l) {
      {
#pragma unroll
        for (int nn = 0; nn < NN; ++nn) {
          const int s = nn * 16 + l15;
          if (b0 + s < B) {
#pragma unroll
            for (int r = 0; r < 4; ++r) hn[(int64_t(l) * B + b0 + s) * H + u0 + r] = hreg[l][nn][r];
          }
        }
      }
    }
  }
}

template <int NN, int LT>
inline int launch_gru_f16_nl(const GruF16Params& Q, const float* x, int B, int T, const float* h0, float* y, float* hn,
                             hipStream_t stream) {
  using G = GruF16Geom<NN>;
  const size_t lds = G::lds_bytes(Q.kpre16, LT);
  if (lds > 160 * 1024) return -4;
  auto kern = gru_f16_kernel<NN, LT>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)) !=
      hipSuccess)
    return -3;
  const int grid = (B + G::MB - 1) / G::MB;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), lds, stream, Q, x, B, T, h0, y, hn);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

template <int NN>
inline int launch_gru_f16_nn(const GruF16Params& Q, const float* x, int B, int T, const float* h0, float* y, float* hn,
                             hipStream_t stream) {
  switch (Q.base.nlayers) {
    case 1: return launch_gru_f16_nl<NN, 1>(Q, x, B, T, h0, y, hn, stream);
    case 2: return launch_gru_f16_nl<NN, 2>(Q, x, B, T, h0, y, hn, stream);
    case 3: return launch_gru_f16_nl<NN, 3>(Q, x, B, T, h0, y, hn, stream);
    case 4: return launch_gru_f16_nl<NN, 4>(Q, x, B, T, h0, y, hn, stream);
    default: return -4;
  }
}

inline int launch_gru_f16(const GruF16Params& Q, const float* x, int B, int T, const float* h0, float* y, float* hn,
                          hipStream_t stream) {
  if (Q.kpre16 > 128 || Q.base.odim > 128) return -4;
  // stream tiles per workgroup: wider tiles amortise the per-step weight stream from L2 (393 KB per layer and step)
  // but need enough streams to keep 256 CUs busy
  if (B >= 64 * 192 && GruF16Geom<4>::lds_bytes(Q.kpre16, Q.base.nlayers) <= 160 * 1024)
    return launch_gru_f16_nn<4>(Q, x, B, T, h0, y, hn, stream);
  if (B >= 32 * 192 && GruF16Geom<2>::lds_bytes(Q.kpre16, Q.base.nlayers) <= 160 * 1024)
    return launch_gru_f16_nn<2>(Q, x, B, T, h0, y, hn, stream);
  return launch_gru_f16_nn<1>(Q, x, B, T, h0, y, hn, stream);
}

}  // namespace wekws
