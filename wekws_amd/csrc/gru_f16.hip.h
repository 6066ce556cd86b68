// Split-precision (3 x fp16 MFMA, fp32 accumulate) GRU forward, layer-major with REGISTER-RESIDENT weights.
// Reference semantics and gate order as gru.hip.h (torch.nn.GRU as built at wekws/model/kws_model.py:128-133).
//
// A GRU step costs ~100 k MACs per stream but needs 393 KB of weights (hi + lo fp16, one layer); streaming them from
// L2 every step bounds the step at the CU's 64 B/clk L1 path (the previous, step-major kernel: 11 us per frame).
// Here one workgroup (8 waves) owns 16*NN streams for the whole call and walks the network LAYER by layer, so that
// each set of weights is read once and then lives in registers for all T steps:
//
//   pass P   in0[t]  = [ReLU](Wpre x[t] + b)                for all t   (time-parallel; Wpre resident, 32 VGPRs)
//   per layer l:
//     pass I gi[t]   = W_ih in_l[t] + b_ih (+ b_hh for r, z) for all t   (time-parallel; W_ih resident, 96 VGPRs)
//     pass R h[t]    = GRU cell(gi[t], h[t-1])               t = 0..T-1  (serial; W_hh resident, 96 VGPRs)
//   pass H   y[t]    = [sigmoid](Wc h_top[t] + bc)           for all t   (time-parallel over waves)
//
// Wave w owns hidden units 16w..16w+15 of all three gates, so r, z, n of a (unit, stream) land in the same lane and
// the cell math is register-local; the f32 state stays in registers.  Only pass R has a per-step dependency: its
// step is 36*NN MFMAs per wave + the cell math + ONE workgroup barrier (the fp16 hi/lo image of h ping-pongs between
// two LDS plane buffers).  Sequences travel between passes through an HBM workspace (per model and stream) in the layout the next
// pass consumes directly: layer inputs as operand planes [t][hi|lo][k-octet][stream][8 halves] (a B fragment is one
// coalesced 16-byte load per lane, no LDS), gate pre-activations in the producing lane's own D-fragment order (the
// lane that wrote them is the only one that reads them).
//
// Block floating point (conv_stack_f16.hip.h): every packed matrix carries a host-chosen power-of-two scale; operands are
// scaled per stream tile before the fp16 split -- the features and the preprocessing output per time step (the exact
// maximum of x[t] is wave-local: every wave holds the whole step; in0[t] is bounded through it), the hidden state by the
// bound max(1, max|h0|) (the cell output is a convex combination of a tanh and the previous state).  The per-step scale
// of in0 travels to pass I through a small workspace array.
#pragma once
#include "conv_stack_f16.hip.h"
#include "gru.hip.h"
#ifndef WEKWS_GRU_MAX_PACKED_WGS
#define WEKWS_GRU_MAX_PACKED_WGS 128
#endif

namespace wekws {

struct GruF16Params {
  GruParams base;           // sizes, f32 biases (b_ih, b_hh, pre_b, head_b offsets)
  int32_t kpre16;           // idim rounded up to 32
  uint32_t pre_a16;         // packed fp16 hi/lo A fragments ([o-tile][k32][hi|lo][lane][8])
  uint32_t a_ih16[kGruMaxLayers], a_hh16[kGruMaxLayers];
  uint32_t head_a16;        // classifier rows padded to a multiple of 16
  // 1 / power-of-two scales of the packed matrices; |Wpre x + b| <= pre_alpha * max|x| + pre_beta
  float pre_inv_s, head_inv_s, ih_inv_s[kGruMaxLayers], hh_inv_s[kGruMaxLayers];
  float pre_alpha, pre_beta;
};

struct GruF16Workspace {
  char* seq[2];             // ping-pong layer sequences, operand planes
  float* gi;                // gate pre-activations of the layer in flight
  float* sc;                // [stream tile][t][16]: 1 / scale of the preprocessing output planes of step t ([0]: the
                            // whole tile; time-packed mode: one per stream slot)
};

template <int NN>
struct GruF16Geom {
  static constexpr int MB = 16 * NN;                       // streams per workgroup
  static constexpr int PLANE_H = (kGruH / 8) * MB * 16;     // bytes of one hi (or lo) plane of an H-wide operand
  static constexpr int SEQ_STEP = 2 * PLANE_H;              // bytes of one step of a layer sequence (per workgroup)
  static constexpr int GI_STEP = 8 * 3 * NN * 256;          // floats of one step of gate pre-activations
  static constexpr int CS = 4;                              // steps of a layer input staged in LDS at a time (pass I)
  static constexpr int LDS_BYTES = 2 * CS * SEQ_STEP;       // pass I: two staging buffers; pass R re-uses the start
                                                            // for its two [hi | lo] plane buffers of h
  static size_t seq_bytes(int tiles, int T) { return size_t(tiles) * T * SEQ_STEP; }       // tiles = workgroups of the call
  static size_t gi_floats(int tiles, int T) { return size_t(tiles) * T * GI_STEP; }
  static size_t sc_floats(int tiles, int T) { return size_t(tiles) * T * 16; }             // per step: 1 (tile) or 16 (columns)
};

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

typedef float gru_f32x8 __attribute__((ext_vector_type(8)));
typedef unsigned gru_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void gru_split4(f32x4 v, f16x4& h, f16x4& l) {
  h = __builtin_convertvector(v, f16x4);
  l = __builtin_convertvector(v - __builtin_convertvector(h, f32x4), f16x4);
}

__device__ __forceinline__ float gru_sigmoid(float v) { return __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
// tanh(v) = 1 - 2 / (1 + e^{2v}); absolute error ~1e-7 (the cancellation near 0 is absolute, not relative)
__device__ __forceinline__ float gru_tanh(float v) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * v)); }

// MODE 0: the whole network in one launch (streaming chunks, or enough stream tiles to fill the GPU).  MODE 1 / 2 / 4:
// only pass P / I (layer `lsel`) / H, over the time chunk blockIdx.y -- the time-parallel passes as their own launches
// over (stream tile x time chunk), so that a mid-sized batch (fewer stream tiles than CUs) still uses every CU for
// them; MODE 3: only pass R of layer `lsel` (grid = stream tiles).
template <int NN, int MODE>
__global__ __launch_bounds__(kThreads) void gru_f16_kernel(const GruF16Params Q, const GruF16Workspace WS,
                                                           const float* __restrict__ x, int B, int T,
                                                           const float* __restrict__ h0, float* __restrict__ y,
                                                           float* __restrict__ hn, int lsel, int tchunk, int spw) {
  using G = GruF16Geom<NN>;
  constexpr int MB = G::MB, H = kGruH, PH = G::PLANE_H;
  constexpr int KSB = 4 * MB * 16;                            // bytes per K step inside a plane
  constexpr int OTS = (H / 32) * 128;                         // uint4 per o-tile of an H-deep matrix (4 K steps)
  const GruParams& P = Q.base;
  extern __shared__ __attribute__((aligned(16))) char gru16_lds[];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  // spw: streams this workgroup owns (MB, or fewer in single-launch mode when there are more CUs than stream tiles:
  // gru_f16_spw); its MFMA tiles still have MB columns
  const int b0 = blockIdx.x * spw;
  const int bend = min(B, b0 + spw);                          // one past the workgroup's last stream
  const float* __restrict__ W = P.w;
  const int K = P.odim, idim = P.idim;
  const int u0 = wave * 16 + lq * 4;                        // first of this lane's 4 hidden units
  const int frag = (lq * MB + l15) * 16;                    // this lane's B-fragment item of stream tile 0, K step 0
  // where this lane's 4 consecutive units of stream tile 0 go inside an H-wide plane (8-byte store)
  const int wr_off = (((u0 >> 3) * MB + l15) * 8 + (u0 & 7)) * 2;
  char* const seq0 = WS.seq[0] + size_t(blockIdx.x) * T * G::SEQ_STEP;
  char* const seq1 = WS.seq[1] + size_t(blockIdx.x) * T * G::SEQ_STEP;
  float* const gi = WS.gi + size_t(blockIdx.x) * T * G::GI_STEP + size_t(wave) * 3 * NN * 256 + lane * 4;
  const int tb = (MODE == 0 || MODE == 3) ? 0 : int(blockIdx.y) * tchunk;       // time range of this workgroup
  const int te = (MODE == 0 || MODE == 3) ? T : min(T, tb + tchunk);
  float* const sc = WS.sc + size_t(blockIdx.x) * T * 16;
  // TIME-PACKED mode of the time-parallel passes (P, I, H) for a handful of streams: with nb <= 8 streams in the tile,
  // an MFMA tile's 16 columns are (step, stream) pairs -- 16 / nb steps per tile -- instead of 16 stream slots of one
  // step of which nb are real.  A 10-frame chunk of ONE stream is then 1 tile per pass instead of 10 (per-phase
  // stamps of that chunk: passes P + I + H were 68 k of its 128 k cycles).  Columns are independent and every operand
  // scale is an exact power of two, so the results are bit-identical to the unpacked passes; operand scales become per
  // column (sc[t][slot]).  The recurrence (pass R) is serial in t and stays one tile per step.
  const int nb = bend - b0;
  const bool packed = MODE == 0 && NN == 1 && nb <= 8;
  const int psh = nb <= 1 ? 0 : nb <= 2 ? 1 : nb <= 4 ? 2 : 3;   // log2 of the stream slots per step
  const int TP = 16 >> psh;                                      // steps per tile
  const int pcs = l15 & ((1 << psh) - 1), pdt = l15 >> psh;      // this column's stream slot and step inside a tile
  float* const gi_w = WS.gi + size_t(blockIdx.x) * T * G::GI_STEP + size_t(wave) * 3 * NN * 256;   // (wave's gates, lane 0)

  // bound of layer l's hidden state over this stream tile: max(1, max|h0[l]|) -- h(t) is a convex combination of a tanh
  // and h(t-1).  Called by all threads (it holds a barrier); the same number in every pass / launch that needs it.
  __shared__ AmaxCell gru_cells[kGruMaxLayers];
  amax_zero<kThreads>(gru_cells, kGruMaxLayers);
  __syncthreads();
  auto h_bound = [&](int l) __attribute__((always_inline)) -> float {
    float m = 0.f;
    if (h0)
      for (int e = tid; e < MB * H; e += kThreads) {
        const int sidx = b0 + e / H;
        if (sidx < bend) m = fmaxf(m, fabsf(nf_clean(h0[(int64_t(l) * B + sidx) * H + (e % H)])));
      }
    amax_publish(gru_cells + l, m);
    __syncthreads();
    return fmaxf(1.f, amax_read(gru_cells + l));
  };
  float hb_prev = 1.f;                                       // bound of the previous layer's output sequence
  // Single-launch mode (streaming chunks): the h0 values behind every layer's bound are requested NOW, so that their
  // trip from memory overlaps pass P; they are reduced and published behind pass P's barrier.
  f32x4 h0v[MODE == 0 ? kGruMaxLayers : 1][NN];
  const bool h0vec = h0 && reinterpret_cast<uintptr_t>(h0) % 16 == 0;
  if constexpr (MODE == 0) {
#pragma unroll
    for (int l = 0; l < kGruMaxLayers; ++l)
#pragma unroll
      for (int nn = 0; nn < NN; ++nn) {
        const int f = (nn * kThreads + tid) * 4, sidx = b0 + f / H;
        h0v[l][nn] = (h0vec && l < P.nlayers && sidx < bend)
                         ? nf_clean_vec<f32x4, 4>(*reinterpret_cast<const f32x4*>(h0 + (int64_t(l) * B + sidx) * H + (f % H)))
                         : f32x4{0.f, 0.f, 0.f, 0.f};
      }
  }

  // =================== pass P: in0[t] = [ReLU](Wpre x[t] + b) -> seq0 (subsampling.py:53-57) ===================
  if constexpr (MODE == 0 || MODE == 1) {
    const int nkp = Q.kpre16 / 32;                            // <= 4 (checked by the launcher)
    const uint4* ap = reinterpret_cast<const uint4*>(W + Q.pre_a16) + size_t(wave) * nkp * 128 + lane;
    F16Frag a[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      if (ks < nkp) a[ks] = load_frag(ap + ks * 128);
    const f32x4 bpre = *reinterpret_cast<const f32x4*>(W + P.pre_b + u0);
    const bool xvec = (idim % 4 == 0) && (reinterpret_cast<uintptr_t>(x) % 16 == 0);
    // this lane's B-fragment source: 8 consecutive features (k = ks*32 + lq*8 ..) of stream nn*16 + l15
    auto load_x = [&](gru_f32x8 (&xr)[4][NN], int t) __attribute__((always_inline)) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int nn = 0; nn < NN; ++nn) {
          gru_f32x8 v = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          const int s = b0 + nn * 16 + l15, k0 = ks * 32 + lq * 8;
          if (ks < nkp && s < bend && t < T && k0 < idim) {
            const float* src = x + (int64_t(s) * T + t) * idim + k0;
            if (xvec && k0 + 8 <= idim) {
              const f32x4 lo4 = *reinterpret_cast<const f32x4*>(src), hi4 = *reinterpret_cast<const f32x4*>(src + 4);
              v = gru_f32x8{lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
            } else {
#pragma unroll
              for (int j = 0; j < 8; ++j)
                if (k0 + j < idim) v[j] = src[j];
            }
          }
          xr[ks][nn] = nf_clean_vec<gru_f32x8, 8>(v);        // (a NaN / Inf feature enters as 0: nonfinite.hip.h)
        }
    };
    if (packed) {
      for (int t0 = 0; t0 < T; t0 += TP) {
        const int t = t0 + pdt;
        const bool cv = t < T && pcs < nb;                    // this column exists
        gru_f32x8 xr[4];
        float ax = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          gru_f32x8 v = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          const int k0 = ks * 32 + lq * 8;
          if (ks < nkp && cv && k0 < idim) {
            const float* src = x + (int64_t(b0 + pcs) * T + t) * idim + k0;
            if (xvec && k0 + 8 <= idim) {
              const f32x4 lo4 = *reinterpret_cast<const f32x4*>(src), hi4 = *reinterpret_cast<const f32x4*>(src + 4);
              v = gru_f32x8{lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
            } else {
#pragma unroll
              for (int j = 0; j < 8; ++j)
                if (k0 + j < idim) v[j] = src[j];
            }
          }
          v = nf_clean_vec<gru_f32x8, 8>(v);                  // (a NaN / Inf feature enters as 0: nonfinite.hip.h)
          xr[ks] = v;
#pragma unroll
          for (int j = 0; j < 8; ++j) ax = fmaxf(ax, fabsf(v[j]));
        }
        ax = fmaxf(ax, __shfl_xor(ax, 16));                   // the column's features are spread over the four k-octet
        ax = fmaxf(ax, __shfl_xor(ax, 32));                   // lane groups
        float cx, inv_s0;
        const float sx = pow2_scale(ax, &cx);
        const float s0 = pow2_scale(fmaf(Q.pre_alpha, ax, Q.pre_beta), &inv_s0);
        cx *= Q.pre_inv_s;
        if (wave == 0 && lq == 0 && cv) sc[t * 16 + pcs] = inv_s0;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          if (ks < nkp) {
            const gru_f32x8 xs = xr[ks] * sx;
            const f16x8 bh = __builtin_convertvector(xs, f16x8);
            const f16x8 bl = __builtin_convertvector(xs - __builtin_convertvector(bh, gru_f32x8), f16x8);
            gru_mfma1(acc, a[ks], bh, bl);
          }
        f32x4 v = acc * cx + bpre;
        if (P.pre_relu) v = __builtin_elementwise_max(v, f32x4{0.f, 0.f, 0.f, 0.f});
        f16x4 vh, vl;
        gru_split4(v * s0, vh, vl);
        if (cv) {
          char* dst = seq0 + size_t(t) * G::SEQ_STEP + (((u0 >> 3) * MB + pcs) * 8 + (u0 & 7)) * 2;
          *reinterpret_cast<f16x4*>(dst) = vh;
          *reinterpret_cast<f16x4*>(dst + PH) = vl;
        }
      }
    } else {
    gru_f32x8 xc[4][NN], xn[4][NN];
    load_x(xc, tb);
    for (int t = tb; t < te; ++t) {
      load_x(xn, t + 1);
      // max|x[t]| over the tile: every wave holds the whole step, so the maximum is wave-local (no barrier)
      float ax = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        if (ks < nkp) {
#pragma unroll
          for (int nn = 0; nn < NN; ++nn)
#pragma unroll
            for (int j = 0; j < 8; ++j) ax = fmaxf(ax, fabsf(xc[ks][nn][j]));
        }
      ax = __uint_as_float(unsigned(__builtin_amdgcn_readlane(int(wave_umax63(__float_as_uint(ax))), 63)));
      float cx, inv_s0;
      const float sx = pow2_scale(ax, &cx);
      const float s0 = pow2_scale(fmaf(Q.pre_alpha, ax, Q.pre_beta), &inv_s0);
      cx *= Q.pre_inv_s;
      if (tid == 0) sc[t * 16] = inv_s0;
      f32x4 acc[NN];
#pragma unroll
      for (int nn = 0; nn < NN; ++nn) acc[nn] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        if (ks < nkp) {
#pragma unroll
          for (int nn = 0; nn < NN; ++nn) {
            const gru_f32x8 xs = xc[ks][nn] * sx;
            const f16x8 bh = __builtin_convertvector(xs, f16x8);
            const f16x8 bl = __builtin_convertvector(xs - __builtin_convertvector(bh, gru_f32x8), f16x8);
            gru_mfma1(acc[nn], a[ks], bh, bl);
          }
        }
#pragma unroll
      for (int nn = 0; nn < NN; ++nn) {
        f32x4 v = acc[nn] * cx + bpre;
        if (P.pre_relu) v = __builtin_elementwise_max(v, f32x4{0.f, 0.f, 0.f, 0.f});
        f16x4 vh, vl;
        gru_split4(v * s0, vh, vl);
        char* dst = seq0 + size_t(t) * G::SEQ_STEP + wr_off + nn * 256;
        *reinterpret_cast<f16x4*>(dst) = vh;
        *reinterpret_cast<f16x4*>(dst + PH) = vl;
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int nn = 0; nn < NN; ++nn) xc[ks][nn] = xn[ks][nn];
    }
    }
  }
  if constexpr (MODE == 0) {
    if (h0vec) {
#pragma unroll
      for (int l = 0; l < kGruMaxLayers; ++l)
        if (l < P.nlayers) {
          float m = 0.f;
#pragma unroll
          for (int nn = 0; nn < NN; ++nn)
#pragma unroll
            for (int r = 0; r < 4; ++r) m = fmaxf(m, fabsf(h0v[l][nn][r]));
          amax_publish(gru_cells + l, m);
        }
    }
    __threadfence_block();
    __syncthreads();
  }

  // ============================== GRU layers ==============================
#pragma unroll 1
  for (int l = (MODE == 0 ? 0 : lsel); l < (MODE == 0 ? P.nlayers : ((MODE == 2 || MODE == 3) ? lsel + 1 : 0)); ++l) {
    const GruLayer gl = P.layer[l];
    const char* const sin = (l & 1) ? seq1 : seq0;            // this layer's input sequence
    char* const sout = (l & 1) ? seq0 : seq1;                 // its output sequence
    // ---------------- pass I: gi[t] = W_ih in[t] + b_ih (+ b_hh for r, z), all t ----------------
    if constexpr (MODE == 0 || MODE == 2) {
      const uint4* aih = reinterpret_cast<const uint4*>(W + Q.a_ih16[l]) + lane;
      F16Frag wi[3][4];
#pragma unroll
      for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) wi[g][ks] = load_frag(aih + (g * 8 + wave) * OTS + ks * 128);
      f32x4 bias[3];
      bias[0] = *reinterpret_cast<const f32x4*>(W + gl.b_ih + u0) + *reinterpret_cast<const f32x4*>(W + gl.b_hh + u0);
      bias[1] = *reinterpret_cast<const f32x4*>(W + gl.b_ih + H + u0) + *reinterpret_cast<const f32x4*>(W + gl.b_hh + H + u0);
      bias[2] = *reinterpret_cast<const f32x4*>(W + gl.b_ih + 2 * H + u0);
      // scale of this layer's input planes: per step for the preprocessing output (layer 0), else the bound of the
      // previous layer's hidden state
      if constexpr (MODE == 2) {
        if (l > 0) hb_prev = h_bound(l - 1);
      }
      float inv_in;
      (void)pow2_scale(hb_prev, &inv_in);
      if (packed) {
        for (int t0 = 0; t0 < T; t0 += TP) {
          const int t = t0 + pdt, tc = min(t, T - 1);
          const bool cv = t < T && pcs < nb;
          const char* p = sin + size_t(tc) * G::SEQ_STEP + (lq * MB + pcs) * 16;   // this column's fragment, K step 0
          f32x4 acc[3];
#pragma unroll
          for (int g = 0; g < 3; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            const f16x8 bh = *reinterpret_cast<const f16x8*>(p + ks * KSB);
            const f16x8 bl = *reinterpret_cast<const f16x8*>(p + PH + ks * KSB);
#pragma unroll
            for (int g = 0; g < 3; ++g) gru_mfma1(acc[g], wi[g][ks], bh, bl);
          }
          const float cin = (l == 0 ? sc[tc * 16 + pcs] : inv_in) * Q.ih_inv_s[l];
          if (cv) {                                           // into the slot pass R's lane (stream pcs, same lq) reads
            float* go = gi_w + size_t(t) * G::GI_STEP + (lq * 16 + pcs) * 4;
#pragma unroll
            for (int g = 0; g < 3; ++g) *reinterpret_cast<f32x4*>(go + g * 256) = acc[g] * cin + bias[g];
          }
        }
        __threadfence_block();
        __syncthreads();
      } else {
      // The input sequence is staged CS steps at a time through LDS (every wave needs all of it): each thread moves
      // CS*NN 16-byte items per chunk, requested one chunk ahead into registers and written to the other buffer after
      // the current chunk's products -- one barrier per chunk.
      constexpr int CS = G::CS, CHUNK = CS * G::SEQ_STEP, IT = CHUNK / (kThreads * 16);
      static_assert(IT * kThreads * 16 == CHUNK, "whole items per thread");
      gru_u32x4 stage[IT];
      // byte offset of this thread's item i inside a chunk: step = off / SEQ_STEP, position inside the step's planes
#define GRU_FETCH(T0)                                                                                     \
  _Pragma("unroll") for (int i = 0; i < IT; ++i) {                                                       \
    const int off = (i * kThreads + tid) * 16;                                                           \
    const int tf = min((T0) + off / G::SEQ_STEP, T - 1);                                                 \
    stage[i] = *reinterpret_cast<const gru_u32x4*>(sin + size_t(tf) * G::SEQ_STEP + off % G::SEQ_STEP);      \
  }
#define GRU_PUT(BUF)                                                                                      \
  _Pragma("unroll") for (int i = 0; i < IT; ++i)                                                         \
      *reinterpret_cast<gru_u32x4*>((BUF) + (i * kThreads + tid) * 16) = stage[i];
      // 1 / scale of each step's input planes, a chunk ahead like the planes themselves (layer 0: per step, from pass P)
      float scur[CS], snext[CS];
#pragma unroll
      for (int dt = 0; dt < CS; ++dt) scur[dt] = l == 0 ? sc[min(tb + dt, T - 1) * 16] : inv_in;
      GRU_FETCH(tb)
      GRU_PUT(gru16_lds)
      __syncthreads();
      for (int t0 = tb, c = 0; t0 < te; t0 += CS, ++c) {
        const char* cur = gru16_lds + (c & 1) * CHUNK;
        GRU_FETCH(t0 + CS)                                   // clamped to the last step: harmless past the end
#pragma unroll
        for (int dt = 0; dt < CS; ++dt) snext[dt] = l == 0 ? sc[min(t0 + CS + dt, T - 1) * 16] : inv_in;
#pragma unroll
        for (int dt = 0; dt < CS; ++dt) {
          const int t = t0 + dt;
          if (t < te) {
            f32x4 acc[3][NN];
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
              for (int nn = 0; nn < NN; ++nn) acc[g][nn] = f32x4{0.f, 0.f, 0.f, 0.f};
            const char* p = cur + dt * G::SEQ_STEP + frag;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
              for (int nn = 0; nn < NN; ++nn) {
                const f16x8 bh = *reinterpret_cast<const f16x8*>(p + ks * KSB + nn * 256);
                const f16x8 bl = *reinterpret_cast<const f16x8*>(p + PH + ks * KSB + nn * 256);
#pragma unroll
                for (int g = 0; g < 3; ++g) gru_mfma1(acc[g][nn], wi[g][ks], bh, bl);
              }
            float* go = gi + size_t(t) * G::GI_STEP;
            const float cin = scur[dt] * Q.ih_inv_s[l];
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
              for (int nn = 0; nn < NN; ++nn)
                *reinterpret_cast<f32x4*>(go + (g * NN + nn) * 256) = acc[g][nn] * cin + bias[g];
          }
        }
        GRU_PUT(gru16_lds + ((c + 1) & 1) * CHUNK)
#pragma unroll
        for (int dt = 0; dt < CS; ++dt) scur[dt] = snext[dt];
        __syncthreads();
      }
#undef GRU_FETCH
#undef GRU_PUT
      }
    }
    // ---------------- pass R: the recurrence ----------------
    if constexpr (MODE == 0 || MODE == 3) {
      const uint4* ahh = reinterpret_cast<const uint4*>(W + Q.a_hh16[l]) + lane;
      F16Frag wh[3][4];
#pragma unroll
      for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) wh[g][ks] = load_frag(ahh + (g * 8 + wave) * OTS + ks * 128);
      const f32x4 b_hn = *reinterpret_cast<const f32x4*>(W + gl.b_hh + 2 * H + u0);
      // single-launch mode: every layer's cell was published behind pass P's barrier (pass I's last chunk ended with a
      // barrier too, so the staging buffers are free); otherwise reduce h0[l] now
      const float hb = (MODE == 0 && (h0vec || !h0)) ? fmaxf(1.f, amax_read(gru_cells + l)) : h_bound(l);
      float chh;
      const float shl = pow2_scale(hb, &chh);
      chh *= Q.hh_inv_s[l];
      hb_prev = hb;
      f32x4 hreg[NN];
#pragma unroll
      for (int nn = 0; nn < NN; ++nn) {
        const int s = b0 + nn * 16 + l15;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (h0 && s < bend) v = nf_clean_vec<f32x4, 4>(*reinterpret_cast<const f32x4*>(h0 + (int64_t(l) * B + s) * H + u0));
        hreg[nn] = v;
        f16x4 vh, vl;
        gru_split4(v * shl, vh, vl);
        char* dst = gru16_lds + wr_off + nn * 256;
        *reinterpret_cast<f16x4*>(dst) = vh;
        *reinterpret_cast<f16x4*>(dst + PH) = vl;
      }
      // gate pre-activations of this lane, two steps ahead (they do not depend on the recurrence)
      f32x4 g0[3][NN], g1[3][NN];
      auto load_g = [&](f32x4 (&gg)[3][NN], int t) __attribute__((always_inline)) {
        const float* gp = gi + size_t(min(t, T - 1)) * G::GI_STEP;
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
          for (int nn = 0; nn < NN; ++nn) gg[g][nn] = *reinterpret_cast<const f32x4*>(gp + (g * NN + nn) * 256);
      };
      load_g(g0, 0);
      load_g(g1, 1);
      __syncthreads();
      for (int t = 0; t < T; ++t) {
        const char* hb = gru16_lds + (t & 1) * 2 * PH + frag;        // image of h(t-1)
        char* const hw = gru16_lds + ((t + 1) & 1) * 2 * PH;          // image of h(t)
        f32x4 acc[3][NN];
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
          for (int nn = 0; nn < NN; ++nn) acc[g][nn] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int nn = 0; nn < NN; ++nn) {
            const f16x8 hh = *reinterpret_cast<const f16x8*>(hb + ks * KSB + nn * 256);
            const f16x8 hl = *reinterpret_cast<const f16x8*>(hb + PH + ks * KSB + nn * 256);
#pragma unroll
            for (int g = 0; g < 3; ++g) gru_mfma1(acc[g][nn], wh[g][ks], hh, hl);
          }
        // cell math, register-local (PyTorch formulation, gate order r, z, n)
#pragma unroll
        for (int nn = 0; nn < NN; ++nn) {
          f32x4 hv;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float rg = gru_sigmoid(fmaf(acc[0][nn][r], chh, g0[0][nn][r]));
            const float zg = gru_sigmoid(fmaf(acc[1][nn][r], chh, g0[1][nn][r]));
            const float ng = gru_tanh(g0[2][nn][r] + rg * fmaf(acc[2][nn][r], chh, b_hn[r]));
            hv[r] = ng + zg * (hreg[nn][r] - ng);               // (1 - z) n + z h
          }
          hreg[nn] = hv;
          f16x4 vh, vl;
          gru_split4(hv * shl, vh, vl);
          char* dst = hw + wr_off + nn * 256;
          *reinterpret_cast<f16x4*>(dst) = vh;
          *reinterpret_cast<f16x4*>(dst + PH) = vl;
          char* gd = sout + size_t(t) * G::SEQ_STEP + wr_off + nn * 256;
          *reinterpret_cast<f16x4*>(gd) = vh;
          *reinterpret_cast<f16x4*>(gd + PH) = vl;
        }
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
          for (int nn = 0; nn < NN; ++nn) g0[g][nn] = g1[g][nn];
        load_g(g1, t + 2);
        __syncthreads();   // h(t) is complete and every wave is done with h(t-1)
      }
      if (hn) {
#pragma unroll
        for (int nn = 0; nn < NN; ++nn) {
          const int s = b0 + nn * 16 + l15;
          if (s < bend) *reinterpret_cast<f32x4*>(hn + (int64_t(l) * B + s) * H + u0) = hreg[nn];
        }
      }
    }
    if constexpr (MODE == 0) {
      __threadfence_block();
      __syncthreads();
    }
  }

  // ================= pass H: y[t] = [sigmoid](Wc h_top[t] + bc), waves take steps round-robin =================
  if constexpr (MODE == 0 || MODE == 4) {
    const char* const stop = (P.nlayers & 1) ? seq1 : seq0;   // output sequence of the last layer
    const int head_tiles = (K + 15) / 16;
    if constexpr (MODE == 4) hb_prev = h_bound(P.nlayers - 1);
    float chd;
    (void)pow2_scale(hb_prev, &chd);
    chd *= Q.head_inv_s;
    for (int ot = 0; ot < head_tiles; ++ot) {
      const uint4* ahd = reinterpret_cast<const uint4*>(W + Q.head_a16) + size_t(ot) * OTS + lane;
      F16Frag a[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) a[ks] = load_frag(ahd + ks * 128);
      const int k0 = ot * 16 + lq * 4;
      f32x4 bc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (k0 + r < K) bc[r] = W[P.head_b + k0 + r];
      if (packed) {
        for (int t0 = wave * TP; t0 < T; t0 += (kThreads / 64) * TP) {
          const int t = t0 + pdt, tc = min(t, T - 1);
          const bool cv = t < T && pcs < nb;
          const char* p = stop + size_t(tc) * G::SEQ_STEP + (lq * MB + pcs) * 16;
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            const f16x8 bh = *reinterpret_cast<const f16x8*>(p + ks * KSB);
            const f16x8 bl = *reinterpret_cast<const f16x8*>(p + PH + ks * KSB);
            gru_mfma1(acc, a[ks], bh, bl);
          }
          if (cv) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (k0 + r < K) {
                float v = fmaf(acc[r], chd, bc[r]);
                if (P.sigmoid) v = sigmoidf_(v);
                y[(int64_t(b0 + pcs) * T + t) * K + k0 + r] = v;
              }
          }
        }
        continue;
      }
      for (int t = tb + wave; t < te; t += kThreads / 64) {
        const char* p = stop + size_t(t) * G::SEQ_STEP + frag;
        f32x4 acc[NN];
#pragma unroll
        for (int nn = 0; nn < NN; ++nn) acc[nn] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int nn = 0; nn < NN; ++nn) {
            const f16x8 bh = *reinterpret_cast<const f16x8*>(p + ks * KSB + nn * 256);
            const f16x8 bl = *reinterpret_cast<const f16x8*>(p + PH + ks * KSB + nn * 256);
            gru_mfma1(acc[nn], a[ks], bh, bl);
          }
#pragma unroll
        for (int nn = 0; nn < NN; ++nn) {
          const int s = b0 + nn * 16 + l15;
          if (s < bend) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (k0 + r < K) {
                float v = fmaf(acc[nn][r], chd, bc[r]);
                if (P.sigmoid) v = sigmoidf_(v);
                y[(int64_t(s) * T + t) * K + k0 + r] = v;
              }
          }
        }
      }
    }
  }
}

// Streams per workgroup of a streaming chunk (T <= 16, single launch).  A workgroup's time does not depend on how many of
// its 16 MFMA columns are real, but the time-parallel passes pack (step, stream) pairs into the columns when it owns <= 8
// streams (gru_f16_kernel: TIME-PACKED mode) -- so with CUs to spare, fewer streams per workgroup is less work per
// workgroup: 256 streams as 128 workgroups of 2 instead of 16 of 16.  Capped at kGruMaxPackedWgs workgroups: every one
// streams the layers' 1.6 MB of weights from L2.
constexpr int kGruMaxPackedWgs = WEKWS_GRU_MAX_PACKED_WGS;
inline int gru_f16_spw(int B, int T, int cus) {
  if (T > 16 || B <= 1) return 16;
  const int wgs = cus < kGruMaxPackedWgs ? cus : kGruMaxPackedWgs;
  int spw = 1;
  while (spw < 16 && spw * wgs < B) spw *= 2;
  return spw;
}

template <int NN, int MODE>
inline int launch_gru_f16_mode(const GruF16Params& Q, const GruF16Workspace& ws, const float* x, int B, int T,
                               const float* h0, float* y, float* hn, int lsel, int tchunk, int nchunks,
                               hipStream_t stream, int spw = GruF16Geom<NN>::MB) {
  using G = GruF16Geom<NN>;
  const int tiles = (B + spw - 1) / spw;
  static DynLdsGrant grant;
  auto kern = gru_f16_kernel<NN, MODE>;
  if (grant_dynamic_lds(kern, int(G::LDS_BYTES), grant)) return -3;
  hipLaunchKernelGGL(kern, dim3(tiles, nchunks), dim3(kThreads), G::LDS_BYTES, stream, Q, ws, x, B, T, h0, y, hn, lsel,
                     tchunk, spw);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

template <int NN>
inline int launch_gru_f16_nn(const GruF16Params& Q, const GruF16Workspace& ws, const float* x, int B, int T,
                             const float* h0, float* y, float* hn, int cus, hipStream_t stream) {
  using G = GruF16Geom<NN>;
  const int tiles = (B + G::MB - 1) / G::MB;
  // few stream tiles and a long input: run the time-parallel passes over (tile x time chunk) so every CU works
  if (2 * tiles <= cus && T >= 32) {
    int nchunks = (2 * cus + tiles - 1) / tiles;
    int tchunk = ((T + nchunks - 1) / nchunks + G::CS - 1) / G::CS * G::CS;
    if (tchunk < 2 * G::CS) tchunk = 2 * G::CS;
    nchunks = (T + tchunk - 1) / tchunk;
    int rc = launch_gru_f16_mode<NN, 1>(Q, ws, x, B, T, h0, y, hn, 0, tchunk, nchunks, stream);
    for (int l = 0; l < Q.base.nlayers && !rc; ++l) {
      rc = launch_gru_f16_mode<NN, 2>(Q, ws, x, B, T, h0, y, hn, l, tchunk, nchunks, stream);
      if (!rc) rc = launch_gru_f16_mode<NN, 3>(Q, ws, x, B, T, h0, y, hn, l, T, 1, stream);
    }
    if (!rc) rc = launch_gru_f16_mode<NN, 4>(Q, ws, x, B, T, h0, y, hn, 0, tchunk, nchunks, stream);
    return rc;
  }
  return launch_gru_f16_mode<NN, 0>(Q, ws, x, B, T, h0, y, hn, 0, T, 1, stream, NN == 1 ? gru_f16_spw(B, T, cus) : G::MB);
}

inline bool gru_f16_supported(const GruF16Params& Q) { return Q.kpre16 <= 128 && Q.base.odim <= 128; }
// stream tiles per workgroup: one (16 streams) until every CU has a workgroup, then two
inline int gru_f16_nn(int B) { return B > 16 * 256 ? 2 : 1; }
// workspace sizes of one call (bytes): each of the two sequence buffers, and the gate pre-activations
inline void gru_f16_workspace_bytes(int B, int T, int cus, size_t* seq_bytes, size_t* gi_bytes, size_t* sc_bytes) {
  if (gru_f16_nn(B) == 2) {
    const int tiles = (B + 31) / 32;
    *seq_bytes = GruF16Geom<2>::seq_bytes(tiles, T);
    *gi_bytes = GruF16Geom<2>::gi_floats(tiles, T) * sizeof(float);
    *sc_bytes = GruF16Geom<2>::sc_floats(tiles, T) * sizeof(float);
  } else {
    const int spw = gru_f16_spw(B, T, cus), tiles = (B + spw - 1) / spw;
    *seq_bytes = GruF16Geom<1>::seq_bytes(tiles, T);
    *gi_bytes = GruF16Geom<1>::gi_floats(tiles, T) * sizeof(float);
    *sc_bytes = GruF16Geom<1>::sc_floats(tiles, T) * sizeof(float);
  }
}

inline int launch_gru_f16(const GruF16Params& Q, const GruF16Workspace& ws, const float* x, int B, int T,
                          const float* h0, float* y, float* hn, int cus, hipStream_t stream) {
  if (!gru_f16_supported(Q)) return -4;
  return gru_f16_nn(B) == 2 ? launch_gru_f16_nn<2>(Q, ws, x, B, T, h0, y, hn, cus, stream)
                            : launch_gru_f16_nn<1>(Q, ws, x, B, T, h0, y, hn, cus, stream);
}

}  // namespace wekws
