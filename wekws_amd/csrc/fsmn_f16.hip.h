// FSMN backbone (wekws/model/fsmn.py:462-495) as ONE fused gfx950 kernel, split-precision fp16 MFMA.
//
//   y = out_linear2(out_linear1( L x [ReLU(affine(memory(proj(h))))] ( ReLU(in_linear2(in_linear1(x))) ) ))
//
// Every dense layer is D[o][t] = sum_k W[o][k] B[k][t] on v_mfma_f32_16x16x32_f16 with both operands split into
// fp16 hi + lo (three MFMAs per product into one fp32 accumulator, see conv_stack_f16.hip.h).  A workgroup owns ONE
// utterance x ONE tile of 16*NT frames; the activations of the tile never leave LDS:
//
//   R0: x planes (idim)                 -> later: linear planes (linear_dim) | memory-output planes (proj_dim)
//   R1: in_linear1 planes (affine_dim)  -> later: projection tile p[proj][SS] f32 (+ left context) -> out_linear1 planes
//
// "planes" = the MFMA B operand layout [k-octet][frame][8 halves], one plane for hi and one for lo; weights are the
// A operand, pre-split and pre-packed on the host ([o-tile][k32][hi|lo][lane][8], Image::put_packed_a16) and streamed
// from L2.  All channel counts are zero-padded to multiples of 32 on the host, so no phase has edge cases.
//
// Memory block (FSMNBlock.forward, fsmn.py:214-253), stride 1 as the reference always builds it (fsmn.py:381-383):
//   out[t] = sum_j taps[j] x_pad[t + j],  x_pad = [cache (P = lorder-1+rorder) | p],  taps = [wl .. wl_last + 1 | wr]
// (the identity path is folded into the taps by wekws_amd/pack.py).  The tile keeps x_pad in f32, column j = x_pad[j]
// (first frame at column P), so the 4-frame run a lane computes reads a 16-byte aligned window; the new cache is the
// last P valid columns.  Cache tensor: (B, proj, P, layers),
// layer index innermost (fsmn.py:495, torch.cat(in_cache, dim=-1)).
//
// Block floating point (conv_stack_f16.hip.h), per packed utterance: every packed matrix carries a host-chosen power-of-two
// scale; every set of operand planes is written as v * s with s from a rigorous bound of the layer's output --
// alpha * max|input| + beta (alpha = largest row 1-norm of the matrix, beta = largest |bias|; the memory block: largest
// tap 1-norm x max(|p|, |cache|)).  Bounds chain through one FSMN layer (projection -> memory -> affine) and are
// re-anchored on the EXACT maximum of the linear planes each layer ends with, which the affine epilogue tracks and hands
// on through an LDS cell: a bound may be 2^18 too loose before the split's error exceeds 2^-22 of the tile maximum, a
// layer's chain overshoots by ~2^12.  Epilogues undo both scales in their bias FMA.  Scales are per utterance, so
// packing utterances into one workgroup does not change any result.
//
// Long inputs are cut into tiles of kFsmnTileFrames by the host, chained through the same cache format; short ones
// (<= 32 frames, e.g. a 1-s utterance at frame_skip 3, or streaming chunks) are packed 2 or 4 utterances to a workgroup.
#pragma once
#include "conv_stack_f16.hip.h"

namespace wekws {

constexpr int kFsmnThreads = 512;
constexpr int kFsmnWaves = kFsmnThreads / 64;
constexpr int kFsmnMaxLayers = 16;
constexpr int kFsmnHeldK = 5;       // layers with <= this many k-steps keep a whole o-tile pair of weights in registers
constexpr int kFsmnMaxTaps = 32;
constexpr int kFsmnTileFrames = 64;
constexpr int kFsmnLdsLimit = 160 * 1024 - 2048;   // (the maxima cells are static LDS beside the dynamic tile)

typedef float f32x8 __attribute__((ext_vector_type(8)));
struct __attribute__((packed, aligned(4))) F32x4U { float v[4]; };   // 16-byte store that only needs dword alignment

// v = hi + lo with hi = fp16(v), lo = fp16(v - hi): v_cvt_pk_f16_f32 / v_pk_add_f32 on gfx950
__device__ __forceinline__ void split16x4(f32x4 v, f16x4& h, f16x4& l) {
  h = __builtin_convertvector(v, f16x4);
  l = __builtin_convertvector(v - __builtin_convertvector(h, f32x4), f16x4);
}
__device__ __forceinline__ void split16x8(f32x8 v, f16x8& h, f16x8& l) {
  h = __builtin_convertvector(v, f16x8);
  l = __builtin_convertvector(v - __builtin_convertvector(h, f32x8), f16x8);
}

// one dense layer's block-floating constants: 1 / scale of its packed matrix; |W a + b| <= alpha * max|a| + beta
struct FsmnDense {
  float inv_s, alpha, beta;
};

struct FsmnLayer {
  uint32_t wp_a;   // proj  (Dp x LINp) packed A16, no bias
  uint32_t taps;   // [Dp][taps_ld] f32, zero padded
  uint32_t wa_a;   // affine (LINp x Dp) packed A16
  uint32_t wa_b;   // [LINp] f32
  FsmnDense wp, wa;
  float taps_l1;   // largest 1-norm of a channel's tap vector
};

// LDS cells of one utterance (maxima handed from phase to phase): [0] features, [1] incoming cache, [2 + l] the
// linear planes entering layer l (l = nlayers: entering out_linear1)
constexpr int kFsmnCells = 3 + kFsmnMaxLayers;

struct FsmnParams {
  const float* w;
  int32_t idim, odim, proj;            // true sizes (x row length, y row length, cache channels)
  int32_t kin, a1p, linp, dp, a2p, op; // padded to multiples of 32
  int32_t nlayers, ntaps, P, taps_ld;
  uint32_t in1_a, in1_b, in2_a, in2_b, out1_a, out1_b, out2_a, out2_b;
  FsmnDense in1, in2, out1, out2;
  FsmnLayer layer[kFsmnMaxLayers];
};

struct FsmnArgs {
  const float* x;         // first frame of this tile, utterance 0
  int64_t xs_b;           // floats between utterances in x
  const float* in_cache;  // (B, proj, P, L) or nullptr
  float* out_cache;       // (B, proj, P, L) or nullptr
  float* y;               // first output row of this tile
  int64_t ys_b;
  int32_t B, T;           // T: valid frames in this tile (1..16*NT)
  int32_t head_slices;    // >= 1: gridDim.y workgroups per tile share the o-tiles of out_linear2 (small calls, below)
  const NfCtx* nf;        // utterances with a non-finite input are re-computed in exact IEEE f32 (nonfinite.hip.h)
};
static __device__ __attribute__((noinline, unused)) void nf_repair_fsmn_call(const NfCtx* nf, const float* x, int64_t xs_b, const float* ic,
                                                                             float* oc, float* y, int64_t ys_b, int T, int b) {
  nf_repair_fsmn(nf, x, xs_b, ic, oc, y, ys_b, T, b);
}

// LDS plan for a tile of TT frames = U utterances x TT / U frames (bytes); shared by host (capacity check) and device
struct FsmnLds {
  int ss, seg, r0, r1, m_off;
  __host__ __device__ static inline FsmnLds make(const FsmnParams& P, int TT, int U) {
    FsmnLds g;
    // x_pad row of the projection tile: per utterance [P cache columns | TT/U frames | slack], column j = x_pad[j], so
    // the window of a 4-frame run starts 16-byte aligned; the slack absorbs the zero-padded tap groups
    g.seg = (TT / U + P.taps_ld + 3) / 4 * 4;
    int ss = U * g.seg;
    ss = (ss + 7) / 8 * 8 + 4;                                  // == 4 (mod 8): 4 rows apart -> 16 banks apart
    g.ss = ss;
    const int xb = P.kin * TT * 4, linb = P.linp * TT * 4, mb = P.dp * TT * 4;
    g.m_off = linb;
    g.r0 = xb > linb + mb ? xb : linb + mb;
    int r1 = P.a1p * TT * 4;
    if (P.dp * ss * 4 > r1) r1 = P.dp * ss * 4;
    if (P.a2p * TT * 4 > r1) r1 = P.a2p * TT * 4;
    g.r1 = r1;
    return g;
  }
  __host__ __device__ inline int bytes() const { return r0 + r1; }
};

// One dense layer: for every pair of o-tiles owned by this wave, acc = W x B over KS k-steps, then epi(ot, acc).
// bh: this lane's 16-byte item of k-step 0 / t-tile 0 in the hi plane, lo plane PLB bytes behind it.
template <int NT, class Epi>
__device__ __attribute__((always_inline)) void fsmn_gemm(const float* __restrict__ W, uint32_t a_off, uint32_t bias_off,
                                                          int MT, int KS, const char* bh, int PLB, int lane, int wave,
                                                          Epi epi, int ot_lo = 0) {
  constexpr int TT = 16 * NT;
  constexpr int KSB = 4 * TT * 16;                       // bytes per k-step inside a plane
  const int ots = KS * 128;                              // uint4 per o-tile
  int ot = ot_lo + wave * 2;                             // o-tiles [ot_lo, MT) (ot_lo even)
  if (ot >= MT) return;
  const uint4* const abase = reinterpret_cast<const uint4*>(W + a_off) + lane;
  const int k1 = min(1, KS - 1);
  // Weight fragments run one pair of k-steps ahead and are software-pipelined ACROSS o-tile pairs: the first two
  // k-steps of the next pair are requested before this pair's epilogue, so the wait for them never has to drain the
  // epilogue's stores (loads and stores share one in-order counter on gfx9).
  F16Frag a0[2], a1[2];
  load_a16<2>(a0, abase + size_t(ot) * ots, ots);
  load_a16<2>(a1, abase + size_t(ot) * ots + k1 * 128, ots);
  for (; ot < MT; ot += 2 * kFsmnWaves) {
    f32x4 acc[2][NT];
    zero_acc(acc);
    // bias of this pair: requested BEFORE the k loop so that the epilogue's wait for it does not drain the younger
    // weight prefetches behind it
    f32x4 bias[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if (bias_off) {
      bias[0] = *reinterpret_cast<const f32x4*>(W + bias_off + ot * 16 + (lane >> 4) * 4);
      bias[1] = *reinterpret_cast<const f32x4*>(W + bias_off + ot * 16 + 16 + (lane >> 4) * 4);
    }
    const uint4* ap = abase + size_t(ot) * ots;
    const int otn = (ot + 2 * kFsmnWaves < MT) ? ot + 2 * kFsmnWaves : ot;
    const uint4* apn = abase + size_t(otn) * ots;
    int ks = 0;
    for (; ks + 2 < KS; ks += 2) {
      mfma16_step<2, NT>(acc, a0, bh + ks * KSB, bh + PLB + ks * KSB);
      load_a16<2>(a0, ap + (ks + 2) * 128, ots);
      mfma16_step<2, NT>(acc, a1, bh + (ks + 1) * KSB, bh + PLB + (ks + 1) * KSB);
      load_a16<2>(a1, ap + min(ks + 3, KS - 1) * 128, ots);
    }
    mfma16_step<2, NT>(acc, a0, bh + ks * KSB, bh + PLB + ks * KSB);
    load_a16<2>(a0, apn, ots);
    if (ks + 1 < KS) mfma16_step<2, NT>(acc, a1, bh + (ks + 1) * KSB, bh + PLB + (ks + 1) * KSB);
    load_a16<2>(a1, apn + k1 * 128, ots);
    epi(ot, acc, bias);
  }
}

// Same contract as fsmn_gemm for layers with at most KH k-steps: ALL weight fragments of an o-tile pair live in
// registers, and each one is re-requested for the wave's NEXT pair right after its last use, so every fragment has a
// whole pair of MFMA work (not one k-step) to arrive and never queues behind the epilogue's stores.
template <int NT, int KH, class Epi>
__device__ __attribute__((always_inline)) void fsmn_gemm_held(const float* __restrict__ W, uint32_t a_off,
                                                               uint32_t bias_off, int MT, int KS, const char* bh,
                                                               int PLB, int lane, int wave, Epi epi, int ot_lo = 0) {
  constexpr int TT = 16 * NT;
  constexpr int KSB = 4 * TT * 16;
  const int ots = KS * 128;
  int ot = ot_lo + wave * 2;
  if (ot >= MT) return;
  const uint4* const abase = reinterpret_cast<const uint4*>(W + a_off) + lane;
  F16Frag a[KH][2];
#pragma unroll
  for (int ks = 0; ks < KH; ++ks)
    if (ks < KS) load_a16<2>(a[ks], abase + size_t(ot) * ots + ks * 128, ots);
  for (; ot < MT; ot += 2 * kFsmnWaves) {
    f32x4 acc[2][NT];
    zero_acc(acc);
    f32x4 bias[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if (bias_off) {
      bias[0] = *reinterpret_cast<const f32x4*>(W + bias_off + ot * 16 + (lane >> 4) * 4);
      bias[1] = *reinterpret_cast<const f32x4*>(W + bias_off + ot * 16 + 16 + (lane >> 4) * 4);
    }
    const int otn = (ot + 2 * kFsmnWaves < MT) ? ot + 2 * kFsmnWaves : ot;
    const uint4* apn = abase + size_t(otn) * ots;
#pragma unroll
    for (int ks = 0; ks < KH; ++ks) {
      if (ks < KS) {
        mfma16_step<2, NT>(acc, a[ks], bh + ks * KSB, bh + PLB + ks * KSB);
        load_a16<2>(a[ks], apn + ks * 128, ots);
      }
    }
    epi(ot, acc, bias);
  }
}

// NT frame tiles per workgroup = U utterances x NT / U tiles each (short inputs are packed U to a workgroup so that a
// weight fragment, whose trip through the CU's 64 B/clk L1 path is the fixed cost of a workgroup, feeds NT MFMA tiles)
template <int NT, int U>
__global__ __launch_bounds__(kFsmnThreads) void fsmn_f16_kernel(const FsmnParams P, const FsmnArgs A) {
  constexpr int TT = 16 * NT, NTU = NT / U, TTU = 16 * NTU;
  static_assert(NTU * U == NT, "tiles split evenly over the packed utterances");
  extern __shared__ __attribute__((aligned(16))) char fsmn_lds[];
  const FsmnLds G = FsmnLds::make(P, TT, U);
  char* const r0 = fsmn_lds;
  char* const r1 = fsmn_lds + G.r0;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: o-tile loops and their branches go scalar
  const int l15 = lane & 15, lq = lane >> 4;
  const int b0 = blockIdx.x * U;                              // first utterance of this workgroup
  const int T = A.T;
  const float* __restrict__ W = P.w;
  const int frag_off = (lq * TT + l15) * 16;

  // ---- block floating point: per-utterance maxima cells; scales of the planes being read (cin: what undoes them and
  //      the matrix scale) and written (sout); running maxima of what an epilogue writes (mtrk)
  __shared__ AmaxCell cells[U * kFsmnCells];
  amax_zero<kFsmnThreads>(cells, U * kFsmnCells);
  __syncthreads();
  for (int u = 0; u < U; ++u)
    if (b0 + u < A.B) {
      amax_publish(cells + u * kFsmnCells, amax_span_bits<kFsmnThreads>(A.x + int64_t(b0 + u) * A.xs_b, T * P.idim, 0.f));
      if (A.in_cache)
        amax_publish(cells + u * kFsmnCells + 1,
                     amax_span_bits<kFsmnThreads>(A.in_cache + int64_t(b0 + u) * P.proj * P.P * P.nlayers, P.proj * P.P * P.nlayers, 0.f));
    }
  __syncthreads();                                           // the staging below scales x with its maximum
  {                                                          // a NaN / Inf feature or cache element among this workgroup's utterances:
    bool bad = false;                                        // the reference's arithmetic for all of them (nonfinite.hip.h)
    for (int u = 0; u < U; ++u)
      bad |= unsigned(__builtin_amdgcn_readfirstlane(int(max(cells[u * kFsmnCells].v, cells[u * kFsmnCells + 1].v)))) >= 0x7f800000u;
    if (bad) {
      if (blockIdx.y == 0)                                   // (head slices: every slice sees it, one of them re-computes)
        for (int u = 0; u < U; ++u)
          if (b0 + u < A.B) nf_repair_fsmn_call(A.nf, A.x, A.xs_b, A.in_cache, A.out_cache, A.y, A.ys_b, T, b0 + u);
      return;
    }
  }
  float cin[U], sout[U], sinv[U], mtrk[U];
  float bnd[U], inv_cur[U];                                  // bound (or exact maximum) and 1 / scale of the planes being read
  // a dense layer reading those planes: cin = 1 / (their scale * the matrix scale); its output planes get the scale
  // of the chained bound alpha * bnd + beta, which becomes the next layer's bnd
  auto set_scales = [&](const FsmnDense& dl) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      cin[u] = inv_cur[u] * dl.inv_s;
      bnd[u] = fmaf(dl.alpha, bnd[u], dl.beta);
      sout[u] = pow2_scale(bnd[u], &sinv[u]);
      inv_cur[u] = sinv[u];
      mtrk[u] = 0.f;
    }
  };
  // re-anchor on the exact maximum of the planes just written (all threads; holds the layer-end barrier)
  auto reanchor = [&](int ci) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < U; ++u) amax_publish(cells + u * kFsmnCells + ci, mtrk[u]);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < U; ++u) bnd[u] = amax_read(cells + u * kFsmnCells + ci);
  };

  // epilogue: acc * cin (+bias) [ReLU] -> tracked, scaled by sout -> hi / lo planes of CH channels at `dst`
  auto to_planes = [&](char* dst, int CH, bool relu) __attribute__((always_inline)) {
    return [&, dst, CH, relu](int ot, f32x4 (&acc)[2][NT], const f32x4 (&bias)[2]) __attribute__((always_inline)) {
      const int plb = CH * TT * 2;
#pragma unroll
      for (int ow = 0; ow < 2; ++ow) {
        const int o = (ot + ow) * 16 + lq * 4;
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
          const int u = tt / NTU;
          f32x4 v = acc[ow][tt] * cin[u] + bias[ow];
          if (relu) v = __builtin_elementwise_max(v, f32x4{0.f, 0.f, 0.f, 0.f});
          mtrk[u] = fmaxf(fmaxf(mtrk[u], fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
          f16x4 vh, vl;
          split16x4(v * sout[u], vh, vl);
          char* d = dst + ((o >> 3) * TT + tt * 16 + l15) * 16 + (o & 4) * 2;
          *reinterpret_cast<f16x4*>(d) = vh;
          *reinterpret_cast<f16x4*>(d + plb) = vl;
        }
      }
    };
  };

  // ---------------- x tile -> planes in R0 (frames beyond T read as zero) ----------------
  {
    const int KO = P.kin / 8;
    const int plb = P.kin * TT * 2;
    const bool xvec = (P.idim % 4 == 0) && (A.xs_b % 4 == 0) && (reinterpret_cast<uintptr_t>(A.x) % 16 == 0);
    float sxu[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float inv_unused;
      sxu[u] = pow2_scale(amax_read(cells + u * kFsmnCells), &inv_unused);
    }
    // item = (k-octet, frame); 8 consecutive lanes take 8 consecutive frames of one octet (conflict-free LDS rows),
    // the next lane bit walks the octets (32-byte neighbours in memory)
    for (int e = tid; e < KO * TT; e += kFsmnThreads) {
      const int tl = e & 7;
      const int q = e >> 3;
      const int koct = q % KO, th = q / KO;
      const int f = th * 8 + tl;                              // frame slot of the tile
      const int u = f / TTU, t = f - u * TTU;
      const int k0 = koct * 8;
      const float* xr = A.x + int64_t(b0 + u) * A.xs_b + int64_t(t) * P.idim + k0;
      f32x8 xv = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (t < T && b0 + u < A.B) {
        if (xvec && k0 + 8 <= P.idim) {
          const f32x4 lo4 = *reinterpret_cast<const f32x4*>(xr), hi4 = *reinterpret_cast<const f32x4*>(xr + 4);
          xv = f32x8{lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i)
            if (k0 + i < P.idim) xv[i] = xr[i];
        }
      }
      float sxl = sxu[0];
#pragma unroll
      for (int q = 1; q < U; ++q) sxl = (u == q) ? sxu[q] : sxl;
      f16x8 vh, vl;
      split16x8(xv * sxl, vh, vl);
      char* d = r0 + (koct * TT + f) * 16;
      *reinterpret_cast<f16x8*>(d) = vh;
      *reinterpret_cast<f16x8*>(d + plb) = vl;
    }
  }
  __syncthreads();
  // ---------------- in_linear1: x planes (R0) -> a1 planes (R1) ----------------
#pragma unroll
  for (int u = 0; u < U; ++u) {
    bnd[u] = amax_read(cells + u * kFsmnCells);
    (void)pow2_scale(bnd[u], &inv_cur[u]);
  }
  set_scales(P.in1);
  fsmn_gemm<NT>(W, P.in1_a, P.in1_b, P.a1p / 16, P.kin / 32, r0 + frag_off, P.kin * TT * 2, lane, wave,
                to_planes(r1, P.a1p, false));
  __syncthreads();
  // ---------------- in_linear2 + ReLU: a1 planes (R1) -> linear planes (R0) ----------------
  set_scales(P.in2);
  fsmn_gemm<NT>(W, P.in2_a, P.in2_b, P.linp / 16, P.a1p / 32, r1 + frag_off, P.a1p * TT * 2, lane, wave,
                to_planes(r0, P.linp, true));
  reanchor(2);

  float* const pt = reinterpret_cast<float*>(r1);           // p[dp][ss] f32
  char* const mpl = r0 + G.m_off;                           // memory-output planes
  const int SS = G.ss, SEG = G.seg, Pc = P.P, L = P.nlayers;
  for (int l = 0; l < L; ++l) {
    const FsmnLayer ly = P.layer[l];
    // left context of this layer's memory block: columns [0, P) of every utterance segment; the slack columns behind
    // its frames are zeroed because the zero-padded tap groups multiply them
    {
      const int per = SEG - TTU;                               // P cache columns + slack
      for (int e = tid; e < P.dp * U * per; e += kFsmnThreads) {
        const int c = e / (U * per), r = e - c * (U * per);
        const int u = r / per, j = r - u * per;
        float v = 0.f;
        if (j < Pc && A.in_cache && c < P.proj && b0 + u < A.B)
          v = A.in_cache[((int64_t(b0 + u) * P.proj + c) * Pc + j) * L + l];
        pt[c * SS + u * SEG + (j < Pc ? j : TTU + j)] = v;
      }
    }
    // projection (no bias): linear planes (R0) -> p tile (R1), f32
#pragma unroll
    for (int u = 0; u < U; ++u) {
      cin[u] = inv_cur[u] * ly.wp.inv_s;
      bnd[u] = ly.wp.alpha * bnd[u];
    }
    fsmn_gemm<NT>(W, ly.wp_a, 0, P.dp / 16, P.linp / 32, r0 + frag_off, P.linp * TT * 2, lane, wave,
                  [&](int ot, f32x4 (&acc)[2][NT], const f32x4 (&)[2]) __attribute__((always_inline)) {
#pragma unroll
                    for (int ow = 0; ow < 2; ++ow) {
                      const int o = (ot + ow) * 16 + lq * 4;
#pragma unroll
                      for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                          pt[(o + r) * SS + (tt / NTU) * SEG + Pc + (tt % NTU) * 16 + l15] = acc[ow][tt][r] * cin[tt / NTU];
                        }
                    }
                  });
    __syncthreads();
    // memory-block output planes: bound = largest tap 1-norm x max(bound of p, |cache|)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      bnd[u] = ly.taps_l1 * fmaxf(bnd[u], amax_read(cells + u * kFsmnCells + 1));
      sout[u] = pow2_scale(bnd[u], &sinv[u]);
      inv_cur[u] = sinv[u];
    }
    // memory block: item = (channel, run of 4 frames).  Lane bits: [0:2] channel within its octet, [3:4] run, [5]
    // octet -> conflict-free 16-byte window reads, 4-way conflicts on the 2-byte plane writes.
    {
      const int plb = P.dp * TT * 2;
      const int ng = P.taps_ld / 4, nq4 = TT / 16;
      for (int e = tid; e < P.dp * (TT / 4); e += kFsmnThreads) {
        const int r = e >> 6;
        const int c = (r / nq4) * 16 + ((e >> 5) & 1) * 8 + (e & 7);
        const int q = (r % nq4) * 4 + ((e >> 3) & 3);     // 4-frame run of the tile; runs never straddle utterances
        const int u = q / (TTU / 4);
        const float* wt = W + ly.taps + c * P.taps_ld;
        const float* src = pt + c * SS + u * SEG + 4 * (q - u * (TTU / 4));
        f32x4 k = *reinterpret_cast<const f32x4*>(wt);
        f32x4 w0 = *reinterpret_cast<const f32x4*>(src);
        f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
        for (int g = 0; g < ng; ++g) {
          const f32x4 kn = *reinterpret_cast<const f32x4*>(wt + 4 * min(g + 1, ng - 1));
          const f32x4 w1 = *reinterpret_cast<const f32x4*>(src + 4 * g + 4);
          acc4[0] = fmaf(k[3], w0[3], fmaf(k[2], w0[2], fmaf(k[1], w0[1], fmaf(k[0], w0[0], acc4[0]))));
          acc4[1] = fmaf(k[3], w1[0], fmaf(k[2], w0[3], fmaf(k[1], w0[2], fmaf(k[0], w0[1], acc4[1]))));
          acc4[2] = fmaf(k[3], w1[1], fmaf(k[2], w1[0], fmaf(k[1], w0[3], fmaf(k[0], w0[2], acc4[2]))));
          acc4[3] = fmaf(k[3], w1[2], fmaf(k[2], w1[1], fmaf(k[1], w1[0], fmaf(k[0], w0[3], acc4[3]))));
          w0 = w1;
          k = kn;
        }
        float su = sout[0];                                 // (u is wave-uniform here: a wave covers 4 aligned runs)
#pragma unroll
        for (int k2 = 1; k2 < U; ++k2) su = (u == k2) ? sout[k2] : su;
        f16x4 vh, vl;
        split16x4(acc4 * su, vh, vl);
        char* d = mpl + ((c >> 3) * TT + 4 * q) * 16 + (c & 7) * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          *reinterpret_cast<_Float16*>(d + i * 16) = vh[i];
          *reinterpret_cast<_Float16*>(d + i * 16 + plb) = vl[i];
        }
      }
      // new cache = last P valid columns of x_pad
      if (A.out_cache && blockIdx.y == 0) {               // head slices recompute the backbone; one of them hands over
        for (int e = tid; e < U * P.proj * Pc; e += kFsmnThreads) {
          const int u = e / (P.proj * Pc), r = e - u * (P.proj * Pc);
          const int c = r / Pc, j = r - c * Pc;
          if (b0 + u < A.B)
            A.out_cache[((int64_t(b0 + u) * P.proj + c) * Pc + j) * L + l] = pt[c * SS + u * SEG + T + j];
        }
      }
    }
    __syncthreads();
    // affine + ReLU: memory planes -> linear planes (R0), whose exact maximum re-anchors the bounds
    set_scales(ly.wa);
    fsmn_gemm<NT>(W, ly.wa_a, ly.wa_b, P.linp / 16, P.dp / 32, mpl + frag_off, P.dp * TT * 2, lane, wave,
                  to_planes(r0, P.linp, true));
    reanchor(3 + l);
  }
  // ---------------- out_linear1: linear planes (R0) -> o1 planes (R1) ----------------
  set_scales(P.out1);
  fsmn_gemm<NT>(W, P.out1_a, P.out1_b, P.a2p / 16, P.linp / 32, r0 + frag_off, P.linp * TT * 2, lane, wave,
                to_planes(r1, P.a2p, false));
#pragma unroll
  for (int u = 0; u < U; ++u) cin[u] = inv_cur[u] * P.out2.inv_s;
  __syncthreads();
  // ---------------- out_linear2: o1 planes (R1) -> y ----------------
  {
    const int K = P.odim;
    auto store_y = [&, K](int ot, f32x4 (&acc)[2][NT], const f32x4 (&bias)[2]) __attribute__((always_inline)) {
                    // rows of y are only dword aligned (odim is odd in the recipes): 16-byte stores through a
                    // 4-byte-aligned type.  Only the last pair can run past odim; that test is wave-uniform.
                    const bool whole = (ot + 2) * 16 <= K;
#pragma unroll
                    for (int ow = 0; ow < 2; ++ow) {
                      const int o = (ot + ow) * 16 + lq * 4;
#pragma unroll
                      for (int tt = 0; tt < NT; ++tt) {
                        const int u = tt / NTU, t = (tt % NTU) * 16 + l15;
                        const f32x4 v = acc[ow][tt] * cin[u] + bias[ow];
                        float* yr = A.y + int64_t(b0 + u) * A.ys_b + int64_t(t) * K + o;
                        const bool ok = t < T && b0 + u < A.B;
                        if (whole) {
                          if (ok) *reinterpret_cast<F32x4U*>(yr) = F32x4U{{v[0], v[1], v[2], v[3]}};
                        } else if (ok) {
#pragma unroll
                          for (int r = 0; r < 4; ++r)
                            if (o + r < K) yr[r] = v[r];
                        }
                      }
                    }
                  };
    // Small calls (fewer tiles than compute units): gridDim.y workgroups run the same tile and split the o-tiles of
    // this layer -- for a CTC vocabulary it holds half of the model's weights (2599 x 140 of 756 k), and a workgroup's
    // time at one tile is the trip of its weights through the CU's 64 B/clk path.  The backbone is recomputed per
    // slice (idle CUs otherwise); results are the same numbers, each y element written by exactly one slice.
    const int MT = P.op / 16;
    const int per = (((MT + A.head_slices - 1) / A.head_slices) + 1) & ~1;
    const int ot_lo = int(blockIdx.y) * per, ot_hi = min(MT, ot_lo + per);
    if (P.a2p / 32 <= kFsmnHeldK)
      fsmn_gemm_held<NT, kFsmnHeldK>(W, P.out2_a, P.out2_b, ot_hi, P.a2p / 32, r1 + frag_off, P.a2p * TT * 2, lane,
                                     wave, store_y, ot_lo);
    else
      fsmn_gemm<NT>(W, P.out2_a, P.out2_b, ot_hi, P.a2p / 32, r1 + frag_off, P.a2p * TT * 2, lane, wave, store_y, ot_lo);
  }
}

template <int NT, int U>
inline int launch_fsmn_nt(const FsmnParams& P, const FsmnArgs& A, hipStream_t stream) {
  const int lds = FsmnLds::make(P, 16 * NT, U).bytes();
  if (lds > kFsmnLdsLimit) return -4;
  static DynLdsGrant grant;
  auto kern = fsmn_f16_kernel<NT, U>;
  if (grant_dynamic_lds(kern, lds, grant)) return -3;
  hipLaunchKernelGGL(kern, dim3((A.B + U - 1) / U, A.head_slices > 1 ? A.head_slices : 1), dim3(kFsmnThreads), lds, stream, P, A);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// nt frame tiles per utterance, u utterances per workgroup (u in {1, 2, 4}, nt * u <= 4)
int launch_fsmn_f16(int nt, int u, const FsmnParams& P, const FsmnArgs& A, hipStream_t stream);

}  // namespace wekws
