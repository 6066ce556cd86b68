// MDTC, hidden_dim 64 (examples/hi_xiaowen/s0/conf/mdtc.yaml, the speech-commands MDTC): 16-wave variant of
// conv_stack_f16_kernel<KIND_MDTC, 64, NT, 5>.  Same arithmetic and results, same LDS footprint (two utterances per
// workgroup: f32 tile 2 x 64 x SS + 2 x 256*TT bytes of operand planes = 114,688 B at NT = 7), different shape:
//
//   * 1024 threads = 4 waves per SIMD.  Per-phase clock64 sums of the 8-wave kernel (B = 1024, tools/probe) put 59 %
//     of a block in the depthwise producer, three times its vector-ALU bound: with two waves per SIMD the producer's
//     LDS round trips are not covered.
//   * wave = (utterance, o-tile, frame half): 16 accumulator registers instead of 28, all 16 waves multiply.
//   * the whole 64-channel depthwise output of both utterances is produced in ONE phase (128 rows = 64 lane-groups x
//     2 rows; lane-group g makes channel g of BOTH utterances, so its taps are loaded once), then GEMM 1 runs over the
//     full K: 4 barriers per block instead of 5.
//
// Block (mdtc.py:95-121): dw conv + BN (no ReLU) -> 1x1 + BN1 + ReLU -> 1x1 + BN2 -> + residual -> ReLU; the output of
// the last block of a stack is added to the stack sum (mdtc.py:270-273), which lives in registers.
#pragma once
#include "conv_stack_f16.hip.h"
#include "ds256_w16.hip.h"

namespace wekws {

// Geometry: the planes of Geom<KIND_MDTC, 64, NT>, the f32 tile with the conflict-free row stride of ds256_w16.hip.h
// (16 NT + 4: epilogue rows four apart and the producer's row pairs land 16 banks apart; the lane-groups' channel rows are
// permuted with w16_row so that the two groups of a 32-lane half write different dwords of the operand planes).
template <int NT>
struct M16Geom {
  using G = Geom<KIND_MDTC, 64, NT>;
  static constexpr int SS = 16 * NT + 4;
  static constexpr int S_FLOATS = G::S_FLOATS;
  static constexpr int H_FLOATS = 2 * 64 * SS;
  static constexpr size_t LDS_BYTES = size_t(S_FLOATS + H_FLOATS) * 4;
};

// LCACHE (streaming steps, NT = 1): both streams' whole caches (64 x 244 floats each) live in LDS behind the tile --
// one coalesced load at entry, taps through a selected LDS address, every slice shifted in place after its block's
// depthwise conv has read it, one coalesced store at the end -- instead of 17 blocks x 64 short strided runs in and
// out per stream with a global-memory latency exposed in every block (see ds256_stream.hip.h for the measurements
// behind this).  HAS_CACHE is ignored then (a missing input cache is a zero-filled one).
template <int NT, bool HAS_CACHE, bool SPLIT, bool LCACHE = false>
__global__ __launch_bounds__(kW16Threads) void mdtc64_w16_kernel(const StackParams P, const CallArgs A) {
  using G = M16Geom<NT>;
  constexpr int C = 64, U = 2, SS = G::SS, TT = 16 * NT, KS = 5;
  constexpr int MPB = Plane<C, TT>::BYTES;                   // one hi (or lo) plane of a 64-channel operand
  constexpr int UB = 2 * MPB;                                // bytes of one utterance's planes
  constexpr int NTW = NT < 4 ? NT : 4;                       // frame tiles per wave (frame half fh: tiles 4 fh ..)
  static_assert(U == G::G::U && U * UB <= G::S_FLOATS * 4, "planes must fit the shared geometry");
  extern __shared__ __attribute__((aligned(16))) float mdtc16_lds[];
  char* const slab = reinterpret_cast<char*>(mdtc16_lds);    // [utt][hi | lo][8 oct][TT][8 halves]
  float* const hbuf = mdtc16_lds + G::S_FLOATS;              // [utt][64][SS] f32 resident activations
  float* const cch = hbuf + G::H_FLOATS;                     // LCACHE: [utt][64][Pc] the streams' caches

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  const int T = A.T;
  const int b0 = blockIdx.x * U;
  const float* __restrict__ W = P.w;
  const int Pc = P.cache_len;
  const int pg = w16_row(tid >> 4), tl = tid & 15;           // producer: 64 lane-groups x 16 lanes; pg = the group's channel
  const int wu = wave >> 3, ot = (wave >> 1) & 3, fh = wave & 1;
  const int ft0 = fh * 4;                                    // first frame tile of this wave
  const bool uok = (b0 + wu) < A.B;
  // frame tiles this wave owns; none if its utterance is past the batch end (odd B, B = 1): such waves, like the second
  // frame half at NT < 5, only keep the barriers company -- no weight loads, no MFMAs, no epilogues (at B = 1 that
  // leaves 4 of 16 waves pulling fragments through the CU's 64 B/clk path)
  const int ntw = !uok ? 0 : NT - ft0 < NTW ? (NT - ft0 < 0 ? 0 : NT - ft0) : NTW;
  const bool active = ntw > 0;
  const int o0 = ot * 16 + lq * 4;                           // this lane's 4 output channels
  char* const slab_u = slab + wu * UB;
  float* const h_w = hbuf + wu * C * SS;
  const int frag_off = (lq * TT + ft0 * 16 + l15) * 16;      // B item of this wave's first tile, K step 0

  f32x4 acc[NTW], zsum[NTW];
#pragma unroll
  for (int tt = 0; tt < NTW; ++tt) zsum[tt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- block floating point (conv_stack_f16.hip.h): per-utterance maxima of the features and of the incoming cache
  __shared__ AmaxCell amax_cells[U * kAmaxCells];
  __shared__ BlockDesc blk[kAmaxMaxBlocks];
  AmaxCell* const cells_w = amax_cells + wu * kAmaxCells;
  amax_zero<kW16Threads>(amax_cells, U * kAmaxCells);
  stage_block_table<kW16Threads>(blk, P.blocks, P.nblocks);
  __syncthreads();
  for (int u = 0; u < U; ++u)
    if (b0 + u < A.B) {
      amax_publish(amax_cells + u * kAmaxCells, amax_span<kW16Threads>(A.x + int64_t(b0 + u) * A.xs_b, T * P.idim, 0.f));
      if (!LCACHE && HAS_CACHE)
        amax_publish(amax_cells + u * kAmaxCells + 1, amax_span<kW16Threads>(A.in_cache + int64_t(b0 + u) * C * Pc, C * Pc, 0.f));
    }

  if constexpr (LCACHE) {                                    // the two streams' caches are contiguous in global memory
    static_assert(NT == 1, "the LDS-resident cache is for single-tile streaming steps");
    const int nu = min(U, A.B - b0);
    const int n4 = (C * Pc) >> 2, tot = U * n4;              // C * Pc % 4 == 0 (host checks)
    const f32x4* src = reinterpret_cast<const f32x4*>(A.in_cache + int64_t(b0) * C * Pc);
    float cm[U] = {0.f, 0.f};
    for (int e = tid; e < tot; e += kW16Threads) {
      const f32x4 q =                                      // streamed once: non-temporal, the weights stay in L2
          (A.in_cache && e < nu * n4) ? __builtin_nontemporal_load(src + e) : f32x4{0.f, 0.f, 0.f, 0.f};
      reinterpret_cast<f32x4*>(cch)[e] = q;
      const float qm = fmaxf(fmaxf(fabsf(q[0]), fabsf(q[1])), fmaxf(fabsf(q[2]), fabsf(q[3])));
      if (e < n4) cm[0] = fmaxf(cm[0], qm); else cm[1] = fmaxf(cm[1], qm);
    }
    amax_publish(amax_cells + 1, cm[0]);
    amax_publish(amax_cells + kAmaxCells + 1, cm[1]);
    // visible to the producers: the preprocessing below ends with a barrier
  }

  // weight fragments of one GEMM (2 K steps, hi | lo)
  auto load_frags = [](F16Frag (&a)[2], const uint4* __restrict__ ap) __attribute__((always_inline)) {
    a[0].h = __builtin_bit_cast(f16x8, ap[0]);   a[0].l = __builtin_bit_cast(f16x8, ap[64]);
    a[1].h = __builtin_bit_cast(f16x8, ap[128]); a[1].l = __builtin_bit_cast(f16x8, ap[192]);
  };
  // Short tiles (streaming steps): a block is a chain of dependent phases and every exposed trip to L2 is block time, so
  // BOTH GEMMs' fragments are requested at the top of the block and arrive behind the depthwise producer (32 registers,
  // free at NT <= 2); long tiles load them at the GEMM, where seven tiles of MFMAs cover the latency.
  constexpr bool kPrefetchW = NT <= 2;
  auto gemm = [&](const F16Frag (&a)[2]) __attribute__((always_inline)) {   // acc = A (2 K steps) x planes of slab_u
#pragma unroll
    for (int tt = 0; tt < NTW; ++tt) acc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int tt = 0; tt < NTW; ++tt)
        if (tt < ntw) {
          const char* q = slab_u + ks * 4 * TT * 16 + frag_off + tt * 256;
          const f16x8 vh = *reinterpret_cast<const f16x8*>(q);
          acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ks].h, vh, acc[tt], 0, 0, 0);
          if constexpr (SPLIT) {                             // !SPLIT = WEKWS_HIP_PRECISION_F16: hi halves only
            const f16x8 vl = *reinterpret_cast<const f16x8*>(q + MPB);
            acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ks].h, vl, acc[tt], 0, 0, 0);
            acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ks].l, vh, acc[tt], 0, 0, 0);
          }
        }
  };

  // ============================ preprocessing: h0 = [ReLU](x Wpre^T + b) ============================
  {
    const int nk = P.kpre16 / 32;                            // K steps of the input (40-d: 2, 80-d MFCC: 3)
    const uint4* ap = reinterpret_cast<const uint4*>(W + P.pre_a16) + size_t(ot) * nk * 128 + lane;
    const float4 bias = *reinterpret_cast<const float4*>(W + P.pre_b + o0);
#pragma unroll
    for (int tt = 0; tt < NTW; ++tt) acc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < nk; k0 += 2) {                     // two K steps staged per pass (= the planes of an utterance)
      const int steps = min(2, nk - k0);
      __syncthreads();                                       // (first pass: the feature maxima are published)
      float sxu[U], inv_unused;                              // feature scale of each utterance
#pragma unroll
      for (int u = 0; u < U; ++u) sxu[u] = pow2_scale(amax_read(amax_cells + u * kAmaxCells), &inv_unused);
      for (int e = tid; e < U * steps * 4 * TT; e += kW16Threads) {   // item = (utt, step, k-octet, frame)
        const int t = e % TT;
        int q = e / TT;
        const int oct = q & 3; q >>= 2;
        const int st = q % steps, u = q / steps;
        const int kf = (k0 + st) * 32 + oct * 8;
        const bool ok = (b0 + u) < A.B && t < T;
        const float* xr = A.x + int64_t(b0 + u) * A.xs_b + int64_t(t) * P.idim + kf;
        const float sx = u ? sxu[1] : sxu[0];
        f16x8 vh, vl;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float v = (ok && kf + i < P.idim) ? xr[i] * sx : 0.f;
          _Float16 h, l;
          split16(v, h, l);
          vh[i] = h; vl[i] = l;
        }
        char* dst = slab + u * UB + ((st * 4 + oct) * TT + t) * 16;
        *reinterpret_cast<f16x8*>(dst) = vh;
        if constexpr (SPLIT) *reinterpret_cast<f16x8*>(dst + MPB) = vl;
      }
      __syncthreads();
      for (int st = 0; st < steps; ++st) {
        F16Frag a;
        a.h = __builtin_bit_cast(f16x8, ap[(k0 + st) * 128]);
        a.l = __builtin_bit_cast(f16x8, ap[(k0 + st) * 128 + 64]);
#pragma unroll
        for (int tt = 0; tt < NTW; ++tt)
          if (tt < ntw) {
            const char* q = slab_u + st * 4 * TT * 16 + frag_off + tt * 256;
            const f16x8 vh = *reinterpret_cast<const f16x8*>(q);
            acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.h, vh, acc[tt], 0, 0, 0);
            if constexpr (SPLIT) {
              const f16x8 vl = *reinterpret_cast<const f16x8*>(q + MPB);
              acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.h, vl, acc[tt], 0, 0, 0);
              acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.l, vh, acc[tt], 0, 0, 0);
            }
          }
      }
    }
    float cpre;
    (void)pow2_scale(amax_read(cells_w), &cpre);
    cpre *= P.pre_inv_s;                                     // 1 / (feature scale * weight scale)
    float hmax = 0.f;
#pragma unroll
    for (int tt = 0; tt < NTW; ++tt)
      if (tt < ntw) {
        const int t = (ft0 + tt) * 16 + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = fmaf(acc[tt][r], cpre, f4c(bias, r));
          if (P.pre_relu) v = fmaxf(v, 0.f);
          h_w[(o0 + r) * SS + t] = v;
          hmax = fmaxf(hmax, fabsf(v));
        }
      }
    amax_publish(cells_w + 2, hmax);
    __syncthreads();
  }

  // ======================================= residual blocks =======================================
  for (int bi = 0; bi < P.nblocks; ++bi) {
    const BlockDesc bd = blk[bi];
    const int d = bd.dil, pad = bd.pad;
    const uint4* ap1 = reinterpret_cast<const uint4*>(W + bd.a1_16) + size_t(ot) * 256 + lane;
    const uint4* ap2 = reinterpret_cast<const uint4*>(W + bd.a2_16) + size_t(ot) * 256 + lane;
    // taps + bias of channel pg (8-float record)
    float dww[KS + 1];
    {
      const float4* src = reinterpret_cast<const float4*>(W + bd.dw_pk + pg * 8);
      const float4 q0 = src[0], q1 = src[1];
      dww[0] = q0.x; dww[1] = q0.y; dww[2] = q0.z; dww[3] = q0.w; dww[4] = q1.x; dww[5] = q1.y;
    }
    const float4 bias1 = *reinterpret_cast<const float4*>(W + bd.b1 + o0);   // (in flight over the producer)
    const float4 bias2 = *reinterpret_cast<const float4*>(W + bd.b2 + o0);
    F16Frag g1[2], g2[2];
    if constexpr (kPrefetchW) {
      if (active) {
        load_frags(g1, ap1);
        load_frags(g2, ap2);
      }
    }
    const bool slide = d <= 16 && (16 % d) == 0;
    const int fbase = slide ? (tl / d) * NT * d + (tl % d) : tl;
    // operand scales of the depthwise rows, per utterance (bound through the maxima of the input tile and the cache)
    // mid tile: the bound chained behind the depthwise one (BlockDesc::mid_alpha)
    float sa[U], c1 = 0.f, sm = 1.f, c2 = 1.f;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float au = fmaxf(amax_read(amax_cells + u * kAmaxCells + 2 + bi), amax_read(amax_cells + u * kAmaxCells + 1));
      const float ba = fmaf(bd.dw_alpha, au, bd.dw_beta);
      float inv;
      sa[u] = pow2_scale(ba, &inv);
      if (u == wu) {
        c1 = inv * bd.inv_s1;
        sm = pow2_scale(fmaf(bd.mid_alpha, ba, bd.mid_beta), &c2);
        c2 *= bd.inv_s2;
      }
    }
    // ---- producer: lane-group pg makes channel pg of both utterances: depthwise dilated conv + folded BN
    //      (mdtc.py:55-58, no ReLU), split to fp16 hi/lo planes, and hands the channel's streaming cache over
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = pg;
      const int hoff = (u * C + c) * SS;
      const bool pok = (b0 + u) < A.B;
      if (!pok) continue;                                    // (workgroup-uniform: no such utterance)
      const int64_t gbase = (int64_t(pok ? b0 + u : 0) * C + c) * Pc + bd.cache_off;
#define fetch(idx_)                                                                      \
  ({                                                                                     \
    const int ix_ = (idx_);                                                              \
    float fv_;                                                                           \
    if constexpr (LCACHE) {                                                              \
      fv_ = *(ix_ >= 0 ? hbuf + hoff + ix_ : crow + pad + ix_);                          \
    } else {                                                                             \
    fv_ = hbuf[hoff + ix_];                                                              \
    if constexpr (HAS_CACHE) {                                                           \
      const float fg_ = A.in_cache[gbase + pad + min(ix_, -1)];                          \
      fv_ = ix_ >= 0 ? fv_ : (pok ? fg_ : 0.f);                                          \
    } else {                                                                             \
      fv_ = ix_ >= 0 ? fv_ : 0.f;                                                        \
    }                                                                                    \
    }                                                                                    \
    fv_;                                                                                 \
  })
      float* const crow = cch + (u * C + c) * Pc + bd.cache_off;   // LCACHE: this block's slice of (stream u, channel c)
      if (!LCACHE && A.out_cache && pok) {
        for (int p = tl; p < pad; p += 16) {
          const int src = T + p - pad;   // index into h (negative: still inside the old cache)
          float cv = hbuf[hoff + max(src, 0)];
          if constexpr (HAS_CACHE) {
            const float g = A.in_cache[gbase + pad + min(src, -1)];
            cv = src >= 0 ? cv : g;
          } else {
            cv = src >= 0 ? cv : 0.f;
          }
          A.out_cache[gbase + p] = cv;
        }
      }
      _Float16* ph = reinterpret_cast<_Float16*>(slab + u * UB) + ((c >> 3) * TT) * 8 + (c & 7);
      _Float16* pl = reinterpret_cast<_Float16*>(slab + u * UB + MPB) + ((c >> 3) * TT) * 8 + (c & 7);
      if (slide) {
        float v[NT + KS - 1];
#pragma unroll
        for (int q = 0; q < NT + KS - 1; ++q) {
          if (q >= KS - 1) v[q] = hbuf[hoff + fbase + (q - (KS - 1)) * d];
          else v[q] = fetch(fbase + (q - (KS - 1)) * d);
        }
#pragma unroll
        for (int m = 0; m < NT; ++m) {
          float o = dww[KS];
#pragma unroll
          for (int j = 0; j < KS; ++j) o = fmaf(dww[j], v[m + j], o);
          const int t = fbase + m * d;
          _Float16 h, l;
          split16s(o, sa[u], h, l);
          ph[t * 8] = h;
          if constexpr (SPLIT) pl[t * 8] = l;
        }
      } else {
#pragma unroll 1
        for (int m = 0; m < NT; ++m) {
          const int t = tl + 16 * m;
          float o = dww[KS];
#pragma unroll
          for (int j = 0; j < KS; ++j) o = fmaf(dww[j], fetch(t - (KS - 1 - j) * d), o);
          _Float16 h, l;
          split16s(o, sa[u], h, l);
          ph[t * 8] = h;
          if constexpr (SPLIT) pl[t * 8] = l;
        }
      }
      if constexpr (LCACHE) {
        // new slice = last pad frames of [slice | chunk] (mdtc.py:111), in place: the reads above and these reads
        // precede every write of the group (same wave, LDS in order); pad <= 32 -> two per lane
        float nv[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) nv[k] = fetch(min(tl + 16 * k, pad - 1) + T - pad);
#pragma unroll
        for (int k = 0; k < 2; ++k)
          if (tl + 16 * k < pad) crow[tl + 16 * k] = nv[k];
      }
#undef fetch
    }
    __syncthreads();
    // ---- GEMM 1 (pointwise) over the full K
    if constexpr (!kPrefetchW) { if (active) load_frags(g1, ap1); }
    if (active) gemm(g1);
    __syncthreads();                                         // every wave is done reading the depthwise planes
    // ---- mid = ReLU(BN1(pointwise)) written in operand order over them (mdtc.py:113-114)
#pragma unroll
    for (int tt = 0; tt < NTW; ++tt)
      if (tt < ntw) {
        const int t = (ft0 + tt) * 16 + l15;
        const f32x4 v = __builtin_elementwise_max(acc[tt] * c1 + f32x4{bias1.x, bias1.y, bias1.z, bias1.w}, f32x4{0.f, 0.f, 0.f, 0.f}) * sm;
        const f16x4 vh = __builtin_convertvector(v, f16x4);
        char* dst = slab_u + (((o0 >> 3) * TT + t) * 8 + (o0 & 7)) * 2;   // 4 consecutive channels = 8 bytes
        *reinterpret_cast<f16x4*>(dst) = vh;
        if constexpr (SPLIT)
          *reinterpret_cast<f16x4*>(dst + MPB) = __builtin_convertvector(v - __builtin_convertvector(vh, f32x4), f16x4);
      }
    __syncthreads();
    // ---- conv2 (1x1) + BN2, residual BEFORE the ReLU (mdtc.py:115-118), in place into h
    if constexpr (!kPrefetchW) { if (active) load_frags(g2, ap2); }
    if (active) gemm(g2);
    float hmax = 0.f;
#pragma unroll
    for (int tt = 0; tt < NTW; ++tt)
      if (tt < ntw) {
        const int t = (ft0 + tt) * 16 + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float* hp = h_w + (o0 + r) * SS + t;
          const float v = fmaxf(fmaf(acc[tt][r], c2, f4c(bias2, r)) + *hp, 0.f);
          if (bd.zadd) zsum[tt][r] += v;
          *hp = v;
          hmax = fmaxf(hmax, v);
        }
      }
    amax_publish(cells_w + 3 + bi, hmax);                    // = the input tile of block bi + 1
    __syncthreads();
  }


  // the backbone output is the sum of the stack outputs (mdtc.py:270-273)
#pragma unroll
  for (int tt = 0; tt < NTW; ++tt)
    if (tt < ntw) {
      const int t = (ft0 + tt) * 16 + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) h_w[(o0 + r) * SS + t] = zsum[tt][r];
    }
  __syncthreads();
  if constexpr (LCACHE) {
    if (A.out_cache) {
      const int n4 = (C * Pc) >> 2, tot = min(U, A.B - b0) * n4;
      f32x4* dst = reinterpret_cast<f32x4*>(A.out_cache + int64_t(b0) * C * Pc);
      for (int e = tid; e < tot; e += kW16Threads) __builtin_nontemporal_store(reinterpret_cast<const f32x4*>(cch)[e], dst + e);
    }
  }
  conv_stack_head<KIND_MDTC, 64, NT, kW16Threads, SS>(P, A, hbuf, reinterpret_cast<float*>(slab), b0);
}

template <int NT, bool HAS_CACHE, bool SPLIT>
inline int launch_mdtc64_w16_ntc(const StackParams& P, const CallArgs& A, hipStream_t stream) {
  using G = M16Geom<NT>;
  static DynLdsGrant grant;
  auto kern = mdtc64_w16_kernel<NT, HAS_CACHE, SPLIT>;
  if (grant_dynamic_lds(kern, int(G::LDS_BYTES), grant)) return -3;
  hipLaunchKernelGGL(kern, dim3((A.B + 1) / 2), dim3(kW16Threads), G::LDS_BYTES, stream, P, A);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

template <int NT>
inline int launch_mdtc64_w16_nt(bool split, const StackParams& P, const CallArgs& A, hipStream_t stream) {
  if (split)
    return A.in_cache ? launch_mdtc64_w16_ntc<NT, true, true>(P, A, stream)
                      : launch_mdtc64_w16_ntc<NT, false, true>(P, A, stream);
  return A.in_cache ? launch_mdtc64_w16_ntc<NT, true, false>(P, A, stream)
                    : launch_mdtc64_w16_ntc<NT, false, false>(P, A, stream);
}

inline size_t mdtc64_stream_lds_bytes(int cache_len) { return M16Geom<1>::LDS_BYTES + size_t(2) * 64 * cache_len * 4; }

template <bool SPLIT>
inline int launch_mdtc64_stream_s(const StackParams& P, const CallArgs& A, hipStream_t stream) {
  static DynLdsGrant grant;
  const size_t lds = mdtc64_stream_lds_bytes(P.cache_len);
  auto kern = mdtc64_w16_kernel<1, false, SPLIT, true>;
  if (grant_dynamic_lds(kern, int(lds), grant)) return -3;
  hipLaunchKernelGGL(kern, dim3((A.B + 1) / 2), dim3(kW16Threads), lds, stream, P, A);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// usable when: hidden_dim 64, kernel size 5 (host checks); split as in launch_ds256_w16
int launch_mdtc64_w16(int nt, bool split, const StackParams& P, const CallArgs& A, hipStream_t stream);
// streaming step (A.T <= 16) with both streams' caches resident in LDS; needs the caches to fit (host checks)
int launch_mdtc64_stream(bool split, const StackParams& P, const CallArgs& A, hipStream_t stream);

}  // namespace wekws
