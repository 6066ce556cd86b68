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
  // short tiles (streaming steps) keep the mid tile's planes beside the depthwise planes: GEMM 1's readers and the mid
  // writers then never meet, which takes one of the four barriers out of every block of the latency chain
  static constexpr bool kMidOwn = NT <= 2;
  static constexpr int S_FLOATS = G::S_FLOATS * (kMidOwn ? 2 : 1);
  static constexpr int H_FLOATS = 2 * 64 * SS;
  static constexpr size_t LDS_BYTES = size_t(S_FLOATS + H_FLOATS) * 4;
};

// (Streaming steps of <= 16 frames with the caches resident in LDS have their own kernel: mdtc64_stream.hip.h.)
template <int NT, bool HAS_CACHE, bool SPLIT>
__global__ __launch_bounds__(kW16Threads) void mdtc64_w16_kernel(const StackParams P, const CallArgs A) {
  using G = M16Geom<NT>;
  constexpr int C = 64, U = 2, SS = G::SS, TT = 16 * NT, KS = 5;
  constexpr int MPB = Plane<C, TT>::BYTES;                   // one hi (or lo) plane of a 64-channel operand
  constexpr int UB = 2 * MPB;                                // bytes of one utterance's planes
  constexpr int NTW = NT < 4 ? NT : 4;                       // frame tiles per wave (frame half fh: tiles 4 fh ..)
  constexpr bool kMidOwn = G::kMidOwn;
  static_assert(U == G::G::U && U * UB * (kMidOwn ? 2 : 1) <= G::S_FLOATS * 4, "planes must fit the shared geometry");
  extern __shared__ __attribute__((aligned(16))) float mdtc16_lds[];
  char* const slab = reinterpret_cast<char*>(mdtc16_lds);    // [utt][hi | lo][8 oct][TT][8 halves]
  float* const hbuf = mdtc16_lds + G::S_FLOATS;              // [utt][64][SS] f32 resident activations

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
  char* const mid_u = kMidOwn ? slab + U * UB + wu * UB : slab_u;   // planes of the mid tile (M16Geom::kMidOwn)
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
  // Features.  Short tiles of 40-d input (<= 2 K steps, one 8-feature item per thread: NT <= 4) pass through registers
  // ONCE: loaded here, their maximum published, scaled / split / stored behind the barrier of the preprocessing; a second
  // trip to memory for the same 3 KB would be step time on the streaming path.  Item = (utt, K step, k-octet, frame); a
  // wave's items belong to one utterance (2 * 4 * TT is a multiple of 64), so the publishing cell is wave-uniform.
  const int nk = P.kpre16 / 32;                              // K steps of the input (40-d: 2, 80-d MFCC: 3)
  const bool one_trip = nk <= 2 && U * 8 * TT <= kW16Threads && w16_x_vec_ok(A.x, A.xs_b, P.idim);
  W16XItem xi;
  xi.dst = -1;
  auto load_item = [&]() __attribute__((always_inline)) {
    const int t = tid % TT;
    int q = tid / TT;
    const int oct = q & 3; q >>= 2;
    const int st = q % nk, u = q / nk;
    if (u >= U) return;                                      // (wave-uniform)
    const int kf = st * 32 + oct * 8;
    xi.dst = u * UB + ((st * 4 + oct) * TT + t) * 16;
    w16_fetch_x(xi, A.x + int64_t(b0 + u) * A.xs_b + int64_t(t) * P.idim + kf, A.x, (b0 + u) < A.B && t < T && kf < P.idim);
    amax_publish(amax_cells + u * kAmaxCells, w16_x_amax_bits(xi));
  };

  if (one_trip) load_item();
  for (int u = 0; u < U; ++u)
    if (b0 + u < A.B) {
      if (!one_trip)
        amax_publish(amax_cells + u * kAmaxCells, amax_span_bits<kW16Threads>(A.x + int64_t(b0 + u) * A.xs_b, T * P.idim, 0.f));
      if (HAS_CACHE)
        amax_publish(amax_cells + u * kAmaxCells + 1, amax_span_bits<kW16Threads>(A.in_cache + int64_t(b0 + u) * C * Pc, C * Pc, 0.f));
    }
  __syncthreads();
  {                                                          // a NaN / Inf feature or cache element among this workgroup's utterances:
    bool bad = false;                                        // the reference's arithmetic for all of them (nonfinite.hip.h)
    for (int u = 0; u < U; ++u) bad |= amax_inputs_bad(amax_cells + u * kAmaxCells);
    if (bad) {
      for (int u = 0; u < U; ++u)
        if (b0 + u < A.B) nf_repair_call(A, b0 + u);
      return;
    }
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
  auto gemm = [&](const F16Frag (&a)[2], const char* planes) __attribute__((always_inline)) {   // acc = A (2 K steps) x planes
#pragma unroll
    for (int tt = 0; tt < NTW; ++tt) acc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int tt = 0; tt < NTW; ++tt)
        if (tt < ntw) {
          const char* q = planes + ks * 4 * TT * 16 + frag_off + tt * 256;
          const f16x8 vh = *reinterpret_cast<const f16x8*>(q);
          acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ks].h, vh, acc[tt], 0, 0, 0);
          if constexpr (SPLIT) {                             // !SPLIT = WEKWS_HIP_PRECISION_F16: hi halves only
            const f16x8 vl = *reinterpret_cast<const f16x8*>(q + MPB);
            acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ks].h, vl, acc[tt], 0, 0, 0);
            acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ks].l, vh, acc[tt], 0, 0, 0);
          }
        }
  };

  // The vector-side constants of a block: taps + bias of channel pg (8-float record) and the two folded BN biases of this
  // lane's output channels.
  struct BlkConst { float4 q0, q1, b1, b2; };
  auto load_consts = [&](int bi) __attribute__((always_inline)) {
    const BlockDesc& nb = blk[bi];
    const float4* src = reinterpret_cast<const float4*>(W + nb.dw_pk + pg * 8);
    BlkConst k;
    k.q0 = src[0]; k.q1 = src[1];
    k.b1 = *reinterpret_cast<const float4*>(W + nb.b1 + o0);
    k.b2 = *reinterpret_cast<const float4*>(W + nb.b2 + o0);
    return k;
  };
  // ============================ preprocessing: h0 = [ReLU](x Wpre^T + b) ============================
  {
    const uint4* ap = reinterpret_cast<const uint4*>(W + P.pre_a16) + size_t(ot) * nk * 128 + lane;
    const float4 bias = *reinterpret_cast<const float4*>(W + P.pre_b + o0);
#pragma unroll
    for (int tt = 0; tt < NTW; ++tt) acc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (one_trip) {
      F16Frag a[2];
      if (active) {
#pragma unroll
        for (int st = 0; st < 2; ++st) {                     // in flight over the barriers (nk = 1: the same step twice)
          const uint4* q = ap + min(st, nk - 1) * 128;
          a[st].h = __builtin_bit_cast(f16x8, q[0]);
          a[st].l = __builtin_bit_cast(f16x8, q[64]);
        }
      }
      __syncthreads();                                       // the feature maxima are published
      if (xi.dst >= 0) {
        float inv_unused;
        const float sx = pow2_scale(amax_read(amax_cells + (xi.dst >= UB ? kAmaxCells : 0)), &inv_unused);
        w16_put_x<SPLIT>(xi, sx, slab, MPB);
      }
      __syncthreads();
#pragma unroll
      for (int st = 0; st < 2; ++st)
        if (st < nk) {
#pragma unroll
          for (int tt = 0; tt < NTW; ++tt)
            if (tt < ntw) {
              const char* q = slab_u + st * 4 * TT * 16 + frag_off + tt * 256;
              const f16x8 vh = *reinterpret_cast<const f16x8*>(q);
              acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[st].h, vh, acc[tt], 0, 0, 0);
              if constexpr (SPLIT) {
                const f16x8 vl = *reinterpret_cast<const f16x8*>(q + MPB);
                acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[st].h, vl, acc[tt], 0, 0, 0);
                acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[st].l, vh, acc[tt], 0, 0, 0);
              }
            }
        }
    } else
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
    const BlkConst kc = load_consts(bi);                     // (in flight over the producer's LDS trips)
    const float dww[KS + 1] = {kc.q0.x, kc.q0.y, kc.q0.z, kc.q0.w, kc.q1.x, kc.q1.y};
    const float4 bias1 = kc.b1, bias2 = kc.b2;
    F16Frag g1[2], g2[2];
    if constexpr (kPrefetchW) {
      if (active) {
        load_frags(g1, ap1);
        load_frags(g2, ap2);
      }
    }
    // lane tl owns the NT outputs fbase + m d (a run at stride d: the taps slide over NT + KS - 1 register-resident
    // inputs); one tile: that is frame tl for every dilation
    const bool slide = NT == 1 || slide_ok(d);
    const int fbase = (NT > 1 && slide) ? slide_base(tl, d, NT) : tl;
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
    //      (mdtc.py:55-58, no ReLU), split to fp16 hi/lo planes, and hands the channel's streaming cache over.
    //      NU = utterances this workgroup really has (a compile-time 1 or 2 per call of the lambda: a run-time skip
    //      inside the loop would fence the two utterances' LDS round trips off from each other).  Single tiles read
    //      everything of BOTH utterances before the first write: the plane writes of one utterance would otherwise order
    //      the other's reads behind them.
    auto produce = [&](auto nu_c) __attribute__((always_inline)) {
      constexpr int NU = decltype(nu_c)::value;
      constexpr bool kTwoPhase = NT == 1;
      const int c = pg;
#define fetch(u_, idx_)                                                                  \
  ({                                                                                     \
    const int ix_ = (idx_);                                                              \
    const int hoff_ = ((u_) * C + c) * SS;                                               \
    float fv_ = hbuf[hoff_ + ix_];                                                       \
    if constexpr (HAS_CACHE) {                                                           \
      const float fg_ = A.in_cache[(int64_t(b0 + (u_)) * C + c) * Pc + bd.cache_off + pad + min(ix_, -1)]; \
      fv_ = ix_ >= 0 ? fv_ : fg_;                                                        \
    } else {                                                                             \
      fv_ = ix_ >= 0 ? fv_ : 0.f;                                                        \
    }                                                                                    \
    fv_;                                                                                 \
  })
      auto hand_over = [&](int u) __attribute__((always_inline)) {   // batch path: the new cache slice goes to global memory
        if (!A.out_cache) return;
        const int hoff = (u * C + c) * SS;
        const int64_t gbase = (int64_t(b0 + u) * C + c) * Pc + bd.cache_off;
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
      };
      auto emit = [&](int u, int t, float o) __attribute__((always_inline)) {
        _Float16* ph = reinterpret_cast<_Float16*>(slab + u * UB) + ((c >> 3) * TT) * 8 + (c & 7);
        _Float16 h, l;
        split16s(o, sa[u], h, l);
        ph[t * 8] = h;
        if constexpr (SPLIT) ph[t * 8 + MPB / 2] = l;
      };
      if (slide) {
        float v[kTwoPhase ? NU : 1][NT + KS - 1];
        auto load = [&](int u, int s) __attribute__((always_inline)) {
#pragma unroll
          for (int q = 0; q < NT + KS - 1; ++q) {
            if (q >= KS - 1) v[s][q] = hbuf[(u * C + c) * SS + fbase + (q - (KS - 1)) * d];
            else v[s][q] = fetch(u, fbase + (q - (KS - 1)) * d);
          }
        };
        auto store = [&](int u, int s) __attribute__((always_inline)) {
#pragma unroll
          for (int m = 0; m < NT; ++m) {
            float o = dww[KS];
#pragma unroll
            for (int j = 0; j < KS; ++j) o = fmaf(dww[j], v[s][m + j], o);
            emit(u, fbase + m * d, o);
          }
        };
        if constexpr (kTwoPhase) {
#pragma unroll
          for (int u = 0; u < NU; ++u) { hand_over(u); load(u, u); }
#pragma unroll
          for (int u = 0; u < NU; ++u) store(u, u);
        } else {
#pragma unroll
          for (int u = 0; u < NU; ++u) { hand_over(u); load(u, 0); store(u, 0); }
        }
      } else {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          hand_over(u);
#pragma unroll 1
          for (int m = 0; m < NT; ++m) {
            const int t = tl + 16 * m;
            float o = dww[KS];
#pragma unroll
            for (int j = 0; j < KS; ++j) o = fmaf(dww[j], fetch(u, t - (KS - 1 - j) * d), o);
            emit(u, t, o);
          }
        }
      }
#undef fetch
    };
    if (b0 + 1 < A.B) produce(std::integral_constant<int, 2>{});
    else produce(std::integral_constant<int, 1>{});
    __syncthreads();
    // ---- GEMM 1 (pointwise) over the full K
    if constexpr (!kPrefetchW) { if (active) load_frags(g1, ap1); }
    if (active) gemm(g1, slab_u);
    if constexpr (!kMidOwn) __syncthreads();                 // every wave is done reading the depthwise planes
    // ---- mid = ReLU(BN1(pointwise)) written in operand order over them (mdtc.py:113-114)
#pragma unroll
    for (int tt = 0; tt < NTW; ++tt)
      if (tt < ntw) {
        const int t = (ft0 + tt) * 16 + l15;
        const f32x4 v = __builtin_elementwise_max(acc[tt] * c1 + f32x4{bias1.x, bias1.y, bias1.z, bias1.w}, f32x4{0.f, 0.f, 0.f, 0.f}) * sm;
        const f16x4 vh = __builtin_convertvector(v, f16x4);
        char* dst = mid_u + (((o0 >> 3) * TT + t) * 8 + (o0 & 7)) * 2;   // 4 consecutive channels = 8 bytes
        *reinterpret_cast<f16x4*>(dst) = vh;
        if constexpr (SPLIT)
          *reinterpret_cast<f16x4*>(dst + MPB) = __builtin_convertvector(v - __builtin_convertvector(vh, f32x4), f16x4);
      }
    __syncthreads();
    // ---- conv2 (1x1) + BN2, residual BEFORE the ReLU (mdtc.py:115-118), in place into h
    if constexpr (!kPrefetchW) { if (active) load_frags(g2, ap2); }
    if (active) gemm(g2, mid_u);
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

// usable when: hidden_dim 64, kernel size 5 (host checks); split as in launch_ds256_w16
int launch_mdtc64_w16(int nt, bool split, const StackParams& P, const CallArgs& A, hipStream_t stream);

}  // namespace wekws
