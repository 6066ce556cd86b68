// DS-TCN, hidden_dim 256, no incoming cache -- ds256_g16.hip.h with its two phases OVERLAPPED (round 3).
//
// ds256_g16 runs, per block, [depthwise conv of all frames: vector units] barrier [pointwise conv: matrix pipe] barrier.
// Its stamps: the matrix phase is bound by the matrix pipe (672 MFMAs per SIMD and block at ~16 cycles), the depthwise
// phase by vector issue, and they take turns: 10.5 k + 5.5 k cycles per block.  Within a block they depend on each
// other -- but only COLUMN-wise: the pointwise conv of frames F needs the depthwise output of frames F (all channels),
// and the depthwise conv of frames F needs h of frames <= F (causal).  So the frame tiles are cut into a front half A
// (tiles 0 .. TA-1) and a back half B, and the block loop becomes a two-stage software pipeline:
//     stage 2i     matrix pipe: pointwise A of block i      vector units: epilogue B of block i-1, depthwise B of block i
//     stage 2i+1   matrix pipe: pointwise B of block i      vector units: epilogue A of block i,   depthwise A of block i+1
// with one barrier per stage (as many as before).  Every wave runs both jobs in one instruction stream: per K step the
// MFMAs of its o-tile x the stage's tiles, then a slice of the vector work (the epilogue, then one channel row of the
// depthwise conv per K step, then the plane stores).  Nothing forces the four waves of a SIMD into step inside a stage,
// so while one issues its vector slice the others keep the matrix pipe fed; the vector work (~360 instructions per wave
// and stage) costs less than the stage's MFMAs, and the stage runs at the matrix pipe's pace.
//
// Block floating point needs the operand scale of a depthwise output BEFORE the tile maximum of its input exists across
// all waves (the epilogue that produces the input now runs in the same stage, with no barrier between).  The scale is
// therefore derived from a bound chained over ONE block, the policy MDTC's mid tile already follows:
//     |h_(i+1)| <= |h_i| + pw_alpha_i (dw_alpha_i |h_i| + dw_beta_i) + pw_beta_i
// on the EXACT maxima of h_i's halves, which earlier stages published (a row-1-norm bound overshoots by 2^3 .. 2^6 per
// level; the split keeps fp32-level accuracy up to 2^18).  Results therefore differ from ds256_g16 / ds256_w16 in the
// last bits (other power-of-two operand scales); parity against the goldens is what is tested.
#pragma once
#include "ds256_g16.hip.h"

namespace wekws {

// depthwise conv + folded BN + ReLU + scale / split of channel row R_ for the tiles TT_ .. TE - 1 (g16_dw_row with an end)
template <int D, int R_, int TT_, int TE, int NT, bool SPLIT>
__device__ __forceinline__ void p16_dw_row(const f32x4 (&hv)[NT], const float (&dww)[9], float sa, unsigned (&ph)[NT][2],
                                           unsigned (&pl)[NT][2]) {
  if constexpr (TT_ < TE) {
    float o = dww[8];
    g16_tap<7 * D, TT_, NT>(o, hv, R_, dww[0]);
    g16_tap<6 * D, TT_, NT>(o, hv, R_, dww[1]);
    g16_tap<5 * D, TT_, NT>(o, hv, R_, dww[2]);
    g16_tap<4 * D, TT_, NT>(o, hv, R_, dww[3]);
    g16_tap<3 * D, TT_, NT>(o, hv, R_, dww[4]);
    g16_tap<2 * D, TT_, NT>(o, hv, R_, dww[5]);
    g16_tap<1 * D, TT_, NT>(o, hv, R_, dww[6]);
    g16_tap<0, TT_, NT>(o, hv, R_, dww[7]);
    o = fmaxf(o, 0.f);
    g16_split_into<(R_ & 1) != 0, SPLIT>(o, sa, ph[TT_][R_ >> 1], pl[TT_][R_ >> 1]);
    p16_dw_row<D, R_, TT_ + 1, TE, NT, SPLIT>(hv, dww, sa, ph, pl);
  }
}
template <int D, int R_, int T0, int TE, int NT, bool SPLIT>
__device__ __forceinline__ void p16_dw_one_row(const f32x4 (&hv)[NT], const float* taps_o0, float sa, unsigned (&ph)[NT][2],
                                               unsigned (&pl)[NT][2]) {
  const float4* src = reinterpret_cast<const float4*>(taps_o0 + R_ * 12);
  const float4 q0 = src[0], q1 = src[1], q2 = src[2];
  const float dww[9] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x};
  p16_dw_row<D, R_, T0, TE, NT, SPLIT>(hv, dww, sa, ph, pl);
}

template <int D, int R_, int T0, int TE, int NT, bool SPLIT>
__device__ __forceinline__ void p16_dw_rows(const f32x4 (&hv)[NT], const float* taps_o0, float sa, unsigned (&ph)[NT][2],
                                            unsigned (&pl)[NT][2]) {
  p16_dw_one_row<D, R_, T0, TE, NT, SPLIT>(hv, taps_o0, sa, ph, pl);
  if constexpr (R_ + 1 < 4) p16_dw_rows<D, R_ + 1, T0, TE, NT, SPLIT>(hv, taps_o0, sa, ph, pl);
}

// One 32-deep K step for one o-tile and the frame tiles T0 .. TE - 1 (B fragments tile by tile: the registers are needed
// for the vector work that shares the stage; the SIMD's other waves cover the LDS latency)
template <int T0, int TE, int NT, bool SPLIT>
__device__ __forceinline__ void p16_mfma_step(f32x4 (&acc)[NT], const F16Frag& a, const char* bh, const char* bl) {
#pragma unroll
  for (int tt = T0; tt < TE; ++tt) {
    const f16x8 vh = *reinterpret_cast<const f16x8*>(bh + tt * 256);
    acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.h, vh, acc[tt], 0, 0, 0);
    if constexpr (SPLIT) {
      const f16x8 vl = *reinterpret_cast<const f16x8*>(bl + tt * 256);
      acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.h, vl, acc[tt], 0, 0, 0);
      acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.l, vh, acc[tt], 0, 0, 0);
    }
  }
}

template <int V>
using p16_ic = std::integral_constant<int, V>;

template <int NT, bool SPLIT>
__global__ __launch_bounds__(kW16Threads) void ds256_p16_kernel(const StackParams P, const CallArgs A) {
  using G = W16Geom<NT>;
  constexpr int C = G::C, SS = G::SS, TT = G::TT, PB = G::PB;
  constexpr int NKS = C / 32;                                // K steps per layer
  constexpr int OTS = NKS * 128;                             // uint4 per o-tile
  constexpr int TA = (NT + 1) / 2;                           // tiles 0 .. TA-1: half A; TA .. NT-1: half B
  static_assert(NT >= 2 && NKS == 8, "two halves; the vector slices are laid out over eight K steps");
  static_assert(size_t(2 * NKS) * PB <= G::LDS_BYTES, "the operand planes of a whole layer live where the f32 tile was");
  extern __shared__ __attribute__((aligned(16))) float w16_lds[];
  char* const planes = reinterpret_cast<char*>(w16_lds);     // [K step][hi | lo][k-octet][frame][8 halves]
  float* const hbuf = w16_lds + G::SLAB / 4;                 // [256][SS] f32 tile -- only for the classifier, at the end

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform (SGPR)
  const int l15 = lane & 15, lq = lane >> 4;
  const int T = A.T;
  const int b = blockIdx.x;                                  // one utterance per workgroup
  const float* __restrict__ W = P.w;
  const int Pc = P.cache_len;
  const int o0 = wave * 16 + lq * 4;                         // this lane's 4 channels: rows of the o-tile AND of the tile h
  const int frag_off = (lq * TT + l15) * 16;
  char* const pst = planes + (wave >> 1) * 2 * PB + ((((wave & 1) * 2 + (lq >> 1)) * TT + l15) * 16 + (lq & 1) * 8);

  f32x4 acc[NT];
  f32x4 hv[NT];                                              // the residual tile: channels o0 .. o0 + 3, frames 16 tt + l15

  // ---- block floating point: cells [0] features, [2 + 2 i] / [3 + 2 i]: max|h_i| over the tiles of half A / half B
  __shared__ AmaxCell amax_cells[kAmaxCells];
  __shared__ BlockDesc blk[kAmaxMaxBlocks];
  // depthwise taps + bias, [256][12] floats per block, double-buffered by block parity (a block's depthwise halves run in
  // two different stages; the next block's taps are staged meanwhile): behind the planes in the dynamic allocation
  float* const taps_base = w16_lds + (2 * NKS * PB) / 4;
  auto taps_of = [&](int blk_i) { return taps_base + (blk_i & 1) * (C * 12); };
  amax_zero<kW16Threads>(amax_cells, kAmaxCells);
  stage_block_table<kW16Threads>(blk, P.blocks, P.nblocks);
  __syncthreads();
  if (tid < C * 3) reinterpret_cast<float4*>(taps_base)[tid] = reinterpret_cast<const float4*>(W + blk[0].dw_pk)[tid];
  const int nk = P.kpre16 / 32;
  const bool one_trip = nk <= 2 && 8 * TT <= kW16Threads && w16_x_vec_ok(A.x, A.xs_b, P.idim);
  W16XItem xi;
  if (one_trip) {
    xi = w16_load_x<TT, PB>(A.x + int64_t(b) * A.xs_b, T, P.idim, nk);
    amax_publish(amax_cells, w16_x_amax(xi));
  } else {
    amax_publish(amax_cells, amax_span<kW16Threads>(A.x + int64_t(b) * A.xs_b, T * P.idim, 0.f));
  }

  // ============================ preprocessing: h0 = [ReLU](x Wpre^T + b) ============================
  {
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) acc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const uint4* ap = reinterpret_cast<const uint4*>(W + P.pre_a16) + size_t(wave) * nk * 128 + lane;
    const float4 bias = *reinterpret_cast<const float4*>(W + P.pre_b + o0);
    float sx = 1.f, cpre = 1.f;
    if (one_trip) {
      F16Frag a[2];
#pragma unroll
      for (int st = 0; st < 2; ++st) {                       // in flight over the barriers (nk = 1: the same step twice)
        const uint4* q = ap + min(st, nk - 1) * 128;
        a[st].h = __builtin_bit_cast(f16x8, q[0]);
        a[st].l = __builtin_bit_cast(f16x8, q[64]);
      }
      __syncthreads();
      sx = pow2_scale(amax_read(amax_cells), &cpre);
      w16_store_x<PB, SPLIT>(xi, sx, planes);
      __syncthreads();
#pragma unroll
      for (int st = 0; st < 2; ++st)
        if (st < nk)
          g16_mfma_step<NT, SPLIT>(acc, a[st], planes + st * 2 * PB + frag_off, planes + st * 2 * PB + PB + frag_off);
    } else
    for (int k0 = 0; k0 < nk; k0 += 2) {                     // two K steps staged per pass
      const int steps = min(2, nk - k0);
      __syncthreads();
      sx = pow2_scale(amax_read(amax_cells), &cpre);
      for (int e = tid; e < steps * 4 * TT; e += kW16Threads) {   // item = (step, k-octet, frame)
        const int t = e % TT;
        const int q = e / TT;
        const int oct = q & 3, st = q >> 2;
        const int kf = (k0 + st) * 32 + oct * 8;
        const bool ok = t < T;
        const float* xr = A.x + int64_t(b) * A.xs_b + int64_t(t) * P.idim + kf;
        f16x8 vh, vl;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float v = (ok && kf + i < P.idim) ? xr[i] * sx : 0.f;
          _Float16 h, l;
          split16(v, h, l);
          vh[i] = h; vl[i] = l;
        }
        char* dst = planes + st * 2 * PB + (oct * TT + t) * 16;
        *reinterpret_cast<f16x8*>(dst) = vh;
        if constexpr (SPLIT) *reinterpret_cast<f16x8*>(dst + PB) = vl;
      }
      __syncthreads();
      for (int st = 0; st < steps; ++st) {
        F16Frag a[1];
        load_a16<1>(a, ap + (k0 + st) * 128, 0);
        g16_mfma_step<NT, SPLIT>(acc, a[0], planes + st * 2 * PB + frag_off, planes + st * 2 * PB + PB + frag_off);
      }
    }
    cpre *= P.pre_inv_s;                                     // 1 / (feature scale * weight scale)
    float hmax = 0.f;
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = fmaf(acc[tt][r], cpre, f4c(bias, r));
        if (P.pre_relu) v = fmaxf(v, 0.f);
        hv[tt][r] = v;
        hmax = fmaxf(hmax, fabsf(v));
      }
    }
    amax_publish(amax_cells + 2, hmax);                  // h_0: one maximum for both halves (cells [2] and [3])
    amax_publish(amax_cells + 3, hmax);
    __syncthreads();                                         // maxima published, planes free, taps of block 0 staged
  }

  // ======================================= residual blocks, pipelined over the frame halves =======================
  // bound of |h_(i+1)| over a half, from the exact maxima of h_i (see the header): m = max|h_i| over the half itself, mc =
  // max|h_i| over everything its depthwise conv reads (the half and what lies in front of it)
  auto grow = [](const BlockDesc& q, float m, float mc) { return m + fmaf(q.mid_alpha, fmaf(q.dw_alpha, mc, q.dw_beta), q.mid_beta); };
  // operand scale of the depthwise output of block `bi` for one half (bound = maximum of its input over what the conv reads)
  auto scale_of = [](const BlockDesc& q, float in_bound, float* c1) {
    float inv;
    const float s = pow2_scale(fmaf(q.dw_alpha, in_bound, q.dw_beta), &inv);
    *c1 = inv * q.inv_s1;
    return s;
  };
  // stores of the packed depthwise outputs of the tiles T0 .. TE-1 (8 bytes = this lane's 4 channels of one frame)
  auto store_planes = [&](auto t0c, auto tec, const unsigned (&ph)[NT][2], const unsigned (&pl)[NT][2]) __attribute__((always_inline)) {
    constexpr int T0 = decltype(t0c)::value, TE = decltype(tec)::value;
#pragma unroll
    for (int tt = T0; tt < TE; ++tt) {
      *reinterpret_cast<uint2*>(pst + tt * 256) = uint2{ph[tt][0], ph[tt][1]};
      if constexpr (SPLIT) *reinterpret_cast<uint2*>(pst + PB + tt * 256) = uint2{pl[tt][0], pl[tt][1]};
    }
  };
  // epilogue of the tiles T0 .. TE-1: folded bias + ReLU + residual (tcn.py:60), registers only; returns max|h| over them
  auto epilogue = [&](auto t0c, auto tec, float c1, const float4& ebias) __attribute__((always_inline)) {
    constexpr int T0 = decltype(t0c)::value, TE = decltype(tec)::value;
    float hmax = 0.f;
#pragma unroll
    for (int tt = T0; tt < TE; ++tt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = fmaxf(fmaf(acc[tt][r], c1, f4c(ebias, r)), 0.f) + hv[tt][r];
        hv[tt][r] = v;
        hmax = fmaxf(hmax, fabsf(v));
      }
    }
    return hmax;
  };
  // streaming-cache slice of block q = last `pad` frames of its input [zeros | h] (tcn.py:45-53), from the registers
  auto hand_over = [&](const BlockDesc& q) __attribute__((always_inline)) {
    if (!A.out_cache) return;
    const int pad = q.pad;
    float* const oc = A.out_cache + (int64_t(b) * C + o0) * Pc + q.cache_off;
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
      if (tt * 16 < T && tt * 16 + 16 > T - pad) {           // (wave-uniform: the tile holds frames of the slice)
        const int p = tt * 16 + l15 - (T - pad);
        if (p >= 0 && p < pad) {
#pragma unroll
          for (int r = 0; r < 4; ++r) oc[r * Pc + p] = hv[tt][r];
        }
      }
    }
    if (T < pad) {                                           // shorter than the slice: zero context in front
      const int nz = pad - T;
      for (int e = lane; e < 16 * nz; e += 64) {
        const int cc = e / nz, p = e - cc * nz;
        A.out_cache[(int64_t(b) * C + wave * 16 + cc) * Pc + q.cache_off + p] = 0.f;
      }
    }
  };

  const int nb = P.nblocks;
  float c1A = 1.f, c1B = 1.f;                                // epilogue factors of the halves of the block in flight
  F16Frag a0, a1;                                            // weight fragments of the even / odd K steps

  // One stage.  First the epilogue of the tiles [V0, VE) (results of the PREVIOUS stage's MFMAs, block with bias offset
  // `eb_off`: `do_epi`) -- before this stage's accumulators come alive, so the two halves' accumulators share registers;
  // the other waves' MFMAs run beside it.  Then the MFMAs of the tiles [M0, ME) over all eight K steps, with the vector
  // slices   ks 0 .. 3: depthwise rows 0 .. 3 of the tiles [V0, VE) with dilation D (`do_dw`)   ks 4: their plane stores
  // ks 5: the next block's taps requested (`tap_src`), stored to LDS behind the loop -- and the weight fragments of the
  // NEXT stage's first two K steps requested behind K steps 6 and 7.
  auto run_stage = [&](auto dc, auto m0c, auto mec, auto v0c, auto vec_, bool do_epi, bool do_dw, float c1e, uint32_t eb_off,
                       float sa, const uint4* ap_cur, const uint4* ap_next, const float* taps_o0, int cell_pub,
                       const float* tap_src, float* tap_dst) __attribute__((always_inline)) {
    constexpr int D = decltype(dc)::value;
    constexpr int M0 = decltype(m0c)::value, ME = decltype(mec)::value, V0 = decltype(v0c)::value, VE = decltype(vec_)::value;
    if (do_epi) {
      const float4 eb = *reinterpret_cast<const float4*>(W + eb_off + o0);
      const float hmax = epilogue(p16_ic<V0>{}, p16_ic<VE>{}, c1e, eb);
      amax_publish(amax_cells + cell_pub, hmax);
    }
    unsigned ph[NT][2], pl[NT][2];
    float4 tap_nx = float4{0.f, 0.f, 0.f, 0.f};
    const bool tap_ld = tap_src != nullptr && tid < C * 3;
#ifdef P16_SKEW
    if ((wave >> 2) & 1) __builtin_amdgcn_s_sleep(P16_SKEW);   // experiment: two of a SIMD's four waves start half a K step late
#endif
#pragma unroll
    for (int tt = M0; tt < ME; ++tt) acc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto kstep = [&](auto ksc) __attribute__((always_inline)) {
      constexpr int ks = decltype(ksc)::value;
      const char* bsrc = planes + ks * 2 * PB + frag_off;
      __builtin_amdgcn_sched_barrier(0);                     // (scheduling window = one K step: else every LDS read of the stage is hoisted)
#ifdef P16_SINGLE_A
      p16_mfma_step<M0, ME, NT, SPLIT>(acc, a0, bsrc, bsrc + PB);
      {
        const uint4* src = ks + 1 < NKS ? ap_cur + (ks + 1) * 128 : ap_next;
        F16Frag t[1];
        load_a16<1>(t, src, 0);
        a0 = t[0];
      }
#else
      p16_mfma_step<M0, ME, NT, SPLIT>(acc, (ks & 1) ? a1 : a0, bsrc, bsrc + PB);
      {                                                      // re-request the fragment two K steps ahead (next stage's at the end)
        const uint4* src = ks + 2 < NKS ? ap_cur + (ks + 2) * 128 : ap_next + (ks + 2 - NKS) * 128;
        F16Frag t[1];
        load_a16<1>(t, src, 0);
        if constexpr (ks & 1) a1 = t[0]; else a0 = t[0];
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (ks <= 3) {
        if (do_dw) p16_dw_one_row<D, ks, V0, VE, NT, SPLIT>(hv, taps_o0, sa, ph, pl);
      } else if constexpr (ks == 4) {
        if (do_dw) store_planes(p16_ic<V0>{}, p16_ic<VE>{}, ph, pl);
      } else if constexpr (ks == 5) {
        if (tap_ld) tap_nx = reinterpret_cast<const float4*>(tap_src)[tid];
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    kstep(p16_ic<0>{}); kstep(p16_ic<1>{}); kstep(p16_ic<2>{}); kstep(p16_ic<3>{});
    kstep(p16_ic<4>{}); kstep(p16_ic<5>{}); kstep(p16_ic<6>{}); kstep(p16_ic<7>{});
    if (tap_ld) reinterpret_cast<float4*>(tap_dst)[tid] = tap_nx;
  };
  // (the dilation is an instruction immediate of the DPP shifts: one stage body per dilation)
#define P16_STAGE(dil, ...)                                                      \
  switch (dil) {                                                                 \
    case 1: run_stage(p16_ic<1>{}, __VA_ARGS__); break;                          \
    case 2: run_stage(p16_ic<2>{}, __VA_ARGS__); break;                          \
    case 4: run_stage(p16_ic<4>{}, __VA_ARGS__); break;                          \
    default: run_stage(p16_ic<8>{}, __VA_ARGS__); break;                         \
  }

  // ---- prologue: depthwise A of block 0 (exact maximum of h_0), fragments of block 0
  {
    const BlockDesc q = blk[0];
    const uint4* ap1 = reinterpret_cast<const uint4*>(W + q.a1_16) + size_t(wave) * OTS + lane;
    F16Frag t[1];
    load_a16<1>(t, ap1, 0); a0 = t[0];
    load_a16<1>(t, ap1 + 128, 0); a1 = t[0];
    const float saA = scale_of(q, amax_read(amax_cells + 2), &c1A);
    unsigned ph[NT][2], pl[NT][2];
    const float* taps_o0 = taps_base + o0 * 12;
    switch (q.dil) {
      case 1: p16_dw_rows<1, 0, 0, TA, NT, SPLIT>(hv, taps_o0, saA, ph, pl); break;
      case 2: p16_dw_rows<2, 0, 0, TA, NT, SPLIT>(hv, taps_o0, saA, ph, pl); break;
      case 4: p16_dw_rows<4, 0, 0, TA, NT, SPLIT>(hv, taps_o0, saA, ph, pl); break;
      default: p16_dw_rows<8, 0, 0, TA, NT, SPLIT>(hv, taps_o0, saA, ph, pl); break;
    }
    store_planes(p16_ic<0>{}, p16_ic<TA>{}, ph, pl);
    __syncthreads();
  }
  uint32_t b1_prev = 0;
  for (int bi = 0; bi < nb; ++bi) {
    const BlockDesc q = blk[bi];
    const bool more = bi + 1 < nb;
    const BlockDesc qn = blk[more ? bi + 1 : bi];
    const uint4* ap1 = reinterpret_cast<const uint4*>(W + q.a1_16) + size_t(wave) * OTS + lane;
    const uint4* apn = reinterpret_cast<const uint4*>(W + qn.a1_16) + size_t(wave) * OTS + lane;
    const float mA = amax_read(amax_cells + 2 + 2 * bi);   // exact max|h_bi| over half A (published >= one barrier ago)
    // ---- stage 2 bi: pointwise A of block bi  |  epilogue B of block bi - 1 (-> h_bi, half B), depthwise B of block bi
    {
      // half B of h_bi is being produced in this very stage: bound it through block bi - 1 (block 0: exact, from the pre)
      float mB;
      if (bi == 0) {
        mB = amax_read(amax_cells + 3);
      } else {
        const BlockDesc qp = blk[bi - 1];
        const float pA = amax_read(amax_cells + 2 * bi), pB = amax_read(amax_cells + 1 + 2 * bi);   // halves of h_(bi-1)
        mB = grow(qp, pB, fmaxf(pA, pB));
      }
      float c1B_new;
      const float saB = scale_of(q, fmaxf(mA, mB), &c1B_new);
      P16_STAGE(q.dil, p16_ic<0>{}, p16_ic<TA>{}, p16_ic<TA>{}, p16_ic<NT>{}, bi > 0, true, c1B, b1_prev, saB, ap1, ap1,
                taps_of(bi) + o0 * 12, 3 + 2 * bi, more ? W + qn.dw_pk : nullptr, taps_of(bi + 1))
      c1B = c1B_new;
#ifndef P16_NO_HANDOVER
      hand_over(q);
#endif
      __syncthreads();
    }
    // ---- stage 2 bi + 1: pointwise B of block bi  |  epilogue A of block bi (-> h_(bi+1), half A), depthwise A of bi + 1
    {
      float c1A_new = 1.f;
      const float saA = scale_of(qn, grow(q, mA, mA), &c1A_new);   // half A reads only itself (zeros in front)
      P16_STAGE(qn.dil, p16_ic<TA>{}, p16_ic<NT>{}, p16_ic<0>{}, p16_ic<TA>{}, true, more, c1A, q.b1, saA, ap1, apn,
                taps_of(bi + 1) + o0 * 12, 2 + 2 * (bi + 1), static_cast<const float*>(nullptr), static_cast<float*>(nullptr))
      c1A = c1A_new;
      __syncthreads();
    }
    b1_prev = q.b1;
  }
#undef P16_STAGE
  // ---- epilogue B of the last block
  {
    const float4 eb = *reinterpret_cast<const float4*>(W + b1_prev + o0);
    (void)epilogue(p16_ic<TA>{}, p16_ic<NT>{}, c1B, eb);
  }

  // ---- the classifier reads the tile from LDS (conv_stack_head): written once, where the planes were
#pragma unroll
  for (int tt = 0; tt < NT; ++tt) {
#pragma unroll
    for (int r = 0; r < 4; ++r) hbuf[(o0 + r) * SS + tt * 16 + l15] = hv[tt][r];
  }
  __syncthreads();
  conv_stack_head<KIND_DS, 256, NT, kW16Threads, SS>(P, A, hbuf, w16_lds, b);
}

template <int NT, bool SPLIT>
inline int launch_ds256_p16_nts(const StackParams& P, const CallArgs& A, hipStream_t stream) {
  using G = W16Geom<NT>;
  static DynLdsGrant grant;
  auto kern = ds256_p16_kernel<NT, SPLIT>;
  constexpr size_t PLANES = size_t(2 * (G::C / 32)) * G::PB + 2 * G::C * 12 * 4;   // operand planes + two tap tables
  constexpr size_t LDS = PLANES > G::LDS_BYTES ? PLANES : G::LDS_BYTES;
  if (grant_dynamic_lds(kern, int(LDS), grant)) return -3;
  hipLaunchKernelGGL(kern, dim3(A.B), dim3(kW16Threads), LDS, stream, P, A);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// Calls WITHOUT an incoming cache, dilations 1 / 2 / 4 / 8, at most (kAmaxCells - 4) / 2 blocks, tiles of 4 or 7 x 16 frames
// (the host checks; everything else: launch_ds256_g16 / launch_ds256_w16).
int launch_ds256_p16(int nt, bool split, const StackParams& P, const CallArgs& A, hipStream_t stream);

}  // namespace wekws
