// MDTC, hidden_dim 64 (examples/hi_xiaowen/s0/conf/mdtc.yaml), calls WITHOUT an incoming cache (whole utterances, first
// chunks): REGISTER-RESIDENT kernel, one utterance per 4-wave workgroup (round 3).  Same arithmetic as mdtc64_w16.hip.h --
// the returned cache is bit-identical -- in a different shape.
//
// What bounded mdtc64_w16 (0.17 of the matrix peak): 17 blocks of four short phases each, one 1024-thread workgroup per
// CU (two utterances, 114 KB of LDS), every phase fenced by a barrier over sixteen waves that all do the same thing at the
// same time -- the matrix pipe idles through the vector phases, the vector units through the matrix phases, everybody
// through the barriers and the trips to L2 at the head of a block.  A block has ~4 k cycles of work and took 13 k.
//
// Here a workgroup is ONE utterance on FOUR waves, one per SIMD: wave w owns output channels 16 w .. 16 w + 15 for all
// frames, in the accumulator layout of the 1x1 convolutions, and keeps the residual tile h, the stack sum and the
// accumulators in registers (3 x 4 NT); LDS holds only the operand planes (64 channels x frames x hi/lo = 28 KB, shared by
// the depthwise output and the mid tile) plus taps -- so FOUR workgroups fit a CU (128 registers, 4 x 32 KB) and run
// independently: while one utterance waits at a barrier or for a weight fragment, the other three use the units.  The
// overlap the 16-wave kernels could not get out of waves in lockstep comes from the hardware scheduler for free.
//   * frame layout as in ds256_g16.hip.h: column 16 tt + l of the matrix products holds frame NT l + tt - off, i.e. every
//     lane owns NT consecutive frames; off = (-T) mod NT aligns the END of the utterance with a lane boundary (frames below
//     zero are kept at zero = the causal left context), so the cache slices (the last 4 d frames) are whole lanes plus one
//     lane's last (4 d) mod NT registers: wide stores;
//   * depthwise taps (k = 5, mdtc.py:55-58): one v_fmac_f32_dpp row_shr per tap and output, a plain FMA when the source
//     stays in the lane, nothing when it leaves the 16-lane row;
//   * block = dw -> planes | barrier | GEMM 1 | barrier | mid = ReLU(BN1) -> planes | barrier | GEMM 2, epilogue in
//     registers (residual before the ReLU, mdtc.py:115-118; stack sum, mdtc.py:270-273) | barrier: four barriers of four
//     waves;
//   * the keyword head (per-frame linear, one or two outputs) from the registers; every other head, inputs with a cache,
//     feature widths above 96: mdtc64_w16.
#pragma once
#include "ds256_g16.hip.h"
#include "mdtc64_w16.hip.h"

namespace wekws {

constexpr int kG4Threads = 256;

// scale + split of TWO values (channel rows 2 p, 2 p + 1 of one column) into one packed hi and one packed lo register
// (g16_split_pair without the ReLU)
template <bool SPLIT>
__device__ __forceinline__ void g4_split_pair(float v0, float v1, float s, unsigned& ph, unsigned& pl) {
  const float t0 = v0 * s, t1 = v1 * s;
  const g16_f16x2 h = __builtin_convertvector(g16_f32x2{t0, t1}, g16_f16x2);
  ph = __builtin_bit_cast(unsigned, h);
  if constexpr (SPLIT) {
    const float d0 = t0 - static_cast<float>(h[0]), d1 = t1 - static_cast<float>(h[1]);
    pl = __builtin_bit_cast(unsigned, __builtin_convertvector(g16_f32x2{d0, d1}, g16_f16x2));
  }
}

// Depthwise dilated conv + folded BN (no ReLU) of the channel-row pair (2 P_, 2 P_ + 1) of the lane's four, all NT frames of
// the lane; taps of a channel: 8-float record {w0 .. w4, bias, -, -}; tap j multiplies the frame (4 - j) dilations back
template <int D, int P_, int NT, bool SPLIT, bool CTX = false>
__device__ __forceinline__ void g4_dw_pair(const f32x4 (&hv)[NT], const f32x4 (&cx)[NT], const float* taps_o0, float sa, char* pst, int lo_off) {
  const float4* src = reinterpret_cast<const float4*>(taps_o0 + 2 * P_ * 8);
  const float4 a0 = src[0], a1 = src[1], b0 = src[2], b1 = src[3];
  constexpr auto tiles = std::make_integer_sequence<int, NT>{};
  constexpr int RA = 2 * P_, RB = 2 * P_ + 1;
  float oa[NT], ob[NT];
#pragma unroll
  for (int tt = 0; tt < NT; ++tt) { oa[tt] = a1.y; ob[tt] = b1.y; }
  if constexpr (CTX) {   // (an incoming cache: the taps that leave the 16-lane row continue in the context tile, ds256_g16.hip.h)
    g16_tapc_tiles<4 * D, NT, RA>(oa, hv, cx, a0.x, tiles); g16_tapc_tiles<4 * D, NT, RB>(ob, hv, cx, b0.x, tiles);
    g16_tapc_tiles<3 * D, NT, RA>(oa, hv, cx, a0.y, tiles); g16_tapc_tiles<3 * D, NT, RB>(ob, hv, cx, b0.y, tiles);
    g16_tapc_tiles<2 * D, NT, RA>(oa, hv, cx, a0.z, tiles); g16_tapc_tiles<2 * D, NT, RB>(ob, hv, cx, b0.z, tiles);
    g16_tapc_tiles<1 * D, NT, RA>(oa, hv, cx, a0.w, tiles); g16_tapc_tiles<1 * D, NT, RB>(ob, hv, cx, b0.w, tiles);
    g16_tapc_tiles<0, NT, RA>(oa, hv, cx, a1.x, tiles);     g16_tapc_tiles<0, NT, RB>(ob, hv, cx, b1.x, tiles);
  } else {
  g16_tap_tiles<4 * D, NT, RA>(oa, hv, a0.x, tiles); g16_tap_tiles<4 * D, NT, RB>(ob, hv, b0.x, tiles);
  g16_tap_tiles<3 * D, NT, RA>(oa, hv, a0.y, tiles); g16_tap_tiles<3 * D, NT, RB>(ob, hv, b0.y, tiles);
  g16_tap_tiles<2 * D, NT, RA>(oa, hv, a0.z, tiles); g16_tap_tiles<2 * D, NT, RB>(ob, hv, b0.z, tiles);
  g16_tap_tiles<1 * D, NT, RA>(oa, hv, a0.w, tiles); g16_tap_tiles<1 * D, NT, RB>(ob, hv, b0.w, tiles);
  g16_tap_tiles<0, NT, RA>(oa, hv, a1.x, tiles);     g16_tap_tiles<0, NT, RB>(ob, hv, b1.x, tiles);
  }
#pragma unroll
  for (int tt = 0; tt < NT; ++tt) {
    // the pair's two halves of column 16 tt + l15: one 4-byte store per plane (holding them for an 8-byte store with the
    // other pair costs 2 NT registers through the second pair's taps -- the kernel is at the 128-register limit)
    unsigned ph, pl;
    g4_split_pair<SPLIT>(oa[tt], ob[tt], sa, ph, pl);
    *reinterpret_cast<unsigned*>(pst + tt * 256 + P_ * 4) = ph;
    if constexpr (SPLIT) *reinterpret_cast<unsigned*>(pst + lo_off + tt * 256 + P_ * 4) = pl;
  }
}
template <int D, int NT, bool SPLIT, bool CTX = false>
__device__ __forceinline__ void g4_dw_rows(const f32x4 (&hv)[NT], const f32x4 (&cx)[NT], const float* taps_o0, float sa, char* pst, int lo_off) {
  g4_dw_pair<D, 0, NT, SPLIT, CTX>(hv, cx, taps_o0, sa, pst, lo_off);
  g4_dw_pair<D, 1, NT, SPLIT, CTX>(hv, cx, taps_o0, sa, pst, lo_off);
}
template <int D, int NT, bool SPLIT>
__device__ __forceinline__ void g4_dw_rows(const f32x4 (&hv)[NT], const float* taps_o0, float sa, char* pst, int lo_off) {
  g4_dw_rows<D, NT, SPLIT, false>(hv, hv, taps_o0, sa, pst, lo_off);
}

// The last S registers of row r (frames NT l + NT - S .. NT l + NT - 1 of the lane) to S consecutive floats
template <int NT, int S>
__device__ __forceinline__ void g4_store_tail(float* dst, const f32x4 (&hv)[NT], int r) {
  struct __attribute__((packed, aligned(4))) V4 { float v[4]; };
  struct __attribute__((packed, aligned(4))) V3 { float v[3]; };
  struct __attribute__((packed, aligned(4))) V2 { float v[2]; };
  static_assert(S >= 1 && S < NT, "a proper suffix");
  constexpr int F = NT - S;
  if constexpr (S >= 4) {
    *reinterpret_cast<V4*>(dst) = V4{{hv[F][r], hv[F + 1][r], hv[F + 2][r], hv[F + 3][r]}};
    if constexpr (S == 5) dst[4] = hv[F + 4][r];
    if constexpr (S == 6) *reinterpret_cast<V2*>(dst + 4) = V2{{hv[F + 4][r], hv[F + 5][r]}};
  } else if constexpr (S == 3) {
    *reinterpret_cast<V3*>(dst) = V3{{hv[F][r], hv[F + 1][r], hv[F + 2][r]}};
  } else if constexpr (S == 2) {
    *reinterpret_cast<V2*>(dst) = V2{{hv[F][r], hv[F + 1][r]}};
  } else {
    dst[0] = hv[F][r];
  }
}
template <int NT, int S = 1>
__device__ __forceinline__ void g4_store_tail_n(int s, float* dst, const f32x4 (&hv)[NT], int r) {
  if constexpr (S < NT) {
    if (s == S) g4_store_tail<NT, S>(dst, hv, r);
    else g4_store_tail_n<NT, S + 1>(s, dst, hv, r);
  }
}

// POOLED: the head is GlobalClassifier / LastClassifier (classifier.py:26-28, :38-40) instead of the per-frame linear one.
// ALIGNED: NT divides T (the 98-frame utterance at NT = 7), i.e. off = 0 at compile time: no frames below zero, so none of
// the masks that keep them at zero (28 compare-selects per block).
// C = 64 (mdtc.yaml; four waves) or 32 (mdtc_small.yaml, round 4: two waves per utterance, eight workgroups per CU).
// CTX (round 5; per-frame linear head, NT >= 4): the call has an incoming cache -- a later chunk of 17 .. 112 frames of a
// stream (shorter chunks: mdtc64_stream).  The blocks' left context continues the lane-major tile to the left as in
// ds256_g16.hip.h's context variant: a second register tile cx (lane p = lane p - 16) reached by a row_shl for the taps that
// leave the row, plus -- where NT does not divide T -- the frames below zero inside lane 0 (registers tt < off), which are
// the slice's last columns instead of zeros.  Until round 5 these calls ran mdtc64_w16 (1.65 .. 1.8 x the first chunk's time).
template <int C, int NT, bool SPLIT, bool POOLED, bool ALIGNED, bool CTX = false>
__global__ __launch_bounds__(C * 4, 4) void mdtc_g4_kernel(const StackParams P, const CallArgs A) {
  static_assert(!CTX || (!POOLED && NT >= 4), "context variant: keyword head, a context of one 16-lane row");
  constexpr int TT = 16 * NT;
  constexpr int NTHR = C * 4;                                // one wave per o-tile of 16 channels
  constexpr int KS = C / 32;                                 // K steps of the block GEMMs = K steps of features staged per pass
  constexpr int MPB = Plane<C, TT>::BYTES;                   // one hi (or lo) plane of the C-channel operand
  constexpr int XK = C == 64 ? 3 : 2;                        // K steps of features at most (96 / 64 dims)
  constexpr int XI = (XK * 4 * TT + NTHR - 1) / NTHR;        // feature items per thread
  extern __shared__ __attribute__((aligned(16))) float g4_lds[];
  char* const planes = reinterpret_cast<char*>(g4_lds);      // [hi | lo][k-octet 0..7][column][8 halves]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  const int T = A.T;
  const int b = blockIdx.x;                                  // one utterance per workgroup
  const float* __restrict__ W = P.w;
  const int Pc = P.cache_len;
  const int o0 = wave * 16 + lq * 4;                         // this lane's 4 channels: rows of the o-tile AND of the tile h
  const int off = ALIGNED ? 0 : (NT - T % NT) % NT;          // frame of column 16 tt + l: NT l + tt - off
  const int frag_off = (lq * TT + l15) * 16;
  // where this lane's 4 channels of column 16 tt + l15 sit in the hi plane: + tt * 256; lo: + MPB
  char* const pst = planes + ((o0 >> 3) * TT + l15) * 16 + (o0 & 7) * 2;

  f32x4 acc[NT], hv[NT];
  f32x4 cx[CTX ? NT : 1];                                    // CTX: the current block's left context (registers shared with acc)
  // The head is linear, so the sum of the stack outputs (mdtc.py:270-273) never has to exist: every stack end adds ITS
  // contribution to the lane's partial head sums  yp[tt][k] = sum_r Wc[k][o0 + r] * out[r][frame tt]  (2 NT registers
  // instead of 4 NT for the sum itself -- the kernel is at the 128-register limit of four workgroups per CU).
  float yp[POOLED ? 1 : NT][2];
  // POOLED heads need even less: the time pooling commutes with the stack sum as well, so a lane keeps one running value per
  // channel -- the sum over its (valid) frames (Global) or its frame T - 1 (Last: register NT - 1 of one lane, the
  // utterance's end is lane-aligned) -- of every stack output.
  float zs[4] = {0.f, 0.f, 0.f, 0.f};

  __shared__ AmaxCell amax_cells[kAmaxCells];
  __shared__ BlockDesc blk[kAmaxMaxBlocks];
  __shared__ __attribute__((aligned(16))) float taps[2][C * 8];   // taps + bias records of the current / next block
#ifdef WEKWS_G4_STAGGER                                     // (A/B builds only, DESIGN.md 3.1b: co-resident workgroups start out of phase)
  for (int k = int((blockIdx.x >> 8) & 3u) * WEKWS_G4_STAGGER; k > 0; --k) __builtin_amdgcn_s_sleep(16);
#endif
  amax_zero<NTHR>(amax_cells, kAmaxCells);
  stage_block_table<NTHR>(blk, P.blocks, P.nblocks);
  // taps of a block: C * 32 bytes copied from the weight image straight into LDS by waves 0 (and 1) (global_load_lds); nobody waits
  // for the copy explicitly -- the issuing waves consume weight fragments they requested AFTER it (loads return in order)
  // before the barrier in front of the depthwise phase that reads the taps
  auto stage_taps = [&](int bi, int ln) __attribute__((always_inline)) {
    if (wave < C / 32)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(W + __builtin_amdgcn_readfirstlane(blk[bi].dw_pk) + (wave * 64 + ln) * 4),
                                       (__attribute__((address_space(3))) void*)(&taps[bi & 1][0] + wave * 256), 16, 0, 0);
  };

  // ---- features: every thread owns up to XI items (K step, k-octet, column) = 8 consecutive features of a frame; they
  //      pass through registers once (maximum published, then scaled, split and stored behind the barrier)
  const int nk = P.kpre16 / 32;                              // K steps of the input (40-d: 2, 80-d: 3)
  const int nitems = nk * 4 * TT;
  W16XItem xi[XI];
  float xmax = 0.f;
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int e = tid + i * NTHR;
    const int n = e % TT, q = e / TT;
    const int f = NT * (n & 15) + (n >> 4) - off;
    const int oct = q & 3, st = q >> 2;
    const int kf = st * 32 + oct * 8;
    const bool has = e < nitems;
    xi[i].dst = has ? ((st % KS) * 4 + oct) * TT * 16 + n * 16 : -1;   // (KS K steps fit the planes: later ones are staged where earlier ones were)
    w16_fetch_x(xi[i], A.x + int64_t(b) * A.xs_b + int64_t(f) * P.idim + kf, A.x, has && f >= 0 && f < T && kf < P.idim);
    xmax = amax_merge(xmax, w16_x_amax_bits(xi[i]));          // (bit patterns: a NaN / Inf stays on top)
  }
  __syncthreads();                                           // cells zeroed, table staged
  stage_taps(0, lane);
  amax_publish(amax_cells, xmax);
  if constexpr (CTX)     // the depthwise rows are bounded through max(tile, incoming cache), like mdtc64_w16
    amax_publish(amax_cells + 1, amax_span_bits<NTHR>(A.in_cache + int64_t(b) * C * Pc, C * Pc, 0.f));

  // ============================ preprocessing: h0 = [ReLU](x Wpre^T + b) ============================
  {
    const uint4* ap = reinterpret_cast<const uint4*>(W + P.pre_a16) + size_t(wave) * nk * 128 + lane;
    const float4 bias = *reinterpret_cast<const float4*>(W + P.pre_b + o0);
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) acc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();                                         // the feature maximum is published
    float cpre;
    const float sx = pow2_scale(amax_read(amax_cells), &cpre);
    for (int k0 = 0; k0 < nk; k0 += KS) {                    // KS K steps (= the planes) per pass
      if (k0) __syncthreads();                               // the K steps before have been multiplied
#pragma unroll
      for (int i = 0; i < XI; ++i) {
        const int st = (tid + i * NTHR) / (4 * TT);          // the item's K step
        if (st >= k0 && st < k0 + KS) w16_put_x<SPLIT>(xi[i], sx, planes, MPB);
      }
      __syncthreads();
      for (int st = k0; st < min(k0 + KS, nk); ++st) {
        F16Frag a;
        a.h = __builtin_bit_cast(f16x8, ap[st * 128]);
        a.l = __builtin_bit_cast(f16x8, ap[st * 128 + 64]);
        const char* bh = planes + (st % KS) * 4 * TT * 16 + frag_off;
        g16_mfma_step<NT, SPLIT>(acc, a, bh, bh + MPB);
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
        if (l15 == 0 && tt < off) v = 0.f;                   // frames below zero: the causal left context
        hv[tt][r] = v;
        hmax = fmaxf(hmax, fabsf(v));
      }
    }
    amax_publish(amax_cells + 2, hmax);
  }

  // ======================================= residual blocks =======================================
  auto frag_ptr = [&](uint32_t a16) __attribute__((always_inline)) {
    return reinterpret_cast<const uint4*>(W + __builtin_amdgcn_readfirstlane(a16)) + size_t(wave) * (KS * 128);
  };
  auto load_frag = [&](F16Frag& a, const uint4* ap, int ln) __attribute__((always_inline)) {   // one K step: hi | lo
    a.h = __builtin_bit_cast(f16x8, ap[ln]);
    a.l = __builtin_bit_cast(f16x8, ap[ln + 64]);
  };
#pragma unroll
  for (int tt = 0; tt < (POOLED ? 1 : NT); ++tt) yp[tt][0] = yp[tt][1] = 0.f;
  F16Frag g1a;                                               // GEMM 1, first K step: requested a block ahead
  __syncthreads();                                           // (A) maximum published, planes free, table / taps visible
  load_frag(g1a, frag_ptr(blk[0].a1_16), lane);
  for (int bi = 0; bi < P.nblocks; ++bi) {
    // (the table comes from LDS, i.e. in vector registers: what is used in addresses and branches is moved to scalar
    // registers, or every bias / cache address becomes a 64-bit per-lane value that lives through the block loop)
    // (and the lane's channel / lane numbers are made opaque per block: the 64-bit per-lane addresses derived from them --
    // biases, fragments, classifier rows, cache rows -- are otherwise all computed once in front of the loop and held)
    int o0b = o0, laneb = lane;
    asm volatile("" : "+v"(o0b), "+v"(laneb));
    // A workgroup that is ahead steps back (priority 3 .. 0 over the network's quarters): a SIMD serves its oldest wave
    // first, so of the four workgroups sharing a CU the oldest races through and the youngest finishes alone, without
    // anybody covering its latencies -- with ONE round of workgroups per CU (1024 utterances on 256 CUs) that tail is the
    // end of the kernel: 0.1425 -> 0.1349 ms at B = 1024 (-5.3 %), -1.6 % at B = 8192.
    {
      const int q = (bi * 4) / P.nblocks;                    // (the instruction takes an immediate)
      if (q == 0) __builtin_amdgcn_s_setprio(3);
      else if (q == 1) __builtin_amdgcn_s_setprio(2);
      else if (q == 2) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
    }
    BlockDesc bd = blk[bi];
    bd.pad = __builtin_amdgcn_readfirstlane(bd.pad);
    bd.dil = __builtin_amdgcn_readfirstlane(bd.dil);
    bd.cache_off = __builtin_amdgcn_readfirstlane(bd.cache_off);
    bd.zadd = __builtin_amdgcn_readfirstlane(bd.zadd);
    bd.b1 = __builtin_amdgcn_readfirstlane(bd.b1);
    bd.b2 = __builtin_amdgcn_readfirstlane(bd.b2);
    const int pad = bd.pad;
    if (bi + 1 < P.nblocks) stage_taps(bi + 1, laneb);
    // ---- operand scales (conv_stack_f16.hip.h): the depthwise rows are bounded through the maximum of the input tile,
    //      the mid tile through the bound chained behind it (BlockDesc::mid_alpha)
    //      (wave-uniform values, moved to scalar registers: the kernel is at the 128-register limit)
    auto uni = [](float v) __attribute__((always_inline)) {
      return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v)));
    };
    float c1v, c2v;
    const float au = CTX ? fmaxf(amax_read(amax_cells + 2 + bi), amax_read(amax_cells + 1)) : amax_read(amax_cells + 2 + bi);
    const float ba = fmaf(bd.dw_alpha, au, bd.dw_beta);
    const float sa = uni(pow2_scale(ba, &c1v));
    const float c1 = uni(c1v * bd.inv_s1);
    const float sm = uni(pow2_scale(fmaf(bd.mid_alpha, ba, bd.mid_beta), &c2v));
    const float c2 = uni(c2v * bd.inv_s2);

    // ---- CTX: the block's left context from its slice of the incoming cache.  Position q = NT lane + tt holds frame q - off;
    //      frame f < 0 is slice column pad + f.  Lane p of cx holds lane p - 16 of the tile (whole lanes as 28-byte runs where
    //      all NT columns exist), and the frames below zero inside lane 0 (tt < off) take the slice's last columns.
    if constexpr (CTX) {
      const float* const ic = A.in_cache + int64_t(b) * C * Pc + bd.cache_off;
      const int c0 = pad - off + NT * (l15 - 16);            // slice column of this lane's first context position
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) cx[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (c0 >= 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) g16_load_run<NT>(ic + unsigned((o0b + r) * Pc + c0), cx, r);
      } else if (c0 + NT > 0) {
#pragma unroll
        for (int tt = 0; tt < NT; ++tt)
          if (c0 + tt >= 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) cx[tt][r] = ic[unsigned((o0b + r) * Pc + c0 + tt)];
          }
      }
      if constexpr (!ALIGNED) {
        if (l15 == 0) {
#pragma unroll
          for (int tt = 0; tt < NT - 1; ++tt)
            if (tt < off) {
              const int col = pad - off + tt;
#pragma unroll
              for (int r = 0; r < 4; ++r) hv[tt][r] = col >= 0 ? ic[unsigned((o0b + r) * Pc + col)] : 0.f;
            }
        }
      }
    }

    // ---- the block's streaming-cache slice = the last `pad` frames of its input tile [zeros | h], from the registers:
    //      whole lanes (NT consecutive columns each) plus one lane's last pad mod NT registers
    if (A.out_cache) {
      float* const ocb = A.out_cache + int64_t(b) * C * Pc + bd.cache_off;   // (wave-uniform base + 32-bit lane offsets)
      const int p0 = NT * l15 - off - (T - pad);             // slice column of this lane's first frame
      if (p0 >= 0 && p0 + NT <= pad) {
#pragma unroll
        for (int r = 0; r < 4; ++r) g16_store_run<NT>(ocb + unsigned((o0b + r) * Pc + p0), hv, r);
      } else if (p0 < 0 && p0 + NT > 0) {
        if constexpr (NT > 1) {
#pragma unroll
          for (int r = 0; r < 4; ++r) g4_store_tail_n<NT>(p0 + NT, ocb + unsigned((o0b + r) * Pc), hv, r);
        }
      }
      if (T + off < pad) {                                   // shorter than the slice: zero context in front
        const int nz = pad - T - off;
        for (int e = lane; e < 16 * nz; e += 64) {
          const int cc = e / nz, p = e - cc * nz;
          const int64_t at = (int64_t(b) * C + wave * 16 + cc) * Pc + bd.cache_off + p;
          A.out_cache[at] = CTX ? A.in_cache[at + T] : 0.f;   // (CTX: what was the tail of the incoming slice moves forward)
        }
      }
    }

    // ---- depthwise dilated conv + folded BN (mdtc.py:55-58, no ReLU), scaled, split, to the operand planes
    {
      const float* taps_o0 = &taps[bi & 1][0] + o0 * 8;
      switch (bd.dil) {                                      // (the host admits this kernel for dilations 1 / 2 / 4 / 8 only)
        case 1: if constexpr (CTX) g4_dw_rows<1, NT, SPLIT, true>(hv, cx, taps_o0, sa, pst, MPB); else g4_dw_rows<1, NT, SPLIT>(hv, taps_o0, sa, pst, MPB); break;
        case 2: if constexpr (CTX) g4_dw_rows<2, NT, SPLIT, true>(hv, cx, taps_o0, sa, pst, MPB); else g4_dw_rows<2, NT, SPLIT>(hv, taps_o0, sa, pst, MPB); break;
        case 4: if constexpr (CTX) g4_dw_rows<4, NT, SPLIT, true>(hv, cx, taps_o0, sa, pst, MPB); else g4_dw_rows<4, NT, SPLIT>(hv, taps_o0, sa, pst, MPB); break;
        case 8: if constexpr (CTX) g4_dw_rows<8, NT, SPLIT, true>(hv, cx, taps_o0, sa, pst, MPB); else g4_dw_rows<8, NT, SPLIT>(hv, taps_o0, sa, pst, MPB); break;
        default: break;                                      // (d = 8 as the `default` arm came out wrong in this kernel: DESIGN.md 3.1a)
      }
    }
    __syncthreads();                                         // (B1) the depthwise planes are written

    // ---- GEMM 1 (pointwise) over the full K = C
    const float4 bias1 = *reinterpret_cast<const float4*>(W + bd.b1 + o0b);
    F16Frag gb;                                              // second K step: arrives behind the first one's MFMAs
    if constexpr (KS == 2) load_frag(gb, frag_ptr(bd.a1_16) + 128, laneb);
    g16_mfma_step<NT, SPLIT, true>(acc, g1a, planes + frag_off, planes + MPB + frag_off);   // (C = 0: no cleared accumulators)
    if constexpr (KS == 2) g16_mfma_step<NT, SPLIT>(acc, gb, planes + 4 * TT * 16 + frag_off, planes + 4 * TT * 16 + MPB + frag_off);
    F16Frag g2a;                                             // GEMM 2, first K step: arrives behind the mid epilogue
    load_frag(g2a, frag_ptr(bd.a2_16), laneb);
    __syncthreads();                                         // (B2) every wave is done reading the depthwise planes

    // ---- mid = ReLU(BN1(pointwise)) written in operand order over them (mdtc.py:113-114); its scale sm is a power of two
    //      and positive, so ReLU(acc c1 + b) sm = ReLU(acc (c1 sm) + b sm) bit for bit: folded into the constants
    const float c1s = uni(c1 * sm);
    const float4 b1s = float4{bias1.x * sm, bias1.y * sm, bias1.z * sm, bias1.w * sm};
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = fmaxf(fmaf(acc[tt][r], c1s, f4c(b1s, r)), 0.f);
      const f16x4 vh = __builtin_convertvector(f32x4{v[0], v[1], v[2], v[3]}, f16x4);
      *reinterpret_cast<f16x4*>(pst + tt * 256) = vh;
      if constexpr (SPLIT) {
        const f32x4 hf = __builtin_convertvector(vh, f32x4);
        *reinterpret_cast<f16x4*>(pst + MPB + tt * 256) =
            __builtin_convertvector(f32x4{v[0] - hf[0], v[1] - hf[1], v[2] - hf[2], v[3] - hf[3]}, f16x4);
      }
    }
    __syncthreads();                                         // (B3) the mid planes are written
    const float4 bias2 = *reinterpret_cast<const float4*>(W + bd.b2 + o0b);   // (arrives behind GEMM 2)

    // ---- conv2 (1x1) + BN2, residual BEFORE the ReLU (mdtc.py:115-118), registers only
    if constexpr (KS == 2) load_frag(gb, frag_ptr(bd.a2_16) + 128, laneb);
    g16_mfma_step<NT, SPLIT, true>(acc, g2a, planes + frag_off, planes + MPB + frag_off);
    if constexpr (KS == 2) g16_mfma_step<NT, SPLIT>(acc, gb, planes + 4 * TT * 16 + frag_off, planes + 4 * TT * 16 + MPB + frag_off);
    if (bi + 1 < P.nblocks) load_frag(g1a, frag_ptr(blk[bi + 1].a1_16), laneb);   // next block's GEMM 1: behind its depthwise phase
    float hmax = 0.f;
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = fmaxf(fmaf(acc[tt][r], c2, f4c(bias2, r)) + hv[tt][r], 0.f);
        if (l15 == 0 && tt < off) v = 0.f;                   // frames below zero stay the (zero) left context
        hv[tt][r] = v;
        hmax = fmaxf(hmax, v);
      }
    }
    if constexpr (POOLED) {
      if (bd.zadd) {
        const int lastl = (T + off) / NT - 1;                // the lane that holds frame T - 1 (in register NT - 1)
        if (P.head == HEAD_GLOBAL) {
#pragma unroll
          for (int tt = 0; tt < NT; ++tt) {
            const bool in = NT * l15 + tt - off < T;         // (frames below zero are zeros already)
#pragma unroll
            for (int r = 0; r < 4; ++r) zs[r] += in ? hv[tt][r] : 0.f;
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) zs[r] += l15 == lastl ? hv[NT - 1][r] : 0.f;
        }
      }
    } else
    if (bd.zadd) {                                           // the block closes a stack: its output enters the head
      const int K = P.odim;
      const float4 w0 = *reinterpret_cast<const float4*>(W + P.head_w + o0b);
      const float4 w1 = *reinterpret_cast<const float4*>(W + P.head_w + (K > 1 ? C : 0) + o0b);
      const HeadPairs hw(w0, w1);                            // (packed, operand selects spelled out: pk_safe.hip.h)
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        pk_f32x2 pp{yp[tt][0], yp[tt][1]};
        head_fma4(pp, hw.x, hw.y, hw.z, hw.w, hv[tt]);
        yp[tt][0] = pp.x; yp[tt][1] = pp.y;
      }
    }
    amax_publish(amax_cells + 3 + bi, hmax);                 // = the input tile of block bi + 1
    __syncthreads();                                         // (B4) maximum published, planes free
  }

  if constexpr (POOLED) {
    // ---- pooled heads: m = mean_t / last frame of the stack sum -> W2 ReLU(W1 m + b1) + b2.  The 16 lanes of a row add
    //      their channel sums (xor butterfly), the 64 pooled values meet in LDS, one small MLP on the vector units.
    float* const mvec = g4_lds;                              // [C] pooled channels, then [head_hidden]
    float* const hid = g4_lds + C;
    int th = threadIdx.x;
    asm volatile("" : "+v"(th));
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = zs[r];
#pragma unroll
      for (int m = 1; m < 16; m <<= 1) v += __shfl_xor(v, m);
      zs[r] = v;
    }
    const int c0 = (th >> 4) * 4;                            // = o0
    if ((th & 15) == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = zs[r];
        if (P.head == HEAD_GLOBAL) {
          if (A.gsum) {                                      // long inputs: running sums across the tiles of the call
            float* gp = A.gsum + int64_t(b) * C + c0 + r;
            if (!A.first_tile) v += *gp;
            if (!A.last_tile) *gp = v;
          }
          v = v / float(A.T_total);
        }
        mvec[c0 + r] = v;
      }
    }
    __syncthreads();
    if (A.last_tile) {
      const int HH = P.head_hidden, K = P.odim;
      for (int j = th; j < HH; j += NTHR) {
        const float* w1 = W + P.head_w + j * C;
        float v = W[P.head_b + j];
        for (int c = 0; c < C; ++c) v = fmaf(w1[c], mvec[c], v);
        hid[j] = fmaxf(v, 0.f);
      }
      __syncthreads();
      for (int k = th; k < K; k += NTHR) {
        const float* w2 = W + P.head_w2 + k * HH;
        float v = W[P.head_b2 + k];
        for (int j = 0; j < HH; ++j) v = fmaf(w2[j], hid[j], v);
        if (P.sigmoid) v = sigmoidf_(v);
        A.y[int64_t(b) * A.ys_b + k] = v;
      }
    }
  } else
  // ---- keyword head (per-frame linear, one or two outputs; classifier.py:63-67) on the sum of the stack outputs
  //      (mdtc.py:270-273): the C / 4 partial sums per output (waves x 4 channel groups) meet in LDS
  {
    const int K = P.odim;
    constexpr int PS = 32 * NT + 16;                          // floats per partial row: 16 lanes x (NT frames x 2 outputs), padded
    float* const part = g4_lds;
    int th = threadIdx.x;                                    // (recomputed behind the block loop instead of carried through it)
    asm volatile("" : "+v"(th));
    float* dst = part + (th >> 4) * PS + 2 * NT * (th & 15);   // row = wave * 4 + lq
#pragma unroll
    for (int tt = 0; tt < NT; ++tt)
      *reinterpret_cast<float2*>(dst + 2 * tt) = float2{yp[tt][0], yp[tt][1]};   // column NT l15 + tt = frame NT l15 + tt - off
    __syncthreads();
    for (int e = th; e < 2 * TT; e += NTHR) {                // item = (column, output)
      const int t = (e >> 1) - off, k = e & 1;
      if (t >= 0 && t < T && k < K) {
        float v = W[P.head_b + k];
#pragma unroll
        for (int i = 0; i < NTHR / 16; ++i) v += part[i * PS + e];
        if (P.sigmoid) v = sigmoidf_(v);
        A.y[int64_t(b) * A.ys_b + int64_t(t) * K + k] = v;
      }
    }
  }
  // A NaN / Inf feature or cache element (cells [0] / [1], untouched since the top, at or above 0x7f800000): the utterance is
  // re-computed with the reference's arithmetic HERE, where nothing is live (ds64_g4.hip.h says why not earlier).  Pooled heads
  // only run on first tiles (no incoming cache), so the running sums the kernel wrote are simply written again.
  if (amax_inputs_bad(amax_cells)) {                         // (workgroup-uniform, scalar)
    __syncthreads();
    nf_repair_call(A, blockIdx.x);
  }
}

// Usable when (the host checks the model side: hidden_dim 64, kernel size 5, dilations 1 / 2 / 4 / 8, pads 4 d): no incoming
// cache, features of <= 96 dims in whole aligned 8-float items, a per-frame linear head with one or two outputs or a pooled
// (Global / Last) head whose hidden layer fits the LDS left over.
// Returns -4 otherwise (the caller then runs mdtc64_w16).
int launch_mdtc64_g4(int nt, bool split, const StackParams& P, const CallArgs& A, hipStream_t stream);
// ... and hidden_dim 32 (mdtc_small.yaml): features of <= 64 dims
int launch_mdtc32_g4(int nt, bool split, const StackParams& P, const CallArgs& A, hipStream_t stream);

}  // namespace wekws
