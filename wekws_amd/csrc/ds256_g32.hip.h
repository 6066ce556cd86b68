// DS-TCN, hidden_dim 256, precision F32 (every product rounded once, like the reference's fp32 arithmetic), keyword
// configuration, no incoming cache: the REGISTER-RESIDENT kernel of ds256_g16.hip.h on v_mfma_f32_16x16x4_f32.
//
// The generic exact-f32 kernel (conv_stack_kernel<KIND_DS, 256, 7, 8>: 8 waves, f32 tile in LDS, 32-channel operand slab,
// one barrier per chunk) spends 54 % of its time in the matrix pipe; the f32-input MFMA issues at the vector-f32 rate
// (32 cycles per 16x16x4), so 56.4 GFLOP per 1024 utterances cannot take less than 0.36 ms.  Everything that made
// ds256_g16 faster than ds256_w16 is independent of the operand type and carries over unchanged:
//   * the residual tile lives in registers in the accumulator layout (wave = 16 channels x all frames, lane = 4 channels x
//     NT consecutive frames: column 16 tt + l holds frame NT l + tt), the depthwise taps are DPP row shifts, the epilogue
//     and the keyword head work from the registers, cache slices leave as dwordx4 + dwordx3 per lane;
//   * the depthwise output of ALL 256 channels goes to LDS as operand planes, two barriers per block;
//   * persistent workgroups, features and taps copied into LDS by global_load_lds.
// What differs: no operand scaling at all (f32 has the range), so no maxima, no split, one barrier less in the
// preprocessing; B planes hold f32 items [16-channel group g][lq][column][4 steps] -- lane (lq, column) reads the B values
// of a group's four MFMAs with one ds_read_b128, channel order inside a group = the host image's (put_packed_a: the A
// fragment of step s, lane l is W[16 ot + (l & 15)][16 g + 4 s + (l >> 4)]).
#pragma once
#include "ds256_g16.hip.h"

namespace wekws {

// One 16-deep K group for one o-tile: four MFMAs per frame tile, the next tile's B item requested before them
template <int NT>
__device__ __forceinline__ void g32_mfma_group(f32x4 (&acc)[NT], const float4 a, const char* bsrc) {
  float4 bv[2];
  bv[0] = *reinterpret_cast<const float4*>(bsrc);
#pragma unroll
  for (int tt = 0; tt < NT; ++tt) {
    if (tt + 1 < NT) bv[(tt + 1) & 1] = *reinterpret_cast<const float4*>(bsrc + (tt + 1) * 256);
    __builtin_amdgcn_sched_barrier(0);
    const float4 q = bv[tt & 1];
    acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, q.x, acc[tt], 0, 0, 0);
    acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, q.y, acc[tt], 0, 0, 0);
    acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, q.z, acc[tt], 0, 0, 0);
    acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, q.w, acc[tt], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Depthwise conv + folded BN + ReLU of the channel-row pair (2 P_, 2 P_ + 1) of the lane's four, all NT frames of the lane
// (g16_dw_pair without the split): row r of column 16 tt + l15 is one float of item (group = wave, lq = r, column),
// step slot = the lane's own lq
template <int D, int P_, int NT>
__device__ __forceinline__ void g32_dw_pair(const f32x4 (&hv)[NT], const float* taps_o0, char* pst) {
  constexpr int TT = 16 * NT;
  const float4* src = reinterpret_cast<const float4*>(taps_o0 + 2 * P_ * 12);
  const float4 a0 = src[0], a1 = src[1], a2 = src[2], b0 = src[3], b1 = src[4], b2 = src[5];
  constexpr auto tiles = std::make_integer_sequence<int, NT>{};
  constexpr int RA = 2 * P_, RB = 2 * P_ + 1;
  float oa[NT], ob[NT];
  __builtin_amdgcn_s_setprio(P_ == 0 ? 3 : 1);               // (a wave that is ahead steps back: ds256_g16.hip.h)
#pragma unroll
  for (int tt = 0; tt < NT; ++tt) { oa[tt] = a2.x; ob[tt] = b2.x; }
  g16_tap_tiles<7 * D, NT, RA>(oa, hv, a0.x, tiles); g16_tap_tiles<7 * D, NT, RB>(ob, hv, b0.x, tiles);
  g16_tap_tiles<6 * D, NT, RA>(oa, hv, a0.y, tiles); g16_tap_tiles<6 * D, NT, RB>(ob, hv, b0.y, tiles);
  g16_tap_tiles<5 * D, NT, RA>(oa, hv, a0.z, tiles); g16_tap_tiles<5 * D, NT, RB>(ob, hv, b0.z, tiles);
  g16_tap_tiles<4 * D, NT, RA>(oa, hv, a0.w, tiles); g16_tap_tiles<4 * D, NT, RB>(ob, hv, b0.w, tiles);
  g16_tap_tiles<3 * D, NT, RA>(oa, hv, a1.x, tiles); g16_tap_tiles<3 * D, NT, RB>(ob, hv, b1.x, tiles);
  g16_tap_tiles<2 * D, NT, RA>(oa, hv, a1.y, tiles); g16_tap_tiles<2 * D, NT, RB>(ob, hv, b1.y, tiles);
  g16_tap_tiles<1 * D, NT, RA>(oa, hv, a1.z, tiles); g16_tap_tiles<1 * D, NT, RB>(ob, hv, b1.z, tiles);
  g16_tap_tiles<0, NT, RA>(oa, hv, a1.w, tiles);     g16_tap_tiles<0, NT, RB>(ob, hv, b1.w, tiles);
  __builtin_amdgcn_s_setprio(P_ == 0 ? 2 : 0);
#pragma unroll
  for (int tt = 0; tt < NT; ++tt) {
    *reinterpret_cast<float*>(pst + (RA * TT + tt * 16) * 16) = fmaxf(oa[tt], 0.f);
    *reinterpret_cast<float*>(pst + (RB * TT + tt * 16) * 16) = fmaxf(ob[tt], 0.f);
  }
}
template <int D, int NT>
__device__ __forceinline__ void g32_dw_rows(const f32x4 (&hv)[NT], const float* taps_o0, char* pst) {
  g32_dw_pair<D, 0, NT>(hv, taps_o0, pst);
  g32_dw_pair<D, 1, NT>(hv, taps_o0, pst);
}

// Launched only for the keyword configuration (the FAST conditions of ds256_g16.hip.h: features of <= 64 dims in whole
// aligned 8-float items, per-frame linear head with one or two outputs) without an incoming cache and with dilations
// 1 / 2 / 4 / 8; everything else in precision F32: conv_stack_kernel.
template <int NT>
__global__ __launch_bounds__(kW16Threads) void ds256_g32_kernel(const StackParams P, const CallArgs A) {
  using G = W16Geom<NT>;
  constexpr int C = G::C, TT = G::TT, PB = G::PB;           // PB = 64 TT bytes: one 16-channel group of f32 items, too
  constexpr int NG = C / 16;                                 // K groups per layer
  static_assert(size_t(NG) * PB + size_t(TT) * 64 * 4 <= G::LDS_BYTES, "planes of a whole layer + feature buffer");
  extern __shared__ __attribute__((aligned(16))) float w16_lds[];
  char* const planes = reinterpret_cast<char*>(w16_lds);     // [K group][lq][column][4 steps] f32
  float* const xbuf = w16_lds + (NG * PB) / 4;               // the utterance's features, frame-major as in memory

  f32x4 acc[NT];
  f32x4 hv[NT];                                              // the residual tile: channels o0 .. o0 + 3, frames NT l15 + tt
  G16_PH_DECL;

  __shared__ BlockDesc blk[kAmaxMaxBlocks];
  __shared__ __attribute__((aligned(16))) float taps[C * 12];   // taps + bias of the current block ([256][12] records)
  __shared__ NfList nfl;                                     // utterances with a NaN / Inf feature (nonfinite.hip.h)
  nf_list_init(nfl);
  {
    const int ntbl = P.nblocks * int(sizeof(BlockDesc) / 4);
    const int t0 = threadIdx.x;
    if (t0 < ntbl) reinterpret_cast<uint32_t*>(blk)[t0] = reinterpret_cast<const uint32_t*>(P.blocks)[t0];
  }
  auto prefetch_x = [&](int bn) __attribute__((always_inline)) {
    const int nitems = (A.T * P.idim) >> 2;                  // 16-byte pieces (idim % 8 == 0)
    const float* src = A.x + int64_t(bn) * A.xs_b;
    const int t0 = threadIdx.x;
    for (int e0 = 0; e0 < nitems; e0 += kW16Threads) {       // (wave-uniform trip count; a wave's pieces are consecutive)
      const int wbase = __builtin_amdgcn_readfirstlane(e0 + (t0 & ~63));
      if (e0 + t0 < nitems)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (e0 + t0) * 4),
                                         (__attribute__((address_space(3))) void*)(xbuf + wbase * 4), 16, 0, 0);
    }
  };
  prefetch_x(blockIdx.x);
  for (int b = blockIdx.x; b < A.B; b += gridDim.x) {          // persistent workgroups (ds256_g16.hip.h)
  const float* Wp = P.w;
  asm volatile("" : "+s"(Wp));                               // (opaque per utterance: nothing is hoisted out of the loop)
  const float* __restrict__ W = Wp;
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  int T = A.T;
  asm volatile("" : "+s"(T));
  const int Pc = P.cache_len;
  const int o0 = wave * 16 + lq * 4;                         // this lane's 4 channels: rows of the o-tile AND of the tile h
  const int frag_off = (lq * TT + l15) * 16;
  char* const pst = planes + wave * PB + l15 * 16 + lq * 4;  // + (r TT + 16 tt) 16: row r, tile tt

  auto stage_taps = [&](const BlockDesc& nb) __attribute__((always_inline)) {
    if (wave < C * 3 / 64)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(W + nb.dw_pk + tid * 4),
                                       (__attribute__((address_space(3))) void*)(taps + wave * 256), 16, 0, 0);
  };
  __builtin_amdgcn_s_waitcnt(0x0F70);                        // vmcnt(0): this wave's pieces of the feature copy have landed
  __syncthreads();                                           // table staged; the utterance before is done with LDS
  stage_taps(blk[0]);

  // ============================ preprocessing: h0 = [ReLU](x Wpre^T + b) ============================
  {
    const int ng = P.kpre / 16;
    // B items of the features: item (group g, lq, column n) = x[frame(n)][16 g + 4 s + lq], s = 0..3
    unsigned xbits = 0;                                      // max of the |x| bit patterns: NaN / Inf end up on top
    for (int e = tid; e < ng * 4 * TT; e += kW16Threads) {
      const int n = e % TT, q = e / TT;
      const int f = NT * (n & 15) + (n >> 4);
      const int k0 = (q >> 2) * 16 + (q & 3);
      const float* xr = xbuf + f * P.idim + k0;
      float4 v;
      v.x = (f < T && k0 < P.idim) ? xr[0] : 0.f;
      v.y = (f < T && k0 + 4 < P.idim) ? xr[4] : 0.f;
      v.z = (f < T && k0 + 8 < P.idim) ? xr[8] : 0.f;
      v.w = (f < T && k0 + 12 < P.idim) ? xr[12] : 0.f;
      xbits = max(max(xbits, nf_abs_bits(v.x)), max(max(nf_abs_bits(v.y), nf_abs_bits(v.z)), nf_abs_bits(v.w)));
      *reinterpret_cast<float4*>(planes + (q >> 2) * PB + ((q & 3) * TT + n) * 16) = v;
    }
    if (nf_bad_bits(xbits)) nfl.flag = 1;
    const float4* ap = reinterpret_cast<const float4*>(W + P.pre_a) + size_t(wave) * ng * 64 + lane;
    const float4 bias = *reinterpret_cast<const float4*>(W + P.pre_b + o0);
    float4 an = ap[0];
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) acc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    if (__builtin_amdgcn_readfirstlane(nfl.flag)) {          // a NaN / Inf feature: noted, re-computed behind the loop
      nf_list_note(nfl, b);
      if (b + int(gridDim.x) < A.B) prefetch_x(b + gridDim.x);
      continue;
    }
    for (int g = 0; g < ng; ++g) {
      const float4 a = an;
      an = ap[min(g + 1, ng - 1) * 64];
      g32_mfma_group<NT>(acc, a, planes + g * PB + frag_off);
    }
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[tt][r] + f4c(bias, r);
        if (P.pre_relu) v = fmaxf(v, 0.f);
        hv[tt][r] = v;
      }
    }
    __syncthreads();                                         // (A) planes free, taps on their way
  }
  G16_PH(0);                                                 // [0] preprocessing
  // ======================================= residual blocks =======================================
  auto frag_base = [&](int i) __attribute__((always_inline)) {
    return reinterpret_cast<const float4*>(W + __builtin_amdgcn_readfirstlane(blk[i].a1)) + size_t(wave) * NG * 64;
  };
  float4 an = frag_base(0)[lane];                            // fragment of K group 0: carried from block to block
  for (int bi = 0; bi < P.nblocks; ++bi) {
    const BlockDesc bd = blk[bi];
    const int pad = bd.pad;
    const float4* ap1 = frag_base(bi);
    const float4* apn = frag_base(min(bi + 1, P.nblocks - 1));
    G16_PH(1);                                               // [1] block top

    // ---- depthwise dilated conv + folded BN + ReLU (tcn.py:102-109), from the registers, to the operand planes
    {
      const float* taps_o0 = taps + o0 * 12;
      switch (bd.dil) {
        case 1: g32_dw_rows<1, NT>(hv, taps_o0, pst); break;
        case 2: g32_dw_rows<2, NT>(hv, taps_o0, pst); break;
        case 4: g32_dw_rows<4, NT>(hv, taps_o0, pst); break;
        case 8: g32_dw_rows<8, NT>(hv, taps_o0, pst); break;
        default: break;
      }
    }
    G16_PH(2);                                               // [2] depthwise conv -> operand planes
    __syncthreads();                                         // (B) the planes of all 256 channels are written
    G16_PH(3);

    const float4 ebias = *reinterpret_cast<const float4*>(W + bd.b1 + o0);
    if (bi + 1 < P.nblocks) stage_taps(blk[bi + 1]);         // (this block's taps were last read before barrier (B))

    // ---- pointwise conv: sixteen K groups back to back, the next group's fragment requested a group ahead (the last
    //      group requests group 0 of the next block)
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) acc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int g = 0; g < NG; ++g) {
      if ((g & 3) == 0) {                                    // (a wave that is ahead steps back: ds256_g16.hip.h)
        if (g == 0) __builtin_amdgcn_s_setprio(3);
        else if (g == 4) __builtin_amdgcn_s_setprio(2);
        else if (g == 8) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
      }
      const float4 a = an;
      an = (g + 1 < NG ? ap1 + (g + 1) * 64 : apn)[lane];
      g32_mfma_group<NT>(acc, a, planes + g * PB + frag_off);
    }
    G16_PH(4);                                               // [4] matrix phase

    // ---- the block's streaming-cache slice: the last `pad` frames of its INPUT tile, from the registers (ds256_g16.hip.h)
    if (A.out_cache) {
      float* const oc = A.out_cache + (int64_t(b) * C + o0) * Pc + bd.cache_off;
      const int p0 = NT * l15 - (T - pad);                   // slice column of this lane's first frame
      if (p0 >= 0 && p0 + NT <= pad) {                       // the lane's NT frames are NT consecutive columns of the slice
#pragma unroll
        for (int r = 0; r < 4; ++r) g16_store_run<NT>(oc + r * Pc + p0, hv, r);
      } else if (p0 + NT > 0 && p0 < pad) {                  // (NT does not divide T: slice boundary inside the lane)
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
          const int p = p0 + tt;
          if (p >= 0 && p < pad) {
#pragma unroll
            for (int r = 0; r < 4; ++r) oc[r * Pc + p] = hv[tt][r];
          }
        }
      }
      if (T < pad) {                                         // shorter than the slice: zero context in front
        const int nz = pad - T;
        for (int e = lane; e < 16 * nz; e += 64) {
          const int cc = e / nz, p = e - cc * nz;
          A.out_cache[(int64_t(b) * C + wave * 16 + cc) * Pc + bd.cache_off + p] = 0.f;
        }
      }
    }
    G16_PH(6);                                               // [6] cache hand-over

    // ---- epilogue: folded bias + ReLU + residual (tcn.py:60: add after the ReLU), registers only
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) hv[tt][r] = fmaxf(acc[tt][r] + f4c(ebias, r), 0.f) + hv[tt][r];
    }
    G16_PH(5);                                               // [5] epilogue
    __syncthreads();                                         // (A) planes free, taps on their way
    G16_PH(3);
  }

  // ---- keyword head from the registers (ds256_g16.hip.h): 64 partial sums per output meet in LDS where the planes were
  {
    const int K = P.odim;
    constexpr int PS = 32 * NT + 16;
    float* const part = w16_lds;
    {
      const float4 w0 = *reinterpret_cast<const float4*>(W + P.head_w + o0);
      const float4 w1 = *reinterpret_cast<const float4*>(W + P.head_w + (K > 1 ? C : 0) + o0);
      if (b + int(gridDim.x) < A.B) prefetch_x(b + gridDim.x);   // (behind the classifier rows: loads return in order)
      float* dst = part + (wave * 4 + lq) * PS + 2 * NT * l15;
      const HeadPairs hw(w0, w1);                            // (packed, operand selects spelled out: pk_safe.hip.h)
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        pk_f32x2 pp{0.f, 0.f};
        head_fma4(pp, hw.x, hw.y, hw.z, hw.w, hv[tt]);
        *reinterpret_cast<float2*>(dst + 2 * tt) = float2{pp.x, pp.y};
      }
    }
    __syncthreads();
    {
      const int e = tid >> 2, qd = tid & 3;                  // e = 2 frame + output; four lanes per e
      const int t = e >> 1, k = e & 1;
      const float* src = part + qd * 16 * PS + e;
      float v = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) v += src[i * PS];
      v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
      v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));
      if (qd == 0 && t < T && k < K) {
        v += W[P.head_b + k];
        if (P.sigmoid) v = sigmoidf_(v);
        A.y[int64_t(b) * A.ys_b + int64_t(t) * K + k] = v;
      }
    }
  }
  G16_PH(7);                                                 // [7] classifier
  }                                                          // next utterance of this workgroup
  nf_list_drain(nfl, A, P.idim, 0, blockIdx.x, gridDim.x);
  G16_PH_DUMP;
}

// nt: frame tiles (1 / 2 / 4 / 7); cus: compute units = the largest grid.  Returns -4 when the call is not one this kernel
// takes (the caller then runs conv_stack_kernel).
int launch_ds256_g32(int nt, const StackParams& P, const CallArgs& A, hipStream_t stream, int cus);

}  // namespace wekws
