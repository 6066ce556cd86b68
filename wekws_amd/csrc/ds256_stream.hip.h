// DS-TCN, hidden_dim 256, STREAMING step (a chunk of T <= 16 frames with the carried cache; 10 frames in
// wekws/bin/stream_kws_ctc.py:486-487, any batch_size in runtime/core/bin/kws_main.cc): the 16-wave kernel of
// ds256_w16.hip.h at NT = 1 with the stream's whole cache resident in LDS.
//
// Why a separate kernel: a chunk is almost all left context -- receptive field 105 frames, 10 new ones -- and the
// reference's cache layout (B, C, 105) interleaves the four blocks' slices inside each channel row (tcn.py:165).  Read
// and written block by block (what the batch kernel does) that is 4 x 256 runs of 28..224 bytes in and out per stream:
// partial sectors, every row fetched four times, and sixteen exposed global latencies per chunk.  Measured on 4096
// streams the cache traffic was 55 % of the step (0.78 ms vs 0.36 ms without it), at 2 TB/s of useful bytes.
// At NT = 1 the activations need 16 KB of LDS, so the cache (256 x 105 x 4 = 107,520 B) fits beside them:
//   load   the stream's cache with ONE coalesced 16-byte-per-lane pass (contiguous 107 KB) while x is staged,
//   use    it from LDS: a tap of frame tau at lag s is hbuf[c][tau - s] or cache[c][off + pad + tau - s] -- one LDS read
//          through a selected address,
//   update each slice in place after its block's depthwise conv has read it (shift left by T, append the chunk),
//   store  the new cache with one coalesced pass at the end.
// Arithmetic and operation order are those of the batch kernel: bit-identical results (tests).
#pragma once
#include "ds256_w16.hip.h"

namespace wekws {

template <bool SPLIT>
__global__ __launch_bounds__(kW16Threads) void ds256_stream_kernel(const StackParams P, const CallArgs A) {
  using G = W16Geom<1>;
  constexpr int C = G::C, SS = G::SS, TT = G::TT, PB = G::PB, KS = 8, NT = 1;
  extern __shared__ __attribute__((aligned(16))) float strm_lds[];
  constexpr int kSlab = 8 * 2 * PB;                          // operand planes of ALL 256 channels: [k step][hi | lo]
  char* const slab = reinterpret_cast<char*>(strm_lds);
  float* const hbuf = strm_lds + kSlab / 4;                  // [256][16] f32 activations of the chunk
  float* const cch = hbuf + G::H_FLOATS;                     // [256][Pc]  the stream's cache, reference layout

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  const int T = A.T;                                         // <= 16
  const int b = blockIdx.x;                                  // one stream per workgroup
  const float* __restrict__ W = P.w;
  const int Pc = P.cache_len;
  const int pg = w16_row(tid >> 4), tl = tid & 15;           // (row permutation: ds256_w16.hip.h)
  const int o0 = wave * 16 + lq * 4;
  const int frag_off = (lq * TT + l15) * 16;
  const int n4 = (C * Pc) >> 2;                              // float4 items of one stream's cache (C * Pc % 4 == 0)

  // ---- the cache comes in with one coalesced pass (or as zeros: kws_model.py:67-69, empty cache == zero padding).
  //      The loads are issued here and committed to LDS after the preprocessing GEMM, so their latency is covered.
  constexpr int kCV = 7;                                     // float4 items per thread: 256 * 105 / 4 / 1024 -> 7
  const bool has_cache = A.in_cache != nullptr;              // unconditional clamped loads: the array stays in registers
  const f32x4* const csrc = has_cache ? reinterpret_cast<const f32x4*>(A.in_cache + int64_t(b) * C * Pc)
                                      : reinterpret_cast<const f32x4*>(W);
  f32x4 cv[kCV];
#pragma unroll
  for (int k = 0; k < kCV; ++k)                              // streamed once: keep it out of the weights' way in L2
    cv[k] = __builtin_nontemporal_load(csrc + min(tid + k * kW16Threads, has_cache ? n4 - 1 : 0));

  f32x4 acc[1][NT];

  // ---- block floating point (conv_stack_f16.hip.h): maximum of the chunk's features now, of the carried cache when
  //      its registers are committed to LDS below
  __shared__ AmaxCell amax_cells[kAmaxCells];
  __shared__ BlockDesc blk[kAmaxMaxBlocks];
  amax_zero<kW16Threads>(amax_cells, kAmaxCells);
  stage_block_table<kW16Threads>(blk, P.blocks, P.nblocks);
  __syncthreads();
  const int nk = P.kpre16 / 32;
  const bool one_trip = nk <= 2;                             // 40-d fbank: the features pass through registers once
  W16XItem xi;
  if (one_trip) {
    xi = w16_load_x<TT, PB>(A.x + int64_t(b) * A.xs_b, T, P.idim, nk);
    amax_publish(amax_cells, w16_x_amax(xi));
  } else {
    amax_publish(amax_cells, amax_span<kW16Threads>(A.x + int64_t(b) * A.xs_b, T * P.idim, 0.f));
  }

  // ============================ preprocessing: h0 = [ReLU](x Wpre^T + b) ============================
  {
    zero_acc(acc);
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
      w16_store_x<PB, SPLIT>(xi, sx, slab);
      __syncthreads();
#pragma unroll
      for (int st = 0; st < 2; ++st)                         // (compile-time indices: a runtime-indexed fragment array spills)
        if (st < nk)
          mfma16_step_nb<NT, SPLIT>(acc[0], a[st], slab + st * 2 * PB + frag_off, slab + st * 2 * PB + PB + frag_off);
    } else
    for (int k0 = 0; k0 < nk; k0 += 2) {
      const int steps = min(2, nk - k0);
      __syncthreads();
      sx = pow2_scale(amax_read(amax_cells), &cpre);
      for (int e = tid; e < steps * 4 * TT; e += kW16Threads) {
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
        char* dst = slab + st * 2 * PB + (oct * TT + t) * 16;
        *reinterpret_cast<f16x8*>(dst) = vh;
        if constexpr (SPLIT) *reinterpret_cast<f16x8*>(dst + PB) = vl;
      }
      __syncthreads();
      for (int st = 0; st < steps; ++st) {
        F16Frag a[1];
        load_a16<1>(a, ap + (k0 + st) * 128, 0);
        mfma16_step_nb<NT, SPLIT>(acc[0], a[0], slab + st * 2 * PB + frag_off, slab + st * 2 * PB + PB + frag_off);
      }
    }
    cpre *= P.pre_inv_s;
    float hmax = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = fmaf(acc[0][0][r], cpre, f4c(bias, r));
      if (P.pre_relu) v = fmaxf(v, 0.f);
      hbuf[(o0 + r) * SS + l15] = v;
      hmax = fmaxf(hmax, fabsf(v));
    }
    amax_publish(amax_cells + 2, hmax);
    float cm = 0.f;
#pragma unroll
    for (int k = 0; k < kCV; ++k) {
      const int e = tid + k * kW16Threads;
      const f32x4 q = has_cache ? cv[k] : f32x4{0.f, 0.f, 0.f, 0.f};
      if (e < n4) reinterpret_cast<f32x4*>(cch)[e] = q;
#pragma unroll
      for (int r = 0; r < 4; ++r) cm = fmaxf(cm, fabsf(q[r]));   // (clamped duplicates past n4 repeat the last item)
    }
    amax_publish(amax_cells + 1, cm);
    __syncthreads();                                         // activations, cache image and both maxima complete
  }

  // ======================================= residual blocks =======================================
  // One 16-frame tile: the operand planes of all 256 channels fit the slab (16 KB), so a block is
  //   [every lane-group produces 4 channel rows] barrier [8 K steps x 3 products] [epilogue] barrier
  // -- two barriers per block instead of the batch kernel's nine; at ten frames of work per step the barriers and the
  // latencies behind them are the step time.
  constexpr int OTS = (C / 32) * 128;
  auto ldfrag = [](F16Frag& f, const uint4* __restrict__ q) __attribute__((always_inline)) {
    f.h = __builtin_bit_cast(f16x8, q[0]);
    f.l = __builtin_bit_cast(f16x8, q[64]);
  };
  for (int bi = 0; bi < P.nblocks; ++bi) {
    const BlockDesc bd = blk[bi];
    const int d = bd.dil, pad = bd.pad;
    const uint4* ap1 = reinterpret_cast<const uint4*>(W + bd.a1_16) + size_t(wave) * OTS + lane;
    const float4 ebias = *reinterpret_cast<const float4*>(W + bd.b1 + o0);
    F16Frag af[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) ldfrag(af[s4], ap1 + s4 * 128);   // K steps 0..3, in flight over the producer
    // operand scale of this block (same rule and same numbers as the batch kernel: bit-identical results)
    float c1;
    const float sa = pow2_scale(fmaf(bd.dw_alpha, fmaxf(amax_read(amax_cells + 2 + bi), amax_read(amax_cells + 1)), bd.dw_beta), &c1);
    c1 *= bd.inv_s1;

    // ---- producer: lane-group pg makes channels pg, pg + 64, pg + 128, pg + 192; lane tl = frame tau of the chunk
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = pg + 64 * i;
      float dww[KS + 1];
      {
        const float4* src = reinterpret_cast<const float4*>(W + bd.dw_pk + c * 12);
        const float4 q0 = src[0], q1 = src[1], q2 = src[2];
        dww[0] = q0.x; dww[1] = q0.y; dww[2] = q0.z; dww[3] = q0.w;
        dww[4] = q1.x; dww[5] = q1.y; dww[6] = q1.z; dww[7] = q1.w;
        dww[8] = q2.x;
      }
      const float* const hrow = hbuf + c * SS;               // frames 0..15 of the chunk
      float* const crow = cch + c * Pc + bd.cache_off;       // this block's slice: frames -pad..-1
      // the padded sequence [slice | chunk] at chunk-relative frame ix, one LDS read through a selected address
      auto at = [&](int ix) __attribute__((always_inline)) -> float { return *(ix >= 0 ? hrow + ix : crow + pad + ix); };
      float o = dww[KS];
#pragma unroll
      for (int j = 0; j < KS; ++j) o = fmaf(dww[j], at(tl - (KS - 1 - j) * d), o);
      // new slice = last pad frames of [slice | chunk] (tcn.py:52), in place: every lane reads before any lane of the
      // group writes (same wave, LDS in order)
      float nv[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) nv[k] = at(min(tl + 16 * k, pad - 1) + T - pad);
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (tl + 16 * k < pad) crow[tl + 16 * k] = nv[k];
      o = fmaxf(o, 0.f);
      _Float16 h, l;
      split16s(o, sa, h, l);
      char* const plane = slab + (c >> 5) * 2 * PB;          // K step c / 32
      _Float16* ph = reinterpret_cast<_Float16*>(plane) + (((c & 31) >> 3) * TT) * 8 + (c & 7);
      _Float16* pl = reinterpret_cast<_Float16*>(plane + PB) + (((c & 31) >> 3) * TT) * 8 + (c & 7);
      ph[tl * 8] = h;
      if constexpr (SPLIT) pl[tl * 8] = l;
    }
    zero_acc(acc);
    __syncthreads();
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      mfma16_step_nb<NT, SPLIT>(acc[0], af[s4], slab + s4 * 2 * PB + frag_off, slab + s4 * 2 * PB + PB + frag_off);
      ldfrag(af[s4], ap1 + (4 + s4) * 128);         // K steps 4..7 follow through the same registers
    }
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
      mfma16_step_nb<NT, SPLIT>(acc[0], af[s4], slab + (4 + s4) * 2 * PB + frag_off,
                                slab + (4 + s4) * 2 * PB + PB + frag_off);
    // epilogue: folded bias + ReLU + residual, in place (tcn.py:60).  Every producer read of hbuf is behind the barrier.
    float hmax = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float* hp = hbuf + (o0 + r) * SS + l15;
      const float v = fmaxf(fmaf(acc[0][0][r], c1, f4c(ebias, r)), 0.f) + *hp;
      *hp = v;
      hmax = fmaxf(hmax, fabsf(v));
    }
    amax_publish(amax_cells + 3 + bi, hmax);
    __syncthreads();
  }

  // ---- the new cache leaves with one coalesced pass; then the classifier
  if (A.out_cache) {
    const f32x4* src = reinterpret_cast<const f32x4*>(cch);
    f32x4* dst = reinterpret_cast<f32x4*>(A.out_cache + int64_t(b) * C * Pc);
    for (int e = tid; e < n4; e += kW16Threads) __builtin_nontemporal_store(src[e], dst + e);
  }
  conv_stack_head<KIND_DS, 256, 1, kW16Threads, SS>(P, A, hbuf, reinterpret_cast<float*>(slab), b);
}

inline size_t ds256_stream_lds_bytes(int cache_len) {       // slab 16 KB + chunk 16 KB + cache
  return size_t(8 * 2 * W16Geom<1>::PB) + size_t(W16Geom<1>::H_FLOATS) * 4 + size_t(256) * cache_len * 4;
}

template <bool SPLIT>
inline int launch_ds256_stream_s(const StackParams& P, const CallArgs& A, hipStream_t stream) {
  static DynLdsGrant grant;
  const size_t lds = ds256_stream_lds_bytes(P.cache_len);
  auto kern = ds256_stream_kernel<SPLIT>;
  if (grant_dynamic_lds(kern, int(lds), grant)) return -3;
  hipLaunchKernelGGL(kern, dim3(A.B), dim3(kW16Threads), lds, stream, P, A);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// usable when: kernel size 8, A.T <= 16, the stream's cache fits LDS beside the chunk (host checks)
int launch_ds256_stream(bool split, const StackParams& P, const CallArgs& A, hipStream_t stream);

}  // namespace wekws
