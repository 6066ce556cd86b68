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
//
// Shape of a block (round 2; per-phase clock64 stamps, build/probe): with ONE output per lane and channel (lane = frame)
// the depthwise phase was 7.3 k of a block's 10.3 k cycles -- ~300 vector instructions per wave of which 32 are the
// taps' FMAs; the rest was per-output tap addressing, cache / tile selects and the slice shift, times four waves per
// SIMD.  Now lane = (channel, quarter): FOUR outputs per lane as a run at stride = dilation (d = 8: two runs of two), the
// eight taps sliding over 11 register-resident inputs, dilation a compile-time constant per switch arm (offsets are
// instruction immediates, and for d >= 4 every left-context input is known to come from the cache: no selects).  The
// taps of the next block are requested a block ahead.  The block's 262 KB of weights stream through the CU's 64 B/clk
// path (4.1 k cycles, the floor of a block): K steps 0..3 behind the depthwise phase, 4..7 behind the matrix phase, through
// the same 32 registers (all eight up front do not fit the 128-register budget of four waves per SIMD).  Tile rows are skewed (row c starts at 20 c + c % 4) so that the 8
// channels x 4 quarters of a 32-lane half read 32 different LDS banks at d = 1.
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
  const int pc = tid >> 2, g = tid & 3;                      // producer: channel, quarter
  auto hoff = [](int c) __attribute__((always_inline)) { return c * SS + (c & 3); };   // skewed tile rows (header)
  const int o0 = wave * 16 + lq * 4;
  const int frag_off = (lq * TT + l15) * 16;
  const int n4 = (C * Pc) >> 2;                              // float4 items of one stream's cache (C * Pc % 4 == 0)

  f32x4 acc[1][NT];

  // ---- block floating point (conv_stack_f16.hip.h): maximum of the chunk's features now, of the carried cache when
  //      its registers are committed to LDS below
  __shared__ AmaxCell amax_cells[kAmaxCells];
  __shared__ BlockDesc blk[kAmaxMaxBlocks];
  // Every load of the prologue is requested before the first barrier -- the block table, the features, the preprocessing
  // fragments, the classifier, and LAST the cache -- so that the step starts with one trip to memory, not one per dependency.
  amax_zero<kW16Threads>(amax_cells, kAmaxCells);
  const int tbl_n = P.nblocks * int(sizeof(BlockDesc) / 4);  // <= 24 x 21 dwords: one per thread
  uint32_t tbl_v = 0;
  if (tid < tbl_n) tbl_v = reinterpret_cast<const uint32_t*>(P.blocks)[tid];
  const int nk = P.kpre16 / 32;
  const bool one_trip = nk <= 2 && w16_x_vec_ok(A.x, A.xs_b, P.idim);   // 40-d fbank: the features pass through registers once
  W16XItem xi;
  if (one_trip) xi = w16_load_x<TT, PB>(A.x + int64_t(b) * A.xs_b, T, P.idim, nk);
  const uint4* const ap_pre = reinterpret_cast<const uint4*>(W + P.pre_a16) + size_t(wave) * nk * 128 + lane;
  const float4 bias_pre = *reinterpret_cast<const float4*>(W + P.pre_b + o0);
  F16Frag a_pre[2];
  if (one_trip) {
#pragma unroll
    for (int st = 0; st < 2; ++st) {                         // (nk = 1: the same step twice)
      const uint4* q = ap_pre + min(st, nk - 1) * 128;
      a_pre[st].h = __builtin_bit_cast(f16x8, q[0]);
      a_pre[st].l = __builtin_bit_cast(f16x8, q[64]);
    }
  }
  const HeadPre head_pre = conv_stack_head_prefetch<KIND_DS, 256, 1, kW16Threads>(P);
  // ---- the cache comes in with one coalesced pass (or as zeros: kws_model.py:67-69, empty cache == zero padding).
  //      The loads are issued here -- BEHIND the table, the features and the preprocessing weights, because loads return in
  //      order and the preprocessing must not wait for this trip to HBM -- and committed to LDS after the preprocessing GEMM.
  constexpr int kCV = 7;                                     // float4 items per thread: 256 * 105 / 4 / 1024 -> 7
  const bool has_cache = A.in_cache != nullptr;              // unconditional clamped loads: the array stays in registers
  const f32x4* const csrc = has_cache ? reinterpret_cast<const f32x4*>(A.in_cache + int64_t(b) * C * Pc)
                                      : reinterpret_cast<const f32x4*>(W);
  f32x4 cv[kCV];
#pragma unroll
  for (int k = 0; k < kCV; ++k)                              // streamed once: keep it out of the weights' way in L2
    cv[k] = __builtin_nontemporal_load(csrc + min(tid + k * kW16Threads, has_cache ? n4 - 1 : 0));

  if (tid < tbl_n) reinterpret_cast<uint32_t*>(blk)[tid] = tbl_v;
  __syncthreads();
  if (one_trip) amax_publish(amax_cells, w16_x_amax_bits(xi));
  else amax_publish(amax_cells, amax_span_bits<kW16Threads>(A.x + int64_t(b) * A.xs_b, T * P.idim, 0.f));

  // ============================ preprocessing: h0 = [ReLU](x Wpre^T + b) ============================
  {
    zero_acc(acc);
    const uint4* ap = ap_pre;
    const float4 bias = bias_pre;
    float sx = 1.f, cpre = 1.f;
    if (one_trip) {
      const F16Frag (&a)[2] = a_pre;
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
      hbuf[hoff(o0 + r) + l15] = v;
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
      for (int r = 0; r < 4; ++r) cm = amax_acc(cm, q[r]);       // (clamped duplicates past n4 repeat the last item; bit-pattern
    }                                                            //  maximum: a NaN / Inf stays on top, see amax_acc)
    amax_publish(amax_cells + 1, cm);
    __syncthreads();                                         // activations, cache image and both maxima complete
  }
  if (amax_inputs_bad(amax_cells)) {                         // a NaN / Inf in the chunk's features or the carried cache: the
    nf_repair_call(A, b);                                    // reference's arithmetic for this stream (nonfinite.hip.h)
    return;
  }

  // ======================================= residual blocks =======================================
  // One 16-frame tile: the operand planes of all 256 channels fit the slab (16 KB), so a block is
  //   [every lane makes 4 outputs of its channel] barrier [8 K steps x 3 products] [epilogue] barrier
  // -- two barriers per block instead of the batch kernel's nine.
  constexpr int OTS = (C / 32) * 128;
  auto ldfrag = [](F16Frag& f, const uint4* __restrict__ q) __attribute__((always_inline)) {
    f.h = __builtin_bit_cast(f16x8, q[0]);
    f.l = __builtin_bit_cast(f16x8, q[64]);
  };
  struct TapRec { float4 q0, q1, q2; };                      // taps + bias of channel pc (12-float record)
  auto load_taps = [&](int bi) __attribute__((always_inline)) {
    const float4* src = reinterpret_cast<const float4*>(W + blk[bi].dw_pk + pc * 12);
    return TapRec{src[0], src[1], src[2]};
  };
  // Producer body for dilation D (compile time).  Lane (pc, g) makes NR runs of R outputs at stride D (R = min(4, 16 / D)):
  // run i = g NR + r starts at frame f0 = (i / D) R D + i % D and reads the R + 7 frames f0 + (q - 7) D -- from the tile
  // where that is >= 0, else from the block's cache slice (pad = 7 D frames, tcn.py:41-52).  Then the slice is shifted in
  // place: new slice = last pad frames of [slice | chunk] (tcn.py:52); a row belongs to one wave and every read precedes
  // the writes (LDS is in order within a wave).
  auto produce = [&](auto d_c, const BlockDesc& bd, const TapRec& tc, float sa) __attribute__((always_inline)) {
    constexpr int D = decltype(d_c)::value;
    constexpr int R = (16 / D) < 4 ? (16 / D) : 4, NR = 4 / R, PAD = 7 * D;
    constexpr int kMaxF0 = D == 1 ? 12 : D == 2 ? 9 : D == 4 ? 3 : 7;   // largest first frame of a run
    const float dww[KS + 1] = {tc.q0.x, tc.q0.y, tc.q0.z, tc.q0.w, tc.q1.x, tc.q1.y, tc.q1.z, tc.q1.w, tc.q2.x};
    const float* hrow = hbuf + hoff(pc);
    float* crow = cch + pc * Pc + bd.cache_off;
    char* const plane = slab + (pc >> 5) * 2 * PB;           // K step pc / 32
    _Float16* ph = reinterpret_cast<_Float16*>(plane) + (((pc & 31) >> 3) * TT) * 8 + (pc & 7);
#pragma unroll
    for (int r = 0; r < NR; ++r) {                           // (run by run: one window of registers at a time)
      const int i = g * NR + r;
      const int f0 = (i / D) * R * D + (i % D);
      const float* tb = hrow + f0 - PAD;                     // tile address of input q = 0 (may lie left of the row)
      const float* cb = crow + f0;                           // cache address of input q = 0
      float win[R + 7];
#pragma unroll
      for (int q = 0; q < R + 7; ++q) {
        if (q >= 7) win[q] = tb[q * D];
        else if (kMaxF0 + (q - 7) * D < 0) win[q] = cb[q * D];             // (compile time: always left of the chunk)
        else win[q] = *((f0 >= (7 - q) * D) ? tb + q * D : cb + q * D);
      }
#pragma unroll
      for (int m = 0; m < R; ++m) {
        float o = dww[KS];
#pragma unroll
        for (int j = 0; j < KS; ++j) o = fmaf(dww[j], win[m + j], o);
        o = fmaxf(o, 0.f);
        const int t = f0 + m * D;
        _Float16 h, l;
        split16s(o, sa, h, l);
        ph[t * 8] = h;
        if constexpr (SPLIT) ph[t * 8 + PB / 2] = l;
      }
    }
    // the slice moves left by T in ascending chunks of four elements per lane: element p is overwritten only after its
    // old value has been read as the source of element p - T (an earlier or the same chunk)
    constexpr int NS = (PAD + 3) / 4;                        // slice elements per lane: p = g + 4 k
#pragma unroll
    for (int k0 = 0; k0 < NS; k0 += 4) {
      float nv[4];
#pragma unroll
      for (int k = k0; k < k0 + 4 && k < NS; ++k) {
        const int sidx = T + g + 4 * k;                      // index into [slice | chunk]
        nv[k - k0] = *(sidx < PAD ? crow + sidx : hrow + (sidx - PAD));   // (p >= PAD, last k only: read, not written)
      }
#pragma unroll
      for (int k = k0; k < k0 + 4 && k < NS; ++k)
        if (PAD % 4 == 0 || k + 1 < NS || g + 4 * k < PAD) crow[g + 4 * k] = nv[k - k0];
    }
  };

  TapRec tnext = {};
  if (P.nblocks > 0) tnext = load_taps(0);
  for (int bi = 0; bi < P.nblocks; ++bi) {
    const BlockDesc& bd = blk[bi];
    const uint4* ap1 = reinterpret_cast<const uint4*>(W + bd.a1_16) + size_t(wave) * OTS + lane;
    const TapRec tc = tnext;
    F16Frag af[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) ldfrag(af[s4], ap1 + s4 * 128);   // K steps 0..3, in flight over the producer
    const float4 ebias = *reinterpret_cast<const float4*>(W + bd.b1 + o0);
    // operand scale of this block (same rule and same numbers as the batch kernel: bit-identical results)
    float c1;
    const float sa = pow2_scale(fmaf(bd.dw_alpha, fmaxf(amax_read(amax_cells + 2 + bi), amax_read(amax_cells + 1)), bd.dw_beta), &c1);
    c1 *= bd.inv_s1;
    switch (bd.dil) {
      case 1: produce(std::integral_constant<int, 1>{}, bd, tc, sa); break;
      case 2: produce(std::integral_constant<int, 2>{}, bd, tc, sa); break;
      case 4: produce(std::integral_constant<int, 4>{}, bd, tc, sa); break;
      default: produce(std::integral_constant<int, 8>{}, bd, tc, sa); break;     // (host: dilations are 1 / 2 / 4 / 8)
    }
    zero_acc(acc);
    tnext = load_taps(min(bi + 1, P.nblocks - 1));           // (a block ahead: the producer cannot start without them)
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
      float* hp = hbuf + hoff(o0 + r) + l15;
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
  conv_stack_head<KIND_DS, 256, 1, kW16Threads, SS, true>(P, A, hbuf, reinterpret_cast<float*>(slab), b, &head_pre);
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
