// Split-precision variant of the fused conv-backbone forward (see conv_stack.hip.h for the structure, the
// reference citations and the LDS geometry, which are shared).
//
// The f32-input matrix instruction of gfx950 issues at the vector-FP32 rate (32 cycles per 16x16x4), 1/16 of the
// fp16 rate, and -- measured, tools/probe/mfma_probe.hip -- a saturated MFMA stream lets only ONE VALU instruction
// of the co-resident wave through per MFMA, so matrix time and vector time add up.  Here every 1x1 / dense
// convolution runs on v_mfma_f32_16x16x32_f16 instead, with fp32-level accuracy recovered by the classic
// two-term split of both operands:
//        w = wh + wl,  a = ah + al      (wh = fp16(w), wl = fp16(w - wh); same for a)
//        w*a ~= wh*ah + wh*al + wl*ah   (the dropped wl*al term is <= 2^-22 relative)
// Products of fp16 values are exact in the fp32 accumulator and the instruction honours fp16 subnormals
// (tools/probe/denorm.hip), so all three terms share ONE accumulator set.  Cost per 32-deep K step: 3 x 17 cycles
// instead of 8 x 32.
//
// BLOCK FLOATING POINT.  hi + lo carries 22 significand bits only while lo stays a NORMAL fp16, i.e. for
// |v| >= 2^-2; below that lo is subnormal (absolute quantum 2^-24), and |v| >= 65520 overflows to inf.  fp16's
// exponent range is therefore managed explicitly, per operand, with exact power-of-two scales:
//   * weights (static): every packed matrix holds W * sw with sw = 2^k chosen on the host so that max|W| * sw lies in
//     [2^14, 2^15) -- the TOP of the fp16 range, 16 binades of full-precision elements below the matrix maximum and an
//     absolute error of 2^-25 (2^-39 of the maximum) below those;
//   * activations (dynamic): every operand tile is multiplied by sa = 2^j before the split, with j from the tile's
//     magnitude: the exact max|.| where it is cheaply known before the tile is written (the features; tiles re-written
//     behind an existing barrier), else a rigorous bound (depthwise output: dw_alpha * max|input| + dw_beta; MDTC mid
//     tile: mid_alpha * that + mid_beta), anchored once per block on the EXACT maximum of the residual tile, which the
//     epilogue that writes the tile tracks -- so bounds compound over at most one block.  bound * sa lies in
//     [2^14, 2^15): no element can overflow; the split's error is max(2^-24 |v|, 2^-25) in scaled units, i.e. below 2^-22
//     of the tile maximum as long as the bound is less than 2^18 too loose (row 1-norms overshoot by 2^3 .. 2^6);
//   * the epilogue multiplies the accumulator by 1 / (sw * sa), exact, as part of the bias FMA.
// Result: the products are scale-invariant -- a model with weights x 2^-12 and activations x 2^12 (or the reverse)
// gives the same numbers as the unscaled one, like fp32 math does (tests/test_hip_parity.py::test_scale_sweep).
// Maxima travel between phases through a few LDS cells (wave reduction + one ds_max_u32 per wave).
//
// Differences from the f32 kernel:
//   - slab (MFMA B operand) holds fp16 hi and lo planes in [k-octet][frame][8] order: one lane's fragment
//     (8 consecutive k of one frame) is a single ds_read_b128 and a 16-lane group covers 16 distinct 16-byte
//     slots -> conflict free; same byte footprint as the f32 slab;
//   - weights are scaled, split and packed on the host into [o-tile][k32][hi|lo][lane][8 halves] (two 16-byte loads
//     per lane, o-tile and K step);
//   - producers convert on the fly (scale, cvt, subtract, cvt) and store halves; the MDTC mid tile is written by the
//     first epilogue directly in operand order.
#pragma once
#include "conv_stack.hip.h"

namespace wekws {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

struct F16Frag {
  f16x8 h, l;
};

__device__ __forceinline__ void split16(float v, _Float16& h, _Float16& l) {
  h = static_cast<_Float16>(v);
  l = static_cast<_Float16>(v - static_cast<float>(h));
}

// Scale and split in three instructions: hi = fp16(v * s) (v_fma_mixlo_f16: one rounding of the exact product),
// d = v * s - hi (v_fma_mix_f32 with hi as an fp16 operand: exact), lo = fp16(d).  s is a power of two.
__device__ __forceinline__ void split16s(float v, float s, _Float16& h, _Float16& l) {
  unsigned hr;
  float d;
  asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hr) : "v"(v), "v"(s));            // (upper half of hr: don't care)
  asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(d) : "v"(v), "v"(s), "v"(hr));
  h = __builtin_bit_cast(_Float16, static_cast<unsigned short>(hr & 0xffffu));
  l = static_cast<_Float16>(d);
}

// ---------------------------------------------------------------------------------------------
// Block floating point (see the header comment).
// ---------------------------------------------------------------------------------------------
// Tile maxima travel between phases through LDS cells (one 32-bit word each, zeroed at kernel entry): the publisher
// reduces inside the wave (six DPP steps) and lane 63 merges the wave's maximum with ds_max_u32; the reader needs ONE
// ds_read_b32.  Every wave reads the cells and derives the scales itself, redundantly, on the VECTOR unit: sixteen
// waves bouncing a handful of values between the vector and the scalar unit (v_readfirstlane and back) measured four
// times slower on the block-to-block critical path of the streaming kernels.
// Cells per utterance: [0] features, [1] incoming cache, [2 + i] the residual tile entering block i (i = nblocks: the
// backbone output).
struct AmaxCell {
  unsigned v;
};
constexpr int kAmaxMaxBlocks = 24;                     // rows of the LDS block table (MDTC 4 x 4 + 1 = 17, DS-TCN 4)
constexpr int kAmaxCells = 3 + kAmaxMaxBlocks;

template <int NTHR>
__device__ __forceinline__ void amax_zero(AmaxCell* cells, int n) {
  for (int e = threadIdx.x; e < n; e += NTHR) cells[e].v = 0u;
}

// s = 2^(14 - floor(log2 bound)): bound * s in [2^14, 2^15); *inv = 1 / s.  Six integer operations, no special cases:
// a zero (or f32-subnormal) bound means every element of the tile is zero (or below 2^-126) and gets s = 2^126; a
// non-finite bound gets s = 2^-114 and the infinities stay infinities, as they would in fp32 arithmetic.
__device__ __forceinline__ float pow2_scale(float bound, float* inv) {
  const uint32_t e = (__float_as_uint(bound) >> 23) & 0xffu;   // biased exponent
  const uint32_t se = min(268u - e, 253u);                      // 127 + 14 - (e - 127), capped so that 1 / s stays normal
  *inv = __uint_as_float((254u - se) << 23);
  return __uint_as_float(se << 23);
}

// x = max(x, x of the lane the DPP control selects): ONE v_max_u32 with a DPP operand (the compiler's own lowering of
// update_dpp + max is a v_mov_dpp, a max and a copy per step).  Written as inline assembly, so the wait states a DPP
// read of a just-written VGPR needs on gfx9 are spelled out.
#define WEKWS_UMAX_DPP(x, ctrl) asm volatile("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 " ctrl " bank_mask:0xf" : "+v"(x))
// max over the wave of non-negative floats (ordered like their bit patterns), valid in lane 63
__device__ __forceinline__ unsigned wave_umax63(unsigned x) {
  WEKWS_UMAX_DPP(x, "quad_perm:[1,0,3,2] row_mask:0xf");
  WEKWS_UMAX_DPP(x, "quad_perm:[2,3,0,1] row_mask:0xf");
  WEKWS_UMAX_DPP(x, "row_half_mirror row_mask:0xf");
  WEKWS_UMAX_DPP(x, "row_mirror row_mask:0xf");          // every lane holds its row's maximum
  WEKWS_UMAX_DPP(x, "row_bcast:15 row_mask:0xa");        // rows 1 and 3 take in rows 0 and 2
  WEKWS_UMAX_DPP(x, "row_bcast:31 row_mask:0xc");        // rows 2 and 3 take in row 1: lane 63 holds the maximum
  return x;
}
// Publish this thread's partial max|.| (v >= 0; a NaN orders above everything and ends up disabling the scale): wave
// reduction, then lane 63 merges into the cell.
// The atomic is inline assembly, i.e. NOT in the compiler's LDS bookkeeping: a barrier behind the publish gets no
// `s_waitcnt lgkmcnt(0)` from it, and a wave that passes the barrier can read the cell before the atomic has landed (found
// in round 4 on ds64_g4.hip.h, four workgroups per CU: ~3 % of the utterances of a large batch were off by 1e-2 -- waves
// of one workgroup had derived DIFFERENT power-of-two scales for rows of the same operand).  The wait is part of the
// publish.
__device__ __forceinline__ void amax_publish(AmaxCell* cell, float v) {
  const unsigned x = wave_umax63(__float_as_uint(v));
  // lane 63 alone issues the ds_max_u32: EXEC is narrowed by hand (five instructions; the compiler's lowering of a
  // one-lane atomicMax adds an active-lane scan loop that costs ~200 cycles of a block boundary)
  const unsigned addr = static_cast<unsigned>(reinterpret_cast<uintptr_t>(&cell->v));   // LDS offset = low half of the flat address
  unsigned long long saved;
  asm volatile(
      "s_mov_b64 %0, exec\n\t"
      "s_mov_b32 exec_lo, 0\n\t"
      "s_mov_b32 exec_hi, 0x80000000\n\t"
      "ds_max_u32 %1, %2\n\t"
      "s_mov_b64 exec, %0\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&s"(saved)
      : "v"(addr), "v"(x)
      : "memory");
}
__device__ __forceinline__ float amax_read(const AmaxCell* cell) { return __uint_as_float(cell->v); }

// INPUT maxima (features, incoming cache) are taken on the bit patterns of |v| -- v_and + v_max_u32 instead of v_max_f32 --: for
// finite values that is the same maximum, and a NaN or an infinity (patterns 0x7f800000 and above) ends up on top instead of being
// dropped by fmaxf.  A cell at or above 0x7f800000 after the barrier = "this utterance's input holds a non-finite value": the
// utterance leaves the fast path (nonfinite.hip.h).  The running maximum `m` is a non-negative float used as its bit pattern.
__device__ __forceinline__ float amax_acc(float m, float v) {
  return __uint_as_float(max(__float_as_uint(m), __float_as_uint(v) & 0x7fffffffu));
}
__device__ __forceinline__ float amax_merge(float a, float b) { return __uint_as_float(max(__float_as_uint(a), __float_as_uint(b))); }
// cells[0] = features, cells[1] = incoming cache (zero when there is none); workgroup-uniform, scalar
__device__ __forceinline__ bool amax_inputs_bad(const AmaxCell* cells) {
#ifdef WEKWS_NF_OFF                                          // (A/B builds only: tools/abvar.sh)
  return false;
#endif
  return unsigned(__builtin_amdgcn_readfirstlane(int(max(cells[0].v, cells[1].v)))) >= 0x7f800000u;
}

// The block table, copied into LDS once: reading a descriptor at a block boundary then costs an LDS round trip instead of
// a trip to L2 that every wave of the workgroup waits for (a uniform global load is not scalarised here: the kernels
// also store to global memory).
template <int NTHR, class BD>
__device__ __forceinline__ void stage_block_table(BD* dst, const BD* __restrict__ src, int nb) {
  const int n = nb * int(sizeof(BD) / 4);
  for (int e = threadIdx.x; e < n; e += NTHR) reinterpret_cast<uint32_t*>(dst)[e] = reinterpret_cast<const uint32_t*>(src)[e];
}

// max|.| over a contiguous run of n floats, this thread's share (stride = workgroup size).  Eight loads are in flight
// per thread before the first is consumed: the run costs one trip to memory per 8 x NTHR elements, not one per element.
template <int NTHR>
__device__ __forceinline__ float amax_span(const float* __restrict__ p, int n, float m) {
  constexpr int DEPTH = 8;
  for (int e0 = threadIdx.x; e0 < n; e0 += NTHR * DEPTH) {
    float v[DEPTH];
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) {
      const int e = e0 + k * NTHR;
      v[k] = p[e < n ? e : e0];                              // (clamped duplicate: harmless for a maximum)
    }
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) m = fmaxf(m, fabsf(v[k]));
  }
  return m;
}
// the same on the bit patterns of |v| (amax_acc): NaN / Inf stay on top (the kernels that look for non-finite inputs themselves)
template <int NTHR>
__device__ __forceinline__ float amax_span_bits(const float* __restrict__ p, int n, float m) {
  constexpr int DEPTH = 8;
  for (int e0 = threadIdx.x; e0 < n; e0 += NTHR * DEPTH) {
    float v[DEPTH];
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) {
      const int e = e0 + k * NTHR;
      v[k] = p[e < n ? e : e0];
    }
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) m = amax_acc(m, v[k]);
  }
  return m;
}

// Byte size of one (utterance, buffer) hi or lo plane holding KCH channels x TT frames
template <int KCH, int TT>
struct Plane {
  static constexpr int BYTES = (KCH / 8) * TT * 16;
};

// acc += A x B for ONE 32-deep K step.  bh / bl: this lane's 16-byte item of t-tile 0 in the hi / lo plane
// ((k-octet = lane>>4, frame = lane&15)); consecutive t-tiles are 16 items (256 B) apart.
template <int OW, int NT>
__device__ __forceinline__ void mfma16_step(f32x4 (&acc)[OW][NT], const F16Frag (&a)[OW], const char* bh, const char* bl) {
  f16x8 vh[2], vl[2];
  vh[0] = *reinterpret_cast<const f16x8*>(bh);
  vl[0] = *reinterpret_cast<const f16x8*>(bl);
#pragma unroll
  for (int tt = 0; tt < NT; ++tt) {
    if (tt + 1 < NT) {
      vh[(tt + 1) & 1] = *reinterpret_cast<const f16x8*>(bh + (tt + 1) * 256);
      vl[(tt + 1) & 1] = *reinterpret_cast<const f16x8*>(bl + (tt + 1) * 256);
    }
    __builtin_amdgcn_sched_barrier(0);  // reads of tile tt+1 stay ahead of the MFMAs of tile tt
#pragma unroll
    for (int ow = 0; ow < OW; ++ow) {
      acc[ow][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ow].h, vh[tt & 1], acc[ow][tt], 0, 0, 0);
      acc[ow][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ow].h, vl[tt & 1], acc[ow][tt], 0, 0, 0);
      acc[ow][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ow].l, vh[tt & 1], acc[ow][tt], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// A fragments of one K step: image order [o-tile][k32][hi|lo][lane][8 halves] = 128 uint4 per (o-tile, k32)
template <int OW>
__device__ __forceinline__ void load_a16(F16Frag (&a)[OW], const uint4* __restrict__ ap, int ot_stride) {
#pragma unroll
  for (int ow = 0; ow < OW; ++ow) {
    const uint4 h = ap[ow * ot_stride], l = ap[ow * ot_stride + 64];
    a[ow].h = __builtin_bit_cast(f16x8, h);
    a[ow].l = __builtin_bit_cast(f16x8, l);
  }
}

// ---------------------------------------------------------------------------------------------
template <int KIND, int C, int NT, int KS>
__global__ __launch_bounds__(kThreads, 2) void conv_stack_f16_kernel(const StackParams P, const CallArgs A) {
  using G = Geom<KIND, C, NT>;
  constexpr int U = G::U, OW = G::OW, SS = G::SS;
  constexpr int TT = 16 * NT;
  constexpr int KC = 32;                                   // one MFMA K step per produced chunk
  constexpr int NBUF = (C >= 64) ? 2 : 1;                  // C = 32: the whole K is one chunk
  constexpr int PB = Plane<KC, TT>::BYTES;                 // one hi (or lo) plane of a chunk
  constexpr int UB = (KIND == KIND_MDTC && C > 2 * KC) ? 2 * Plane<C, TT>::BYTES : NBUF * 2 * PB;  // bytes per utterance
  static_assert(U * UB <= G::S_FLOATS * 4, "fp16 slab must fit the shared geometry");
  constexpr int RP = (U * KC) / (kThreads / 16);           // slab rows per 16-lane group per chunk
  static_assert(RP * (kThreads / 16) == U * KC && RP >= 1, "producer decomposition");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // the operand slab sits BELOW the activation tile so that a (slightly) negative frame index on channel row 0
  // still addresses valid LDS: left-context reads need no index clamp, only the select
  char* const slab = reinterpret_cast<char*>(lds);
  float* const hbuf = lds + G::S_FLOATS;                   // [U][C][SS] f32 resident activations

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform (SGPR)
  const int wo = wave % G::WO, wu = wave / G::WO;
  const int l15 = lane & 15, lq = lane >> 4;
  const int T = A.T;
  const int b0 = blockIdx.x * U;
  const float* __restrict__ W = P.w;
  const int Pc = P.cache_len;
  const int pg = tid >> 4, tl = tid & 15;
  const int o_base = wo * OW * 16;
  float* const h_w = hbuf + wu * C * SS;
  char* const slab_u = slab + wu * UB;                     // this wave's utterance
  const int frag_off = (lq * TT + l15) * 16;               // this lane's item of t-tile 0 inside a plane

  f32x4 acc[OW][NT];
  f32x4 zsum[KIND == KIND_MDTC ? OW : 1][KIND == KIND_MDTC ? NT : 1];
  if constexpr (KIND == KIND_MDTC) zero_acc(zsum);

  // ---- block floating point: per-utterance maxima of the features and of the incoming cache
  __shared__ AmaxCell amax_cells[U * kAmaxCells];
  __shared__ BlockDesc blk[kAmaxMaxBlocks];
  AmaxCell* const cells_w = amax_cells + wu * kAmaxCells;  // this wave's utterance
  amax_zero<kThreads>(amax_cells, U * kAmaxCells);
  stage_block_table<kThreads>(blk, P.blocks, P.nblocks);
  __syncthreads();
  for (int u = 0; u < U; ++u) {
    if (b0 + u < A.B) {                                    // workgroup-uniform
      amax_publish(amax_cells + u * kAmaxCells, amax_span_bits<kThreads>(A.x + int64_t(b0 + u) * A.xs_b, T * P.idim, 0.f));
      if (A.in_cache)
        amax_publish(amax_cells + u * kAmaxCells + 1, amax_span_bits<kThreads>(A.in_cache + int64_t(b0 + u) * C * Pc, C * Pc, 0.f));
    }
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

  // ============================ preprocessing: h0 = [ReLU](x Wpre^T + b) ============================
  {
    zero_acc(acc);
    const int nk = P.kpre16 / 32;                          // K steps (idim rounded up to 32)
    const int ot_stride = nk * 128;
    const uint4* ap = reinterpret_cast<const uint4*>(W + P.pre_a16) + (wo * OW) * ot_stride + lane;
    float4 bias[OW];
#pragma unroll
    for (int ow = 0; ow < OW; ++ow) bias[ow] = *reinterpret_cast<const float4*>(W + P.pre_b + o_base + ow * 16 + lq * 4);
    for (int k0 = 0; k0 < nk; k0 += NBUF) {                // NBUF K steps staged per pass
      const int steps = min(NBUF, nk - k0);
      __syncthreads();
      float sxu[U];                                        // feature scale of each utterance
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float inv_unused;
        sxu[u] = pow2_scale(amax_read(amax_cells + u * kAmaxCells), &inv_unused);
      }
      // item = (utterance, step, k-octet, frame): 8 consecutive features of one frame -> one 16-byte hi + lo store
      for (int e = tid; e < U * steps * 4 * TT; e += kThreads) {
        const int t = e % TT;
        int q = e / TT;
        const int oct = q & 3; q >>= 2;
        const int st = q % steps, u = q / steps;
        const int kf = (k0 + st) * 32 + oct * 8;
        const bool ok = (b0 + u) < A.B && t < T;
        const float* xr = A.x + int64_t(b0 + u) * A.xs_b + int64_t(t) * P.idim + kf;
        float sx = sxu[0];
#pragma unroll
        for (int q = 1; q < U; ++q) sx = (u == q) ? sxu[q] : sx;
        f16x8 vh, vl;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float v = (ok && kf + i < P.idim) ? xr[i] * sx : 0.f;
          _Float16 h, l;
          split16(v, h, l);
          vh[i] = h; vl[i] = l;
        }
        char* dst = slab + u * UB + st * 2 * PB + (oct * TT + t) * 16;
        *reinterpret_cast<f16x8*>(dst) = vh;
        *reinterpret_cast<f16x8*>(dst + PB) = vl;
      }
      __syncthreads();
      for (int st = 0; st < steps; ++st) {
        F16Frag a[OW];
        load_a16<OW>(a, ap + (k0 + st) * 128, ot_stride);
        mfma16_step<OW, NT>(acc, a, slab_u + st * 2 * PB + frag_off, slab_u + st * 2 * PB + PB + frag_off);
      }
    }
    float cpre;
    (void)pow2_scale(amax_read(cells_w), &cpre);
    cpre *= P.pre_inv_s;                                   // 1 / (feature scale * weight scale)
    float hmax = 0.f;
#pragma unroll
    for (int ow = 0; ow < OW; ++ow) {
      const int o = o_base + ow * 16 + lq * 4;
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        const int t = tt * 16 + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = fmaf(acc[ow][tt][r], cpre, f4c(bias[ow], r));
          if (P.pre_relu) v = fmaxf(v, 0.f);
          h_w[(o + r) * SS + t] = v;
          hmax = fmaxf(hmax, fabsf(v));
        }
      }
    }
    amax_publish(cells_w + 2, hmax);
    __syncthreads();
  }

  // ======================================= residual blocks =======================================
  for (int bi = 0; bi < P.nblocks; ++bi) {
    const BlockDesc bd = blk[bi];
    const int d = bd.dil, pad = bd.pad;
    constexpr int K1 = (KIND == KIND_TCN) ? C * KS : C;
    constexpr int nch = K1 / KC;
    static_assert(NBUF == 1 ? nch == 1 : nch % 2 == 0, "chunk pipeline shape");
    const int ot_stride1 = nch * 128;
    const uint4* ap1 = reinterpret_cast<const uint4*>(W + bd.a1_16) + (wo * OW) * ot_stride1 + lane;

    float dww[KIND == KIND_TCN ? 1 : RP][KIND == KIND_TCN ? 1 : KS + 1];
    auto load_dw = [&](int n) __attribute__((always_inline)) {
      if constexpr (KIND != KIND_TCN) {
#pragma unroll
        for (int i = 0; i < RP; ++i) {
          const int item = pg + i * (kThreads / 16);
          const int c = n * KC + (item % KC);
          constexpr int DWP = (KS + 1 + 3) / 4 * 4;
          const float4* src = reinterpret_cast<const float4*>(W + bd.dw_pk + c * DWP);
#pragma unroll
          for (int q = 0; q < DWP / 4; ++q) {
            const float4 w4 = src[q];
            if (q * 4 + 0 <= KS) dww[i][q * 4 + 0] = w4.x;
            if (q * 4 + 1 <= KS) dww[i][q * 4 + 1] = w4.y;
            if (q * 4 + 2 <= KS) dww[i][q * 4 + 2] = w4.z;
            if (q * 4 + 3 <= KS) dww[i][q * 4 + 3] = w4.w;
          }
        }
      }
    };
    F16Frag a0[OW], a1[OW];
    load_dw(0);
    load_a16<OW>(a0, ap1, ot_stride1);
    float4 ebias[OW];
#pragma unroll
    for (int ow = 0; ow < OW; ++ow)
      ebias[ow] = *reinterpret_cast<const float4*>(W + (KIND == KIND_MDTC ? bd.b2 : bd.b1) + o_base + ow * 16 + lq * 4);

    const bool slide = slide_ok(d);
    const int fbase = slide ? slide_base(tl, d, NT) : tl;

    // ---- operand scale of this block, per utterance: the producer's rows are bounded by the maximum of the input
    //      tile (published by the epilogue that wrote it) and of the incoming cache
    float sa[U], c1 = 0.f, sm = 1.f, c2 = 1.f;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float au = fmaxf(amax_read(amax_cells + u * kAmaxCells + 2 + bi), amax_read(amax_cells + u * kAmaxCells + 1));
      const float ba = KIND == KIND_TCN ? au : fmaf(bd.dw_alpha, au, bd.dw_beta);
      float inv;
      sa[u] = pow2_scale(ba, &inv);
      if (u == wu) {
        c1 = inv * bd.inv_s1;
        if constexpr (KIND == KIND_MDTC) {
          sm = pow2_scale(fmaf(bd.mid_alpha, ba, bd.mid_beta), &c2);
          c2 *= bd.inv_s2;
        }
      }
    }
    (void)sm; (void)c2;

    // ---- producer of K-chunk n into slab buffer `buf` (fp16 hi / lo planes, [k-octet][frame][8])
    auto produce_impl = [&](int n, int buf, auto has_cache_tag) __attribute__((always_inline)) {
      constexpr bool HAS_CACHE = decltype(has_cache_tag)::value;
#define fetch(idx_)                                                                      \
  ({                                                                                     \
    const int ix_ = (idx_);                                                              \
    float fv_ = hbuf[hoff + ix_];                                                        \
    if constexpr (HAS_CACHE) {                                                           \
      const float fg_ = A.in_cache[gbase + pad + min(ix_, -1)];                          \
      fv_ = ix_ >= 0 ? fv_ : (uok ? fg_ : 0.f);                                          \
    } else {                                                                             \
      fv_ = ix_ >= 0 ? fv_ : 0.f;                                                        \
    }                                                                                    \
    fv_;                                                                                 \
  })
#pragma unroll
      for (int i = 0; i < RP; ++i) {
        const int item = pg + i * (kThreads / 16);
        const int u = item / KC, r = item % KC;
        const bool uok = (b0 + u) < A.B;
        float sau = sa[0];
#pragma unroll
        for (int q = 1; q < U; ++q) sau = (u == q) ? sa[q] : sau;
        char* const plane = slab + u * UB + buf * 2 * PB;       // hi plane; lo plane at + PB
        if constexpr (KIND == KIND_TCN) {
          // dense conv as GEMM over K' = (c, j): the 8 taps of channel c are one k-octet   (tcn.py:76-80)
          // 16 lanes cover the 8 taps x 2 frame halves; one lane stores one half (2 bytes) per frame.
          const int kk = n * KC + r;
          const int c = kk / KS, j0 = kk % KS;
          const int hoff = (u * C + c) * SS;
          const int64_t gbase = (int64_t(uok ? b0 + u : 0) * C + c) * Pc + bd.cache_off;
          if (A.out_cache && uok && j0 == 0) {
            for (int p = tl; p < pad; p += 16) {
              const int src = T + p - pad;
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
          const int sh = (KS - 1 - j0) * d;
          _Float16* ph = reinterpret_cast<_Float16*>(plane) + ((r >> 3) * TT) * 8 + (r & 7);
          _Float16* pl = reinterpret_cast<_Float16*>(plane + PB) + ((r >> 3) * TT) * 8 + (r & 7);
#pragma unroll
          for (int m = 0; m < NT; ++m) {
            const int t = tl + 16 * m;
            const float v = fetch(t - sh);
            _Float16 h, l;
            split16s(v, sau, h, l);
            ph[t * 8] = h;
            pl[t * 8] = l;
          }
        } else {
          // depthwise dilated conv + folded BN (+ReLU for DS-TCN)        (tcn.py:102-109, mdtc.py:55-58)
          const int c = n * KC + r;
          const int hoff = (u * C + c) * SS;
          const int64_t gbase = (int64_t(uok ? b0 + u : 0) * C + c) * Pc + bd.cache_off;
          if (A.out_cache && uok) {
            for (int p = tl; p < pad; p += 16) {
              const int src = T + p - pad;
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
          _Float16* ph = reinterpret_cast<_Float16*>(plane) + ((r >> 3) * TT) * 8 + (r & 7);
          _Float16* pl = reinterpret_cast<_Float16*>(plane + PB) + ((r >> 3) * TT) * 8 + (r & 7);
          if (slide) {
            float v[NT + KS - 1];
#pragma unroll
            for (int q = 0; q < NT + KS - 1; ++q) v[q] = fetch(fbase + (q - (KS - 1)) * d);
#pragma unroll
            for (int m = 0; m < NT; ++m) {
              float o = dww[i][KS];
#pragma unroll
              for (int j = 0; j < KS; ++j) o = fmaf(dww[i][j], v[m + j], o);
              if (KIND == KIND_DS) o = fmaxf(o, 0.f);
              const int t = fbase + m * d;
              _Float16 h, l;
              split16s(o, sau, h, l);
              ph[t * 8] = h;
              pl[t * 8] = l;
            }
          } else {
#pragma unroll 1
            for (int m = 0; m < NT; ++m) {
              const int t = tl + 16 * m;
              float o = dww[i][KS];
#pragma unroll
              for (int j = 0; j < KS; ++j) o = fmaf(dww[i][j], fetch(t - (KS - 1 - j) * d), o);
              if (KIND == KIND_DS) o = fmaxf(o, 0.f);
              _Float16 h, l;
              split16s(o, sau, h, l);
              ph[t * 8] = h;
              pl[t * 8] = l;
            }
          }
        }
      }
#undef fetch
    };
    const bool has_cache = A.in_cache != nullptr;
    auto produce = [&](int n, int buf) __attribute__((always_inline)) {
      if (has_cache) produce_impl(n, buf, std::true_type{});
      else produce_impl(n, buf, std::false_type{});
    };

    // ---- GEMM 1 over K: one MFMA K step per chunk, double-buffered planes, one barrier per chunk
    zero_acc(acc);
    produce(0, 0);
    load_dw(min(1, nch - 1));
    __syncthreads();
    if constexpr (NBUF == 2) {
      for (int n = 0; n < nch; n += 2) {
        produce(n + 1, 1);
        load_a16<OW>(a1, ap1 + (n + 1) * 128, ot_stride1);
        load_dw(min(n + 2, nch - 1));
        __builtin_amdgcn_sched_barrier(0);
        mfma16_step<OW, NT>(acc, a0, slab_u + frag_off, slab_u + PB + frag_off);
        __syncthreads();
        if (n + 2 < nch) produce(n + 2, 0);
        load_a16<OW>(a0, ap1 + min(n + 2, nch - 1) * 128, ot_stride1);
        load_dw(min(n + 3, nch - 1));
        __builtin_amdgcn_sched_barrier(0);
        mfma16_step<OW, NT>(acc, a1, slab_u + 2 * PB + frag_off, slab_u + 3 * PB + frag_off);
        __syncthreads();
      }
    } else {
      mfma16_step<OW, NT>(acc, a0, slab_u + frag_off, slab_u + PB + frag_off);
      __syncthreads();
      (void)a1;
    }

    if constexpr (KIND == KIND_MDTC) {
      // ---- mid = ReLU(BN1(pointwise)) written straight in operand order, then conv2 (1x1) + BN2   (mdtc.py:113-116)
      constexpr int NK2 = C / 32;
      constexpr int MPB = Plane<C, TT>::BYTES;                 // hi plane of the full-width mid tile
      static_assert(2 * MPB <= UB, "mid tile must fit the per-utterance slab");
      F16Frag a2[NK2][OW];
      const uint4* ap2 = reinterpret_cast<const uint4*>(W + bd.a2_16) + (wo * OW) * (NK2 * 128) + lane;
#pragma unroll
      for (int ks = 0; ks < NK2; ++ks) load_a16<OW>(a2[ks], ap2 + ks * 128, NK2 * 128);
      // mid = ReLU(BN1(pointwise)): scaled by the bound chained behind the depthwise one (BlockDesc::mid_alpha)
#pragma unroll
      for (int ow = 0; ow < OW; ++ow) {
        const int o = o_base + ow * 16 + lq * 4;
        const float4 bias = *reinterpret_cast<const float4*>(W + bd.b1 + o);
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
          const int t = tt * 16 + l15;
          f16x4 vh, vl;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = fmaxf(fmaf(acc[ow][tt][r], c1, f4c(bias, r)), 0.f);
            _Float16 h, l;
            split16(v * sm, h, l);
            vh[r] = h; vl[r] = l;
          }
          char* dst = slab_u + (((o >> 3) * TT + t) * 8 + (o & 7)) * 2;   // 4 consecutive channels = 8 bytes
          *reinterpret_cast<f16x4*>(dst) = vh;
          *reinterpret_cast<f16x4*>(dst + MPB) = vl;
        }
      }
      __syncthreads();
      zero_acc(acc);
#pragma unroll
      for (int ks = 0; ks < NK2; ++ks)
        mfma16_step<OW, NT>(acc, a2[ks], slab_u + ks * 4 * TT * 16 + frag_off, slab_u + MPB + ks * 4 * TT * 16 + frag_off);
      c1 = c2;                                               // the epilogue below undoes conv2's scales
    }

    // ---- epilogue: bias (+ReLU) + residual, in place into h
    float hmax = 0.f;
#pragma unroll
    for (int ow = 0; ow < OW; ++ow) {
      const int o = o_base + ow * 16 + lq * 4;
      const float4 bias = ebias[ow];
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        const int t = tt * 16 + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = fmaf(acc[ow][tt][r], c1, f4c(bias, r));
          float* hp = h_w + (o + r) * SS + t;
          if constexpr (KIND == KIND_MDTC) {
            v = fmaxf(v + *hp, 0.f);
            if (bd.zadd) zsum[ow][tt][r] += v;
          } else {
            v = fmaxf(v, 0.f) + *hp;
          }
          *hp = v;
          hmax = fmaxf(hmax, fabsf(v));
        }
      }
    }
    amax_publish(cells_w + 3 + bi, hmax);              // = the input tile of block bi + 1
    __syncthreads();
  }

  if constexpr (KIND == KIND_MDTC) {
#pragma unroll
    for (int ow = 0; ow < OW; ++ow) {
      const int o = o_base + ow * 16 + lq * 4;
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        const int t = tt * 16 + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) h_w[(o + r) * SS + t] = zsum[ow][tt][r];
      }
    }
    __syncthreads();
  }

  conv_stack_head<KIND, C, NT>(P, A, hbuf, reinterpret_cast<float*>(slab), b0);
}

template <int KIND>
int launch_conv_stack_f16(int C, int nt, const StackParams& P, const CallArgs& A, hipStream_t stream);
template <> int launch_conv_stack_f16<KIND_DS>(int, int, const StackParams&, const CallArgs&, hipStream_t);
template <> int launch_conv_stack_f16<KIND_TCN>(int, int, const StackParams&, const CallArgs&, hipStream_t);
template <> int launch_conv_stack_f16<KIND_MDTC>(int, int, const StackParams&, const CallArgs&, hipStream_t);

template <int KIND, int C, int NT>
inline int launch_one_f16(const StackParams& P, const CallArgs& A, hipStream_t stream) {
  using G = Geom<KIND, C, NT>;
  constexpr int KS = (KIND == KIND_MDTC) ? 5 : 8;
  if (P.ksize != KS) return -4;
  if constexpr (KIND == KIND_TCN && C < 64) {
    return -4;  // K = 8*C needs a double-buffered slab that does not fit Geom<> at C = 32; served by dense_stack_f16
  } else {
  static DynLdsGrant grant;
  auto kern = conv_stack_f16_kernel<KIND, C, NT, KS>;
  if (grant_dynamic_lds(kern, int(G::LDS_BYTES), grant)) return -3;
  const int grid = (A.B + G::U - 1) / G::U;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), G::LDS_BYTES, stream, P, A);
  return hipGetLastError() == hipSuccess ? 0 : -3;
  }
}

#define WEKWS_DISPATCH_NT_F16(KIND, CC)                                        \
  switch (nt) {                                                                \
    case 1: return launch_one_f16<KIND, CC, 1>(P, A, stream);                  \
    case 2: return launch_one_f16<KIND, CC, 2>(P, A, stream);                  \
    case 4: return launch_one_f16<KIND, CC, 4>(P, A, stream);                  \
    case 7: return launch_one_f16<KIND, CC, 7>(P, A, stream);                  \
    default: return -1;                                                        \
  }

#define WEKWS_DEFINE_LAUNCHER_F16(KIND, WITH256)                               \
  template <>                                                                  \
  int launch_conv_stack_f16<KIND>(int C, int nt, const StackParams& P, const CallArgs& A, hipStream_t stream) { \
    switch (C) {                                                               \
      case 32: WEKWS_DISPATCH_NT_F16(KIND, 32)                                 \
      case 64: WEKWS_DISPATCH_NT_F16(KIND, 64)                                 \
      case 128: WEKWS_DISPATCH_NT_F16(KIND, 128)                               \
      case 256: if constexpr (WITH256) { WEKWS_DISPATCH_NT_F16(KIND, 256) } else return -4; \
      default: return -4;                                                      \
    }                                                                          \
  }

}  // namespace wekws
