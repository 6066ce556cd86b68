// DS-TCN, hidden_dim 256 (the reference's hi_xiaowen ds_tcn.yaml, the headline model): 16-wave variant of
// conv_stack_f16_kernel<KIND_DS, 256, NT, 8>.  Same arithmetic (3 x fp16 MFMA on hi/lo-split operands, fp32
// accumulate), same LDS footprint (f32 tile 256 x SS + 28,672 B of operand planes = 143,360 B at NT = 7, one
// workgroup per CU), same results -- different occupancy: 1024 threads = 4 waves per SIMD instead of 2.
//
// Why: after the move to fp16 matrix products the kernel is bound by vector-instruction ISSUE, not by the matrix pipe
// (profiles/r01c: MFMA pipe ~25 % busy), and tools/probe/valu_rate.hip measures what a SIMD issues per VALU
// instruction: 5.5 cycles with one wave, 3.0 with two, 1.9 with four.  The LDS tile allows only one workgroup per CU,
// so the only way to four waves per SIMD is a 16-wave workgroup: each wave owns ONE 16-channel o-tile for all NT
// frame tiles (28 accumulator registers at NT = 7, the whole kernel fits the 128-VGPR budget of 4 waves/SIMD).
//
// K is consumed in intervals of 64 channels = the whole operand slab (two 32-deep K steps), single-buffered:
//   [all 64 lane-groups produce one row each] barrier [every wave: 2 K steps x 3 products x NT tiles] barrier
// A double buffer would buy nothing here: a saturated MFMA stream lets one VALU instruction of the co-resident waves
// through per MFMA (tools/probe/mfma_probe.hip), so producer and matrix phases do not overlap on a SIMD anyway.
#pragma once
#include "conv_stack_f16.hip.h"

namespace wekws {

// One 32-deep K step for one o-tile, B fragments loaded tile by tile (no double buffer: with four waves per SIMD the
// LDS latency of a tile is covered by the other waves, and the registers are needed elsewhere).
// SPLIT = false is WEKWS_HIP_PRECISION_F16: weights and activations enter the product as plain fp16 (the hi halves
// only), one MFMA per product instead of three; the lo planes are neither written nor read.
template <int NT, bool SPLIT = true>
__device__ __forceinline__ void mfma16_step_nb(f32x4 (&acc)[NT], const F16Frag& a, const char* bh, const char* bl) {
#pragma unroll
  for (int tt = 0; tt < NT; ++tt) {
    const f16x8 vh = *reinterpret_cast<const f16x8*>(bh + tt * 256);
    acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.h, vh, acc[tt], 0, 0, 0);
    if constexpr (SPLIT) {
      const f16x8 vl = *reinterpret_cast<const f16x8*>(bl + tt * 256);
      acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.h, vl, acc[tt], 0, 0, 0);
      acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.l, vh, acc[tt], 0, 0, 0);
    }
  }
}

constexpr int kW16Threads = 1024;
constexpr int kW16Waves = 16;

// Preprocessing input of one utterance for feature dims up to 64 (two K steps = one slab fill): every thread owns at most
// ONE item (step, k-octet, frame) = 8 consecutive features of a frame.  The features make one trip from memory: they are
// loaded into registers, their maximum is published (block floating point), and after the barrier the same registers are
// scaled, split and stored as operand planes.
// The load is UNCONDITIONAL and branch-free (an item without features reads a valid dummy address and is zeroed where it
// is used): two control-flow arms producing the item made the compiler wait for the load right where the arms merge --
// one exposed trip to memory at the head of every kernel.  That needs whole, 16-byte aligned items: w16_x_vec_ok().
typedef float w16_f32x8 __attribute__((ext_vector_type(8)));   // (a plain float[8] member ends up in scratch)
struct W16XItem {
  w16_f32x8 v;
  int dst;                                                   // byte offset of the item's hi slot in the slab, -1: no item
  bool ok;                                                   // v holds features (else: dummy data, the item is zeros)
};
__device__ __forceinline__ bool w16_x_vec_ok(const float* x, int64_t xs_b, int idim) {   // (kernel-uniform)
  return idim % 8 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && xs_b % 4 == 0;
}
__device__ __forceinline__ void w16_fetch_x(W16XItem& it, const float* xr, const float* dummy, bool ok) {
  const float* p = ok ? xr : dummy;
  const float4 a = *reinterpret_cast<const float4*>(p), c = *reinterpret_cast<const float4*>(p + 4);
  it.v = w16_f32x8{a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
  it.ok = ok;
}
template <int TT, int PB>
__device__ __forceinline__ W16XItem w16_load_x(const float* __restrict__ xb, int T, int idim, int nk) {
  W16XItem it;
  const int e = threadIdx.x;
  const int t = e % TT, q = e / TT;
  const int oct = q & 3, st = q >> 2;
  const int kf = st * 32 + oct * 8;
  const bool has = e < nk * 4 * TT;
  it.dst = has ? st * 2 * PB + (oct * TT + t) * 16 : -1;
  w16_fetch_x(it, xb + int64_t(t) * idim + kf, xb, has && t < T && kf < idim);
  return it;
}
__device__ __forceinline__ float w16_x_amax(const W16XItem& it) {
  float m = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) m = fmaxf(m, fabsf(it.v[i]));
  return it.ok ? m : 0.f;
}
// the same on the bit patterns of |v| (amax_acc): a NaN / Inf among the item's values stays on top -- the kernels that look for
// non-finite inputs themselves (nonfinite.hip.h) publish this one
__device__ __forceinline__ float w16_x_amax_bits(const W16XItem& it) {
  float m = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) m = amax_acc(m, it.v[i]);
  return it.ok ? m : 0.f;
}
// scale, split and store the item at slab + dst (hi) / + lo_off (lo)
template <bool SPLIT>
__device__ __forceinline__ void w16_put_x(const W16XItem& it, float sx, char* slab, int lo_off) {
  if (it.dst < 0) return;
  f16x8 vh, vl;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    _Float16 h, l;
    split16s(it.ok ? it.v[i] : 0.f, sx, h, l);
    vh[i] = h; vl[i] = l;
  }
  *reinterpret_cast<f16x8*>(slab + it.dst) = vh;
  if constexpr (SPLIT) *reinterpret_cast<f16x8*>(slab + it.dst + lo_off) = vl;
}
template <int PB, bool SPLIT>
__device__ __forceinline__ void w16_store_x(const W16XItem& it, float sx, char* slab) {
  w16_put_x<SPLIT>(it, sx, slab, PB);
}

// Which of the 64 channel rows of a K interval lane-group pg (= tid >> 4) produces.  A wave's four lane-groups are the
// two 32-lane halves LDS instructions are serviced in; the permutation makes the two groups of a half produce rows 4
// apart -- different dwords of the operand planes (rows r, r + 1 share one: their 2-byte stores collided) and 16 banks
// apart in the f32 tile.  Within an octet: groups (0,1 | 2,3) of the even wave -> rows (0,4 | 1,5), odd wave -> (2,6 | 3,7).
__device__ __forceinline__ int w16_row(int pg) { return (pg & ~7) | (((pg & 1) << 2) + ((pg >> 1) & 1) + ((pg >> 2) & 1) * 2); }

template <int NT>
struct W16Geom {
  static constexpr int C = 256;
  static constexpr int TT = 16 * NT;
  // Row stride of the f32 tile: TT + 4.  LDS banks of 4-byte accesses are (a / 4) mod 32, serviced per 32-lane half:
  //   epilogue   a half = 16 frames x 2 row groups FOUR rows apart: 4 * SS == 16 (mod 32) -> 32 distinct banks
  //   producer   a half = 2 lane-groups; w16_row() puts them four channel rows apart as well (same 16-bank offset), and
  //              the 16 lanes of a group walk strided frame runs that cover 16 distinct banks for every dilation
  // (TT itself, == 16 mod 32, made the epilogue 2-way and the producer's 2-byte plane stores 4-way conflicted:
  // profiles/r01f: 22 % of the LDS cycles)
  static constexpr int SS = 16 * NT + 4;
  static constexpr int PB = Plane<32, TT>::BYTES;            // one hi (or lo) plane of one 32-channel K step
  static constexpr int SLAB = 4 * PB;                        // [kstep 0 hi | kstep 0 lo | kstep 1 hi | kstep 1 lo]
  static constexpr int H_FLOATS = C * SS;
  static constexpr size_t LDS_BYTES = size_t(SLAB) + size_t(H_FLOATS) * 4;
};

// HAS_CACHE: left context from the streaming cache in global memory (true) or zeros (false).  A template parameter,
// not a run-time branch: with both producer variants in one kernel the register allocation exceeds the 128-VGPR
// budget of four waves per SIMD and spills.
template <int NT, bool HAS_CACHE, bool SPLIT>
__global__ __launch_bounds__(kW16Threads) void ds256_w16_kernel(const StackParams P, const CallArgs A) {
  using G = W16Geom<NT>;
  constexpr int C = G::C, SS = G::SS, TT = G::TT, PB = G::PB, KS = 8;
  extern __shared__ __attribute__((aligned(16))) float w16_lds[];
  char* const slab = reinterpret_cast<char*>(w16_lds);       // operand planes, below the tile (no index clamp needed)
  float* const hbuf = w16_lds + G::SLAB / 4;                 // [256][SS] f32 resident activations

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform (SGPR)
  const int l15 = lane & 15, lq = lane >> 4;
  const int T = A.T;
  const int b = blockIdx.x;                                  // one utterance per workgroup
  const float* __restrict__ W = P.w;
  const int Pc = P.cache_len;
  const int pg = w16_row(tid >> 4), tl = tid & 15;           // producer: 64 lane-groups x 16 lanes; pg = the row it makes
  const int o0 = wave * 16 + lq * 4;                         // this lane's 4 output channels (o-tile = wave)
  const int frag_off = (lq * TT + l15) * 16;

  f32x4 acc[1][NT];

  // ---- block floating point (conv_stack_f16.hip.h): maxima of the feature tile and of the incoming cache
  __shared__ AmaxCell amax_cells[kAmaxCells];
  __shared__ BlockDesc blk[kAmaxMaxBlocks];
  amax_zero<kW16Threads>(amax_cells, kAmaxCells);
  stage_block_table<kW16Threads>(blk, P.blocks, P.nblocks);
  __syncthreads();
  const int nk = P.kpre16 / 32;
  // 40-d fbank: the features pass through registers once
  const bool one_trip = nk <= 2 && 8 * TT <= kW16Threads && w16_x_vec_ok(A.x, A.xs_b, P.idim);
  W16XItem xi;
  if (one_trip) {
    xi = w16_load_x<TT, PB>(A.x + int64_t(b) * A.xs_b, T, P.idim, nk);
    amax_publish(amax_cells, w16_x_amax_bits(xi));
  } else {
    amax_publish(amax_cells, amax_span_bits<kW16Threads>(A.x + int64_t(b) * A.xs_b, T * P.idim, 0.f));
  }
  if constexpr (HAS_CACHE)
    amax_publish(amax_cells + 1, amax_span_bits<kW16Threads>(A.in_cache + int64_t(b) * C * Pc, C * Pc, 0.f));
  __syncthreads();
  if (amax_inputs_bad(amax_cells)) {                         // a NaN / Inf feature or cache element: the reference's arithmetic
    nf_repair_call(A, b);
    return;
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
    for (int k0 = 0; k0 < nk; k0 += 2) {                     // two K steps staged per pass (= the slab)
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
    cpre *= P.pre_inv_s;                                     // 1 / (feature scale * weight scale)
    float hmax = 0.f;
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
      const int t = tt * 16 + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = fmaf(acc[0][tt][r], cpre, f4c(bias, r));
        if (P.pre_relu) v = fmaxf(v, 0.f);
        hbuf[(o0 + r) * SS + t] = v;
        hmax = fmaxf(hmax, fabsf(v));
      }
    }
    amax_publish(amax_cells + 2, hmax);
    __syncthreads();
  }
  // ======================================= residual blocks =======================================
  constexpr int NIV = C / 64;                                // K intervals per layer
  constexpr int OTS = (C / 32) * 128;                        // uint4 per o-tile (8 K steps)
  for (int bi = 0; bi < P.nblocks; ++bi) {
    const BlockDesc bd = blk[bi];
    const int d = bd.dil, pad = bd.pad;
    const uint4* ap1 = reinterpret_cast<const uint4*>(W + bd.a1_16) + size_t(wave) * OTS + lane;
    const float4 ebias = *reinterpret_cast<const float4*>(W + bd.b1 + o0);
    // taps + bias of the row this lane-group produces in the coming interval (padded 12-float record)
    float dww[KS + 1];
    auto load_dw = [&](int iv) __attribute__((always_inline)) {
      const float4* src = reinterpret_cast<const float4*>(W + bd.dw_pk + (iv * 64 + pg) * 12);
      const float4 q0 = src[0], q1 = src[1], q2 = src[2];
      dww[0] = q0.x; dww[1] = q0.y; dww[2] = q0.z; dww[3] = q0.w;
      dww[4] = q1.x; dww[5] = q1.y; dww[6] = q1.z; dww[7] = q1.w;
      dww[8] = q2.x;
    };
    // weight fragments of the two K steps of an interval; each is reloaded with the next interval's right after its
    // MFMAs have been issued, i.e. a whole producer phase ahead of its next use
    F16Frag a0[1], a1[1];
    load_dw(0);
    load_a16<1>(a0, ap1, 0);
    load_a16<1>(a1, ap1 + 128, 0);

    const bool slide = slide_ok(d);
    const int fbase = slide ? slide_base(tl, d, NT) : tl;

    // ---- operand scale of this block: the depthwise rows are bounded through the maximum of the input tile (published
    //      by the epilogue that wrote it) and of the incoming cache
    float c1;
    const float au = HAS_CACHE ? fmaxf(amax_read(amax_cells + 2 + bi), amax_read(amax_cells + 1))
                               : amax_read(amax_cells + 2 + bi);
    const float sa = pow2_scale(fmaf(bd.dw_alpha, au, bd.dw_beta), &c1);
    c1 *= bd.inv_s1;

    // ---- producer: lane-group pg makes slab row pg of interval iv: depthwise dilated conv + folded BN + ReLU of
    //      channel iv*64 + pg (tcn.py:102-109), split to fp16 hi/lo, and hands the channel's streaming cache over.
    auto produce_iv = [&](int iv) __attribute__((always_inline)) {
      const int r = pg, c = iv * 64 + pg;
      const int hoff = c * SS;
      const int64_t gbase = (int64_t(b) * C + c) * Pc + bd.cache_off;
#define fetch(idx_)                                                                      \
  ({                                                                                     \
    const int ix_ = (idx_);                                                              \
    float fv_ = hbuf[hoff + ix_];                                                        \
    if constexpr (HAS_CACHE) {                                                           \
      const float fg_ = A.in_cache[gbase + pad + min(ix_, -1)];                          \
      fv_ = ix_ >= 0 ? fv_ : fg_;                                                        \
    } else {                                                                             \
      fv_ = ix_ >= 0 ? fv_ : 0.f;                                                        \
    }                                                                                    \
    fv_;                                                                                 \
  })
      // the row's new streaming-cache slice = last `pad` frames of [old slice | h]
      if constexpr (HAS_CACHE) {
        // (with an incoming cache the kernel sits at the 128-register budget of four waves per SIMD: element by element)
        if (A.out_cache) {
          for (int p = tl; p < pad; p += 16) {
            const int src = T + p - pad;   // index into h (negative: still inside the old cache)
            const float hv = hbuf[hoff + max(src, 0)];
            const float g = A.in_cache[gbase + pad + min(src, -1)];
            A.out_cache[gbase + p] = src >= 0 ? hv : g;
          }
        }
      } else if (A.out_cache) {
        // lane tl hands over columns 4 tl .. 4 tl + 3 (+ 64 per pass: one pass for the recipes' four blocks, pad <= 56; a fifth
        // block has 112 columns) with ONE 16-byte store where the four exist -- rows of the (B, C, 105) cache are only 4-byte
        // aligned, so through a 4-byte-aligned type -- instead of four 4-byte stores in four passes
        struct __attribute__((packed, aligned(4))) V4 { float v[4]; };
        for (int p0 = 4 * tl; p0 < pad; p0 += 64) {
          float cv[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int src = T + p0 + k - pad;   // index into h (negative: left of the first frame -> zero context)
            const float v = hbuf[hoff + max(src, 0)];
            cv[k] = src >= 0 ? v : 0.f;
          }
          float* dst = A.out_cache + gbase + p0;
          if (p0 + 4 <= pad) {
            *reinterpret_cast<V4*>(dst) = V4{{cv[0], cv[1], cv[2], cv[3]}};
          } else {
#pragma unroll
            for (int k = 0; k < 3; ++k)
              if (p0 + k < pad) dst[k] = cv[k];
          }
        }
      }
      // row r -> K step r>>5, k-octet (r&31)>>3, half (r&7) of the [k-octet][frame][8] planes
      char* const plane = slab + (r >> 5) * 2 * PB;
      _Float16* ph = reinterpret_cast<_Float16*>(plane) + (((r & 31) >> 3) * TT) * 8 + (r & 7);
      _Float16* pl = reinterpret_cast<_Float16*>(plane + PB) + (((r & 31) >> 3) * TT) * 8 + (r & 7);
      if (slide) {
        float v[NT + KS - 1];
#pragma unroll
        for (int q = 0; q < NT + KS - 1; ++q) {
          // slots q >= KS-1 sit at frame fbase + (q-KS+1)*d >= 0: never left context, plain read, no select
          if (q >= KS - 1) v[q] = hbuf[hoff + fbase + (q - (KS - 1)) * d];
          else v[q] = fetch(fbase + (q - (KS - 1)) * d);
        }
#pragma unroll
        for (int m = 0; m < NT; ++m) {
          float o = dww[KS];
#pragma unroll
          for (int j = 0; j < KS; ++j) o = fmaf(dww[j], v[m + j], o);
          o = fmaxf(o, 0.f);
          const int t = fbase + m * d;
          _Float16 h, l;
          split16s(o, sa, h, l);
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
          o = fmaxf(o, 0.f);
          _Float16 h, l;
          split16s(o, sa, h, l);
          ph[t * 8] = h;
          if constexpr (SPLIT) pl[t * 8] = l;
        }
      }
#undef fetch
    };
    zero_acc(acc);
#pragma unroll 1
    for (int iv = 0; iv < NIV; ++iv) {
      const int nx = min(iv + 1, NIV - 1);                   // clamped: the last interval re-reads itself
      produce_iv(iv);
      load_dw(nx);
      __syncthreads();
      mfma16_step_nb<NT, SPLIT>(acc[0], a0[0], slab + frag_off, slab + PB + frag_off);
      load_a16<1>(a0, ap1 + (2 * nx) * 128, 0);
      mfma16_step_nb<NT, SPLIT>(acc[0], a1[0], slab + 2 * PB + frag_off, slab + 3 * PB + frag_off);
      load_a16<1>(a1, ap1 + (2 * nx + 1) * 128, 0);
      __syncthreads();
    }

    // ---- epilogue: folded bias + ReLU + residual, in place (tcn.py:60: add after the ReLU)
    float hmax = 0.f;
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
      const int t = tt * 16 + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float* hp = hbuf + (o0 + r) * SS + t;
        const float v = fmaxf(fmaf(acc[0][tt][r], c1, f4c(ebias, r)), 0.f) + *hp;
        *hp = v;
        hmax = fmaxf(hmax, fabsf(v));
      }
    }
    amax_publish(amax_cells + 3 + bi, hmax);             // = the input tile of block bi + 1
    __syncthreads();
  }

  conv_stack_head<KIND_DS, 256, NT, kW16Threads, SS>(P, A, hbuf, reinterpret_cast<float*>(slab), b);

}

template <int NT, bool HAS_CACHE, bool SPLIT>
inline int launch_ds256_w16_ntc(const StackParams& P, const CallArgs& A, hipStream_t stream) {
  using G = W16Geom<NT>;
  static DynLdsGrant grant;
  auto kern = ds256_w16_kernel<NT, HAS_CACHE, SPLIT>;
  if (grant_dynamic_lds(kern, int(G::LDS_BYTES), grant)) return -3;
  hipLaunchKernelGGL(kern, dim3(A.B), dim3(kW16Threads), G::LDS_BYTES, stream, P, A);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

template <int NT>
inline int launch_ds256_w16_nt(bool split, const StackParams& P, const CallArgs& A, hipStream_t stream) {
  if (split)
    return A.in_cache ? launch_ds256_w16_ntc<NT, true, true>(P, A, stream)
                      : launch_ds256_w16_ntc<NT, false, true>(P, A, stream);
  return A.in_cache ? launch_ds256_w16_ntc<NT, true, false>(P, A, stream)
                    : launch_ds256_w16_ntc<NT, false, false>(P, A, stream);
}

// split: three fp16 products per MAC on hi/lo operands (F16X3) or one on the hi halves (F16)
int launch_ds256_w16(int nt, bool split, const StackParams& P, const CallArgs& A, hipStream_t stream);

}  // namespace wekws
