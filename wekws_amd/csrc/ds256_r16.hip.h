// DS-TCN, hidden_dim 256, headline shape -- ROLE-SPLIT 16-wave kernel (round 3).  Same arithmetic, same LDS layout, same
// weight image and the same results bit for bit as ds256_w16.hip.h; a different division of labour.
//
// What round 2's counters said about ds256_w16 (all sixteen waves produce a 64-channel operand slab, barrier, all
// sixteen multiply it, barrier): per utterance 132 k cycles, of which the LDS array is busy 69 k, the matrix pipe needs
// 31 k (tools/probe/overlap_probe.hip: a 16x16x32 fp16 MFMA issues every 10.4 cycles from four waves per SIMD, not 16) and
// the vector units 28 k -- every unit idles more than half of the time because the phases are serial and, inside a
// phase, all waves hit the same unit at the same moment (14 LDS reads, then ~100 vector instructions, then 14 LDS
// stores, sixteen times over).  The same probe shows what a SIMD does when two of its four waves only multiply and
// the other two only run vector / LDS work: the multiplying waves keep their full rate, the others keep 60 .. 85 % of
// theirs.  Hence fixed roles:
//   waves 0..7   M-waves: wave w owns o-tiles 2w, 2w+1 (32 output channels) x all NT frame tiles: 8 NT accumulator
//                registers; one B fragment read from LDS feeds two o-tiles, so the operand slab is read 8 times per K
//                step instead of 16 (B reads were 30 k of the 69 k LDS cycles);
//   waves 8..15  P-waves: 32 lane-groups make the 32 channel rows of one K step (depthwise dilated conv + folded BN +
//                ReLU, block-floating scale, fp16 hi/lo split) and hand the rows' streaming cache over.
// The slab's two K-step buffers form a ring: P-waves fill buffer (k+1)&1 while M-waves multiply buffer k&1; ONE barrier
// per 32-channel K step (nine per block instead of eight pairs).  Block boundary: the M-waves' epilogue (bias, ReLU,
// residual, in place) runs while the P-waves wait, and the first K step of a block is produced while the M-waves wait
// -- the bubble the fixed roles cost; it is smaller than the phases they overlap.
#pragma once
#include "ds256_w16.hip.h"

namespace wekws {

// Development instrumentation (tools/probe/stamps_r16.py; -DWEKWS_R16_STAMPS, never in the product build): wave 0 (M) and
// wave 8 (P) of workgroup 0 accumulate clock64() per phase and leave the sums in the first floats of the returned cache.
#ifdef WEKWS_R16_STAMPS
#define R16_PH_DECL long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long tlast = clock64()
#define R16_PH(id) do { long long now_ = clock64(); tph[id] += now_ - tlast; tlast = now_; } while (0)
#define R16_PH_DUMP                                                                                    \
  do {                                                                                                 \
    __syncthreads();                                                                                   \
    if (b == 0 && A.out_cache && lane == 0 && (wave == 0 || wave == 8))                                \
      for (int i = 0; i < 8; ++i) A.out_cache[(wave ? 8 : 0) + i] = float(tph[i]);                     \
  } while (0)
#else
#define R16_PH_DECL
#define R16_PH(id)
#define R16_PH_DUMP
#endif

// One 32-deep K step for TWO o-tiles: a B fragment read from LDS feeds both.  The fragments are loaded tile by tile
// (no double buffer: the SIMD's other M-wave covers the LDS latency, and the 56 accumulators + two K steps of A fragments
// leave no registers for one); the two o-tiles' MFMAs alternate so that the three products of one accumulator are never
// back to back.
template <int NT, bool SPLIT>
__device__ __forceinline__ void r16_mfma_step(f32x4 (&acc)[2][NT], const F16Frag (&a)[2], const char* bh, const char* bl) {
#pragma unroll
  for (int tt = 0; tt < NT; ++tt) {
    const f16x8 vh = *reinterpret_cast<const f16x8*>(bh + tt * 256);
    acc[0][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0].h, vh, acc[0][tt], 0, 0, 0);
    acc[1][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1].h, vh, acc[1][tt], 0, 0, 0);
    if constexpr (SPLIT) {
      const f16x8 vl = *reinterpret_cast<const f16x8*>(bl + tt * 256);
      acc[0][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0].h, vl, acc[0][tt], 0, 0, 0);
      acc[1][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1].h, vl, acc[1][tt], 0, 0, 0);
      acc[0][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0].l, vh, acc[0][tt], 0, 0, 0);
      acc[1][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1].l, vh, acc[1][tt], 0, 0, 0);
    }
  }
}

template <int NT, bool HAS_CACHE, bool SPLIT>
__global__ __launch_bounds__(kW16Threads) void ds256_r16_kernel(const StackParams P, const CallArgs A) {
  using G = W16Geom<NT>;
  constexpr int C = G::C, SS = G::SS, TT = G::TT, PB = G::PB, KS = 8;
  extern __shared__ __attribute__((aligned(16))) float w16_lds[];
  char* const slab = reinterpret_cast<char*>(w16_lds);       // two K-step buffers: [hi | lo] [hi | lo]
  float* const hbuf = w16_lds + G::SLAB / 4;                 // [256][SS] f32 resident activations

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform (SGPR)
  const bool m_role = wave < 8;                              // waves 0..7 multiply, 8..15 produce
  const int l15 = lane & 15, lq = lane >> 4;
  const int T = A.T;
  const int b = blockIdx.x;                                  // one utterance per workgroup
  const float* __restrict__ W = P.w;
  const int Pc = P.cache_len;
  const int mw = wave & 7;                                   // index inside the role
  const int o0 = mw * 32 + lq * 4;                           // M: this lane's 4 output channels of o-tile 2 mw (+16: 2 mw + 1)
  const int frag_off = (lq * TT + l15) * 16;
  const int pg = w16_row((tid & 511) >> 4), tl = tid & 15;   // P: the row (0..31) of a K step this lane-group makes

  f32x4 acc[2][NT];
  R16_PH_DECL;

  // ---- block floating point (conv_stack_f16.hip.h): maxima of the feature tile and of the incoming cache
  __shared__ AmaxCell amax_cells[kAmaxCells];
  __shared__ BlockDesc blk[kAmaxMaxBlocks];
  amax_zero<kW16Threads>(amax_cells, kAmaxCells);
  stage_block_table<kW16Threads>(blk, P.blocks, P.nblocks);
  __syncthreads();
  const int nk = P.kpre16 / 32;
  const bool one_trip = nk <= 2 && 8 * TT <= kW16Threads && w16_x_vec_ok(A.x, A.xs_b, P.idim);
  W16XItem xi;
  if (one_trip) {
    xi = w16_load_x<TT, PB>(A.x + int64_t(b) * A.xs_b, T, P.idim, nk);
    amax_publish(amax_cells, w16_x_amax(xi));
  } else {
    amax_publish(amax_cells, amax_span<kW16Threads>(A.x + int64_t(b) * A.xs_b, T * P.idim, 0.f));
  }
  if constexpr (HAS_CACHE)
    amax_publish(amax_cells + 1, amax_span<kW16Threads>(A.in_cache + int64_t(b) * C * Pc, C * Pc, 0.f));

  // ============================ preprocessing: h0 = [ReLU](x Wpre^T + b) ============================
  // every thread stages its share of the features; the M-waves multiply (two o-tiles each) and write h0
  {
    const uint4* ap = reinterpret_cast<const uint4*>(W + P.pre_a16) + size_t(2 * mw) * nk * 128 + lane;
    float sx = 1.f, cpre = 1.f;
    if (one_trip) {
      // 40-d fbank: the features sit in registers; one slab fill, at most two K steps
      __syncthreads();
      sx = pow2_scale(amax_read(amax_cells), &cpre);
      w16_store_x<PB, SPLIT>(xi, sx, slab);
      __syncthreads();
      if (m_role) {
        zero_acc(acc);
#pragma unroll 1
        for (int st = 0; st < nk; ++st) {
          F16Frag a[2];
          load_a16<2>(a, ap + st * 128, nk * 128);
          r16_mfma_step<NT, SPLIT>(acc, a, slab + st * 2 * PB + frag_off, slab + st * 2 * PB + PB + frag_off);
        }
      }
    } else {
      if (m_role) zero_acc(acc);
      for (int k0 = 0; k0 < nk; k0 += 2) {                   // two K steps staged per pass (= the slab)
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
        if (m_role) {
#pragma unroll 1
          for (int st = 0; st < steps; ++st) {
            F16Frag a[2];
            load_a16<2>(a, ap + (k0 + st) * 128, nk * 128);
            r16_mfma_step<NT, SPLIT>(acc, a, slab + st * 2 * PB + frag_off, slab + st * 2 * PB + PB + frag_off);
          }
        }
      }
    }
    if (m_role) {
      cpre *= P.pre_inv_s;                                   // 1 / (feature scale * weight scale)
      float hmax = 0.f;
#pragma unroll
      for (int ow = 0; ow < 2; ++ow) {
        const float4 bias = *reinterpret_cast<const float4*>(W + P.pre_b + o0 + ow * 16);
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
          const int t = tt * 16 + l15;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = fmaf(acc[ow][tt][r], cpre, f4c(bias, r));
            if (P.pre_relu) v = fmaxf(v, 0.f);
            hbuf[(o0 + ow * 16 + r) * SS + t] = v;
            hmax = fmaxf(hmax, fabsf(v));
          }
        }
      }
      amax_publish(amax_cells + 2, hmax);
    }
    __syncthreads();
  }
  R16_PH(0);                                                 // [0] preprocessing
  // ======================================= residual blocks =======================================
  constexpr int NKS = C / 32;                                // K steps per layer
  constexpr int OTS = NKS * 128;                             // uint4 per o-tile
  for (int bi = 0; bi < P.nblocks; ++bi) {
    const BlockDesc bd = blk[bi];
    // ---- operand scale of this block: the depthwise rows are bounded through the maximum of the input tile (published
    //      by the epilogue that wrote it) and of the incoming cache
    float c1;
    const float au = HAS_CACHE ? fmaxf(amax_read(amax_cells + 2 + bi), amax_read(amax_cells + 1))
                               : amax_read(amax_cells + 2 + bi);
    const float sa = pow2_scale(fmaf(bd.dw_alpha, au, bd.dw_beta), &c1);
    c1 *= bd.inv_s1;

    if (m_role) {
      // ------------------------------------------ M-waves ------------------------------------------
      const uint4* ap1 = reinterpret_cast<const uint4*>(W + bd.a1_16) + size_t(2 * mw) * OTS + lane;
      F16Frag a0[2], a1[2];                                  // fragments of the even / odd K steps, each re-requested
      load_a16<2>(a0, ap1, OTS);                             // right after its MFMAs: a whole K step ahead of its use
      load_a16<2>(a1, ap1 + 128, OTS);
      zero_acc(acc);
      R16_PH(1);                                             // [1] block top
      __syncthreads();                                       // K step 0 is in buffer 0
      R16_PH(2);                                             // [2] wait for the first K step
#pragma unroll 1
      for (int ks = 0; ks < NKS; ks += 2) {
        r16_mfma_step<NT, SPLIT>(acc, a0, slab + frag_off, slab + PB + frag_off);
        load_a16<2>(a0, ap1 + min(ks + 2, NKS - 2) * 128, OTS);
        R16_PH(3);                                           // [3] multiply
        __syncthreads();                                     // K step ks + 1 is in buffer 1; buffer 0 is free
        R16_PH(4);                                           // [4] wait inside the K loop
        r16_mfma_step<NT, SPLIT>(acc, a1, slab + 2 * PB + frag_off, slab + 3 * PB + frag_off);
        load_a16<2>(a1, ap1 + min(ks + 3, NKS - 1) * 128, OTS);
        R16_PH(3);
        if (ks + 2 < NKS) __syncthreads();                   // K step ks + 2 is in buffer 0; buffer 1 is free
        R16_PH(4);
      }
      // ---- epilogue: folded bias + ReLU + residual, in place (tcn.py:60: add after the ReLU)
      float4 ebias[2];
      ebias[0] = *reinterpret_cast<const float4*>(W + bd.b1 + o0);
      ebias[1] = *reinterpret_cast<const float4*>(W + bd.b1 + o0 + 16);
      float hmax = 0.f;
#pragma unroll
      for (int ow = 0; ow < 2; ++ow) {
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
          const int t = tt * 16 + l15;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* hp = hbuf + (o0 + ow * 16 + r) * SS + t;
            const float v = fmaxf(fmaf(acc[ow][tt][r], c1, f4c(ebias[ow], r)), 0.f) + *hp;
            *hp = v;
            hmax = fmaxf(hmax, fabsf(v));
          }
        }
      }
      amax_publish(amax_cells + 3 + bi, hmax);           // = the input tile of block bi + 1
      R16_PH(5);                                             // [5] epilogue
      __syncthreads();                                       // the block's output tile is written
      R16_PH(6);                                             // [6] barrier behind the epilogue
    } else {
      // ------------------------------------------ P-waves ------------------------------------------
      const int d = bd.dil, pad = bd.pad;
      // taps + bias of the row this lane-group produces in the coming K step (padded 12-float record)
      float dww[KS + 1];
      auto load_dw = [&](int ks) __attribute__((always_inline)) {
        const float4* src = reinterpret_cast<const float4*>(W + bd.dw_pk + (ks * 32 + pg) * 12);
        const float4 q0 = src[0], q1 = src[1], q2 = src[2];
        dww[0] = q0.x; dww[1] = q0.y; dww[2] = q0.z; dww[3] = q0.w;
        dww[4] = q1.x; dww[5] = q1.y; dww[6] = q1.z; dww[7] = q1.w;
        dww[8] = q2.x;
      };
      const bool slide = slide_ok(d);
      const int fbase = slide ? slide_base(tl, d, NT) : tl;
      // row pg of K step ks -> k-octet pg >> 3, half pg & 7 of buffer ks & 1
      auto produce = [&](int ks) __attribute__((always_inline)) {
        const int c = ks * 32 + pg;
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
          if (A.out_cache) {
            for (int p = tl; p < pad; p += 16) {
              const int src = T + p - pad;   // index into h (negative: still inside the old cache)
              const float hv = hbuf[hoff + max(src, 0)];
              const float g = A.in_cache[gbase + pad + min(src, -1)];
              A.out_cache[gbase + p] = src >= 0 ? hv : g;
            }
          }
        } else if (A.out_cache) {
          // lane tl hands over columns 4 tl .. 4 tl + 3 (pad <= 64: one pass) with ONE 16-byte store where the four exist
          struct __attribute__((packed, aligned(4))) V4 { float v[4]; };
          const int p0 = 4 * tl;
          if (p0 < pad) {
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
        char* const plane = slab + (ks & 1) * 2 * PB;
        _Float16* ph = reinterpret_cast<_Float16*>(plane) + ((pg >> 3) * TT) * 8 + (pg & 7);
        _Float16* pl = reinterpret_cast<_Float16*>(plane + PB) + ((pg >> 3) * TT) * 8 + (pg & 7);
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
      load_dw(0);
      R16_PH(1);                                             // [1] block top
#pragma unroll 1
      for (int ks = 0; ks < NKS; ++ks) {
        produce(ks);
        load_dw(min(ks + 1, NKS - 1));                       // clamped: the last K step re-reads itself
        R16_PH(3);                                           // [3] produce
        __syncthreads();                                     // K step ks is in its buffer; the other buffer is free
        R16_PH(4);                                           // [4] wait inside the K loop
      }
      __syncthreads();                                       // the block's output tile is written (M-waves' epilogue)
      R16_PH(6);                                             // [6] wait for the epilogue
    }
  }

  conv_stack_head<KIND_DS, 256, NT, kW16Threads, SS>(P, A, hbuf, reinterpret_cast<float*>(slab), b);
  R16_PH(7);                                                 // [7] classifier
  R16_PH_DUMP;
}

template <int NT, bool HAS_CACHE, bool SPLIT>
inline int launch_ds256_r16_ntc(const StackParams& P, const CallArgs& A, hipStream_t stream) {
  using G = W16Geom<NT>;
  static DynLdsGrant grant;
  auto kern = ds256_r16_kernel<NT, HAS_CACHE, SPLIT>;
  if (grant_dynamic_lds(kern, int(G::LDS_BYTES), grant)) return -3;
  hipLaunchKernelGGL(kern, dim3(A.B), dim3(kW16Threads), G::LDS_BYTES, stream, P, A);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

template <int NT>
inline int launch_ds256_r16_nt(bool split, const StackParams& P, const CallArgs& A, hipStream_t stream) {
  if (split)
    return A.in_cache ? launch_ds256_r16_ntc<NT, true, true>(P, A, stream)
                      : launch_ds256_r16_ntc<NT, false, true>(P, A, stream);
  return A.in_cache ? launch_ds256_r16_ntc<NT, true, false>(P, A, stream)
                    : launch_ds256_r16_ntc<NT, false, false>(P, A, stream);
}

// split: three fp16 products per MAC on hi/lo operands (F16X3) or one on the hi halves (F16)
int launch_ds256_r16(int nt, bool split, const StackParams& P, const CallArgs& A, hipStream_t stream);

}  // namespace wekws
