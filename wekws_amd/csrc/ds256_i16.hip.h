// DS-TCN, hidden_dim 256, headline shape -- INTERLEAVED 16-wave kernel (round 3).  Same arithmetic, LDS layout, weight
// image and results (bit for bit) as ds256_w16.hip.h; every wave now multiplies K step k WHILE it produces its share of
// K step k + 1, in one instruction stream.
//
// Measurements behind the structure (tools/probe/overlap_probe.hip, per-phase stamps of ds256_w16 / ds256_r16):
//   * ds256_w16 alternates [all 16 waves produce 64 rows] barrier [all multiply] barrier.  Per utterance (132 k cycles) the
//     LDS array is busy 69 k, the matrix pipe needs 31 k (a 16x16x32 fp16 MFMA issues every 10.4 cycles from four waves
//     per SIMD) and the vector units 28 k: every unit idles more than half of the time -- the phases are serial and
//     inside a phase all waves queue at the same unit (14 LDS reads, ~100 vector instructions, 14 LDS stores, x 16).
//   * one wave issues at most one vector instruction per ~8 cycles, whatever else the SIMD does; the vector rate of a SIMD
//     scales with the NUMBER of waves issuing vector work (5.5 / 3.0 / 1.9 cycles per instruction at 1 / 2 / 4 waves).
//     That is why fixed roles (ds256_r16: 8 multiplying + 8 producing waves) gained only 3 .. 6 %: a producer pass on two
//     waves per SIMD takes 2.2 k cycles against 1.3 k for the K step it runs beside.
//   * inside ONE wave's stream two vector instructions per 16x16x32 MFMA are free and further ones cost ~2.4 cycles each
//     (the same as without the MFMA) -- when the instructions are interleaved one by one.
// Hence: all sixteen waves keep both jobs, K is consumed in 32-channel steps through the slab's two K-step buffers as a
// ring (ONE barrier per K step), and the producer is cut so that every wave has the same share of every K step:
//   a wave makes TWO channel rows per K step; lanes 0..31 (two 16-lane groups, one per row) compute the first four
//   outputs of each lane's seven-frame run, lanes 32..63 the last three (same rows, window shifted by 4 dilations).
#pragma once
#include "ds256_w16.hip.h"

namespace wekws {

#ifdef WEKWS_I16_STAMPS
#define I16_PH_DECL long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long tlast = clock64()
#define I16_PH(id) do { long long now_ = clock64(); tph[id] += now_ - tlast; tlast = now_; } while (0)
#define I16_PH_DUMP                                                                                    \
  do {                                                                                                 \
    __syncthreads();                                                                                   \
    if (b == 0 && A.out_cache && lane == 0 && (wave == 0 || wave == 9))                                \
      for (int i = 0; i < 8; ++i) A.out_cache[(wave ? 8 : 0) + i] = float(tph[i]);                     \
  } while (0)
#else
#define I16_PH_DECL
#define I16_PH(id)
#define I16_PH_DUMP
#endif

template <int NT, bool HAS_CACHE, bool SPLIT>
__global__ __launch_bounds__(kW16Threads) void ds256_i16_kernel(const StackParams P, const CallArgs A) {
  using G = W16Geom<NT>;
  constexpr int C = G::C, SS = G::SS, TT = G::TT, PB = G::PB, KS = 8;
  constexpr int MA = (NT + 1) / 2;                           // outputs per lane: lanes 0..31 make MA, lanes 32..63 NT - MA
  constexpr int NW = MA + KS - 1;                            // window slots
  extern __shared__ __attribute__((aligned(16))) float w16_lds[];
  char* const slab = reinterpret_cast<char*>(w16_lds);       // two K-step buffers: [hi | lo] [hi | lo]
  float* const hbuf = w16_lds + G::SLAB / 4;                 // [256][SS] f32 resident activations

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform (SGPR)
  const int l15 = lane & 15, lq = lane >> 4;
  const int T = A.T;
  const int b = blockIdx.x;                                  // one utterance per workgroup
  const float* __restrict__ W = P.w;
  const int Pc = P.cache_len;
  const int o0 = wave * 16 + lq * 4;                         // this lane's 4 output channels (o-tile = wave)
  const int frag_off = (lq * TT + l15) * 16;
  // producer: the wave makes rows (q 8 + i, q 8 + i + 4) of a K step, q = wave >> 2, i = wave & 3; 16-lane group parity
  // picks the row (the two groups of a 32-lane half four rows apart: conflict-free, ds256_w16.hip.h), the half the outputs
  const int pr = (wave >> 2) * 8 + (wave & 3) + ((lane >> 4) & 1) * 4;
  const int tl = lane & 15;
  const bool upper = lane >= 32;
  const int mo = upper ? MA : 0;                             // this lane's first output of the NT-run

  // Within a K step the multiply and the producer arithmetic are independent: two of a SIMD's four waves (waves w, w+4,
  // w+8, w+12 share a SIMD) run the vector part first, the other two the matrix part, so that the SIMD always has both
  // kinds of work to issue instead of four waves queueing at the same unit.
#ifndef WEKWS_I16_ORDER
#define WEKWS_I16_ORDER 1
#endif
  const bool vfirst = WEKWS_I16_ORDER == 1 ? ((wave >> 2) & 1) != 0 : WEKWS_I16_ORDER == 2 ? ((wave >> 3) & 1) != 0
                    : WEKWS_I16_ORDER == 3 ? (wave & 1) != 0 : false;

  f32x4 acc[1][NT];
  I16_PH_DECL;

  // ---- block floating point (conv_stack_f16.hip.h): maxima of the feature tile and of the incoming cache
  __shared__ AmaxCell amax_cells[kAmaxCells];
  __shared__ BlockDesc blk[kAmaxMaxBlocks];
  // depthwise taps + bias of the CURRENT block, [256][12] floats (the 12-float records of BlockDesc::dw_pk): staged for block
  // bi + 1 behind block bi's last production (block 0: during the preprocessing), read back as three 16-byte broadcasts
  // where a row is computed -- nine registers per lane are not carried across the matrix part of a K step
  __shared__ __attribute__((aligned(16))) float taps[C * 12];
  auto stage_taps = [&](const BlockDesc& nb) __attribute__((always_inline)) {
    if (tid < C * 3) reinterpret_cast<float4*>(taps)[tid] = reinterpret_cast<const float4*>(W + nb.dw_pk)[tid];
  };
  amax_zero<kW16Threads>(amax_cells, kAmaxCells);
  stage_block_table<kW16Threads>(blk, P.blocks, P.nblocks);
  __syncthreads();
  stage_taps(blk[0]);
  const int nk = P.kpre16 / 32;
  // 40-d fbank: the features pass through registers once
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
  I16_PH(0);                                                 // [0] preprocessing
  // ======================================= residual blocks =======================================
  constexpr int NKS = C / 32;                                // K steps per layer
  constexpr int OTS = NKS * 128;                             // uint4 per o-tile
  static_assert(NKS % 2 == 0, "K steps are consumed in pairs");
  for (int bi = 0; bi < P.nblocks; ++bi) {
    const BlockDesc bd = blk[bi];
    const int d = bd.dil, pad = bd.pad;                      // (the host admits this kernel only for dilations 2^j <= 16)
    const uint4* ap1 = reinterpret_cast<const uint4*>(W + bd.a1_16) + size_t(wave) * OTS + lane;
    // ---- operand scale of this block: the depthwise rows are bounded through the maximum of the input tile (published
    //      by the epilogue that wrote it) and of the incoming cache
    float c1;
    const float au = HAS_CACHE ? fmaxf(amax_read(amax_cells + 2 + bi), amax_read(amax_cells + 1))
                               : amax_read(amax_cells + 2 + bi);
    const float sa = pow2_scale(fmaf(bd.dw_alpha, au, bd.dw_beta), &c1);
    c1 *= bd.inv_s1;

    // this lane's first output frame: the lane's NT-run at stride d (slide_base) entered at output mo
    const int fb = slide_base(tl, d, NT) + mo * d;

    // ---- producer, in two halves so that the MFMAs of a K step can sit between the window reads and the arithmetic:
    //      row pr of K step ks = depthwise dilated conv + folded BN + ReLU of channel ks*32 + pr (tcn.py:102-109), split
    //      to fp16 hi/lo into buffer ks & 1; lanes 0..31 also hand the channel's streaming cache over.
    float v[NW];
    auto window = [&](int ks) __attribute__((always_inline)) {
      const int c = ks * 32 + pr;
      const int hoff = c * SS;
      const int64_t gbase = (int64_t(b) * C + c) * Pc + bd.cache_off;
#pragma unroll
      for (int q = 0; q < NW; ++q) {
        const int ix = fb + (q - (KS - 1)) * d;
        // slots q >= KS-1 sit at frame fb + (q-KS+1)*d >= 0: never left context, plain read, no select
        float fv = hbuf[hoff + ix];
        if (q < KS - 1) {
          if constexpr (HAS_CACHE) {
            const float fg = A.in_cache[gbase + pad + min(ix, -1)];
            fv = ix >= 0 ? fv : fg;
          } else {
            fv = ix >= 0 ? fv : 0.f;
          }
        }
        v[q] = fv;
      }
      // the row's new streaming-cache slice = last `pad` frames of [old slice | h]
      if (A.out_cache && !upper) {
        if constexpr (HAS_CACHE) {
          for (int p = tl; p < pad; p += 16) {
            const int src = T + p - pad;   // index into h (negative: still inside the old cache)
            const float hv = hbuf[hoff + max(src, 0)];
            const float g = A.in_cache[gbase + pad + min(src, -1)];
            A.out_cache[gbase + p] = src >= 0 ? hv : g;
          }
        } else {
          // lane tl hands over columns 4 tl .. 4 tl + 3 (pad <= 64: one pass) with ONE 16-byte store where the four exist
          struct __attribute__((packed, aligned(4))) V4 { float v[4]; };
          const int p0 = 4 * tl;
          if (p0 < pad) {
            float cv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int src = T + p0 + k - pad;   // index into h (negative: left of the first frame -> zero context)
              const float hv = hbuf[hoff + max(src, 0)];
              cv[k] = src >= 0 ? hv : 0.f;
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
      }
    };
    auto outputs = [&](int ks) __attribute__((always_inline)) {
      // taps + bias of the row (padded 12-float record): three LDS broadcasts
      float dww[KS + 1];
      {
        const float4* src = reinterpret_cast<const float4*>(taps + (ks * 32 + pr) * 12);
        const float4 q0 = src[0], q1 = src[1], q2 = src[2];
        dww[0] = q0.x; dww[1] = q0.y; dww[2] = q0.z; dww[3] = q0.w;
        dww[4] = q1.x; dww[5] = q1.y; dww[6] = q1.z; dww[7] = q1.w;
        dww[8] = q2.x;
      }
      char* const plane = slab + (ks & 1) * 2 * PB;
      _Float16* ph = reinterpret_cast<_Float16*>(plane) + ((pr >> 3) * TT) * 8 + (pr & 7);
      _Float16* pl = reinterpret_cast<_Float16*>(plane + PB) + ((pr >> 3) * TT) * 8 + (pr & 7);
#pragma unroll
      for (int m = 0; m < MA; ++m) {
        float o = dww[KS];
#pragma unroll
        for (int j = 0; j < KS; ++j) o = fmaf(dww[j], v[m + j], o);
        o = fmaxf(o, 0.f);
        const int t = fb + m * d;
        _Float16 h, l;
        split16s(o, sa, h, l);
        if (m + MA < NT || !upper) {                         // (the upper lanes' run is NT - MA long)
          ph[t * 8] = h;
          if constexpr (SPLIT) pl[t * 8] = l;
        }
      }
    };

    // ONE weight fragment in flight: it is re-requested for the next K step right after its MFMAs have been issued and
    // has that step's producer arithmetic and the barrier to arrive (ds256_w16 needed two: it multiplies two K steps back
    // to back)
    F16Frag a0[1];
    load_a16<1>(a0, ap1, 0);
    zero_acc(acc);
    I16_PH(1);                                               // [1] block top
    // Software pipeline over K steps -1 .. NKS-1: step ks multiplies K step ks (buffer ks & 1) and produces K step ks + 1
    // (buffer (ks + 1) & 1); the first step only produces, the last only multiplies.  One barrier per step.
#pragma unroll 1
    for (int ks = -1; ks < NKS; ++ks) {
      const int nx = ks + 1;                                 // the K step produced
      const bool prod = nx < NKS;
      if (prod) window(nx);
      const char* bsrc = slab + (ks & 1) * 2 * PB + frag_off;
      if (vfirst) {
        if (prod) {
          outputs(nx);
        }
        if (ks >= 0) {
          mfma16_step_nb<NT, SPLIT>(acc[0], a0[0], bsrc, bsrc + PB);
          load_a16<1>(a0, ap1 + min(nx, NKS - 1) * 128, 0);
        }
      } else {
        if (ks >= 0) {
          mfma16_step_nb<NT, SPLIT>(acc[0], a0[0], bsrc, bsrc + PB);
          load_a16<1>(a0, ap1 + min(nx, NKS - 1) * 128, 0);
        }
        if (prod) {
          outputs(nx);
        }
      }
      if (prod) {
        I16_PH(4);                                           // [4] pipelined K steps
        __syncthreads();
        I16_PH(3);                                           // [3] barrier waits
      }
    }
    I16_PH(5);                                               // [5] last K step (nothing to overlap with)

    // ---- epilogue: folded bias + ReLU + residual, in place (tcn.py:60: add after the ReLU)
    const float4 ebias = *reinterpret_cast<const float4*>(W + bd.b1 + o0);
    float hmax = 0.f;
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
      const int t = tt * 16 + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float* hp = hbuf + (o0 + r) * SS + t;
        const float vv = fmaxf(fmaf(acc[0][tt][r], c1, f4c(ebias, r)), 0.f) + *hp;
        *hp = vv;
        hmax = fmaxf(hmax, fabsf(vv));
      }
    }
    amax_publish(amax_cells + 3 + bi, hmax);             // = the input tile of block bi + 1
    if (bi + 1 < P.nblocks) stage_taps(blk[bi + 1]);       // (this block's taps were last read a barrier ago)
    I16_PH(6);                                               // [6] epilogue
    __syncthreads();
    I16_PH(3);
  }

  conv_stack_head<KIND_DS, 256, NT, kW16Threads, SS>(P, A, hbuf, reinterpret_cast<float*>(slab), b);
  I16_PH(7);                                                 // [7] classifier
  I16_PH_DUMP;
}

template <int NT, bool HAS_CACHE, bool SPLIT>
inline int launch_ds256_i16_ntc(const StackParams& P, const CallArgs& A, hipStream_t stream) {
  using G = W16Geom<NT>;
  static DynLdsGrant grant;
  auto kern = ds256_i16_kernel<NT, HAS_CACHE, SPLIT>;
  // (+64 B: the upper lanes' unused fourth output reads up to three floats past a row; the last row's land here)
  constexpr size_t LDS = G::LDS_BYTES + 64;
  if (grant_dynamic_lds(kern, int(LDS), grant)) return -3;
  hipLaunchKernelGGL(kern, dim3(A.B), dim3(kW16Threads), LDS, stream, P, A);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

template <int NT>
inline int launch_ds256_i16_nt(bool split, const StackParams& P, const CallArgs& A, hipStream_t stream) {
  if (split)
    return A.in_cache ? launch_ds256_i16_ntc<NT, true, true>(P, A, stream)
                      : launch_ds256_i16_ntc<NT, false, true>(P, A, stream);
  return A.in_cache ? launch_ds256_i16_ntc<NT, true, false>(P, A, stream)
                    : launch_ds256_i16_ntc<NT, false, false>(P, A, stream);
}

// split: three fp16 products per MAC on hi/lo operands (F16X3) or one on the hi halves (F16).
// Requires every block's dilation to be a power of two <= 16 (the host checks: wekws_hip.hip).
int launch_ds256_i16(int nt, bool split, const StackParams& P, const CallArgs& A, hipStream_t stream);

}  // namespace wekws
