// DS-TCN, hidden_dim 64 (examples/hey_snips/s0/conf/ds_tcn.yaml, examples/hi_xiaowen/s0/conf/ds_tcn.yaml; the trained model the
// reference ships for Android is this shape), calls WITHOUT an incoming cache: REGISTER-RESIDENT kernel, one utterance per
// 4-wave workgroup (round 4).  The recipe of mdtc64_g4.hip.h on the arithmetic of tcn.py:101-114 / :35-61:
//
//   block:  u = [zeros | h];  a = ReLU(BN1(dw_k8_dil_d(u)))  -> operand planes | barrier |
//           p = ReLU(BN2(pointwise(a)))  (one 64 x 64 GEMM, two K steps);  h = p + h  (residual AFTER the ReLU, tcn.py:60) |
//           barrier
//
// The generic 8-wave kernel (conv_stack_f16.hip.h) keeps the f32 tile in LDS and walks a 32-channel operand slab through
// nine barriers per block with 512 threads in lockstep: at 4.1 MFLOP per utterance the model is all latency, and it ran at
// 0.07 of the matrix roofline and 0.08 of HBM (round-3 review).  Here wave w owns output channels 16 w .. 16 w + 15 for all
// frames in the accumulator layout of the 1x1 convolution, the residual tile lives in 4 NT registers, LDS holds the 28 KB of
// operand planes, and FOUR independent workgroups share a CU (<= 128 registers): the overlap comes from the hardware scheduler.
//   * frames lane-major with the utterance's END aligned to a lane boundary (mdtc64_g4.hip.h): the cache slices (the last
//     7 d frames of every block's input) are whole lanes plus one lane's tail registers;
//   * depthwise taps (k = 8): g16_dw_rows of ds256_g16.hip.h, one v_fmac_f32_dpp row_shr per tap and output;
//   * the keyword head (per-frame linear, one or two outputs) from the registers.
// Every other call (incoming cache, other heads, wider features): the generic kernel.  Same operand scaling, same products,
// same sums per output as the generic kernel's (block floating point: conv_stack_f16.hip.h); the head's sum order differs.
#pragma once
#include "mdtc64_g4.hip.h"

namespace wekws {

// CTX (round 5, NT >= 4): the call has an incoming cache -- a later chunk of 17 .. 112 frames of a stream, e.g. the Android
// caller's 80 (runtime/android/app/src/main/cpp/wekws.cc:84-97; shorter chunks: the generic kernel).  The blocks' left context
// as in mdtc64_g4.hip.h's context variant: a second register tile cx (lane p = lane p - 16 of the tile, reached by a row_shl
// for the taps that leave the row) and, where NT does not divide T, the frames below zero inside lane 0 from the slice's last
// columns.  Until round 5 these calls ran the generic 8-wave kernel: 2.6 .. 3 x the first chunk's time.
template <int NT, bool SPLIT, bool ALIGNED, bool CTX = false>
__global__ __launch_bounds__(kG4Threads, 4) void ds64_g4_kernel(const StackParams P, const CallArgs A) {
  static_assert(!CTX || NT >= 4, "the context tile is one 16-lane row (paddings up to 56 frames)");
  constexpr int C = 64, TT = 16 * NT;
  constexpr int MPB = Plane<C, TT>::BYTES;                   // one hi (or lo) plane of the 64-channel operand
  constexpr int XI = (3 * 4 * TT + kG4Threads - 1) / kG4Threads;   // feature items per thread (<= 3 K steps)
  extern __shared__ __attribute__((aligned(16))) float d4_lds[];
  char* const planes = reinterpret_cast<char*>(d4_lds);      // [hi | lo][k-octet 0..7][column][8 halves]

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
  char* const pst = planes + ((o0 >> 3) * TT + l15) * 16 + (o0 & 7) * 2;

  f32x4 acc[NT], hv[NT];
  f32x4 cx[CTX ? NT : 1];                                    // CTX: the current block's left context (registers shared with acc)
  __shared__ AmaxCell amax_cells[kAmaxCells];
  __shared__ BlockDesc blk[kAmaxMaxBlocks];
  __shared__ __attribute__((aligned(16))) float taps[2][C * 12];   // taps + bias records of the current / next block
  amax_zero<kG4Threads>(amax_cells, kAmaxCells);
  stage_block_table<kG4Threads>(blk, P.blocks, P.nblocks);
  // taps of a block: 3 KB copied from the weight image straight into LDS by waves 0 .. 2 (global_load_lds); nobody waits for
  // the copy explicitly -- the issuing waves consume weight fragments they requested AFTER it before the barrier in front of
  // the depthwise phase that reads the taps
  auto stage_taps = [&](int bi, int ln) __attribute__((always_inline)) {
    if (wave < 3)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(W + __builtin_amdgcn_readfirstlane(blk[bi].dw_pk) + (wave * 64 + ln) * 4),
                                       (__attribute__((address_space(3))) void*)(&taps[bi & 1][0] + wave * 256), 16, 0, 0);
  };

  // ---- features (as mdtc64_g4): every thread owns up to XI items = 8 consecutive features of a frame
  const int nk = P.kpre16 / 32;
  const int nitems = nk * 4 * TT;
  W16XItem xi[XI];
  float xmax = 0.f;
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int e = tid + i * kG4Threads;
    const int n = e % TT, q = e / TT;
    const int f = NT * (n & 15) + (n >> 4) - off;
    const int oct = q & 3, st = q >> 2;
    const int kf = st * 32 + oct * 8;
    const bool has = e < nitems;
    xi[i].dst = has ? ((st & 1) * 4 + oct) * TT * 16 + n * 16 : -1;
    w16_fetch_x(xi[i], A.x + int64_t(b) * A.xs_b + int64_t(f) * P.idim + kf, A.x, has && f >= 0 && f < T && kf < P.idim);
    xmax = amax_merge(xmax, w16_x_amax_bits(xi[i]));          // (bit patterns: a NaN / Inf stays on top)
  }
  __syncthreads();                                           // cells zeroed, table staged
  stage_taps(0, lane);
  amax_publish(amax_cells, xmax);
  if constexpr (CTX)     // the depthwise rows are bounded through max(tile, incoming cache), like conv_stack_f16.hip.h
    amax_publish(amax_cells + 1, amax_span_bits<kG4Threads>(A.in_cache + int64_t(b) * C * Pc, C * Pc, 0.f));

  // ============================ preprocessing: h0 = [ReLU](x Wpre^T + b) ============================
  {
    const uint4* ap = reinterpret_cast<const uint4*>(W + P.pre_a16) + size_t(wave) * nk * 128 + lane;
    const float4 bias = *reinterpret_cast<const float4*>(W + P.pre_b + o0);
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) acc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();                                         // the feature maximum is published
#ifdef WEKWS_D64_EARLY                                        // (diagnosis builds only: the placement that used to bring the rare wrong posteriors back)
    if (amax_inputs_bad(amax_cells)) { nf_repair_call(A, blockIdx.x); return; }
#endif
    float cpre;
    const float sx = pow2_scale(amax_read(amax_cells), &cpre);
    for (int k0 = 0; k0 < nk; k0 += 2) {
      if (k0) __syncthreads();
#pragma unroll
      for (int i = 0; i < XI; ++i) {
        const int st = (tid + i * kG4Threads) / (4 * TT);
        if (st >= k0 && st < k0 + 2) w16_put_x<SPLIT>(xi[i], sx, planes, MPB);
      }
      __syncthreads();
      for (int st = k0; st < min(k0 + 2, nk); ++st) {
        F16Frag a;
        a.h = __builtin_bit_cast(f16x8, ap[st * 128]);
        a.l = __builtin_bit_cast(f16x8, ap[st * 128 + 64]);
        const char* bh = planes + (st & 1) * 4 * TT * 16 + frag_off;
        g16_mfma_step<NT, SPLIT>(acc, a, bh, bh + MPB);
      }
    }
    cpre *= P.pre_inv_s;
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
    return reinterpret_cast<const uint4*>(W + __builtin_amdgcn_readfirstlane(a16)) + size_t(wave) * 256;
  };
  auto load_frag = [&](F16Frag& a, const uint4* ap, int ln) __attribute__((always_inline)) {
    a.h = __builtin_bit_cast(f16x8, ap[ln]);
    a.l = __builtin_bit_cast(f16x8, ap[ln + 64]);
  };
  float yp[NT][2];
#pragma unroll
  for (int tt = 0; tt < NT; ++tt) yp[tt][0] = yp[tt][1] = 0.f;
  F16Frag g1a;                                               // first K step of the block's GEMM: requested a block ahead
  __syncthreads();                                           // (A) maximum published, planes free, table / taps visible
  load_frag(g1a, frag_ptr(blk[0].a1_16), lane);
  for (int bi = 0; bi < P.nblocks; ++bi) {
    int o0b = o0, laneb = lane;                              // (opaque per block: see mdtc64_g4.hip.h)
    asm volatile("" : "+v"(o0b), "+v"(laneb));

    BlockDesc bd = blk[bi];
    bd.pad = __builtin_amdgcn_readfirstlane(bd.pad);
    bd.dil = __builtin_amdgcn_readfirstlane(bd.dil);
    bd.cache_off = __builtin_amdgcn_readfirstlane(bd.cache_off);
    bd.b1 = __builtin_amdgcn_readfirstlane(bd.b1);
    const int pad = bd.pad;
    if (bi + 1 < P.nblocks) stage_taps(bi + 1, laneb);
    auto uni = [](float v) __attribute__((always_inline)) {
      return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v)));
    };
    float c1v;
    const float au = CTX ? fmaxf(amax_read(amax_cells + 2 + bi), amax_read(amax_cells + 1)) : amax_read(amax_cells + 2 + bi);
    const float sa = uni(pow2_scale(fmaf(bd.dw_alpha, au, bd.dw_beta), &c1v));
    const float c1 = uni(c1v * bd.inv_s1);

    // ---- CTX: the block's left context from its slice of the incoming cache (position q = NT lane + tt holds frame q - off;
    //      frame f < 0 is slice column pad + f): lane p of cx = lane p - 16 of the tile, lane 0's registers tt < off
    if constexpr (CTX) {
      const float* const ic = A.in_cache + int64_t(b) * C * Pc + bd.cache_off;
      const int c0 = pad - off + NT * (l15 - 16);
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

    // ---- the block's streaming-cache slice = the last `pad` frames of its input tile [zeros | h], from the registers
    if (A.out_cache) {
      float* const ocb = A.out_cache + int64_t(b) * C * Pc + bd.cache_off;
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

    // ---- depthwise dilated conv + folded BN + ReLU (tcn.py:102-108), scaled, split, to the operand planes
    {
      const float* taps_o0 = &taps[bi & 1][0] + o0 * 12;
      switch (bd.dil) {                                      // (the host admits this kernel for dilations 1 / 2 / 4 / 8 only)
        case 1: if constexpr (CTX) g16_dw_rows<1, NT, SPLIT, true>(hv, cx, taps_o0, sa, pst, MPB); else g16_dw_rows<1, NT, SPLIT>(hv, taps_o0, sa, pst, MPB); break;
        case 2: if constexpr (CTX) g16_dw_rows<2, NT, SPLIT, true>(hv, cx, taps_o0, sa, pst, MPB); else g16_dw_rows<2, NT, SPLIT>(hv, taps_o0, sa, pst, MPB); break;
        case 4: if constexpr (CTX) g16_dw_rows<4, NT, SPLIT, true>(hv, cx, taps_o0, sa, pst, MPB); else g16_dw_rows<4, NT, SPLIT>(hv, taps_o0, sa, pst, MPB); break;
        case 8: if constexpr (CTX) g16_dw_rows<8, NT, SPLIT, true>(hv, cx, taps_o0, sa, pst, MPB); else g16_dw_rows<8, NT, SPLIT>(hv, taps_o0, sa, pst, MPB); break;
        default: break;
      }
    }
    __syncthreads();                                         // (B1) the depthwise planes are written

    // ---- pointwise conv over the full K = 64, + BN + ReLU, + residual (tcn.py:109-112, :60), registers only
    const float4 bias1 = *reinterpret_cast<const float4*>(W + bd.b1 + o0b);
    F16Frag gb;                                              // second K step: arrives behind the first one's MFMAs
    load_frag(gb, frag_ptr(bd.a1_16) + 128, laneb);
    g16_mfma_step<NT, SPLIT, true>(acc, g1a, planes + frag_off, planes + MPB + frag_off);
    g16_mfma_step<NT, SPLIT>(acc, gb, planes + 4 * TT * 16 + frag_off, planes + 4 * TT * 16 + MPB + frag_off);
    if (bi + 1 < P.nblocks) load_frag(g1a, frag_ptr(blk[bi + 1].a1_16), laneb);   // next block's first K step
    float hmax = 0.f;
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = fmaxf(fmaf(acc[tt][r], c1, f4c(bias1, r)), 0.f) + hv[tt][r];
        if (l15 == 0 && tt < off) v = 0.f;                   // frames below zero stay the (zero) left context
        hv[tt][r] = v;
        hmax = fmaxf(hmax, fabsf(v));
      }
    }
    amax_publish(amax_cells + 3 + bi, hmax);                 // = the input tile of block bi + 1
    __syncthreads();                                         // (B2) maximum published, planes free
  }

  // ---- the backbone's output enters the head: this lane's partial sums over its four channels, as packed FMAs with their operand
  //      selects spelled out (head_fma4, pk_safe.hip.h).  History: rounds 4-6 saw ~2 % of the utterances of a large batch come out
  //      1e-2 off in ONE output whenever several workgroups shared the CU, and could only move the effect around (where the sums
  //      were computed, where the non-finite check sat, the register budget).  Round 6 found it: the compiler had lowered this chain
  //      to v_pk_fma_f32 ... op_sel:[0,1,0], and gfx950 returns lanes 48..63 of that instruction's low result without the product
  //      while another workgroup's waves issue MFMAs on the SIMD (tools/probe/d64_dump.py, pk_opsel_probe4.hip).
  {
    const int o0b = o0;
    {
      const int K = P.odim;
      const float4 w0 = *reinterpret_cast<const float4*>(W + P.head_w + o0b);
      const float4 w1 = *reinterpret_cast<const float4*>(W + P.head_w + (K > 1 ? C : 0) + o0b);
      const HeadPairs hw(w0, w1);                            // (packed, operand selects spelled out: pk_safe.hip.h)
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        pk_f32x2 pp{0.f, 0.f};
        head_fma4(pp, hw.x, hw.y, hw.z, hw.w, hv[tt]);
        yp[tt][0] = pp.x; yp[tt][1] = pp.y;
      }
#ifdef WEKWS_D64_DUMP                                         // (diagnosis builds only: the head's inputs and partial sums over the returned cache)
      if (A.out_cache) {
        float* dbg = A.out_cache + int64_t(b) * C * Pc + threadIdx.x * 16;
        dbg[0] = yp[0][0]; dbg[1] = yp[0][1]; dbg[2] = hv[0][0]; dbg[3] = hv[0][1]; dbg[4] = hv[0][2]; dbg[5] = hv[0][3];
        dbg[6] = w0.x; dbg[7] = w0.y; dbg[8] = w0.z; dbg[9] = w0.w;
        if constexpr (NT > 1) { dbg[10] = yp[1][0]; dbg[11] = yp[1][1]; }
      }
#endif
    }
  }
  // ---- keyword head (per-frame linear, one or two outputs; classifier.py:63-67): the 16 partial sums per output (4 waves x 4
  //      channel groups) meet in LDS
  {
    const int K = P.odim;
    constexpr int PS = 32 * NT + 16;                          // floats per partial row: 16 lanes x (NT frames x 2 outputs), padded
    float* const part = d4_lds;
    int th = threadIdx.x;
    asm volatile("" : "+v"(th));
    float* dst = part + (th >> 4) * PS + 2 * NT * (th & 15);   // row = wave * 4 + lq
#pragma unroll
    for (int tt = 0; tt < NT; ++tt)
      *reinterpret_cast<float2*>(dst + 2 * tt) = float2{yp[tt][0], yp[tt][1]};   // column NT l15 + tt = frame NT l15 + tt - off
    __syncthreads();
    const int t = (th >> 1) - off, k = th & 1;               // thread = (column, output)
    if (th < 2 * TT && t >= 0 && t < T && k < K) {
      float v = W[P.head_b + k];
#ifdef WEKWS_D64_DUMP
      if (A.out_cache && th < 28) {                            // columns 0 .. 13 (lane 0's frames), both outputs: the 16 partial rows each
        float* dbg = A.out_cache + int64_t(b) * C * Pc + 4096 + th * 16;
        for (int i = 0; i < 16; ++i) dbg[i] = part[i * PS + th];
      }
#endif
#pragma unroll
      for (int i = 0; i < 16; ++i) v += part[i * PS + th];
      if (P.sigmoid) v = sigmoidf_(v);
      A.y[int64_t(b) * A.ys_b + int64_t(t) * K + k] = v;
    }
  }
  // A NaN / Inf feature or cache element (cells [0] / [1], untouched since the top, at or above 0x7f800000): what the kernel
  // computed for this utterance is garbage (and touched nothing else); the utterance is re-computed with the reference's arithmetic
  // HERE, where nothing is live.  (An early exit near the top is equally right since the head's packed FMAs are written out --
  // -DWEKWS_D64_EARLY builds that placement, the one that used to bring the rare wrong posteriors back: 0 differing utterances in
  // tools/probe/d64_diag.py now -- but saves nothing for finite inputs.)
  if (amax_inputs_bad(amax_cells)) {                         // (workgroup-uniform, scalar)
    __syncthreads();
    nf_repair_call(A, blockIdx.x);
  }
}

// Usable when (the host checks the model side: DS-TCN, hidden_dim 64, kernel size 8, dilations 1 / 2 / 4 / 8): no incoming
// cache, features of <= 96 dims in whole aligned 8-float items, a per-frame linear head with one or two outputs.  Returns -4
// otherwise (the caller then runs the generic kernel).
int launch_ds64_g4(int nt, bool split, const StackParams& P, const CallArgs& A, hipStream_t stream);

}  // namespace wekws
