// MDTC h64, streaming step (chunks of <= 16 frames, both streams' caches resident in LDS): the latency kernel.
//
// Same arithmetic, operand layout and results as the batch kernel mdtc64_w16_kernel<1, ...> fed the same chunks
// (bit-identical: the tests compare them), different division of labour.  Per-phase clock64 stamps of the previous
// streaming kernel -- the batch kernel's structure with the caches in LDS -- on a 10-frame chunk (B = 1: 88 k cycles per step) put 56 % of the step into "top of block + depthwise producer": 16 waves x
// ~200 vector instructions of which five are the arithmetic -- every lane made ONE output, so every output paid its own
// tap addressing, cache / tile selects and scale bookkeeping, and at four waves per SIMD and 4 cycles per wave64 vector
// instruction that is ~2.9 k cycles of pure issue per block, 17 blocks per step.  A streaming step has so little work
// per block that the cost is the number of wave-instructions, not lanes; so here the waves take ROLES:
//
//   waves 8..15  producers: wave = (stream, 16 channels), lane = (channel, quarter g): FOUR outputs per lane -- a run at
//                stride = dilation, the five taps sliding over 8 register-resident inputs (d = 8: two runs of two) --
//                with the dilation a compile-time constant per switch arm, so tap offsets are instruction immediates;
//                the same lanes shift the block's cache slice in place;
//   waves 0..7   matrix waves: wave = (stream, o-tile); both GEMMs' weight fragments and the biases are requested one
//                phase ahead; mid tile into its own planes (no barrier between GEMM 1's reads and the mid writes).
//   Three barriers per block.  Each role derives only the scales it needs from the LDS maxima cells.
//
// Usable when every block has dilation 1 / 2 / 4 / 8 (the reference recipes: stack_size 4, mdtc.py:181-198), kernel size
// 5, <= 128 input features in whole 16-byte aligned octets; anything else runs the batch kernel on the chunk (host decides).
#pragma once
#include "mdtc64_w16.hip.h"

namespace wekws {

struct MdtcStreamGeom {
  static constexpr int C = 64, U = 2, TT = 16, SS = 20, KS = 5;
  static constexpr int MPB = Plane<C, TT>::BYTES;            // one hi (or lo) plane of a 64-channel operand: 2 KB
  static constexpr int UB = 2 * MPB;                         // one stream's planes
  static constexpr int PLANES = 2 * U * UB;                  // [depthwise planes of both streams | mid planes]
  static constexpr int H_FLOATS = U * C * SS;
  static constexpr size_t lds_bytes(int cache_len) {
    return size_t(PLANES) + size_t(H_FLOATS) * 4 + size_t(U) * C * cache_len * 4;
  }
};

template <bool SPLIT>
__global__ __launch_bounds__(kW16Threads) void mdtc64_stream_kernel(const StackParams P, const CallArgs A) {
  using G = MdtcStreamGeom;
  constexpr int C = G::C, U = G::U, TT = G::TT, SS = G::SS, KS = G::KS, MPB = G::MPB, UB = G::UB;
  extern __shared__ __attribute__((aligned(16))) float mdtcs_lds[];
  char* const slab = reinterpret_cast<char*>(mdtcs_lds);     // [dw | mid][stream][hi | lo][8 oct][TT][8 halves]
  float* const hbuf = mdtcs_lds + G::PLANES / 4;             // [stream][64][SS] f32 resident activations
  float* const cch = hbuf + G::H_FLOATS;                     // [stream][64][Pc] the streams' caches

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  const int T = A.T;
  const int b0 = blockIdx.x * U;
  const int nu = min(U, A.B - b0);                           // streams this workgroup really has
  const float* __restrict__ W = P.w;
  const int Pc = P.cache_len;
  const bool mx_wave = wave < 8;                             // role (wave-uniform)
  // matrix role
  const int wu = (wave >> 2) & 1, ot = wave & 3;
  const bool active = mx_wave && wu < nu;
  const int o0 = ot * 16 + lq * 4;                           // this lane's 4 output channels
  char* const dw_u = slab + wu * UB;
  char* const mid_u = slab + U * UB + wu * UB;
  float* const h_w = hbuf + wu * C * SS;
  const int frag_off = (lq * TT + l15) * 16;                 // B item of the tile, K step 0
  // producer role: stream pu, channel pc, quarter g
  const int pu = (wave >> 2) & 1;
  const int pc = (tid & 255) >> 2, g = tid & 3;
  const bool producing = !mx_wave && pu < nu;

  __shared__ AmaxCell amax_cells[U * kAmaxCells];
  __shared__ BlockDesc blk[kAmaxMaxBlocks];
  amax_zero<kW16Threads>(amax_cells, U * kAmaxCells);
  stage_block_table<kW16Threads>(blk, P.blocks, P.nblocks);
  __syncthreads();

  // ---- one trip to memory for everything the step reads: the caches (8 x 16 bytes per thread in flight), the features
  //      (one 8-feature item per thread: item = (stream, K step, k-octet, frame); a wave's items belong to one stream),
  //      the preprocessing fragments and the first block's constants
  const int nk = P.kpre16 / 32;                              // K steps of the input (40-d: 2, 80-d MFCC: 3; host: <= 4)
  const int n4 = (C * Pc) >> 2, tot = U * n4;                // C * Pc % 4 == 0 (host checks)
  const f32x4* csrc = reinterpret_cast<const f32x4*>(A.in_cache + int64_t(b0) * C * Pc);
  constexpr int kInFlight = 8;
  float cm[U] = {0.f, 0.f};
  W16XItem xi;
  xi.dst = -1;
  xi.ok = true;                                              // (an item without features is loaded as zeros)
  F16Frag apre[4];
  struct TapConst { float4 q0, q1; };                        // producer: taps + bias of channel pc (8-float record)
  struct BiasConst { float4 b1, b2; };                       // matrix waves: the folded BN biases of this lane's channels
  auto load_taps = [&](int bi) __attribute__((always_inline)) {
    const float4* src = reinterpret_cast<const float4*>(W + blk[bi].dw_pk + pc * 8);
    return TapConst{src[0], src[1]};
  };
  auto load_bias = [&](int bi) __attribute__((always_inline)) {
    return BiasConst{*reinterpret_cast<const float4*>(W + blk[bi].b1 + o0), *reinterpret_cast<const float4*>(W + blk[bi].b2 + o0)};
  };
  TapConst tnext = {};
  BiasConst bnext = {};
  f32x4 q[kInFlight];
  auto cache_request = [&](int e0) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < kInFlight; ++k) {
      const int e = e0 + k * kW16Threads;                    // streamed once: non-temporal, the weights stay in L2
      q[k] = (A.in_cache && e < nu * n4) ? __builtin_nontemporal_load(csrc + e) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto cache_commit = [&](int e0) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < kInFlight; ++k) {
      const int e = e0 + k * kW16Threads;
      if (e < tot) {
        reinterpret_cast<f32x4*>(cch)[e] = q[k];
        const float qm = amax_acc(amax_acc(amax_acc(amax_acc(0.f, q[k][0]), q[k][1]), q[k][2]), q[k][3]);   // (bit patterns: NaN / Inf on top)
        if (e < n4) cm[0] = amax_merge(cm[0], qm); else cm[1] = amax_merge(cm[1], qm);
      }
    }
  };
  cache_request(tid);
  {                                                          // everything else travels with the first trip
    const int t = tid % TT;
    int qq = tid / TT;
    const int oct = qq & 3; qq >>= 2;
    const int st = qq % nk, u = qq / nk;
    if (u < U) {                                             // (wave-uniform)
      const int kf = st * 32 + oct * 8;
      const bool row = u < nu && t < T;
      const float* xr = A.x + int64_t(b0 + u) * A.xs_b + int64_t(t) * P.idim + kf;
      // K steps 0 / 1 sit in the stream's depthwise planes, 2 / 3 in its mid planes
      xi.dst = (st >> 1) * (U * UB) + u * UB + (((st & 1) * 4 + oct) * TT + t) * 16;
      if (row && kf + 8 <= P.idim && (reinterpret_cast<uintptr_t>(xr) & 15) == 0) {
        const float4 a = *reinterpret_cast<const float4*>(xr), c = *reinterpret_cast<const float4*>(xr + 4);
        xi.v = w16_f32x8{a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
      } else {
        w16_f32x8 v = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (row && kf + i < P.idim) ? xr[i] : 0.f;
        xi.v = v;
      }
    }
    if (active) {
      const uint4* ap = reinterpret_cast<const uint4*>(W + P.pre_a16) + size_t(ot) * nk * 128 + lane;
#pragma unroll
      for (int st2 = 0; st2 < 4; ++st2) {
        const uint4* qa = ap + min(st2, nk - 1) * 128;
        apre[st2].h = __builtin_bit_cast(f16x8, qa[0]);
        apre[st2].l = __builtin_bit_cast(f16x8, qa[64]);
      }
    }
    if (P.nblocks > 0) {
      if (mx_wave) bnext = load_bias(0); else tnext = load_taps(0);
    }
    if (xi.dst >= 0) amax_publish(amax_cells + (tid >= nk * 4 * TT ? kAmaxCells : 0), w16_x_amax_bits(xi));
  }
  cache_commit(tid);
  for (int e0 = tid + kInFlight * kW16Threads; e0 < tot; e0 += kInFlight * kW16Threads) {   // (longer caches)
    cache_request(e0);
    cache_commit(e0);
  }
  amax_publish(amax_cells + 1, cm[0]);
  amax_publish(amax_cells + kAmaxCells + 1, cm[1]);

  // ============================ preprocessing: h0 = [ReLU](x Wpre^T + b) ============================
  f32x4 acc, zsum = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    const float4 bias = *reinterpret_cast<const float4*>(W + P.pre_b + o0);
    __syncthreads();                                         // the feature (and cache) maxima are published
    if (amax_inputs_bad(amax_cells) || amax_inputs_bad(amax_cells + kAmaxCells)) {   // a NaN / Inf in a stream's chunk or cache:
      for (int u = 0; u < nu; ++u) nf_repair_call(A, b0 + u);                         // the reference's arithmetic for both streams
      return;
    }
    if (xi.dst >= 0) {
      float inv_unused;
      const float sx = pow2_scale(amax_read(amax_cells + (tid >= nk * 4 * TT ? kAmaxCells : 0)), &inv_unused);
      f16x8 vh, vl;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        _Float16 h, l;
        split16s(xi.v[i], sx, h, l);
        vh[i] = h; vl[i] = l;
      }
      *reinterpret_cast<f16x8*>(slab + xi.dst) = vh;
      if constexpr (SPLIT) *reinterpret_cast<f16x8*>(slab + xi.dst + MPB) = vl;
    }
    __syncthreads();
    if (active) {
      acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int st = 0; st < 4; ++st)
        if (st < nk) {
          const char* q = slab + (st >> 1) * (U * UB) + wu * UB + (st & 1) * 4 * TT * 16 + frag_off;
          const f16x8 vh = *reinterpret_cast<const f16x8*>(q);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(apre[st].h, vh, acc, 0, 0, 0);
          if constexpr (SPLIT) {
            const f16x8 vl = *reinterpret_cast<const f16x8*>(q + MPB);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(apre[st].h, vl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(apre[st].l, vh, acc, 0, 0, 0);
          }
        }
      float cpre;
      (void)pow2_scale(amax_read(amax_cells + wu * kAmaxCells), &cpre);
      cpre *= P.pre_inv_s;                                   // 1 / (feature scale * weight scale)
      float hmax = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = fmaf(acc[r], cpre, f4c(bias, r));
        if (P.pre_relu) v = fmaxf(v, 0.f);
        h_w[(o0 + r) * SS + l15] = v;
        hmax = fmaxf(hmax, fabsf(v));
      }
      amax_publish(amax_cells + wu * kAmaxCells + 2, hmax);
    }
    __syncthreads();
  }

  // ======================================= residual blocks =======================================
  // weight fragments of one GEMM (2 K steps, hi | lo)
  auto load_frags = [](F16Frag (&a)[2], const uint4* __restrict__ ap) __attribute__((always_inline)) {
    a[0].h = __builtin_bit_cast(f16x8, ap[0]);   a[0].l = __builtin_bit_cast(f16x8, ap[64]);
    a[1].h = __builtin_bit_cast(f16x8, ap[128]); a[1].l = __builtin_bit_cast(f16x8, ap[192]);
  };
  auto gemm = [&](const F16Frag (&a)[2], const char* planes) __attribute__((always_inline)) {
    acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const char* q = planes + ks * 4 * TT * 16 + frag_off;
      const f16x8 vh = *reinterpret_cast<const f16x8*>(q);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ks].h, vh, acc, 0, 0, 0);
      if constexpr (SPLIT) {                                 // !SPLIT = WEKWS_HIP_PRECISION_F16: hi halves only
        const f16x8 vl = *reinterpret_cast<const f16x8*>(q + MPB);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ks].h, vl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ks].l, vh, acc, 0, 0, 0);
      }
    }
  };

  // Producer body for dilation D (compile time): lane (pc, g) makes the outputs of NR runs of R frames at stride D
  // (R = min(4, 16 / D)): run i = g NR + r starts at frame f0 = (i / D) R D + i % D; its inputs are the R + 4 frames
  // f0 + (q - 4) D, from the tile where that is >= 0, else from the block's cache slice (pad = 4 D frames, mdtc.py:104-111).
  // Then the slice is shifted in place: new slice = last pad frames of [slice | chunk] (mdtc.py:111); every read of the
  // row (one wave owns it) precedes the writes.
  auto produce = [&](auto d_c, const BlockDesc& bd, const TapConst& tc, float sa) __attribute__((always_inline)) {
    constexpr int D = decltype(d_c)::value;
    constexpr int R = (16 / D) < 4 ? (16 / D) : 4, NR = 4 / R, PAD = 4 * D;
    const float dww[KS + 1] = {tc.q0.x, tc.q0.y, tc.q0.z, tc.q0.w, tc.q1.x, tc.q1.y};
    const float* hrow = hbuf + (pu * C + pc) * SS;
    float* crow = cch + (pu * C + pc) * Pc + bd.cache_off;
    float win[NR][R + 4];
    int f0[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int i = g * NR + r;
      f0[r] = (i / D) * R * D + (i % D);
      const float* tb = hrow + f0[r] - PAD;                  // tile address of input q = 0 (may lie left of the row)
      const float* cb = crow + f0[r];                        // cache address of input q = 0
#pragma unroll
      for (int q = 0; q < R + 4; ++q) {
        if (q >= 4) win[r][q] = tb[q * D];
        else win[r][q] = *((f0[r] >= (4 - q) * D) ? tb + q * D : cb + q * D);
      }
    }
    constexpr int NS = D;                                    // slice elements per lane: p = g + 4 k
    float nv[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      const int s = T + g + 4 * k;                           // index into [slice | chunk]
      nv[k] = *(s < PAD ? crow + s : hrow + (s - PAD));
    }
    _Float16* ph = reinterpret_cast<_Float16*>(slab + pu * UB) + ((pc >> 3) * TT) * 8 + (pc & 7);
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
      for (int m = 0; m < R; ++m) {
        float o = dww[KS];
#pragma unroll
        for (int j = 0; j < KS; ++j) o = fmaf(dww[j], win[r][m + j], o);
        const int t = f0[r] + m * D;
        _Float16 h, l;
        split16s(o, sa, h, l);
        ph[t * 8] = h;
        if constexpr (SPLIT) ph[t * 8 + MPB / 2] = l;
      }
#pragma unroll
    for (int k = 0; k < NS; ++k) crow[g + 4 * k] = nv[k];
  };

  for (int bi = 0; bi < P.nblocks; ++bi) {
    const BlockDesc& bd = blk[bi];
    const int nbi = min(bi + 1, P.nblocks - 1);
    if (mx_wave) {
      // ---------------- matrix waves ----------------
      const BiasConst bc = bnext;
      F16Frag g1[2], g2[2];
      float c1 = 0.f, sm = 1.f, c2 = 1.f;
      if (active) {
        load_frags(g1, reinterpret_cast<const uint4*>(W + bd.a1_16) + size_t(ot) * 256 + lane);
        load_frags(g2, reinterpret_cast<const uint4*>(W + bd.a2_16) + size_t(ot) * 256 + lane);
        bnext = load_bias(nbi);
        // operand scales (conv_stack_f16.hip.h): depthwise rows bounded through the maxima of the input tile and the
        // cache; mid tile: the bound chained behind that one
        const float au = fmaxf(amax_read(amax_cells + wu * kAmaxCells + 2 + bi), amax_read(amax_cells + wu * kAmaxCells + 1));
        const float ba = fmaf(bd.dw_alpha, au, bd.dw_beta);
        float inv;
        (void)pow2_scale(ba, &inv);
        c1 = inv * bd.inv_s1;
        sm = pow2_scale(fmaf(bd.mid_alpha, ba, bd.mid_beta), &c2);
        c2 *= bd.inv_s2;
      }
      __syncthreads();                                       // (A) the depthwise planes are written
      if (active) {
        gemm(g1, dw_u);
        // mid = ReLU(BN1(pointwise)) in operand order into its own planes (mdtc.py:113-114)
        const f32x4 v = __builtin_elementwise_max(acc * c1 + f32x4{bc.b1.x, bc.b1.y, bc.b1.z, bc.b1.w}, f32x4{0.f, 0.f, 0.f, 0.f}) * sm;
        const f16x4 vh = __builtin_convertvector(v, f16x4);
        char* dst = mid_u + (((o0 >> 3) * TT + l15) * 8 + (o0 & 7)) * 2;   // 4 consecutive channels = 8 bytes
        *reinterpret_cast<f16x4*>(dst) = vh;
        if constexpr (SPLIT)
          *reinterpret_cast<f16x4*>(dst + MPB) = __builtin_convertvector(v - __builtin_convertvector(vh, f32x4), f16x4);
      }
      __syncthreads();                                       // (B) the mid planes are written
      if (active) {
        // conv2 (1x1) + BN2, residual BEFORE the ReLU (mdtc.py:115-118), in place into h
        gemm(g2, mid_u);
        float hmax = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float* hp = h_w + (o0 + r) * SS + l15;
          const float v = fmaxf(fmaf(acc[r], c2, f4c(bc.b2, r)) + *hp, 0.f);
          if (bd.zadd) zsum[r] += v;
          *hp = v;
          hmax = fmaxf(hmax, v);
        }
        amax_publish(amax_cells + wu * kAmaxCells + 3 + bi, hmax);   // = the input tile of block bi + 1
      }
      __syncthreads();                                       // (C) the block's output tile is written
    } else {
      // ---------------- producers ----------------
      const TapConst tc = tnext;
      if (producing) {
        tnext = load_taps(nbi);
        const float au = fmaxf(amax_read(amax_cells + pu * kAmaxCells + 2 + bi), amax_read(amax_cells + pu * kAmaxCells + 1));
        float inv_unused;
        const float sa = pow2_scale(fmaf(bd.dw_alpha, au, bd.dw_beta), &inv_unused);
        switch (bd.dil) {
          case 1: produce(std::integral_constant<int, 1>{}, bd, tc, sa); break;
          case 2: produce(std::integral_constant<int, 2>{}, bd, tc, sa); break;
          case 4: produce(std::integral_constant<int, 4>{}, bd, tc, sa); break;
          default: produce(std::integral_constant<int, 8>{}, bd, tc, sa); break;   // (host: dilations are 1 / 2 / 4 / 8)
        }
      }
      __syncthreads();                                       // (A)
      __syncthreads();                                       // (B)
      __syncthreads();                                       // (C)
    }
  }

  // the backbone output is the sum of the stack outputs (mdtc.py:270-273)
  if (active) {
#pragma unroll
    for (int r = 0; r < 4; ++r) h_w[(o0 + r) * SS + l15] = zsum[r];
  }
  __syncthreads();
  if (A.out_cache) {
    const int otot = nu * n4;
    f32x4* dst = reinterpret_cast<f32x4*>(A.out_cache + int64_t(b0) * C * Pc);
    for (int e = tid; e < otot; e += kW16Threads) __builtin_nontemporal_store(reinterpret_cast<const f32x4*>(cch)[e], dst + e);
  }
  conv_stack_head<KIND_MDTC, 64, 1, kW16Threads, SS>(P, A, hbuf, reinterpret_cast<float*>(slab), b0);
}

inline size_t mdtc64_stream_lds_bytes(int cache_len) { return MdtcStreamGeom::lds_bytes(cache_len); }

template <bool SPLIT>
inline int launch_mdtc64_stream_s(const StackParams& P, const CallArgs& A, hipStream_t stream) {
  static DynLdsGrant grant;
  const size_t lds = mdtc64_stream_lds_bytes(P.cache_len);
  auto kern = mdtc64_stream_kernel<SPLIT>;
  if (grant_dynamic_lds(kern, int(lds), grant)) return -3;
  hipLaunchKernelGGL(kern, dim3((A.B + 1) / 2), dim3(kW16Threads), lds, stream, P, A);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// streaming step (A.T <= 16) with both streams' caches resident in LDS.  The host checks (wekws_hip.hip): kernel size 5,
// every dilation in {1, 2, 4, 8}, <= 128 input features, 64 * cache_len % 4 == 0, the caches fit into LDS.
int launch_mdtc64_stream(bool split, const StackParams& P, const CallArgs& A, hipStream_t stream);

}  // namespace wekws
