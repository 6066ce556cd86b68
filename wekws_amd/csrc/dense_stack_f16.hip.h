// Fused forward for the plain TCN backbone (wekws/model/tcn.py:67-88, `ds: false`): every block is a DENSE dilated
// convolution, i.e. a GEMM with K = ksize*C whose operands are frame-shifted views of the activations themselves.
// Split-precision arithmetic as in conv_stack_f16.hip.h, but the resident tile is kept DIRECTLY in operand form:
//   - h lives in LDS as fp16 hi and lo planes [channel-octet][HALO + frames][8 channels]; the f32 value is hi + lo
//     (exact to 2^-22) and is rebuilt only for the residual add;
//   - the HALO columns in front of frame 0 physically hold the block's left context (its slice of the streaming
//     cache, or zeros), so the B fragment of tap j is ONE ds_read_b128 at frame offset -(ksize-1-j)*dilation with
//     no bounds handling, conflict free for any shift (16 lanes = 16 consecutive 16-byte slots);
//   - per block: [halo fill / cache hand-over] -> GEMM over K = (tap, channel) straight from the h planes -> barrier ->
//     epilogue rewrites the h planes in place -> barrier.  No producer, no operand slab, no f32 tile: 0.118 ms per
//     1024 utterances for the reference tcn.yaml model vs 0.213 ms with the generic kernel (shifted-copy producer).
//   - the per-frame linear head is a matrix product too (classifier rows as padded o-tiles).
// Tried and dropped: running MDTC through this kernel by folding its depthwise conv (+BN) into the first 1x1 conv
// (legal: no nonlinearity in between, mdtc.py:55-59).  Correct, but K grows 5x and the merged GEMM costs more than
// the depthwise producer it removes (0.340 vs 0.260 ms per 1024 utterances), so MDTC stays on conv_stack_f16.
// Geometry (waves, utterances per workgroup, o-tiles per wave) is Geom<> of conv_stack.hip.h.
#pragma once
#include "conv_stack_f16.hip.h"

namespace wekws {

struct DenseBlock {
  int32_t dil, pad, cache_off, zadd;
  uint32_t a1;   // packed fp16 hi/lo A fragments, K = ksize*C in (tap, channel) order
  uint32_t b1;   // f32 folded bias [C]
  uint32_t a2, b2;  // unused (kept for layout stability)
  float inv_s1;     // 1 / power-of-two scale of the packed matrix (block floating point, conv_stack_f16.hip.h)
};

struct DenseParams {
  const float* w;
  const DenseBlock* blocks;
  int32_t nblocks, idim, kpre16, ksize, odim, pre_relu;
  uint32_t pre_a16, pre_b;
  int32_t head, head_hidden, sigmoid;
  uint32_t head_a16;          // LINEAR: classifier rows as packed fragments (rows padded to 16)
  uint32_t head_w, head_b;    // f32: LINEAR bias; GLOBAL/LAST: W1[hh][C], b1
  uint32_t head_w2, head_b2;  // GLOBAL/LAST: W2[odim][hh], b2
  int32_t cache_len;
  float pre_inv_s, head_inv_s;  // 1 / power-of-two scales of pre_a16 and head_a16
};

template <int KIND, int C, int NT>
struct DenseGeom {
  using G = Geom<KIND, C, NT>;
  static constexpr int U = G::U;
  static constexpr int TT = 16 * NT;
  static constexpr int HALO = 56;                                  // max (ksize-1)*dilation of the reference recipe (7*8)
  static constexpr int FR = HALO + TT;                             // frames per octet row of an h plane
  static constexpr int HPLANE = (C / 8) * FR * 16;                 // one hi (or lo) plane of h, bytes
  static constexpr int XPLANE = 4 * TT * 16;                       // one K step (32 features) of staged input
  static constexpr int SCRATCH = 2 * XPLANE;
  static constexpr int UB = 2 * HPLANE + SCRATCH;                  // bytes per utterance
  static constexpr size_t LDS_BYTES = size_t(U) * UB;
};

// f32 value of 4 consecutive channels of one frame from the hi/lo planes (8-byte loads)
// (the planes hold h * s, s = the tile's power-of-two scale: `inv` = 1 / s on the way out, `s` on the way in)
__device__ __forceinline__ void load_h4(const char* p, int lo_off, float inv, float (&v)[4]) {
  const f16x4 h = *reinterpret_cast<const f16x4*>(p);
  const f16x4 l = *reinterpret_cast<const f16x4*>(p + lo_off);
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = (static_cast<float>(h[r]) + static_cast<float>(l[r])) * inv;
}

__device__ __forceinline__ void store_h4(char* p, int lo_off, float s, const float (&v)[4]) {
  f16x4 h, l;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    _Float16 a, b;
    split16(v[r] * s, a, b);
    h[r] = a; l[r] = b;
  }
  *reinterpret_cast<f16x4*>(p) = h;
  *reinterpret_cast<f16x4*>(p + lo_off) = l;
}

template <int KIND, int C, int NT, int KS>
__global__ __launch_bounds__(kThreads, 2) void dense_stack_f16_kernel(const DenseParams P, const CallArgs A) {
  using G = Geom<KIND, C, NT>;
  using D = DenseGeom<KIND, C, NT>;
  constexpr int U = G::U, OW = G::OW, TT = D::TT, HALO = D::HALO, FR = D::FR;
  constexpr int HP = D::HPLANE, XP = D::XPLANE, UB = D::UB;
  extern __shared__ __attribute__((aligned(16))) char dense_lds[];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wo = wave % G::WO, wu = wave / G::WO;
  const int l15 = lane & 15, lq = lane >> 4;
  const int T = A.T;
  const int b0 = blockIdx.x * U;
  const float* __restrict__ W = P.w;
  const int Pc = P.cache_len;
  const int o_base = wo * OW * 16;
  char* const hp_w = dense_lds + wu * UB;                  // this wave's utterance: h hi plane (lo at + HP)
  char* const sc_w = hp_w + 2 * HP;                        // scratch: mid planes / staged input
  // this lane's D-fragment rows (4 consecutive channels o..o+3 of o-tile ow) at frame t of the h planes:
  //   byte = ((o>>3)*FR + HALO + t)*16 + (o&7)*2
  auto h_elem = [&](int o, int t) -> int { return ((o >> 3) * FR + HALO + t) * 16 + (o & 7) * 2; };

  f32x4 acc[OW][NT];
  static_assert(KIND == KIND_TCN, "dense stack serves the plain TCN");

  // ---- block floating point (conv_stack_f16.hip.h): per-utterance maxima of the features and the incoming cache.
  //      The h planes (and the halo in front of them) of block i carry the scale of max(cell[2 + 2i], cache maximum).
  __shared__ AmaxCell amax_cells[U * kAmaxCells];
  __shared__ DenseBlock blk[kAmaxMaxBlocks];
  AmaxCell* const cells_w = amax_cells + wu * kAmaxCells;
  amax_zero<kThreads>(amax_cells, U * kAmaxCells);
  stage_block_table<kThreads>(blk, P.blocks, P.nblocks);
  __syncthreads();
  for (int u = 0; u < U; ++u)
    if (b0 + u < A.B) {
      amax_publish(amax_cells + u * kAmaxCells, amax_span_bits<kThreads>(A.x + int64_t(b0 + u) * A.xs_b, T * P.idim, 0.f));
      if (A.in_cache)
        amax_publish(amax_cells + u * kAmaxCells + 1, amax_span_bits<kThreads>(A.in_cache + int64_t(b0 + u) * C * Pc, C * Pc, 0.f));
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
  // scale of utterance u's h planes while they are the input of block bi (bi = nblocks: the backbone output)
  auto h_scale = [&](int u, int bi, float* inv) __attribute__((always_inline)) -> float {       // u wave-uniform
    return pow2_scale(fmaxf(amax_read(amax_cells + u * kAmaxCells + 2 + bi), amax_read(amax_cells + u * kAmaxCells + 1)), inv);
  };
  // the same for every utterance of the workgroup (per-lane u: select with pick())
  auto h_scales = [&](int bi, float (&sc)[U], float (&inv)[U]) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < U; ++u) sc[u] = h_scale(u, bi, &inv[u]);
  };
  auto pick = [](const float (&a)[U], int u) __attribute__((always_inline)) -> float {
    float v = a[0];
#pragma unroll
    for (int q = 1; q < U; ++q) v = (u == q) ? a[q] : v;
    return v;
  };

  // ---- zero the left halo of every h plane once (no-cache left context; overwritten per block when caching)
  for (int e = tid; e < U * 2 * (C / 8) * HALO; e += kThreads) {
    const int f = e % HALO;
    const int row = e / HALO;                              // (u, plane, octet)
    const int u = row / (2 * (C / 8)), pr = row % (2 * (C / 8));
    *reinterpret_cast<uint4*>(dense_lds + u * UB + (pr / (C / 8)) * HP + ((pr % (C / 8)) * FR + f) * 16) = uint4{0, 0, 0, 0};
  }

  // ============================ preprocessing: h0 = [ReLU](x Wpre^T + b) ============================
  {
    zero_acc(acc);
    const int nk = P.kpre16 / 32;
    const int ot_stride = nk * 128;
    const uint4* ap = reinterpret_cast<const uint4*>(W + P.pre_a16) + (wo * OW) * ot_stride + lane;
    float4 bias[OW];
#pragma unroll
    for (int ow = 0; ow < OW; ++ow) bias[ow] = *reinterpret_cast<const float4*>(W + P.pre_b + o_base + ow * 16 + lq * 4);
    for (int ks = 0; ks < nk; ++ks) {                      // one 32-feature K step staged per pass
      __syncthreads();
      float sxu[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float inv_unused;
        sxu[u] = pow2_scale(amax_read(amax_cells + u * kAmaxCells), &inv_unused);
      }
      for (int e = tid; e < U * 4 * TT; e += kThreads) {  // item = (utterance, k-octet, frame)
        const int t = e % TT;
        const int q = e / TT;
        const int oct = q & 3, u = q >> 2;
        const int kf = ks * 32 + oct * 8;
        const bool ok = (b0 + u) < A.B && t < T;
        const float* xr = A.x + int64_t(b0 + u) * A.xs_b + int64_t(t) * P.idim + kf;
        const float sx = pick(sxu, u);
        f16x8 vh, vl;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float v = (ok && kf + i < P.idim) ? xr[i] * sx : 0.f;
          _Float16 h, l;
          split16(v, h, l);
          vh[i] = h; vl[i] = l;
        }
        char* dst = dense_lds + u * UB + 2 * HP + (oct * TT + t) * 16;
        *reinterpret_cast<f16x8*>(dst) = vh;
        *reinterpret_cast<f16x8*>(dst + XP) = vl;
      }
      __syncthreads();
      F16Frag a[OW];
      load_a16<OW>(a, ap + ks * 128, ot_stride);
      mfma16_step<OW, NT>(acc, a, sc_w + (lq * TT + l15) * 16, sc_w + XP + (lq * TT + l15) * 16);
    }
    float cpre;
    (void)pow2_scale(amax_read(cells_w), &cpre);
    cpre *= P.pre_inv_s;
    float hmax = 0.f;
#pragma unroll
    for (int ow = 0; ow < OW; ++ow)
#pragma unroll
      for (int tt = 0; tt < NT; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = fmaf(acc[ow][tt][r], cpre, f4c(bias[ow], r));
          if (P.pre_relu) v = fmaxf(v, 0.f);
          acc[ow][tt][r] = v;
          hmax = fmaxf(hmax, fabsf(v));
        }
    amax_publish(cells_w + 2, hmax);
    __syncthreads();                                       // the tile's exact maximum is known before it is written
    float inv_h;
    const float sh = h_scale(wu, 0, &inv_h);
#pragma unroll
    for (int ow = 0; ow < OW; ++ow) {
      const int o = o_base + ow * 16 + lq * 4;
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        const float v[4] = {acc[ow][tt][0], acc[ow][tt][1], acc[ow][tt][2], acc[ow][tt][3]};
        store_h4(hp_w + h_elem(o, tt * 16 + l15), HP, sh, v);
      }
    }
    __syncthreads();
  }

  // ======================================= blocks =======================================
  const int frag_h = (lq * FR + HALO + l15) * 16;          // this lane's fragment item: octet lq, frame l15 of tile 0
  for (int bi = 0; bi < P.nblocks; ++bi) {
    const DenseBlock bd = blk[bi];
    const int d = bd.dil, pad = bd.pad;
    constexpr int NK1 = KS * (C / 32);                     // K steps of GEMM1: (tap, 32-channel half)
    const int ot_stride1 = NK1 * 128;
    const uint4* ap1 = reinterpret_cast<const uint4*>(W + bd.a1) + (wo * OW) * ot_stride1 + lane;
    F16Frag a0[OW], a1[OW];
    load_a16<OW>(a0, ap1, ot_stride1);
    float4 ebias[OW];
#pragma unroll
    for (int ow = 0; ow < OW; ++ow) ebias[ow] = *reinterpret_cast<const float4*>(W + bd.b1 + o_base + ow * 16 + lq * 4);

    // ---- left context of this block into the halo, and the streaming-cache hand-over   (tcn.py:49-54, mdtc.py:108-112)
    float shs[U], inv_shs[U];
    h_scales(bi, shs, inv_shs);
    if (A.in_cache) {
      for (int e = tid; e < U * C * pad; e += kThreads) {
        const int p = e % pad;
        const int uc = e / pad;
        const int u = uc / C, c = uc % C;
        float v = 0.f;
        if (b0 + u < A.B) v = A.in_cache[(int64_t(b0 + u) * C + c) * Pc + bd.cache_off + p];
        _Float16 h, l;
        split16(v * pick(shs, u), h, l);
        char* dst = dense_lds + u * UB + ((c >> 3) * FR + HALO - pad + p) * 16 + (c & 7) * 2;
        *reinterpret_cast<_Float16*>(dst) = h;
        *reinterpret_cast<_Float16*>(dst + HP) = l;
      }
      __syncthreads();
    }
    if (A.out_cache) {
      // new_cache = last `pad` frames of [cache | h] = plane frames T-pad .. T-1 (negative frames are the halo)
      for (int e = tid; e < U * C * pad; e += kThreads) {
        const int p = e % pad;
        const int uc = e / pad;
        const int u = uc / C, c = uc % C;
        if (b0 + u < A.B) {
          const char* src = dense_lds + u * UB + ((c >> 3) * FR + HALO + T - pad + p) * 16 + (c & 7) * 2;
          const float inv_hu = pick(inv_shs, u);
          const float v = (static_cast<float>(*reinterpret_cast<const _Float16*>(src)) +
                           static_cast<float>(*reinterpret_cast<const _Float16*>(src + HP))) * inv_hu;
          A.out_cache[(int64_t(b0 + u) * C + c) * Pc + bd.cache_off + p] = v;
        }
      }
    }

    // ---- GEMM1: dense dilated conv, K = (tap j, channel), B = h planes shifted by (KS-1-j)*d frames
    zero_acc(acc);
#pragma unroll 1
    for (int ks = 0; ks < NK1; ks += 2) {
      {
        const int j = ks / (C / 32), m = ks % (C / 32);
        const int off = frag_h + (m * 4 * FR - (KS - 1 - j) * d) * 16;
        load_a16<OW>(a1, ap1 + min(ks + 1, NK1 - 1) * 128, ot_stride1);
        __builtin_amdgcn_sched_barrier(0);
        mfma16_step<OW, NT>(acc, a0, hp_w + off, hp_w + HP + off);
      }
      if (ks + 1 < NK1) {
        const int j = (ks + 1) / (C / 32), m = (ks + 1) % (C / 32);
        const int off = frag_h + (m * 4 * FR - (KS - 1 - j) * d) * 16;
        load_a16<OW>(a0, ap1 + min(ks + 2, NK1 - 1) * 128, ot_stride1);
        __builtin_amdgcn_sched_barrier(0);
        mfma16_step<OW, NT>(acc, a1, hp_w + off, hp_w + HP + off);
      }
    }

    // ---- epilogue: bias, ReLU, residual (h = (hi + lo) / scale); the new tile's exact maximum is published before
    //      the barrier behind which the h planes are rewritten in place with their new scale
    float inv_h;
    (void)h_scale(wu, bi, &inv_h);
    const float c1 = inv_h * bd.inv_s1;
    float hmax = 0.f;
#pragma unroll
    for (int ow = 0; ow < OW; ++ow) {
      const int o = o_base + ow * 16 + lq * 4;
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        float hold[4];
        load_h4(hp_w + h_elem(o, tt * 16 + l15), HP, inv_h, hold);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = fmaxf(fmaf(acc[ow][tt][r], c1, f4c(ebias[ow], r)), 0.f) + hold[r];   // y + x, nothing after the add (tcn.py:60)
          acc[ow][tt][r] = v;
          hmax = fmaxf(hmax, fabsf(v));
        }
      }
    }
    amax_publish(cells_w + 3 + bi, hmax);
    __syncthreads();  // every wave has finished reading the h planes before they are rewritten in place
    const float sh_new = h_scale(wu, bi + 1, &inv_h);
#pragma unroll
    for (int ow = 0; ow < OW; ++ow) {
      const int o = o_base + ow * 16 + lq * 4;
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        const float v[4] = {acc[ow][tt][0], acc[ow][tt][1], acc[ow][tt][2], acc[ow][tt][3]};
        store_h4(hp_w + h_elem(o, tt * 16 + l15), HP, sh_new, v);
      }
    }
    __syncthreads();
  }

  // ============================================ head ============================================
  const int K = P.odim;
  if (P.head == HEAD_LINEAR) {
    // y[t][k] = act(sum_c Wc[k][c] h[c][t] + bc[k]) as a matrix product: classifier rows are o-tiles (padded to 16),
    // (utterance, o-tile) pairs are dealt to the waves                       (classifier.py:63-67)
    constexpr int NKH = C / 32;
    const int tiles = (K + 15) / 16;
    for (int job = wave; job < U * tiles; job += kWaves) {
      const int u = job / tiles, ot = job % tiles;
      const char* hu = dense_lds + u * UB;
      const uint4* ah = reinterpret_cast<const uint4*>(W + P.head_a16) + size_t(ot) * NKH * 128 + lane;
      f32x4 hacc[1][NT];
      zero_acc(hacc);
#pragma unroll
      for (int ks = 0; ks < NKH; ++ks) {
        F16Frag a[1];
        load_a16<1>(a, ah + ks * 128, 0);
        mfma16_step<1, NT>(hacc, a, hu + ks * 4 * FR * 16 + frag_h, hu + HP + ks * 4 * FR * 16 + frag_h);
      }
      if (b0 + u < A.B) {
        float ch;
        (void)h_scale(u, P.nblocks, &ch);
        ch *= P.head_inv_s;
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
          const int t = tt * 16 + l15;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int k = ot * 16 + lq * 4 + r;
            if (t < T && k < K) {
              float v = fmaf(hacc[0][tt][r], ch, W[P.head_b + k]);
              if (P.sigmoid) v = sigmoidf_(v);
              A.y[int64_t(b0 + u) * A.ys_b + int64_t(t) * K + k] = v;
            }
          }
        }
      }
    }
  } else if (P.head == HEAD_IDENTITY) {
    float shs[U], inv_shs[U];
    h_scales(P.nblocks, shs, inv_shs);
    for (int e = tid; e < U * T * C; e += kThreads) {
      const int c = e % C;
      const int ut = e / C;
      const int u = ut / T, t = ut - u * T;
      if (b0 + u >= A.B) continue;
      const char* src = dense_lds + u * UB + ((c >> 3) * FR + HALO + t) * 16 + (c & 7) * 2;
      const float inv_hu = pick(inv_shs, u);
      float v = (static_cast<float>(*reinterpret_cast<const _Float16*>(src)) +
                 static_cast<float>(*reinterpret_cast<const _Float16*>(src + HP))) * inv_hu;
      if (P.sigmoid) v = sigmoidf_(v);
      A.y[int64_t(b0 + u) * A.ys_b + int64_t(t) * C + c] = v;
    }
  } else {
    // GLOBAL: m = mean_t h ; LAST: m = h[:, -1]  ->  W2 ReLU(W1 m + b1) + b2   (classifier.py:26-28, :38-40)
    float* mvec = reinterpret_cast<float*>(dense_lds + 2 * HP);     // utterance 0's scratch: [U][C] then [U][hh]
    float* hid = mvec + U * C;
    const int HH = P.head_hidden;
    float shs[U], inv_shs[U];
    h_scales(P.nblocks, shs, inv_shs);
    for (int e = tid; e < U * C; e += kThreads) {
      const int u = e / C, c = e - u * C;
      const char* row = dense_lds + u * UB + ((c >> 3) * FR + HALO) * 16 + (c & 7) * 2;
      const float inv_hu = pick(inv_shs, u);
      auto at = [&](int t) -> float {
        return (static_cast<float>(*reinterpret_cast<const _Float16*>(row + t * 16)) +
                static_cast<float>(*reinterpret_cast<const _Float16*>(row + HP + t * 16))) * inv_hu;
      };
      float s;
      if (P.head == HEAD_GLOBAL) {
        s = 0.f;
        for (int t = 0; t < T; ++t) s += at(t);
        if (A.gsum && (b0 + u) < A.B) {
          float* gp = A.gsum + int64_t(b0 + u) * C + c;
          if (!A.first_tile) s += *gp;
          if (!A.last_tile) *gp = s;
        }
        s = s / float(A.T_total);
      } else {
        s = at(T - 1);
      }
      mvec[e] = s;
    }
    __syncthreads();
    if (A.last_tile) {
      for (int e = tid; e < U * HH; e += kThreads) {
        const int u = e / HH, j = e - u * HH;
        const float* w1 = W + P.head_w + j * C;
        float s = W[P.head_b + j];
        for (int c = 0; c < C; ++c) s = fmaf(w1[c], mvec[u * C + c], s);
        hid[e] = nf_relu(s);                                 // (a NaN carried in by the running sums of an earlier tile stays one)
      }
      __syncthreads();
      for (int e = tid; e < U * K; e += kThreads) {
        const int u = e / K, k = e - u * K;
        if (b0 + u >= A.B) continue;
        const float* w2 = W + P.head_w2 + k * HH;
        float s = W[P.head_b2 + k];
        for (int j = 0; j < HH; ++j) s = fmaf(w2[j], hid[u * HH + j], s);
        if (P.sigmoid) s = sigmoidf_(s);
        A.y[int64_t(b0 + u) * A.ys_b + k] = s;
      }
    }
  }
}

template <int KIND>
int launch_dense_stack_f16(int C, int nt, const DenseParams& P, const CallArgs& A, hipStream_t stream);
template <> int launch_dense_stack_f16<KIND_TCN>(int, int, const DenseParams&, const CallArgs&, hipStream_t);

template <int KIND, int C, int NT>
inline int launch_dense_one(const DenseParams& P, const CallArgs& A, hipStream_t stream) {
  using D = DenseGeom<KIND, C, NT>;
  constexpr int KS = 8;
  if (P.ksize != KS) return -4;
  if (D::LDS_BYTES > 160 * 1024) return -4;
  static DynLdsGrant grant;
  auto kern = dense_stack_f16_kernel<KIND, C, NT, KS>;
  if (grant_dynamic_lds(kern, int(D::LDS_BYTES), grant)) return -3;
  const int grid = (A.B + D::U - 1) / D::U;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), D::LDS_BYTES, stream, P, A);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

#define WEKWS_DISPATCH_NT_DENSE(KIND, CC)                                      \
  switch (nt) {                                                                \
    case 1: return launch_dense_one<KIND, CC, 1>(P, A, stream);                \
    case 2: return launch_dense_one<KIND, CC, 2>(P, A, stream);                \
    case 4: return launch_dense_one<KIND, CC, 4>(P, A, stream);                \
    case 7: return launch_dense_one<KIND, CC, 7>(P, A, stream);                \
    default: return -1;                                                        \
  }

#define WEKWS_DEFINE_LAUNCHER_DENSE(KIND)                                      \
  template <>                                                                  \
  int launch_dense_stack_f16<KIND>(int C, int nt, const DenseParams& P, const CallArgs& A, hipStream_t stream) { \
    switch (C) {                                                               \
      case 32: WEKWS_DISPATCH_NT_DENSE(KIND, 32)                               \
      case 64: WEKWS_DISPATCH_NT_DENSE(KIND, 64)                               \
      case 128: WEKWS_DISPATCH_NT_DENSE(KIND, 128)                             \
      default: return -4;                                                      \
    }                                                                          \
  }

}  // namespace wekws
