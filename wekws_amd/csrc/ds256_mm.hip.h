// DS-TCN, hidden_dim 256, per-frame linear head (the headline model and its CTC twin): depthwise convolution ON THE
// MATRIX CORES, and a matrix-core classifier of any width.
//
// ds256_w16.hip.h spends ~60 % of its time in the depthwise producer: 8 FMAs + split + stores per output on the vector
// ALU, which on a SIMD does not overlap with MFMA work (one vector instruction issues per MFMA of a co-resident wave).
// Here the depthwise dilated conv (tcn.py:102-109) is itself a matrix product:
//
//   D[16 ch][16 frames] = sum over tap pairs p of  A_p (16 x 32)  x  B_p (32 x 16)
//   A_p = [diag(w[:, 2p]) | diag(w[:, 2p+1])]      the 16 channels' two taps on two diagonals (94 % zeros; the matrix
//                                                  pipe is 16x the vector FMA rate, so the waste is affordable)
//   B_p = [x[ch][t - s(2p) d] ; x[ch][t - s(2p+1) d]],  s(j) = 7 - j   the same 16 channels at two frame shifts
//
// with both operands split into fp16 hi + lo like every other product (3 MFMAs per A x B).  The point is B: the
// activations live in LDS ONLY as fp16 hi/lo operand planes [channel octet][frame][8 halves] (4 bytes per element like
// f32; h = hi + lo carries 22 mantissa bits), so a B fragment of any tap is ONE ds_read_b128 at a shifted frame index --
// no vector arithmetic at all.  Per (16 ch x 16 frames) tile: 4 tap pairs x 3 = 12 MFMAs replace 256 x 8 FMAs + 256
// splits.  The 16 x 16 D fragment (+ folded bias, ReLU) is split once and stored as 8-byte items into the pointwise
// GEMM's operand slab, which is unchanged from ds256_w16 (wave = o-tile, 64-channel K intervals).
//
// LDS (NT = 7): activation planes 2 x 57,344 + slab 28,672 + left-context planes of one 64-channel interval
// 2 x 7,168 = 157,696 B.  Frames < 0 (the causal left context) come from the streaming cache, staged one interval
// ahead as the same kind of planes, or zeros; only frame tiles that can reach them (16 ft < 7 d) pay a per-lane select.
// The residual is added from the planes (hi + lo) and written back in place.
#pragma once
#include "ds256_w16.hip.h"

namespace wekws {

template <int NT>
struct MmGeom {
  static constexpr int C = 256, TT = 16 * NT, PADMAX = 56;
  static constexpr int PB = Plane<32, TT>::BYTES;            // one hi (or lo) plane of one 32-channel K step
  static constexpr int SLAB = 4 * PB;
  static constexpr int HALO_P = 8 * PADMAX * 16;             // one plane of the interval's left context (8 octets)
  static constexpr int HP = (C / 8) * TT * 16;               // one plane of the activations
  static constexpr size_t LDS_BYTES = size_t(2 * HALO_P) + SLAB + size_t(2) * HP;
};

template <int NT, bool HAS_CACHE>
__global__ __launch_bounds__(kW16Threads) void ds256_mm_kernel(const StackParams P, const CallArgs A, uint32_t head_a16) {
  using G = MmGeom<NT>;
  constexpr int C = G::C, TT = G::TT, PB = G::PB, KS = 8, PADMAX = G::PADMAX, HALO_P = G::HALO_P, HP = G::HP;
  extern __shared__ __attribute__((aligned(16))) char mm_lds[];
  char* const halo = mm_lds;                                 // [hi | lo][8 octets][PADMAX][8 halves]
  char* const slab = halo + 2 * HALO_P;                      // pointwise B operand: [kstep][hi | lo][4 oct][TT][8]
  char* const hpl = slab + G::SLAB;                          // activations: [hi | lo][32 oct][TT][8]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  const int T = A.T;
  const int b = blockIdx.x;
  const float* __restrict__ W = P.w;
  const int Pc = P.cache_len;
  const int o0 = wave * 16 + lq * 4;                         // pointwise / preprocessing: this lane's 4 output channels
  const int frag_off = (lq * TT + l15) * 16;
  // where this lane's 4 channels o0..o0+3 of frame tile 0 sit inside an activation plane (8-byte item)
  const int hwr = ((o0 >> 3) * TT + l15) * 16 + (o0 & 7) * 2;

  f32x4 acc[1][NT];

  // ---- block floating point (conv_stack_f16.hip.h).  The activation planes (and the left-context planes in front of
  //      them) of block i carry the power-of-two scale of max(cell[2 + 2i], cache maximum).
  __shared__ AmaxCell amax_cells[kAmaxCells];
  __shared__ BlockDesc blk[kAmaxMaxBlocks];
  amax_zero<kW16Threads>(amax_cells, kAmaxCells);
  stage_block_table<kW16Threads>(blk, P.blocks, P.nblocks);
  __syncthreads();
  amax_publish(amax_cells, amax_span_bits<kW16Threads>(A.x + int64_t(b) * A.xs_b, T * P.idim, 0.f));
  if constexpr (HAS_CACHE)
    amax_publish(amax_cells + 1, amax_span_bits<kW16Threads>(A.in_cache + int64_t(b) * C * Pc, C * Pc, 0.f));
  __syncthreads();
  if (amax_inputs_bad(amax_cells)) {                         // a NaN / Inf feature or cache element: the reference's arithmetic
    if (blockIdx.y == 0) nf_repair_call(A, b);               // (head slices: every slice sees it, one of them re-computes)
    return;
  }
  auto h_amax = [&](int bi) __attribute__((always_inline)) -> float {
    return HAS_CACHE ? fmaxf(amax_read(amax_cells + 2 + bi), amax_read(amax_cells + 1)) : amax_read(amax_cells + 2 + bi);
  };

  if constexpr (!HAS_CACHE) {                                // zero left context, once
    for (int e = tid; e < 2 * HALO_P / 16; e += kW16Threads) *reinterpret_cast<uint4*>(halo + e * 16) = uint4{0, 0, 0, 0};
  }
  // left context of interval iv of a block: cache slice -> fp16 hi/lo planes (x_pad index j = frame j - pad)
  auto stage_halo = [&](const BlockDesc& nb, int iv, float sh) __attribute__((always_inline)) {
    (void)sh;
    if constexpr (HAS_CACHE) {
      const int pad = nb.pad;
      for (int e = tid; e < 64 * pad; e += kW16Threads) {
        const int cl = e / pad, j = e - cl * pad;
        const float v = A.in_cache[(int64_t(b) * C + iv * 64 + cl) * Pc + nb.cache_off + j] * sh;
        _Float16 h, l;
        split16(v, h, l);
        char* d = halo + ((cl >> 3) * PADMAX + j) * 16 + (cl & 7) * 2;
        *reinterpret_cast<_Float16*>(d) = h;
        *reinterpret_cast<_Float16*>(d + HALO_P) = l;
      }
    }
  };

  // ============================ preprocessing: h0 = [ReLU](x Wpre^T + b) -> planes ============================
  {
    zero_acc(acc);
    const int nk = P.kpre16 / 32;
    const uint4* ap = reinterpret_cast<const uint4*>(W + P.pre_a16) + size_t(wave) * nk * 128 + lane;
    const float4 bias = *reinterpret_cast<const float4*>(W + P.pre_b + o0);
    float sx = 1.f, cpre = 1.f;
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
        *reinterpret_cast<f16x8*>(dst + PB) = vl;
      }
      __syncthreads();
      for (int st = 0; st < steps; ++st) {
        F16Frag a[1];
        load_a16<1>(a, ap + (k0 + st) * 128, 0);
        mfma16_step_nb<NT>(acc[0], a[0], slab + st * 2 * PB + frag_off, slab + st * 2 * PB + PB + frag_off);
      }
    }
    const f32x4 b4 = {bias.x, bias.y, bias.z, bias.w};
    cpre *= P.pre_inv_s;
    float hmax = 0.f;
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
      f32x4 v = acc[0][tt] * cpre + b4;
      if (P.pre_relu) v = __builtin_elementwise_max(v, f32x4{0.f, 0.f, 0.f, 0.f});
      acc[0][tt] = v;
#pragma unroll
      for (int r = 0; r < 4; ++r) hmax = fmaxf(hmax, fabsf(v[r]));
    }
    amax_publish(amax_cells + 2, hmax);
    __syncthreads();                                         // the tile's exact maximum is known before it is written
    float inv_unused;
    const float sh0 = pow2_scale(h_amax(0), &inv_unused);
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
      const f32x4 v = acc[0][tt] * sh0;
      const f16x4 vh = __builtin_convertvector(v, f16x4);
      const f16x4 vl = __builtin_convertvector(v - __builtin_convertvector(vh, f32x4), f16x4);
      *reinterpret_cast<f16x4*>(hpl + hwr + tt * 256) = vh;
      *reinterpret_cast<f16x4*>(hpl + HP + hwr + tt * 256) = vl;
    }
    if (P.nblocks > 0) stage_halo(blk[0], 0, sh0);
    __syncthreads();
  }

  // ======================================= residual blocks =======================================
  constexpr int NIV = C / 64;
  constexpr int OTS = (C / 32) * 128;                        // uint4 per o-tile (8 K steps)
  const int ct = wave & 3, fq = wave >> 2;                   // depthwise: channel tile of the interval, frame-tile lane
  BlockDesc bdn = blk[0];
  for (int bi = 0; bi < P.nblocks; ++bi) {
    const BlockDesc bd = bdn;
    bdn = blk[min(bi + 1, P.nblocks - 1)];
    const int d = bd.dil, pad = bd.pad;
    const uint4* ap1 = reinterpret_cast<const uint4*>(W + bd.a1_16) + size_t(wave) * OTS + lane;
    F16Frag a0[1], a1[1];
    load_a16<1>(a0, ap1, 0);
    // byte distance of this lane's tap of pair p from the tile's own frame: s(tap) * d * 16 with tap = 2p + (lq >> 1),
    // s(j) = KS - 1 - j:  sd0 - p * 32 d
    const int sd0 = (KS - 1 - (lq >> 1)) * d * 16;
    // scales of this block: activation planes (exact maximum), depthwise output (bound), and what undoes them
    float inv_sh, c1;
    const float s_h = pow2_scale(h_amax(bi), &inv_sh);         // (named apart from the lane shift `sh` below)
    const float sa = pow2_scale(fmaf(bd.dw_alpha, h_amax(bi), bd.dw_beta), &c1);
    c1 *= bd.inv_s1;
    const float cdw = inv_sh * bd.dw_tap_inv * sa;             // depthwise accumulator -> operand units of the slab

    // taps of this lane's row channel (ct*16 + l15) and folded biases of its 4 D rows, one interval ahead
    f32x4 nq0, nq1, nbias;
    auto load_taps = [&](const BlockDesc& tb, int iv) __attribute__((always_inline)) {
      const int ch = iv * 64 + ct * 16;
      const float* rec = W + tb.dw_pk + (ch + l15) * 12;      // 8 taps (+ bias), padded to 12 floats
      nq0 = *reinterpret_cast<const f32x4*>(rec);
      nq1 = *reinterpret_cast<const f32x4*>(rec + 4);
      nbias = *reinterpret_cast<const f32x4*>(W + tb.dw_b + ch + lq * 4);
    };
    if (bi == 0) load_taps(bd, 0);

    zero_acc(acc);
#pragma unroll 1
    for (int iv = 0; iv < NIV; ++iv) {
      // The taps of this lane's row channel and the folded biases of its 4 D rows were requested during the previous
      // pointwise phase; this lane multiplies the even taps (lq >> 1 == 0) or the odd ones.
      const f32x4 tw = ((lq >> 1) ? f32x4{nq0[1], nq0[3], nq1[1], nq1[3]} : f32x4{nq0[0], nq0[2], nq1[0], nq1[2]}) * bd.dw_tap_s;
      const f32x4 dwb = nbias * sa;
      // ---- the new streaming cache = last `pad` columns of [left context | h].  item = (channel octet, column): one
      //      16-byte hi + lo item -> 8 channels, 8 coalesced 4-byte stores.  Usual case (T >= pad: no column comes
      //      from the old context): all 32 octets at once in the block's first interval; otherwise interval by
      //      interval, because only the current interval's left context is staged.
      if (A.out_cache && blockIdx.y == 0 && (T >= pad ? iv == 0 : true)) {   // head slices: one of them hands over
        const int noct = T >= pad ? C / 8 : 8, oct0 = T >= pad ? 0 : iv * 8;
        float* const ob = A.out_cache + int64_t(b) * C * Pc + bd.cache_off;      // wave-uniform base
        for (int e = tid; e < noct * pad; e += kW16Threads) {
          const int oc = e / pad, j = e - oc * pad;
          const int src = T + j - pad;                        // frame index; negative = still inside the old context
          const char* ph = src >= 0 ? hpl + ((oct0 + oc) * TT + src) * 16 : halo + (oc * PADMAX + (T + j)) * 16;
          const f16x8 vh = *reinterpret_cast<const f16x8*>(ph);
          const f16x8 vl = *reinterpret_cast<const f16x8*>(ph + (src >= 0 ? HP : HALO_P));
          const uint32_t o = uint32_t((oct0 + oc) * 8 * Pc + j);
#pragma unroll
          for (int i = 0; i < 8; ++i) ob[o + uint32_t(i * Pc)] = (float(vh[i]) + float(vl[i])) * inv_sh;
        }
      }
      // ---- depthwise on the matrix cores: tiles (ct, ft), ft = fq + 4 rd.  Tap pair by tap pair: the diagonal A
      //      fragment of pair p (row m = l15 = channel ct*16 + m; this lane's k-octet holds tap 2p + (lq>>1) of channels
      //      8*(lq&1)..+7, i.e. ONE non-zero half, element m & 7, iff (m >> 3) == (lq & 1)) is built once and used by
      //      both of the wave's tiles, so only one fragment is live at a time.
      constexpr int RD = (NT + 3) / 4;
      f32x4 dacc[RD];
      const char* tb[RD];                                      // this lane's B item of tile rd at shift sd0
#pragma unroll
      for (int rd = 0; rd < RD; ++rd) {
        dacc[rd] = f32x4{0.f, 0.f, 0.f, 0.f};
        tb[rd] = hpl + ((iv * 8 + ct * 2 + (lq & 1)) * TT + (fq + 4 * rd) * 16 + l15) * 16 - sd0;
      }
      const bool own = (l15 >> 3) == (lq & 1);
      const int dsel = (l15 & 7) >> 1, sh = (l15 & 1) * 16;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        F16Frag dg;
        {
          _Float16 h, l;
          split16(tw[p], h, l);
          const uint32_t hb = uint32_t(__builtin_bit_cast(unsigned short, h)) << sh;
          const uint32_t lb = uint32_t(__builtin_bit_cast(unsigned short, l)) << sh;
          typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
          u32x4_t vh, vl;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            vh[i] = (own && dsel == i) ? hb : 0u;
            vl[i] = (own && dsel == i) ? lb : 0u;
          }
          dg.h = __builtin_bit_cast(f16x8, vh);
          dg.l = __builtin_bit_cast(f16x8, vl);
        }
#pragma unroll
        for (int rd = 0; rd < RD; ++rd) {
          const int ft = fq + 4 * rd;
          if (ft < NT) {
            const int tn = ft * 16 + l15;
            const char* q = tb[rd] + p * 32 * d;
            int lo = HP;
            if (ft * 16 < (KS - 1) * d) {                      // wave-uniform: some tap of this tile reaches frame < 0
              const bool neg = tn * 16 < (sd0 - p * 32 * d);
              const char* qx = halo + ((ct * 2 + (lq & 1)) * PADMAX + pad + tn) * 16 - (sd0 - p * 32 * d);
              q = neg ? qx : q;
              lo = neg ? HALO_P : HP;
            }
            const f16x8 bh = *reinterpret_cast<const f16x8*>(q);
            const f16x8 bl = *reinterpret_cast<const f16x8*>(q + lo);
            dacc[rd] = __builtin_amdgcn_mfma_f32_16x16x32_f16(dg.h, bh, dacc[rd], 0, 0, 0);
            dacc[rd] = __builtin_amdgcn_mfma_f32_16x16x32_f16(dg.h, bl, dacc[rd], 0, 0, 0);
            dacc[rd] = __builtin_amdgcn_mfma_f32_16x16x32_f16(dg.l, bh, dacc[rd], 0, 0, 0);
          }
        }
      }
      // folded BN bias + ReLU (tcn.py:108-109), split, 8-byte item of the slab: channel ct*16 + 4 lq + r
#pragma unroll
      for (int rd = 0; rd < RD; ++rd) {
        const int ft = fq + 4 * rd;
        if (ft < NT) {
          const int tn = ft * 16 + l15;
          const f32x4 v = __builtin_elementwise_max(dacc[rd] * cdw + dwb, f32x4{0.f, 0.f, 0.f, 0.f});   // = sa * ReLU(dw + b)
          const f16x4 vh = __builtin_convertvector(v, f16x4);
          const f16x4 vl = __builtin_convertvector(v - __builtin_convertvector(vh, f32x4), f16x4);
          char* dst = slab + (ct >> 1) * 2 * PB + (((ct & 1) * 2 + (lq >> 1)) * TT + tn) * 16 + (lq & 1) * 8;
          *reinterpret_cast<f16x4*>(dst) = vh;
          *reinterpret_cast<f16x4*>(dst + PB) = vl;
        }
      }
      __syncthreads();
      // ---- pointwise: two K steps of this interval; the next interval's left context is staged meanwhile
      const int nx = min(iv + 1, NIV - 1);
      load_a16<1>(a1, ap1 + (2 * iv + 1) * 128, 0);           // second K step: requested now, used after 3 NT MFMAs
      mfma16_step_nb<NT>(acc[0], a0[0], slab + frag_off, slab + PB + frag_off);
      load_a16<1>(a0, ap1 + (2 * nx) * 128, 0);                // first K step of the next interval: a whole phase ahead
      mfma16_step_nb<NT>(acc[0], a1[0], slab + 2 * PB + frag_off, slab + 3 * PB + frag_off);
      if (iv + 1 < NIV) { load_taps(bd, iv + 1); stage_halo(bd, iv + 1, s_h); }
      else if (bi + 1 < P.nblocks) { load_taps(bdn, 0); }     // (its left context waits for the new planes' scale)
      __syncthreads();
    }

    // ---- epilogue: folded bias + ReLU + residual (tcn.py:60: add after the ReLU), in place in the planes
    const f32x4 eb = *reinterpret_cast<const f32x4*>(W + bd.b1 + o0);
    float hmax = 0.f;
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
      const char* hp = hpl + hwr + tt * 256;
      const f32x4 hold = (__builtin_convertvector(*reinterpret_cast<const f16x4*>(hp), f32x4) +
                          __builtin_convertvector(*reinterpret_cast<const f16x4*>(hp + HP), f32x4)) * inv_sh;
      const f32x4 v = __builtin_elementwise_max(acc[0][tt] * c1 + eb, f32x4{0.f, 0.f, 0.f, 0.f}) + hold;
      acc[0][tt] = v;
#pragma unroll
      for (int r = 0; r < 4; ++r) hmax = fmaxf(hmax, fabsf(v[r]));
    }
    amax_publish(amax_cells + 3 + bi, hmax);
    __syncthreads();                                         // the new tile's exact maximum sets the planes' new scale
    float inv_unused;
    const float sh_new = pow2_scale(h_amax(bi + 1), &inv_unused);
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
      char* hp = hpl + hwr + tt * 256;
      const f32x4 v = acc[0][tt] * sh_new;
      const f16x4 vh = __builtin_convertvector(v, f16x4);
      const f16x4 vl = __builtin_convertvector(v - __builtin_convertvector(vh, f32x4), f16x4);
      *reinterpret_cast<f16x4*>(hp) = vh;
      *reinterpret_cast<f16x4*>(hp + HP) = vl;
    }
    if (bi + 1 < P.nblocks) stage_halo(bdn, 0, sh_new);
    __syncthreads();
  }

  // ============ head: y[t] = [sigmoid](Wc h[t] + bc) on the matrix cores, straight from the planes ============
  const int K = P.odim;
  float chead;
  (void)pow2_scale(h_amax(P.nblocks), &chead);
  chead *= P.head_inv_s;                                     // 1 / (planes' scale * classifier scale)
  if (K <= 16) {
    // keyword heads (1..16 outputs): one padded o-tile; wave = frame tile
    if (wave < NT) {
      const uint4* ah = reinterpret_cast<const uint4*>(W + head_a16) + lane;
      f32x4 hacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < C / 32; ++ks) {
        F16Frag a[1];
        load_a16<1>(a, ah + ks * 128, 0);
        const char* q = hpl + ((ks * 4 + lq) * TT + wave * 16 + l15) * 16;
        const f16x8 bh = *reinterpret_cast<const f16x8*>(q);
        const f16x8 bl = *reinterpret_cast<const f16x8*>(q + HP);
        hacc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0].h, bh, hacc, 0, 0, 0);
        hacc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0].h, bl, hacc, 0, 0, 0);
        hacc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0].l, bh, hacc, 0, 0, 0);
      }
      const int t = wave * 16 + l15;
      if (t < T) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int k = lq * 4 + r;
          if (k < K) {
            float v = fmaf(hacc[r], chead, W[P.head_b + k]);
            if (P.sigmoid) v = sigmoidf_(v);
            A.y[int64_t(b) * A.ys_b + int64_t(t) * K + k] = v;
          }
        }
      }
    }
  } else {
    // CTC-sized heads (ds_tcn_ctc.yaml: thousands of tokens): waves walk pairs of o-tiles (the host pads the row count
    // to a multiple of 32), weights streamed from L2 one K step ahead, B fragments shared by the two tiles
    constexpr int NK = C / 32;
    const int HT = (K + 31) / 32 * 2;
    struct __attribute__((packed, aligned(4))) Y4 { float v[4]; };   // rows of y are only dword aligned
    float* const yb = A.y + int64_t(b) * A.ys_b;
    // few utterances on many CUs (streaming a handful of streams): gridDim.y workgroups run the same utterance and
    // split these o-tile pairs -- the head is 70 % of this model's weights (2599 x 256 of 955 k), and one workgroup's
    // time on a short chunk is the trip of its weights through the CU's 64 B/clk path.  Same numbers either way.
    const int S = A.head_slices > 1 ? A.head_slices : 1;
    const int per = ((HT / 2 + S - 1) / S) * 2;
    const int ot_lo = int(blockIdx.y) * per, ot_hi = min(HT, ot_lo + per);
    for (int ot = ot_lo + wave * 2; ot < ot_hi; ot += 2 * kW16Waves) {
      const uint4* ap = reinterpret_cast<const uint4*>(W + head_a16) + size_t(ot) * NK * 128 + lane;
      f32x4 hacc[2][NT];
#pragma unroll
      for (int ow = 0; ow < 2; ++ow)
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) hacc[ow][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 hb[2];
#pragma unroll
      for (int ow = 0; ow < 2; ++ow)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int k = (ot + ow) * 16 + lq * 4 + r;
          hb[ow][r] = k < K ? W[P.head_b + k] : 0.f;
        }
      F16Frag ha0[2], ha1[2];
      load_a16<2>(ha0, ap, NK * 128);
#pragma unroll 1
      for (int ks = 0; ks < NK; ks += 2) {
        load_a16<2>(ha1, ap + (ks + 1) * 128, NK * 128);
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
          const char* q = hpl + ((ks * 4 + lq) * TT + tt * 16 + l15) * 16;
          const f16x8 bh = *reinterpret_cast<const f16x8*>(q);
          const f16x8 bl = *reinterpret_cast<const f16x8*>(q + HP);
#pragma unroll
          for (int ow = 0; ow < 2; ++ow) {
            hacc[ow][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha0[ow].h, bh, hacc[ow][tt], 0, 0, 0);
            hacc[ow][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha0[ow].h, bl, hacc[ow][tt], 0, 0, 0);
            hacc[ow][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha0[ow].l, bh, hacc[ow][tt], 0, 0, 0);
          }
        }
        load_a16<2>(ha0, ap + min(ks + 2, NK - 1) * 128, NK * 128);
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
          const char* q = hpl + (((ks + 1) * 4 + lq) * TT + tt * 16 + l15) * 16;
          const f16x8 bh = *reinterpret_cast<const f16x8*>(q);
          const f16x8 bl = *reinterpret_cast<const f16x8*>(q + HP);
#pragma unroll
          for (int ow = 0; ow < 2; ++ow) {
            hacc[ow][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha1[ow].h, bh, hacc[ow][tt], 0, 0, 0);
            hacc[ow][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha1[ow].h, bl, hacc[ow][tt], 0, 0, 0);
            hacc[ow][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha1[ow].l, bh, hacc[ow][tt], 0, 0, 0);
          }
        }
      }
      const bool whole = (ot + 2) * 16 <= K;                 // wave-uniform: only the last pair can run past odim
#pragma unroll
      for (int ow = 0; ow < 2; ++ow) {
        const int k0 = (ot + ow) * 16 + lq * 4;
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
          const int t = tt * 16 + l15;
          f32x4 v = hacc[ow][tt] * chead + hb[ow];
          if (P.sigmoid) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = sigmoidf_(v[r]);
          }
          float* yr = yb + int64_t(t) * K + k0;
          if (whole) {
            if (t < T) *reinterpret_cast<Y4*>(yr) = Y4{{v[0], v[1], v[2], v[3]}};
          } else if (t < T) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (k0 + r < K) yr[r] = v[r];
          }
        }
      }
    }
  }
}

template <int NT, bool HAS_CACHE>
inline int launch_ds256_mm_ntc(const StackParams& P, const CallArgs& A, uint32_t head_a16, hipStream_t stream) {
  using G = MmGeom<NT>;
  static DynLdsGrant grant;
  auto kern = ds256_mm_kernel<NT, HAS_CACHE>;
  if (grant_dynamic_lds(kern, int(G::LDS_BYTES), grant)) return -3;
  hipLaunchKernelGGL(kern, dim3(A.B, A.head_slices > 1 ? A.head_slices : 1), dim3(kW16Threads), G::LDS_BYTES, stream, P, A,
                     head_a16);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

template <int NT>
inline int launch_ds256_mm_nt(const StackParams& P, const CallArgs& A, uint32_t head_a16, hipStream_t stream) {
  return A.in_cache ? launch_ds256_mm_ntc<NT, true>(P, A, head_a16, stream)
                    : launch_ds256_mm_ntc<NT, false>(P, A, head_a16, stream);
}

// usable when: kernel size 8, every block's padding <= 56 frames, per-frame linear head with odim <= 16 (host checks)
int launch_ds256_mm(int nt, const StackParams& P, const CallArgs& A, uint32_t head_a16, hipStream_t stream);

}  // namespace wekws
