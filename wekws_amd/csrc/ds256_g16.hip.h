// DS-TCN, hidden_dim 256, no incoming cache (whole utterances, first chunks) -- REGISTER-RESIDENT 16-wave kernel (round 3).
// Same arithmetic and the same results bit for bit as ds256_w16.hip.h; a different home for the activations.
//
// What bounded ds256_w16 (round-2 counters + round-3 probes): per utterance 132 k cycles of which the LDS
// array is busy 69 k, the matrix pipe needs 31 k and the vector units 28 k -- nine barriers per block keep the three units
// taking turns, and most of the LDS work is the f32 activation tile going round in circles: the epilogue writes h (and
// reads it for the residual), the depthwise producer reads every element of it twice, writes the fp16 operand planes 64
// channels at a time, and sixteen waves read those back.  Three attempts to overlap the phases (ds256_r16: fixed roles;
// ds256_i16: every wave multiplies K step k while it produces k + 1; o-tile pairs) ran into the same two walls: one wave
// issues at most one vector instruction per ~8 cycles, so any arrangement that leaves the vector work to fewer waves is
// slower, and everything that adds LDS instructions gives back what the overlap gains.
//
// Here the f32 residual tile h never touches LDS.  Wave w owns output channels 16 w .. 16 w + 15 for ALL frames -- the
// accumulator layout of the 1x1 convolution (lane = 4 consecutive channels x frame 16 tt + (lane & 15)) -- and keeps h in
// that layout in 4 NT registers for the whole kernel:
//   * epilogue: h += ReLU(acc * c + bias), registers only;
//   * depthwise dilated conv (tcn.py:102-109): the frames a tap needs are in the SAME registers one 16-lane row over, so
//     x[t - s] is a DPP row shift -- v_fmac_f32_dpp row_shr:(s % 16) on tile tt - s/16 for the lanes that stay inside the
//     row, row_shl:(16 - s % 16) on the tile before for the lanes that cross it (bound_ctrl off: a lane whose source
//     falls outside the row is disabled), one plain FMA when 16 | s; left context = tiles < 0 = zeros = no instruction.
//     Two vector instructions per tap instead of one FMA + an LDS read, no addressing, no selects;
//   * the conv's output (+ folded BN, ReLU, block-floating scale, fp16 hi/lo split: v_fma_mixlo / mixhi write the two
//     halves of a register, so four channels are two registers) goes to LDS as MFMA B-operand planes for the WHOLE K = 256
//     -- the tile's place: 16 planes x 7 KB = 112 KB -- with 8-byte stores;
//   * the matrix phase then runs all eight K steps back to back (168 MFMAs per wave, B fragments double-buffered).
// Two barriers per block instead of nine; LDS instructions per block and wave: 14 stores + 112 fragment reads (was
// 56 + 56 tile accesses in the epilogue, 72 reads + 56 two-byte stores in the producer, 112 fragment reads).
// The streaming cache is handed over from the registers; the classifier head reads the tile from LDS as before (it is
// written there once, after the last block).
#pragma once
#include <utility>

#include "ds256_w16.hip.h"

namespace wekws {

#ifdef WEKWS_G16_STAMPS
#define G16_PH_DECL long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long tlast = clock64()
#define G16_PH(id) do { long long now_ = clock64(); tph[id] += now_ - tlast; tlast = now_; } while (0)
#define G16_PH_DUMP                                                                                    \
  do {                                                                                                 \
    __syncthreads();                                                                                   \
    if (blockIdx.x == 0 && A.out_cache && (threadIdx.x & 63) == 0)                                    \
      for (int i = 0; i < 8; ++i) A.out_cache[(threadIdx.x >> 6) * 8 + i] = float(tph[i]);               \
  } while (0)
#else
#define G16_PH_DECL
#define G16_PH(id)
#define G16_PH_DUMP
#endif

// o += w * x[lane - S] for the lanes whose source stays inside their 16-lane row (the others keep o)
template <int S>
__device__ __forceinline__ void g16_fmac_shr(float& o, float x, float w) {
  static_assert(S >= 1 && S <= 15, "row shift");
#define G16_SHR(n) if constexpr (S == n) asm("v_fmac_f32_dpp %0, %1, %2 row_shr:" #n " row_mask:0xf bank_mask:0xf" : "+v"(o) : "v"(x), "v"(w));
  G16_SHR(1) G16_SHR(2) G16_SHR(3) G16_SHR(4) G16_SHR(5) G16_SHR(6) G16_SHR(7) G16_SHR(8)
  G16_SHR(9) G16_SHR(10) G16_SHR(11) G16_SHR(12) G16_SHR(13) G16_SHR(14) G16_SHR(15)
#undef G16_SHR
}
// Frame layout of the register-resident tile (round 3, second version): MFMA column n = 16 tt + l (tile tt, lane l of the
// 16-lane row) holds FRAME  f(n) = NT l + tt  -- every lane owns NT CONSECUTIVE frames, one per register.  The matrix
// products never look at what a column means (columns are independent), so the permutation costs nothing there; it only
// shows where frames are named: the feature staging, the depthwise taps, the cache slices and the classifier tile.
//   * a tap s frames back, s = q NT + m: register tt - m of the lane q places to the left (tt >= m), else register
//     tt - m + NT of the lane q + 1 places to the left: ONE v_fmac_f32_dpp row_shr per tap and output (the first version, frame
//     = 16 tt + l, needed a row_shr on one tile plus a row_shl on the tile before: 13 instead of 7 DPP operations per
//     output -- and a DPP operation costs 3.3 SIMD cycles at four waves per SIMD against 2.3 for a plain v_fmac,
//     tools/probe/valu_rate.hip), a plain FMA when q = 0, nothing when the lane shift leaves the 16-lane row (left context
//     = zeros, bound_ctrl off);
//   * the slice of the streaming cache a block hands over is pad = 7 d frames = whole lanes when NT | T: 28 contiguous
//     bytes per lane and channel (dwordx4 + dwordx3) instead of 4-byte stores.
template <int S, int TT_, int NT, int R_>
__device__ __forceinline__ void g16_tap(float& o, const f32x4 (&hv)[NT], float w) {
  constexpr int Q = S / NT, M = S % NT;
  constexpr int REG = TT_ >= M ? TT_ - M : TT_ - M + NT;
  constexpr int SH = TT_ >= M ? Q : Q + 1;
  if constexpr (SH == 0) o = fmaf(w, hv[REG][R_], o);
  else if constexpr (SH <= 15) g16_fmac_shr<SH>(o, hv[REG][R_], w);
}
template <int S, int NT, int R_, int... TTs>
__device__ __forceinline__ void g16_tap_tiles(float (&o)[NT], const f32x4 (&hv)[NT], float w, std::integer_sequence<int, TTs...>) {
  (g16_tap<S, TTs, NT, R_>(o[TTs], hv, w), ...);
}

// ---- calls WITH an incoming cache (round 5).  The left context of a block -- the last `pad` frames of its input in the call
// before, its slice of the streaming cache (tcn.py:45-53) -- continues the lane-major tile to the left: frame g < 0 belongs to
// "lane" floor(g / NT) < 0.  It is kept in a second register tile cx whose lane p holds lane p - 16 (only the last
// ceil(pad / NT) lanes are non-zero), so the source of a tap that leaves the 16-lane row to the left is cx, SH lanes back =
// 16 - SH lanes FORWARD in cx: one more v_fmac_f32_dpp, row_shl, for exactly the lanes the row_shr left untouched (bound_ctrl
// off: a lane whose source is outside the row is disabled).  cx shares its registers with the accumulators (dead outside the
// matrix phase and the epilogue).
// o += w * x[lane + S] for the lanes whose source stays inside their 16-lane row (the others keep o)
template <int S>
__device__ __forceinline__ void g16_fmac_shl(float& o, float x, float w) {
  static_assert(S >= 1 && S <= 15, "row shift");
#define G16_SHL(n) if constexpr (S == n) asm("v_fmac_f32_dpp %0, %1, %2 row_shl:" #n " row_mask:0xf bank_mask:0xf" : "+v"(o) : "v"(x), "v"(w));
  G16_SHL(1) G16_SHL(2) G16_SHL(3) G16_SHL(4) G16_SHL(5) G16_SHL(6) G16_SHL(7) G16_SHL(8)
  G16_SHL(9) G16_SHL(10) G16_SHL(11) G16_SHL(12) G16_SHL(13) G16_SHL(14) G16_SHL(15)
#undef G16_SHL
}
// PART 0: the tile's share of the tap (the lanes whose source stays in the row), PART 1: the context's share (the others).  A
// tap's two instructions on one output depend on each other through the accumulator: they are issued as two passes over the NT
// outputs, never back to back.
template <int S, int TT_, int NT, int R_, int PART>
__device__ __forceinline__ void g16_tapc(float& o, const f32x4 (&hv)[NT], const f32x4 (&cx)[NT], float w) {
  constexpr int Q = S / NT, M = S % NT;
  constexpr int REG = TT_ >= M ? TT_ - M : TT_ - M + NT;
  constexpr int SH = TT_ >= M ? Q : Q + 1;
  static_assert(SH <= 15, "the context is one 16-lane row: NT >= 4 for paddings up to 56 frames");
  if constexpr (SH == 0) {
    if constexpr (PART == 0) o = fmaf(w, hv[REG][R_], o);
  } else if constexpr (PART == 0) {
    g16_fmac_shr<SH>(o, hv[REG][R_], w);
  } else {
    g16_fmac_shl<16 - SH>(o, cx[REG][R_], w);
  }
}
template <int S, int NT, int R_, int... TTs>
__device__ __forceinline__ void g16_tapc_tiles(float (&o)[NT], const f32x4 (&hv)[NT], const f32x4 (&cx)[NT], float w,
                                               std::integer_sequence<int, TTs...>) {
  (g16_tapc<S, TTs, NT, R_, 0>(o[TTs], hv, cx, w), ...);
  (g16_tapc<S, TTs, NT, R_, 1>(o[TTs], hv, cx, w), ...);
}

typedef float g16_f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 g16_f16x2 __attribute__((ext_vector_type(2)));

// scale + split of TWO depthwise outputs (channel rows 2 p and 2 p + 1 of one frame) into one packed hi and one packed lo
// register: t = v s (exact, s is a power of two), hi = fp16(t), lo = fp16(t - hi) -- split16s(), two at a time with
// v_cvt_pk_f16_f32: 8 vector operations per output pair, none of them slow.  (The first version used the
// mixed-precision FMAs, v_fma_mixlo / mixhi_f16 and v_fma_mix_f32, three per output: tools/probe/valu_rate.hip measures
// 6.0 SIMD cycles for one of those at four waves per SIMD against 1.9 for a v_fma_f32 and 3.3 for a packed operation --
// the split was 36 % of the depthwise phase.)  Same roundings, same bits.
template <bool SPLIT>
__device__ __forceinline__ void g16_split_pair(float v0, float v1, float s, unsigned& ph, unsigned& pl) {
  const float t0 = fmaxf(v0, 0.f) * s, t1 = fmaxf(v1, 0.f) * s;
  const g16_f16x2 h = __builtin_convertvector(g16_f32x2{t0, t1}, g16_f16x2);
  ph = __builtin_bit_cast(unsigned, h);
  if constexpr (SPLIT) {
    const float d0 = t0 - static_cast<float>(h[0]), d1 = t1 - static_cast<float>(h[1]);
    pl = __builtin_bit_cast(unsigned, __builtin_convertvector(g16_f32x2{d0, d1}, g16_f16x2));
  }
}

// Depthwise conv + folded BN + ReLU + scale / split of the channel-row pair (2 P_, 2 P_ + 1) of the lane's four, all NT
// frames of the lane at once: 2 NT independent accumulators per tap, so consecutive instructions never depend on one
// another.  Tap j multiplies the frame (KS - 1 - j) dilations back; j ascending like the reference's (and ds256_w16's) sum.
template <int D, int P_, int NT, bool SPLIT, bool CTX = false>
__device__ __forceinline__ void g16_dw_pair(const f32x4 (&hv)[NT], const f32x4 (&cx)[NT], const float* taps_o0, float sa, char* pst, int lo_off) {
  // taps + bias of channels o0 + 2 P_, + 1 (padded 12-float records): six LDS broadcasts
  const float4* src = reinterpret_cast<const float4*>(taps_o0 + 2 * P_ * 12);
  const float4 a0 = src[0], a1 = src[1], a2 = src[2], b0 = src[3], b1 = src[4], b2 = src[5];
  constexpr auto tiles = std::make_integer_sequence<int, NT>{};
  constexpr int RA = 2 * P_, RB = 2 * P_ + 1;
  float oa[NT], ob[NT];
  __builtin_amdgcn_s_setprio(P_ == 0 ? 3 : 1);               // (a wave that is ahead steps back: see the matrix phase)
#pragma unroll
  for (int tt = 0; tt < NT; ++tt) { oa[tt] = a2.x; ob[tt] = b2.x; }
  if constexpr (CTX) {
    g16_tapc_tiles<7 * D, NT, RA>(oa, hv, cx, a0.x, tiles); g16_tapc_tiles<7 * D, NT, RB>(ob, hv, cx, b0.x, tiles);
    g16_tapc_tiles<6 * D, NT, RA>(oa, hv, cx, a0.y, tiles); g16_tapc_tiles<6 * D, NT, RB>(ob, hv, cx, b0.y, tiles);
    g16_tapc_tiles<5 * D, NT, RA>(oa, hv, cx, a0.z, tiles); g16_tapc_tiles<5 * D, NT, RB>(ob, hv, cx, b0.z, tiles);
    g16_tapc_tiles<4 * D, NT, RA>(oa, hv, cx, a0.w, tiles); g16_tapc_tiles<4 * D, NT, RB>(ob, hv, cx, b0.w, tiles);
    g16_tapc_tiles<3 * D, NT, RA>(oa, hv, cx, a1.x, tiles); g16_tapc_tiles<3 * D, NT, RB>(ob, hv, cx, b1.x, tiles);
    g16_tapc_tiles<2 * D, NT, RA>(oa, hv, cx, a1.y, tiles); g16_tapc_tiles<2 * D, NT, RB>(ob, hv, cx, b1.y, tiles);
    g16_tapc_tiles<1 * D, NT, RA>(oa, hv, cx, a1.z, tiles); g16_tapc_tiles<1 * D, NT, RB>(ob, hv, cx, b1.z, tiles);
    g16_tapc_tiles<0, NT, RA>(oa, hv, cx, a1.w, tiles);     g16_tapc_tiles<0, NT, RB>(ob, hv, cx, b1.w, tiles);
  } else {
  g16_tap_tiles<7 * D, NT, RA>(oa, hv, a0.x, tiles); g16_tap_tiles<7 * D, NT, RB>(ob, hv, b0.x, tiles);
  g16_tap_tiles<6 * D, NT, RA>(oa, hv, a0.y, tiles); g16_tap_tiles<6 * D, NT, RB>(ob, hv, b0.y, tiles);
  g16_tap_tiles<5 * D, NT, RA>(oa, hv, a0.z, tiles); g16_tap_tiles<5 * D, NT, RB>(ob, hv, b0.z, tiles);
  g16_tap_tiles<4 * D, NT, RA>(oa, hv, a0.w, tiles); g16_tap_tiles<4 * D, NT, RB>(ob, hv, b0.w, tiles);
  g16_tap_tiles<3 * D, NT, RA>(oa, hv, a1.x, tiles); g16_tap_tiles<3 * D, NT, RB>(ob, hv, b1.x, tiles);
  g16_tap_tiles<2 * D, NT, RA>(oa, hv, a1.y, tiles); g16_tap_tiles<2 * D, NT, RB>(ob, hv, b1.y, tiles);
  g16_tap_tiles<1 * D, NT, RA>(oa, hv, a1.z, tiles); g16_tap_tiles<1 * D, NT, RB>(ob, hv, b1.z, tiles);
  g16_tap_tiles<0, NT, RA>(oa, hv, a1.w, tiles);     g16_tap_tiles<0, NT, RB>(ob, hv, b1.w, tiles);
  }
  __builtin_amdgcn_s_setprio(P_ == 0 ? 2 : 0);
#pragma unroll
  for (int tt = 0; tt < NT; ++tt) {
    // the pair's two halves of column 16 tt + l15: one 4-byte store per plane (waiting for the other pair to make it an
    // 8-byte store holds 2 NT registers through the second pair's taps: scratch)
    unsigned ph, pl;
    g16_split_pair<SPLIT>(oa[tt], ob[tt], sa, ph, pl);
    *reinterpret_cast<unsigned*>(pst + tt * 256 + P_ * 4) = ph;
    if constexpr (SPLIT) *reinterpret_cast<unsigned*>(pst + lo_off + tt * 256 + P_ * 4) = pl;
  }
}
template <int D, int NT, bool SPLIT, bool CTX = false>
__device__ __forceinline__ void g16_dw_rows(const f32x4 (&hv)[NT], const f32x4 (&cx)[NT], const float* taps_o0, float sa, char* pst, int lo_off) {
  g16_dw_pair<D, 0, NT, SPLIT, CTX>(hv, cx, taps_o0, sa, pst, lo_off);
  g16_dw_pair<D, 1, NT, SPLIT, CTX>(hv, cx, taps_o0, sa, pst, lo_off);
}
template <int D, int NT, bool SPLIT>
__device__ __forceinline__ void g16_dw_rows(const f32x4 (&hv)[NT], const float* taps_o0, float sa, char* pst, int lo_off) {
  g16_dw_rows<D, NT, SPLIT, false>(hv, hv, taps_o0, sa, pst, lo_off);
}

// NT consecutive floats from a dword-aligned address into row r of the lane's registers, as wide loads (g16_store_run's twin)
template <int NT>
__device__ __forceinline__ void g16_load_run(const float* src, f32x4 (&cv)[NT], int r) {
  struct __attribute__((packed, aligned(4))) V4 { float v[4]; };
  struct __attribute__((packed, aligned(4))) V3 { float v[3]; };
  if constexpr (NT == 7) {
    const V4 a = *reinterpret_cast<const V4*>(src);
    const V3 c = *reinterpret_cast<const V3*>(src + 4);
    cv[0][r] = a.v[0]; cv[1][r] = a.v[1]; cv[2][r] = a.v[2]; cv[3][r] = a.v[3];
    cv[4][r] = c.v[0]; cv[5][r] = c.v[1]; cv[6][r] = c.v[2];
  } else if constexpr (NT == 4) {
    const V4 a = *reinterpret_cast<const V4*>(src);
    cv[0][r] = a.v[0]; cv[1][r] = a.v[1]; cv[2][r] = a.v[2]; cv[3][r] = a.v[3];
  } else {
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) cv[tt][r] = src[tt];
  }
}

// The same item from the utterance's features in LDS (FAST: copied there ahead of time, frame-major like in memory)
template <int NT, int PB>
__device__ __forceinline__ W16XItem g16_take_x(const float* xbuf, int T, int idim, int nk, int e) {
  constexpr int TT = 16 * NT;
  W16XItem it;
  const int n = e % TT, q = e / TT;
  const int f = NT * (n & 15) + (n >> 4);
  const int oct = q & 3, st = q >> 2;
  const int kf = st * 32 + oct * 8;
  const bool has = e < nk * 4 * TT;
  it.dst = has ? st * 2 * PB + (oct * TT + n) * 16 : -1;
  it.ok = has && f < T && kf < idim;
  const float* p = xbuf + (it.ok ? f * idim + kf : 0);
  const float4 a = *reinterpret_cast<const float4*>(p), c = *reinterpret_cast<const float4*>(p + 4);
  it.v = w16_f32x8{a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
  return it;
}

// NT consecutive floats (row r of the lane's registers) to a dword-aligned address, as wide stores
template <int NT>
__device__ __forceinline__ void g16_store_run(float* dst, const f32x4 (&hv)[NT], int r) {
  struct __attribute__((packed, aligned(4))) V4 { float v[4]; };
  struct __attribute__((packed, aligned(4))) V3 { float v[3]; };
  struct __attribute__((packed, aligned(4))) V2 { float v[2]; };
  if constexpr (NT == 7) {
    *reinterpret_cast<V4*>(dst) = V4{{hv[0][r], hv[1][r], hv[2][r], hv[3][r]}};
    *reinterpret_cast<V3*>(dst + 4) = V3{{hv[4][r], hv[5][r], hv[6][r]}};
  } else if constexpr (NT == 4) {
    *reinterpret_cast<V4*>(dst) = V4{{hv[0][r], hv[1][r], hv[2][r], hv[3][r]}};
  } else if constexpr (NT == 2) {
    *reinterpret_cast<V2*>(dst) = V2{{hv[0][r], hv[1][r]}};
  } else {
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) dst[tt] = hv[tt][r];
  }
}

// The features of one utterance, one item (K step, k-octet, column n) per thread like w16_load_x, but column n of the operand
// planes is frame NT (n % 16) + n / 16
template <int NT, int PB>
__device__ __forceinline__ W16XItem g16_load_x(const float* __restrict__ xb, int T, int idim, int nk, int e) {
  constexpr int TT = 16 * NT;
  W16XItem it;
  const int n = e % TT, q = e / TT;
  const int f = NT * (n & 15) + (n >> 4);
  const int oct = q & 3, st = q >> 2;
  const int kf = st * 32 + oct * 8;
  const bool has = e < nk * 4 * TT;
  it.dst = has ? st * 2 * PB + (oct * TT + n) * 16 : -1;
  w16_fetch_x(it, xb + int64_t(f) * idim + kf, xb, has && f < T && kf < idim);
  return it;
}

// One 32-deep K step for one o-tile, B fragments of tile tt + 1 requested before the MFMAs of tile tt.
// FIRST: the accumulators start at zero -- the first MFMA of every tile takes the constant 0 as its C operand instead of
// NT x 4 registers that somebody had to clear.
template <int NT, bool SPLIT, bool FIRST = false>
__device__ __forceinline__ void g16_mfma_step(f32x4 (&acc)[NT], const F16Frag& a, const char* bh, const char* bl) {
  f16x8 vh[2], vl[2];
  vh[0] = *reinterpret_cast<const f16x8*>(bh);
  if constexpr (SPLIT) vl[0] = *reinterpret_cast<const f16x8*>(bl);
#pragma unroll
  for (int tt = 0; tt < NT; ++tt) {
    if (tt + 1 < NT) {
      vh[(tt + 1) & 1] = *reinterpret_cast<const f16x8*>(bh + (tt + 1) * 256);
      if constexpr (SPLIT) vl[(tt + 1) & 1] = *reinterpret_cast<const f16x8*>(bl + (tt + 1) * 256);
    }
    __builtin_amdgcn_sched_barrier(0);
    acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.h, vh[tt & 1], FIRST ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[tt], 0, 0, 0);
    if constexpr (SPLIT) {
      acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.h, vl[tt & 1], acc[tt], 0, 0, 0);
      acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.l, vh[tt & 1], acc[tt], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// FAST: the configuration of the keyword recipes -- features of <= 64 dims in whole aligned 8-float items (one trip through
// registers) and a per-frame linear head with one or two outputs (from the registers) -- with PERSISTENT workgroups; the
// general paths (any feature layout, every head through conv_stack_head) stay in the one-utterance-per-workgroup
// instantiation: inside the utterance loop their index arithmetic is loop-invariant, gets hoisted in front of the loop and
// the kernel spills (168 bytes of scratch per lane with both in one kernel).
// CTX (FAST only, NT >= 4): the call has an incoming cache -- a later chunk of a stream (17 .. 112 frames; shorter chunks run
// ds256_stream).  Until round 5 those calls fell back to the LDS-tile kernel ds256_w16 (measured at 80-frame chunks, the
// Android caller's size: 1.45 .. 1.5 x the time of the first, cache-less chunk).
template <int NT, bool SPLIT, bool FAST, bool CTX = false>
__global__ __launch_bounds__(kW16Threads) void ds256_g16_kernel(const StackParams P, const CallArgs A) {
  static_assert(!CTX || (FAST && NT >= 4), "the context tile is one 16-lane row");
  using G = W16Geom<NT>;
  constexpr int C = G::C, SS = G::SS, TT = G::TT, PB = G::PB;
  constexpr int NKS = C / 32;                                // K steps per layer
  constexpr int OTS = NKS * 128;                             // uint4 per o-tile
  static_assert(size_t(2 * NKS) * PB <= G::LDS_BYTES, "the operand planes of a whole layer live where the f32 tile was");
  extern __shared__ __attribute__((aligned(16))) float w16_lds[];
  char* const planes = reinterpret_cast<char*>(w16_lds);     // [K step][hi | lo][k-octet][frame][8 halves]
  float* const hbuf = w16_lds + G::SLAB / 4;                 // [256][SS] f32 tile -- only for the classifier, at the end

  f32x4 acc[NT];
  f32x4 hv[NT];                                              // the residual tile: channels o0 .. o0 + 3, frames NT l15 + tt
  f32x4 cx[CTX ? NT : 1];                                    // CTX: the current block's left context, lane p = lane p - 16 of the tile
  G16_PH_DECL;

  // ---- block floating point (conv_stack_f16.hip.h): maximum of the feature tile
  __shared__ AmaxCell amax_cells[kAmaxCells];
  __shared__ BlockDesc blk[kAmaxMaxBlocks];
  // utterances of this workgroup with a NaN / Inf input (nonfinite.hip.h): they leave the fast path and are re-computed behind the
  // loop, where nothing is live
  __shared__ NfList nfl;
  nf_list_init(nfl);
  // depthwise taps + bias of the CURRENT block, [256][12] floats (the 12-float records of BlockDesc::dw_pk): staged for
  // block bi + 1 behind block bi's matrix phase (block 0: during the preprocessing), read back as 16-byte broadcasts
  __shared__ __attribute__((aligned(16))) float taps[C * 12];
  const int nk = P.kpre16 / 32;
  // 40-d fbank: the features pass through registers once
  const bool one_trip = FAST || (nk <= 2 && 8 * TT <= kW16Threads && w16_x_vec_ok(A.x, A.xs_b, P.idim));
  static_assert(kAmaxMaxBlocks * sizeof(BlockDesc) / 4 <= kW16Threads, "one table dword per thread");
  {
    const int ntbl = P.nblocks * int(sizeof(BlockDesc) / 4);
    const int t0 = threadIdx.x;
    if (t0 < ntbl) reinterpret_cast<uint32_t*>(blk)[t0] = reinterpret_cast<const uint32_t*>(P.blocks)[t0];
  }
  // PERSISTENT workgroups: the grid is one workgroup per compute unit (the 160 KB tile admits no second one) and every
  // workgroup walks over utterances b, b + grid, ...  Per-phase stamps of the one-utterance-per-workgroup version add up to
  // 104 k cycles where the kernel takes 114 k per utterance and CU: a tenth of the time went into workgroup turnover
  // (16 waves and 160 KB of LDS to release, allocate and start four times per CU).
  // FAST: the NEXT utterance's features are copied from HBM straight into LDS (global_load_lds: no registers) behind the
  // classifier of the current one, into the 32 KB of the dynamic allocation above the planes (where the general
  // instantiation keeps the classifier's f32 tile) -- a linear copy of T * idim floats in 16-byte pieces, 1 KiB per wave
  // and instruction.  The trip to HBM that used to open every utterance (features -> registers -> maximum) now ends
  // before the utterance starts.
  float* const xbuf = w16_lds + (2 * NKS * PB) / 4;
  static_assert(!FAST || size_t(2 * NKS) * PB + size_t(TT) * 64 * 4 <= G::LDS_BYTES, "feature buffer above the planes");
  auto prefetch_x = [&](int bn) __attribute__((always_inline)) {
    const int nitems = (A.T * P.idim) >> 2;                  // 16-byte pieces (idim % 8 == 0)
    const float* src = A.x + int64_t(bn) * A.xs_b;
    const int t0 = threadIdx.x;
    for (int e0 = 0; e0 < nitems; e0 += kW16Threads) {       // (wave-uniform trip count; a wave's pieces are consecutive)
      const int wbase = __builtin_amdgcn_readfirstlane(e0 + (t0 & ~63));
      if (e0 + t0 < nitems)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (e0 + t0) * 4),
                                         (__attribute__((address_space(3))) void*)(xbuf + wbase * 4), 16, 0, 0);
    }
  };
  if constexpr (FAST) prefetch_x(blockIdx.x);
  for (int b = blockIdx.x; b < A.B; b += gridDim.x) {          // (not FAST: the grid is B, one pass)
  // (the weight pointer is re-made opaque for every utterance: with a loop-invariant __restrict__ pointer the compiler
  // hoists the preprocessing fragments, biases and classifier rows out of the loop and keeps them in registers for the
  // whole kernel -- 288 bytes of scratch per lane)
  const float* Wp = P.w;
  if constexpr (FAST) asm volatile("" : "+s"(Wp));
  const float* __restrict__ W = Wp;
  // (likewise the lane coordinates and T: everything derived from them -- plane addresses, the division constants of the
  // rarely taken paths -- would be computed once in front of the loop and held in registers)
  int tid = threadIdx.x;
  if constexpr (FAST) asm volatile("" : "+v"(tid));
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform (SGPR)
  const int l15 = lane & 15, lq = lane >> 4;
  int T = A.T;
  if constexpr (FAST) asm volatile("" : "+s"(T));
  const int Pc = P.cache_len;
  const int o0 = wave * 16 + lq * 4;                         // this lane's 4 channels: rows of the o-tile AND of the tile h
  const int frag_off = (lq * TT + l15) * 16;
  // where this lane's 4 channels of frame 16 tt + l15 sit in a hi plane: K step wave >> 1, k-octet (wave & 1) * 2 + lq / 2,
  // halves (lq & 1) * 4 .. + 3 of the 16-byte item
  char* const pst = planes + (wave >> 1) * 2 * PB + ((((wave & 1) * 2 + (lq >> 1)) * TT + l15) * 16 + (lq & 1) * 8);

  // taps + biases of a block: copied from the weight image straight into LDS (global_load_lds, 12 waves x 1 KiB), no
  // registers in between.  Nobody waits for the copy explicitly: every wave consumes weight fragments it requested
  // AFTER it (loads return in order) before it reaches the barrier in front of the depthwise phase that reads the taps.
  auto stage_taps = [&](const BlockDesc& nb) __attribute__((always_inline)) {
    if (wave < C * 3 / 64)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(W + nb.dw_pk + tid * 4),
                                       (__attribute__((address_space(3))) void*)(taps + wave * 256), 16, 0, 0);
  };
  // CTX: the left context of block `bi` into cx -- lane p holds the NT frames of lane p - 16, i.e. columns pad + NT (p - 16) ..
  // of the block's cache slice; whole lanes when NT divides the padding (NT = 7: pad = 7 d), element-wise otherwise.  Requested
  // at the top of the block (measured: requesting it behind the epilogue of the block before, to overlap the trip with the
  // barrier, keeps 28 more registers live across the block top and spills -- 0.30 ms per 1024 x 98 frames against 0.24).
  auto load_ctx = [&](int bi) __attribute__((always_inline)) {
    if constexpr (CTX) {
      const int padn = __builtin_amdgcn_readfirstlane(blk[bi].pad), offn = __builtin_amdgcn_readfirstlane(blk[bi].cache_off);
      const float* const ic = A.in_cache + (int64_t(b) * C + o0) * Pc + offn;
      const int c0 = padn + NT * (l15 - 16);                   // slice column of this lane's first context frame
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) cx[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (c0 >= 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) g16_load_run<NT>(ic + r * Pc + c0, cx, r);
      } else if (c0 + NT > 0) {                                // (the slice begins inside this lane: NT does not divide pad)
#pragma unroll
        for (int tt = 0; tt < NT; ++tt)
          if (c0 + tt >= 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) cx[tt][r] = ic[r * Pc + c0 + tt];
          }
      }
    }
  };
  // A NaN / Inf among this utterance's features or incoming cache (cells [0] / [1] at or above 0x7f800000): noted for the tail, the
  // next utterance's features requested as the classifier would have, on to the next one.  Every wave takes this branch or none
  // does (they read the same cells behind the same barrier), and the barrier inside keeps the cells until all of them have.
  auto skip_bad = [&]() __attribute__((always_inline)) {
    nf_list_note(nfl, b);
    if constexpr (FAST) {
      if (b + int(gridDim.x) < A.B) prefetch_x(b + gridDim.x);
    }
  };
  amax_zero<kW16Threads>(amax_cells, kAmaxCells);
  // The features are requested before the barrier: the table's trip to L2 (first utterance), the barrier and the request
  // for block 0's taps (whose address is in the table) all happen while the features are on their way from HBM.
  W16XItem xi;
  if constexpr (!FAST) {
    if (one_trip) xi = g16_load_x<NT, PB>(A.x + int64_t(b) * A.xs_b, T, P.idim, nk, tid);
  } else {
    __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0): this wave's pieces of the feature copy have landed
  }
  __syncthreads();                                           // table staged, cells zeroed; the utterance before is done with LDS
  stage_taps(blk[0]);
  if constexpr (CTX)     // block floating point: the depthwise rows are bounded through max(tile, incoming cache), like ds256_w16
    amax_publish(amax_cells + 1, amax_span_bits<kW16Threads>(A.in_cache + int64_t(b) * C * Pc, C * Pc, 0.f));
  if constexpr (FAST) xi = g16_take_x<NT, PB>(xbuf, T, P.idim, nk, tid);
  if (one_trip) {
    amax_publish(amax_cells, w16_x_amax_bits(xi));
  } else {
    amax_publish(amax_cells, amax_span_bits<kW16Threads>(A.x + int64_t(b) * A.xs_b, T * P.idim, 0.f));
  }

  // ============================ preprocessing: h0 = [ReLU](x Wpre^T + b) ============================
  {
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) acc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
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
      if (amax_inputs_bad(amax_cells)) { skip_bad(); continue; }
      sx = pow2_scale(amax_read(amax_cells), &cpre);
      w16_store_x<PB, SPLIT>(xi, sx, planes);
      __syncthreads();
#pragma unroll
      for (int st = 0; st < 2; ++st)                         // (compile-time indices: a runtime-indexed fragment array spills)
        if (st < nk)
          g16_mfma_step<NT, SPLIT>(acc, a[st], planes + st * 2 * PB + frag_off, planes + st * 2 * PB + PB + frag_off);
    } else {
    __syncthreads();
    if (amax_inputs_bad(amax_cells)) { skip_bad(); continue; }
    for (int k0 = 0; k0 < nk; k0 += 2) {                     // two K steps staged per pass
      const int steps = min(2, nk - k0);
      __syncthreads();
      sx = pow2_scale(amax_read(amax_cells), &cpre);
      for (int e = tid; e < steps * 4 * TT; e += kW16Threads) {   // item = (step, k-octet, column)
        const int n = e % TT;
        const int t = NT * (n & 15) + (n >> 4);              // the column's frame
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
        char* dst = planes + st * 2 * PB + (oct * TT + n) * 16;
        *reinterpret_cast<f16x8*>(dst) = vh;
        if constexpr (SPLIT) *reinterpret_cast<f16x8*>(dst + PB) = vl;
      }
      __syncthreads();
      for (int st = 0; st < steps; ++st) {
        F16Frag a[1];
        load_a16<1>(a, ap + (k0 + st) * 128, 0);
        g16_mfma_step<NT, SPLIT>(acc, a[0], planes + st * 2 * PB + frag_off, planes + st * 2 * PB + PB + frag_off);
      }
    }
    }
    cpre *= P.pre_inv_s;                                     // 1 / (feature scale * weight scale)
    float hmax = 0.f;
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = fmaf(acc[tt][r], cpre, f4c(bias, r));
        if (P.pre_relu) v = fmaxf(v, 0.f);
        hv[tt][r] = v;
        hmax = fmaxf(hmax, fabsf(v));
      }
    }
    amax_publish(amax_cells + 2, hmax);
    __syncthreads();                                         // (A) maximum published, planes free, taps staged
  }
  G16_PH(0);                                                 // [0] preprocessing
  // ======================================= residual blocks =======================================
  F16Frag a0;                                                // weight fragment of the even K steps; K step 0: carried over
  auto frag_base = [&](int i) __attribute__((always_inline)) {
    return reinterpret_cast<const uint4*>(W + __builtin_amdgcn_readfirstlane(blk[i].a1_16)) + size_t(wave) * OTS;
  };
  {
    const uint4* ap0 = frag_base(0);
    F16Frag t[1];
    load_a16<1>(t, ap0 + lane, 0); a0 = t[0];
  }
  for (int bi = 0; bi < P.nblocks; ++bi) {
    const BlockDesc bd = blk[bi];
    const int pad = bd.pad;
    // (wave-uniform bases in scalar registers: the fragment addresses are base + lane, one shared vector offset)
    const uint4* ap1 = frag_base(bi);
    // the fragment of the NEXT block's first K step is requested by this block's last pass (below)
    const uint4* apn = frag_base(min(bi + 1, P.nblocks - 1));
    // ---- operand scale of this block: the depthwise rows are bounded through the maximum of the input tile (published
    //      by the epilogue that produced it)
    float c1;
    const float au = CTX ? fmaxf(amax_read(amax_cells + 2 + bi), amax_read(amax_cells + 1)) : amax_read(amax_cells + 2 + bi);
    const float sa = pow2_scale(fmaf(bd.dw_alpha, au, bd.dw_beta), &c1);
    c1 *= bd.inv_s1;

    // ---- the block's streaming-cache slice = the last `pad` frames of its input tile [zeros | h] (tcn.py:45-53), from
    //      the registers (still the block's INPUT: the epilogue below is what changes them): frame t = 16 tt + l15 of
    //      channel o0 + r is column t - (T - pad).  WHERE the stores are issued matters more than how many there are: loads
    //      and stores share one in-order counter, so any load waited for behind them waits for their trip to HBM.  At the
    //      head of the block (or of the matrix phase) that load is the next weight fragment and the matrix phase stalls;
    //      here nothing is requested behind them until the next block's fragments, which are not needed before ITS matrix
    //      phase -- a whole depthwise phase later.  (A detour through LDS for 16-byte row segments, 22 store instructions
    //      instead of 48, was measured 2 % slower than the direct stores; so was spreading the whole-lane runs over the
    //      depthwise phase, two store instructions between every four taps: +2.7 %, their addresses live across the phase.
    //      Round 5: the same runs as buffer stores (scalar base, 32-bit offsets), plain and non-temporal: 0.2025 / 0.2009 ms
    //      against 0.2023 / 0.2011 -- noise; gpurun_out/r05b_handover_ab.txt.)
    auto hand_over = [&]() __attribute__((always_inline)) {
      if (A.out_cache) {
        float* const oc = A.out_cache + (int64_t(b) * C + o0) * Pc + bd.cache_off;
        const int p0 = NT * l15 - (T - pad);                   // slice column of this lane's first frame
        if (p0 >= 0 && p0 + NT <= pad) {                       // the lane's NT frames are NT consecutive columns of the slice
#pragma unroll
          for (int r = 0; r < 4; ++r) g16_store_run<NT>(oc + r * Pc + p0, hv, r);
        } else if (p0 + NT > 0 && p0 < pad) {                  // (NT does not divide T: slice boundary inside the lane)
#pragma unroll
          for (int tt = 0; tt < NT; ++tt) {
            const int p = p0 + tt;
            if (p >= 0 && p < pad) {
#pragma unroll
              for (int r = 0; r < 4; ++r) oc[r * Pc + p] = hv[tt][r];
            }
          }
        }
        if (T < pad) {                                         // shorter than the slice: in front of it zeros, or (CTX) what was the
          const int nz = pad - T;                              // tail of the incoming slice: [cache | h][-pad:] (tcn.py:45-53)
          for (int e = lane; e < 16 * nz; e += 64) {
            const int cc = e / nz, p = e - cc * nz;
            const int64_t at = (int64_t(b) * C + wave * 16 + cc) * Pc + bd.cache_off + p;
            A.out_cache[at] = CTX ? A.in_cache[at + T] : 0.f;
          }
        }
      }
    };
    load_ctx(bi);
    G16_PH(1);                                               // [1] block top

    // ---- depthwise dilated conv + folded BN + ReLU (tcn.py:102-109) of this lane's 4 channels x NT frames, from the
    //      registers; scale, split, store as operand planes of K step wave >> 1
    {
      const float* taps_o0 = taps + o0 * 12;
      switch (bd.dil) {                                      // (the host admits this kernel for dilations 1 / 2 / 4 / 8 only)
        case 1: if constexpr (CTX) g16_dw_rows<1, NT, SPLIT, true>(hv, cx, taps_o0, sa, pst, PB); else g16_dw_rows<1, NT, SPLIT>(hv, taps_o0, sa, pst, PB); break;
        case 2: if constexpr (CTX) g16_dw_rows<2, NT, SPLIT, true>(hv, cx, taps_o0, sa, pst, PB); else g16_dw_rows<2, NT, SPLIT>(hv, taps_o0, sa, pst, PB); break;
        case 4: if constexpr (CTX) g16_dw_rows<4, NT, SPLIT, true>(hv, cx, taps_o0, sa, pst, PB); else g16_dw_rows<4, NT, SPLIT>(hv, taps_o0, sa, pst, PB); break;
        case 8: if constexpr (CTX) g16_dw_rows<8, NT, SPLIT, true>(hv, cx, taps_o0, sa, pst, PB); else g16_dw_rows<8, NT, SPLIT>(hv, taps_o0, sa, pst, PB); break;
        default: break;
      }
    }
    G16_PH(2);                                               // [2] depthwise conv -> operand planes
    __syncthreads();                                         // (B) the planes of all 256 channels are written
    G16_PH(3);                                               // [3] barrier waits

    // the fragments of the odd K steps are requested here (K step 1 arrives behind K step 0's MFMAs: the depthwise phase
    // above needs the registers), the even ones a pass ahead -- K step 0 by the block before; the epilogue's bias too
    F16Frag a1;
    {
      F16Frag t[1];
      load_a16<1>(t, ap1 + 128 + lane, 0); a1 = t[0];
    }
    const float4 ebias = *reinterpret_cast<const float4*>(W + bd.b1 + o0);
    // next block's taps: on their way into LDS during the matrix phase
    if (bi + 1 < P.nblocks) stage_taps(blk[bi + 1]);         // (this block's taps were last read before barrier (B))

    // ---- pointwise conv: all eight K steps back to back
    auto kpass = [&](int ks, auto first_c) __attribute__((always_inline)) {
      // A wave that is ahead steps back (s_setprio 3 .. 0 over the four passes; likewise over the four quarters of the
      // depthwise phase, g16_dw_pair).  The SIMD serves its OLDEST wave first: without this the four waves of a SIMD finish a
      // phase one after the other -- all-wave stamps: the oldest is through the matrix phase after 6.9 k cycles, the youngest,
      // alone at the end with nobody to cover its LDS latencies, after 14.1 k, for 10.75 k of matrix pipe; with it 12.7 k
      // (depthwise phase 6.3 k -> 5.8 k).  -1.8 % in time (-5 % in cycles: the clock gives some of it back).
      // (Chaining the B-fragment requests across K steps -- the last tile of a step requesting the first tile of the next, so
      // that no step opens with an exposed LDS round trip -- keeps the fragment buffers alive across the weight loads: 36
      // bytes of scratch, +4 %.  Dropped.)
      if (ks == 0) __builtin_amdgcn_s_setprio(3);
      else if (ks == 2) __builtin_amdgcn_s_setprio(2);
      else if (ks == 4) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
      const char* bsrc = planes + ks * 2 * PB + frag_off;
      const uint4* nx = ks + 2 < NKS ? ap1 + (ks + 2) * 128 : apn;   // (last pass: K step 0 of the next block)
      g16_mfma_step<NT, SPLIT, decltype(first_c)::value>(acc, a0, bsrc, bsrc + PB);   // (first pass: C = 0, nothing to clear)
      {
        F16Frag t[1];
        load_a16<1>(t, nx + lane, 0); a0 = t[0];
      }
      g16_mfma_step<NT, SPLIT>(acc, a1, bsrc + 2 * PB, bsrc + 3 * PB);
      if (ks + 2 < NKS) {
        F16Frag t[1];
        load_a16<1>(t, nx + 128 + lane, 0); a1 = t[0];
      }
    };
    kpass(0, std::true_type{});
#pragma unroll 1
    for (int ks = 2; ks < NKS; ks += 2) kpass(ks, std::false_type{});
    G16_PH(4);                                               // [4] matrix phase

    hand_over();
    G16_PH(6);                                               // [6] cache hand-over

    // ---- epilogue: folded bias + ReLU + residual (tcn.py:60: add after the ReLU), registers only
    float hmax = 0.f;
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = fmaxf(fmaf(acc[tt][r], c1, f4c(ebias, r)), 0.f) + hv[tt][r];
        hv[tt][r] = v;
        hmax = fmaxf(hmax, fabsf(v));
      }
    }
    amax_publish(amax_cells + 3 + bi, hmax);             // = the input tile of block bi + 1
    G16_PH(5);                                               // [5] epilogue
    __syncthreads();                                         // (A) maximum published, planes free, taps staged
    G16_PH(3);
  }

  // ---- classifier
  if (FAST || (P.head == HEAD_LINEAR && P.odim <= 2)) {
    // keyword heads (one or two outputs per frame; classifier.py:63-67): from the registers.  Every lane multiplies its
    // 4 channels x NT frames with the classifier rows (8 NT FMAs), the 64 partial sums per output -- 16 waves x 4
    // channel groups -- meet in LDS where the planes were, four lanes add 16 of them each, a quad reduction and the
    // sigmoid finish the frame: one barrier, ~60 vector and ~25 LDS instructions per wave instead of writing the tile to
    // LDS for conv_stack_head (two more barriers, 80 LDS reads per output part; 8 k of the kernel's 110 k cycles).
    const int K = P.odim;
    constexpr int PS = 32 * NT + 16;                          // floats per partial row: 16 lanes x (NT frames x 2 outputs), padded
    float* const part = w16_lds;
    {
      const float4 w0 = *reinterpret_cast<const float4*>(W + P.head_w + o0);
      const float4 w1 = *reinterpret_cast<const float4*>(W + P.head_w + (K > 1 ? C : 0) + o0);
      if constexpr (FAST) {                                  // (behind the classifier rows: loads return in order)
        if (b + int(gridDim.x) < A.B) prefetch_x(b + gridDim.x);
      }
      float* dst = part + (wave * 4 + lq) * PS + 2 * NT * l15;
      const HeadPairs hw(w0, w1);                            // (packed, operand selects spelled out: pk_safe.hip.h)
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        pk_f32x2 pp{0.f, 0.f};
        head_fma4(pp, hw.x, hw.y, hw.z, hw.w, hv[tt]);
        *reinterpret_cast<float2*>(dst + 2 * tt) = float2{pp.x, pp.y};   // frame NT l15 + tt: outputs (0, 1)
      }
    }
    __syncthreads();
    {
      const int e = tid >> 2, qd = tid & 3;                  // e = 2 frame + output; four lanes per e
      const int t = e >> 1, k = e & 1;
      const float* src = part + qd * 16 * PS + e;
      float v = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) v += src[i * PS];
      v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
      v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
      if (qd == 0 && t < T && k < K) {
        v += W[P.head_b + k];
        if (P.sigmoid) v = sigmoidf_(v);
        A.y[int64_t(b) * A.ys_b + int64_t(t) * K + k] = v;
      }
    }
  } else if constexpr (!FAST) {
    // every other head reads the tile from LDS (conv_stack_head): written once, where the planes were
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) hbuf[(o0 + r) * SS + NT * l15 + tt] = hv[tt][r];
    }
    __syncthreads();
    conv_stack_head<KIND_DS, 256, NT, kW16Threads, SS>(P, A, hbuf, w16_lds, b);
  }
  G16_PH(7);                                                 // [7] classifier
  if constexpr (!FAST) break;                                // (one utterance per workgroup: no loop for the compiler to hoist out of)
  }                                                          // next utterance of this workgroup
  // ---- utterances with non-finite inputs: the reference's IEEE arithmetic (nonfinite.hip.h).  Cold; nothing is live here.
  nf_list_drain(nfl, A, P.idim, C * P.cache_len, blockIdx.x, gridDim.x);
  G16_PH_DUMP;                                               // (stamp builds: sums over this workgroup's utterances)
}

template <int NT, bool SPLIT, bool FAST, bool CTX = false>
inline int launch_ds256_g16_ntsf(const StackParams& P, const CallArgs& A, hipStream_t stream, int grid) {
  using G = W16Geom<NT>;
  static DynLdsGrant grant;
  auto kern = ds256_g16_kernel<NT, SPLIT, FAST, CTX>;
  if (grant_dynamic_lds(kern, int(G::LDS_BYTES), grant)) return -3;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kW16Threads), G::LDS_BYTES, stream, P, A);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
template <int NT, bool SPLIT>
inline int launch_ds256_g16_nts(const StackParams& P, const CallArgs& A, hipStream_t stream, int cus) {
  const bool fast = P.head == HEAD_LINEAR && P.odim <= 2 && P.kpre16 <= 64 && 8 * 16 * NT <= kW16Threads &&
                    P.idim % 8 == 0 && (reinterpret_cast<uintptr_t>(A.x) & 15) == 0 && A.xs_b % 4 == 0;   // = w16_x_vec_ok
  if (A.in_cache) {                                          // a later chunk of a stream: the context variant, where one is built
    if constexpr (NT >= 4) {
      if (fast) return launch_ds256_g16_ntsf<NT, SPLIT, true, true>(P, A, stream, A.B < cus ? A.B : cus);
    }
    return -4;                                               // (other heads / feature layouts / shorter tiles: ds256_w16)
  }
  return fast ? launch_ds256_g16_ntsf<NT, SPLIT, true>(P, A, stream, A.B < cus ? A.B : cus)
              : launch_ds256_g16_ntsf<NT, SPLIT, false>(P, A, stream, A.B);
}

// Calls whose blocks all have dilation 1, 2, 4 or 8 (the host checks; everything else: launch_ds256_w16): without an incoming
// cache, or -- keyword heads, tiles of >= 64 columns -- with one (returns -4 where no context variant is built).  split: three fp16 products per MAC on hi/lo operands (F16X3) or one on the hi halves (F16).
// cus: compute units of the device = the largest grid (persistent workgroups).
int launch_ds256_g16(int nt, bool split, const StackParams& P, const CallArgs& A, hipStream_t stream, int cus);

}  // namespace wekws
