// Log-mel filterbank front-end for gfx950.
//
// Reference semantics (paths relative to the reference tree):
//   wenet::Fbank::Compute                       runtime/core/frontend/fbank.h:138-198
//   mel bank / window construction              runtime/core/frontend/fbank.h:33-97
//   framing (snip edges, no centering)          runtime/core/frontend/fbank.h:141-142
//   int16-scale float input, no normalisation   runtime/core/frontend/feature_pipeline.cc:49-55
//
// GPU design (not the reference's scalar table-driven radix-2): one wave64 per 25 ms frame, persistent waves.
//   1. lane l loads the sample PAIRS (2n, 2n+1), n = l + 64m -- the real frame packed as a 256-point COMPLEX sequence
//      (a real 512-point FFT costs one 256-point complex FFT + a twiddle pass); the DC mean is a wave shuffle
//      reduction, pre-emphasis takes its left neighbour from the next lane down (one shuffle), the window values of a
//      lane's eight samples live in registers for the whole launch, like its twiddles;
//   2. 256 = 4^4: four radix-4 DIF stages, one butterfly per lane per stage; the first runs on the lane's own four
//      values, the others exchange through a 2.5 KiB LDS strip owned by the wave whose element e sits at e + (e >> 2)
//      (conflict-free for the stride-4 and stride-16 stages); the last stage stores X[k] in natural order;
//   3. lane l rebuilds X[k], k = l + 64m, from Z[k] and conj(Z[256-k]), squares it, and the 256 power bins go back to
//      LDS;
//   4. mel: the triangular filters are cut into SLOTS of at most 16 consecutive taps (a filter wider than 16 bins
//      gets two or more); a lane owns a slot for the whole launch with its 16 weights in registers, so a frame costs it
//      16 LDS reads + 16 FMAs, ascending in k like the reference; slot sums are combined per filter in slot order (the
//      only departure from the reference's single ascending loop: one float reassociation per extra slot), floored
//      at FLT_EPSILON, logf, stored as (T, num_bins) rows.  (LDS float atomics were tried for this step and cost 4x
//      the whole rest of the kernel: tens of lanes serialise on one accumulator.)
// PMC of the previous gather version (one lane per mel bin walking its taps, tables re-read from LDS every frame):
// vector ALU 59 % and LDS 74 % busy, 46 % of the LDS time in bank conflicts.
// Twiddles / window / mel weights are built on the host in double precision.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cfloat>
#include <cmath>
#include <vector>

namespace wekws {

constexpr int kFbankMaxBins = 128;
constexpr int kFbankMaxFft = 512;   // frames of up to 512 samples; the 512-point transform serves the reference's 128 / 256 / 512-point cases
#ifndef WEKWS_FBANK_WAVES
#define WEKWS_FBANK_WAVES 4
#endif
constexpr int kFbankWaves = WEKWS_FBANK_WAVES;      // waves per workgroup (each with kFbankFW frames in flight)

struct FbankParams {
  const float* tables;  // device: [tw256: 256 x (cos,sin)] [tw512: 256 x (cos,sin)] [window: 512] [mel...]
  int32_t num_bins;
  int32_t frame_length;
  int32_t frame_shift;
  int32_t mel_first_off;  // float offset of int-as-float first index per bin [num_bins]
  int32_t mel_size_off;   // sizes per bin [num_bins]
  int32_t mel_start_off;  // start offset of each bin's weights inside the weight run [num_bins]
  int32_t mel_w_off;      // concatenated weights
  int32_t mel_w_count;
  int32_t nslots;         // gather slots: runs of <= 16 taps of one filter (a lane owns a slot for the whole launch)
  int32_t slot_first_off; // first FFT bin of the slot (as float) [nslots]
  int32_t slot_bin_off;   // mel bin of the slot (as float) [nslots]
  int32_t slot_w_off;     // 16 weights per slot, zero padded [nslots][16]
  int32_t table_floats;
};

// Host: tables for the kernel.  Mel bank follows fbank.h:51-88 in float32 like the reference; the
// window follows fbank.h:90-96 (double, stored as float); twiddles are exact-rounded from double.
// Returns -1, or the index of the first mel filter that covers no FFT bin (too many bins for this sample rate / frame length:
// the reference CHECK-fails in its constructor, fbank.h:81; the caller refuses the configuration).
inline int fbank_build_tables(int num_bins, int sample_rate, int frame_length, int frame_shift, int window,
                              FbankParams* fp, std::vector<float>* out) {
  const int N = 512, NB = N / 2;
  std::vector<float>& t = *out;
  t.clear();
  const double two_pi = 6.283185307179586476925286766559;
  for (int j = 0; j < 256; ++j) {  // e^{-2 pi i j / 256}
    t.push_back(float(std::cos(two_pi * j / 256.0)));
    t.push_back(float(-std::sin(two_pi * j / 256.0)));
  }
  for (int k = 0; k < 256; ++k) {  // e^{-2 pi i k / 512}
    t.push_back(float(std::cos(two_pi * k / 512.0)));
    t.push_back(float(-std::sin(two_pi * k / 512.0)));
  }
  const double a = two_pi / (frame_length - 1);
  for (int i = 0; i < N; ++i) {
    double w = 0.0;
    if (i < frame_length) {
      if (window == 0) w = 0.54 - 0.46 * std::cos(a * double(i));
      else w = std::pow(0.5 - 0.5 * std::cos(a * double(i)), 0.85);
    }
    t.push_back(float(w));
  }
  // The reference transforms UpperPowerOfTwo(frame_length) points (fbank.h:43,117-119).  The kernel always runs its 512-point
  // transform on the zero-padded frame: bin k of an Nref-point DFT of a frame of <= Nref samples IS bin k * 512 / Nref of
  // its 512-point DFT, so shorter frames (8 kHz audio: 200 samples, 256 points) only change WHERE the mel bank looks -- the
  // filters are computed over the reference's Nref / 2 bins and laid out over every stride-th bin of the 512-point spectrum.
  int Nref = 64;
  while (Nref < frame_length) Nref *= 2;
  const int stride = N / Nref;
  auto mel = [](float f) { return 1127.0f * logf(1.0f + f / 700.0f); };
  const float bin_width = float(sample_rate) / Nref;
  const float mel_lo = mel(20.0f), mel_hi = mel(float(sample_rate / 2));
  const float delta = (mel_hi - mel_lo) / (num_bins + 1);
  std::vector<float> first(num_bins), size(num_bins), start(num_bins), weights;
  for (int b = 0; b < num_bins; ++b) {
    const float left = mel_lo + b * delta, center = mel_lo + (b + 1) * delta, right = mel_lo + (b + 2) * delta;
    int fi = -1, li = -1;
    std::vector<float> w(NB, 0.f);
    for (int i = 0; i < Nref / 2; ++i) {
      const float m = mel(bin_width * i);
      if (m > left && m < right) {
        w[i] = (m <= center) ? (m - left) / (center - left) : (right - m) / (right - center);
        if (fi < 0) fi = i;
        li = i;
      }
    }
    if (fi < 0) return b;             // empty filter: fbank.h:81
    first[b] = float(fi * stride);
    size[b] = float(li >= fi ? (li - fi) * stride + 1 : 0);
    start[b] = float(weights.size());
    for (int i = fi; i <= li; ++i) {
      weights.push_back(w[i]);
      if (i < li) weights.insert(weights.end(), size_t(stride - 1), 0.f);   // (the bins between two of the reference's)
    }
  }
  fp->num_bins = num_bins;
  fp->frame_length = frame_length;
  fp->frame_shift = frame_shift;
  fp->mel_first_off = int(t.size()); t.insert(t.end(), first.begin(), first.end());
  fp->mel_size_off = int(t.size()); t.insert(t.end(), size.begin(), size.end());
  fp->mel_start_off = int(t.size()); t.insert(t.end(), start.begin(), start.end());
  fp->mel_w_off = int(t.size()); t.insert(t.end(), weights.begin(), weights.end());
  fp->mel_w_count = int(weights.size());
  // gather slots: a filter of up to 16 taps is one slot, wider ones are cut into runs of 16; one lane per slot
  std::vector<float> sfirst, sbin, sw;
  for (int b = 0; b < num_bins; ++b)
    for (int k0 = 0; k0 < int(size[b]); k0 += 16) {
      sfirst.push_back(float(int(first[b]) + k0));
      sbin.push_back(float(b));
      for (int u = 0; u < 16; ++u) sw.push_back(k0 + u < int(size[b]) ? weights[size_t(start[b]) + k0 + u] : 0.f);
    }
  fp->nslots = int(sfirst.size());
  fp->slot_first_off = int(t.size()); t.insert(t.end(), sfirst.begin(), sfirst.end());
  fp->slot_bin_off = int(t.size()); t.insert(t.end(), sbin.begin(), sbin.end());
  fp->slot_w_off = int(t.size()); t.insert(t.end(), sw.begin(), sw.end());
  fp->table_floats = int(t.size());
  return -1;
}

// The LDS strip of a frame is private to one wave and a wave's LDS operations execute in program order, so
// hand-offs between lanes of the wave need no workgroup barrier: a wave-scope fence (orders the compiler and
// drains the wave's outstanding LDS traffic) is enough, and the four waves of a workgroup run unsynchronised.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

__device__ __forceinline__ int rev4_256(int k) {  // reverse the four base-4 digits of k
  return ((k & 3) << 6) | (((k >> 2) & 3) << 4) | (((k >> 4) & 3) << 2) | ((k >> 6) & 3);
}

// Position of element e of the FFT strip: an XOR swizzle instead of round 1's padding e + (e >> 2).  The 8-byte accesses of
// a wave are served in two groups of 32 lanes over 32 bank pairs; with bits 5 and 6 of e folded into the low five bits (6 =
// 0b00110, 25 = 0b11001) every access pattern of the four radix-4 stages -- runs of 64 (stage 0, untangle), 16-runs 64
// apart (q = 16), 4-runs 16 apart (q = 4), stride 4 (q = 1), the digit-reversed last stores -- hits 32 different pairs
// (derivation: tools/probe/fbank_swizzle.py; only the reversed untangle reads keep one 2-way pair).  PMC, round 3 layout:
// 10.4 M of 32.1 M LDS-active cycles were bank conflicts.
__device__ __forceinline__ int pz(int e) { return e ^ (((e >> 5) & 1) * 6) ^ (((e >> 6) & 1) * 25); }

constexpr int kFbankStrip = 704;   // floats per frame in flight: 256 complex + 192 slot sums

// DPP forms of the wave-level exchanges (round 3 used __shfl_*: ds_bpermute, i.e. 13 of the 61 LDS instructions of a frame)
template <int CTRL, int RMASK, bool BC>
__device__ __forceinline__ float fb_dppx(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, RMASK, 0xf, BC));
}
// sum over the wave, the same value in every lane (uniform: it comes back through a scalar register)
__device__ __forceinline__ float fb_wave_sum(float s) {
  s += fb_dppx<0xB1, 0xf, true>(s);                         // quad_perm [1,0,3,2]
  s += fb_dppx<0x4E, 0xf, true>(s);                         // quad_perm [2,3,0,1]
  s += fb_dppx<0x141, 0xf, true>(s);                        // row_half_mirror
  s += fb_dppx<0x140, 0xf, true>(s);                        // row_mirror: every lane holds its row's sum
  s += fb_dppx<0x142, 0xa, false>(s);                       // row_bcast:15 -> rows 1 and 3 take in rows 0 and 2
  s += fb_dppx<0x143, 0xc, false>(s);                       // row_bcast:31 -> rows 2 and 3 take in row 1: lane 63 holds the sum
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s), 63));
}
// ---- round 6: the butterflies as PACKED f32 instructions, written out.  A complex value is one 64-bit register pair and every
// complex operation of the transform is one or two v_pk_*_f32 with the right op_sel / neg modifiers: -i (a1 - a3) is an operand
// swap + one sign inside the add that consumes it, a complex product is v_pk_mul + v_pk_fma.  The compiler's own lowering of the
// float2 source found the packed adds but built a complex product from v_pk_mov (swap the twiddle) + v_pk_mul + TWO v_pk_fma (one
// per sign pattern) + a v_mov to pick a half of each, and transposed the inputs of the first stages through v_mov / v_pk_mov pairs:
// 32 instead of 14 vector instructions per twiddled butterfly, 78 register moves per frame (ISA of round 6a).
// OPERAND ORDER: no instruction here takes its LOW result from (first register source's low half, second register source's high half)
// -- on gfx950 that pattern returns lanes 48..63 without the second source's contribution whenever MFMA waves share the SIMD (another
// stream's kernel is enough; constants and scalar registers do not count as sources): pk_safe.hip.h.  The commutative operands are
// written the other way round: the operand whose high half feeds the low result comes first.
// gfx940+ forwarding hazard: the result of a packed (VOP3P) instruction must not be read by the very next VALU instruction
// (LLVM's hasDstSelForwardingHazard: it puts an s_nop 0 there itself -- but cannot look into an asm block).  Inside the blocks
// dependent instructions are >= 2 apart; every block starts and ends with an s_nop 0 for the instructions around it.
typedef float fb_f2 __attribute__((ext_vector_type(2)));
#define FB_SUB " neg_lo:[0,1] neg_hi:[0,1]"
#define FB_CMUL1(r, a, w) "v_pk_mul_f32 %[" #r "], %[" #a "], %[" #w "] op_sel_hi:[0,1]\n\t"
#define FB_CMUL2(r, a, w) "v_pk_fma_f32 %[" #r "], %[" #a "], %[" #w "], %[" #r "] op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t"
// radix-4 DIF butterfly: o0 = a0 + a1 + a2 + a3, r1 = w1 (a0 - i a1 - a2 + i a3), r2 = w2 (a0 - a1 + a2 - a3), r3 = w3 (a0 + i a1 - a2 - i a3)
// Registers are re-used by hand (an asm block's early-clobber outputs cannot share registers with its inputs, and the kernel lives
// at 128): a0 -> t0 -> o2, a1 -> t2 -> r3, a2 -> o1, a3 -> o3, n1 = t1 -> o0, n2 = d -> r1, n3 = r2: seven pairs.
template <bool TW>
__device__ __forceinline__ void fb_bfly(fb_f2 a0, fb_f2 a1, fb_f2 a2, fb_f2 a3, fb_f2 w1, fb_f2 w2, fb_f2 w3,
                                        fb_f2& o0, fb_f2& r1, fb_f2& r2, fb_f2& r3) {
  fb_f2 n1, n2, n3;
  if constexpr (TW) {
    asm("s_nop 0\n\t"
        "v_pk_add_f32 %[n1], %[a0], %[a2]" FB_SUB "\n\t"                                     // t1
        "v_pk_add_f32 %[n2], %[a1], %[a3]" FB_SUB "\n\t"                                     // d
        "v_pk_add_f32 %[a0], %[a0], %[a2]\n\t"                                               // t0
        "v_pk_add_f32 %[a1], %[a1], %[a3]\n\t"                                               // t2
        "v_pk_add_f32 %[a2], %[n2], %[n1] op_sel:[1,0] op_sel_hi:[0,1] neg_hi:[1,0]\n\t"      // o1 = (d.y, -d.x) + t1
        "v_pk_add_f32 %[a3], %[n2], %[n1] op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[1,0]\n\t"      // o3 = (-d.y, d.x) + t1
        "v_pk_add_f32 %[n1], %[a0], %[a1]\n\t"                                               // o0
        "v_pk_add_f32 %[a0], %[a0], %[a1]" FB_SUB "\n\t"                                     // o2
        FB_CMUL1(n2, a2, w1) FB_CMUL1(a1, a3, w3) FB_CMUL1(n3, a0, w2)
        FB_CMUL2(n2, a2, w1) FB_CMUL2(a1, a3, w3) FB_CMUL2(n3, a0, w2)
        "s_nop 0"
        : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [n1] "=&v"(n1), [n2] "=&v"(n2), [n3] "=&v"(n3)
        : [w1] "v"(w1), [w2] "v"(w2), [w3] "v"(w3));
    o0 = n1; r1 = n2; r2 = n3; r3 = a1;
  } else {
    asm("s_nop 0\n\t"
        "v_pk_add_f32 %[n1], %[a0], %[a2]" FB_SUB "\n\t"
        "v_pk_add_f32 %[n2], %[a1], %[a3]" FB_SUB "\n\t"
        "v_pk_add_f32 %[a0], %[a0], %[a2]\n\t"
        "v_pk_add_f32 %[a1], %[a1], %[a3]\n\t"
        "v_pk_add_f32 %[a2], %[n2], %[n1] op_sel:[1,0] op_sel_hi:[0,1] neg_hi:[1,0]\n\t"
        "v_pk_add_f32 %[a3], %[n2], %[n1] op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[1,0]\n\t"
        "v_pk_add_f32 %[n1], %[a0], %[a1]\n\t"
        "v_pk_add_f32 %[a0], %[a0], %[a1]" FB_SUB "\n\t"
        "s_nop 0"
        : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [n1] "=&v"(n1), [n2] "=&v"(n2));
    o0 = n1; r1 = a2; r2 = a0; r3 = a3;
  }
}
// real-FFT untangle of TWO bins (fbank.h:173-175): X[k] = (Zk + conj(Zn)) / 2 - i w^k (Zk - conj(Zn)) / 2, n = 256 - k.  With
// e = Zk + conj(Zn), o = Zk - conj(Zn) and the twiddle held as u = -i w^k / 2 (exact):  X = e / 2 + u o  -- 5 packed instructions per
// bin, all with plain operand order in the last one (the float2 source compiled to 14 plain instructions).  ROT: bins k + 128, whose
// twiddle is -i times the bins k's -- u' = -i u, taken from the same registers with other operand selects:
// u' o = (u.y o.x + u.x o.y, u.y o.y - u.x o.x).  zk -> e -> X, zn -> the product.  (The first version kept w^k / 2 and swapped the
// product's halves inside the last FMA -- x = e * 0.5 + (p.hi, -p.lo), op_sel:[0,0,1] next to a constant: that IS the hazard of
// pk_safe.hip.h, the constant does not count as a register source -- found by tests/test_hip_fbank.py's tenant test.)
template <bool ROT>
__device__ __forceinline__ void fb_untangle2(fb_f2 zka, fb_f2 zna, fb_f2 ua, fb_f2 zkb, fb_f2 znb, fb_f2 ub, fb_f2& xa, fb_f2& xb) {
  fb_f2 oa, ob;
  if constexpr (!ROT) {
    asm("s_nop 0\n\t"
        "v_pk_add_f32 %[oa], %[zka], %[zna] neg_lo:[0,1]\n\t"
        "v_pk_add_f32 %[ob], %[zkb], %[znb] neg_lo:[0,1]\n\t"
        "v_pk_add_f32 %[zka], %[zka], %[zna] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %[zkb], %[zkb], %[znb] neg_hi:[0,1]\n\t"
        "v_pk_mul_f32 %[zna], %[oa], %[ua] op_sel_hi:[1,0]\n\t"                                               // (o.x u.x, o.y u.x)
        "v_pk_mul_f32 %[znb], %[ob], %[ub] op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %[zna], %[oa], %[ua], %[zna] op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n\t"      // (-o.y u.y + ., o.x u.y + .)
        "v_pk_fma_f32 %[znb], %[ob], %[ub], %[znb] op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n\t"
        "v_pk_fma_f32 %[zka], %[zka], 0.5, %[zna] op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %[zkb], %[zkb], 0.5, %[znb] op_sel_hi:[1,0,1]\n\t"
        "s_nop 0"
        : [zka] "+v"(zka), [zna] "+v"(zna), [zkb] "+v"(zkb), [znb] "+v"(znb), [oa] "=&v"(oa), [ob] "=&v"(ob)
        : [ua] "v"(ua), [ub] "v"(ub));
  } else {
    asm("s_nop 0\n\t"
        "v_pk_add_f32 %[oa], %[zka], %[zna] neg_lo:[0,1]\n\t"
        "v_pk_add_f32 %[ob], %[zkb], %[znb] neg_lo:[0,1]\n\t"
        "v_pk_add_f32 %[zka], %[zka], %[zna] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %[zkb], %[zkb], %[znb] neg_hi:[0,1]\n\t"
        "v_pk_mul_f32 %[zna], %[ua], %[oa] op_sel:[1,0] op_sel_hi:[1,1]\n\t"                                  // (u.y o.x, u.y o.y)
        "v_pk_mul_f32 %[znb], %[ub], %[ob] op_sel:[1,0] op_sel_hi:[1,1]\n\t"
        "v_pk_fma_f32 %[zna], %[oa], %[ua], %[zna] op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_hi:[1,0,0]\n\t"      // (o.y u.x + ., -o.x u.x + .)
        "v_pk_fma_f32 %[znb], %[ob], %[ub], %[znb] op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_hi:[1,0,0]\n\t"
        "v_pk_fma_f32 %[zka], %[zka], 0.5, %[zna] op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %[zkb], %[zkb], 0.5, %[znb] op_sel_hi:[1,0,1]\n\t"
        "s_nop 0"
        : [zka] "+v"(zka), [zna] "+v"(zna), [zkb] "+v"(zkb), [znb] "+v"(znb), [oa] "=&v"(oa), [ob] "=&v"(ob)
        : [ua] "v"(ua), [ub] "v"(ub));
  }
  xa = zka; xb = zkb;
}
#undef FB_UNT
#undef FB_SUB
#undef FB_CMUL1
#undef FB_CMUL2
#ifndef WEKWS_FBANK_FW
#define WEKWS_FBANK_FW 1
#endif
constexpr int kFbankFW = WEKWS_FBANK_FW;   // frames per wave, interleaved through every phase (round 4)

// S: sample type of the PCM in device memory -- float (int16 scale, what wav.h:98-102 hands over) or int16_t (what
// FeaturePipeline::AcceptWaveform(const std::vector<int16_t>&), feature_pipeline.cc:49-55, receives: converted here,
// in registers, so the upload and the HBM read are 2 bytes per sample)
//
// Round 4: a wave works on TWO frames at a time (g, g + 1), phase by phase.  The kernel is bound by neither unit (by
// instruction counts ~40 % vector, ~50 % LDS) but by the chain load -> butterfly -> store -> wave-level sync of every FFT
// stage; with two independent frames between two syncs each lane has twice the work to cover the LDS round trips, the syncs
// per frame halve, and the window / twiddle / mel-weight registers serve both frames.  Same arithmetic per frame, bit for bit.
// (two slot rounds -- 80 mel bins: 140 registers = three waves per SIMD; capped at 128 = four, 24 bytes of scratch: 548 -> 528 us per
//  8192 x 1 s, round 6.  Three rounds would spill 136 bytes: left alone.)
template <int ROUNDS, typename S>
__global__ __launch_bounds__(64 * kFbankWaves, ROUNDS == 2 ? 4 : 1) void fbank_kernel(const FbankParams P, const S* __restrict__ pcm,
                                                                 int B, int nsamp, int nframes,
                                                                 float* __restrict__ feats) {
  constexpr int FW = kFbankFW;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* const strip = lds + wave * FW * kFbankStrip;       // [FW] strips of this wave
  const float* __restrict__ tab = P.tables;
  const float2* tw256 = reinterpret_cast<const float2*>(tab);
  const float2* tw512 = reinterpret_cast<const float2*>(tab + 512);
  const float* win = tab + 1024;
  const int FL = P.frame_length;
  const float inv_fl = 1.f / float(FL);

  // ---- per-lane constants of the whole launch (registers): window of its 8 samples, stage twiddles, untangle twiddles,
  //      scatter targets of its 4 FFT bins
  float wn[8];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int n = lane + 64 * m;
    wn[2 * m] = win[2 * n];
    wn[2 * m + 1] = win[2 * n + 1];
  }
  fb_f2 tws[3][3];                                          // stages 0..2: w^j, w^2j, w^3j of the stage's block size
#pragma unroll
  for (int st = 0; st < 3; ++st) {
    const int q = 64 >> (2 * st), L = 4 * q, j = lane % q, tstep = 256 / L;
#pragma unroll
    for (int r = 1; r <= 3; ++r) {
      const float2 t = tw256[(r * j * tstep) & 255];
      tws[st][r - 1] = fb_f2{t.x, t.y};
    }
  }
  fb_f2 twh[2];                                             // untangle twiddles of bins lane, lane + 64: u = -i w^k / 2 (exact); bins + 128: -i u
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const float2 t = tw512[lane + 64 * m];
    twh[m] = fb_f2{0.5f * t.y, -0.5f * t.x};
  }
  // mel slots of this lane: first FFT bin, mel bin, 16 weights (zero padded)
  int sfirst[ROUNDS], sbin[ROUNDS], scount[ROUNDS];          // scount: slots of the filter if this slot is its first, else 0
  float sw[ROUNDS][16];
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const int slot = lane + 64 * r;
    const bool ok = slot < P.nslots;
    sfirst[r] = ok ? int(tab[P.slot_first_off + slot]) : 0;
    sbin[r] = ok ? int(tab[P.slot_bin_off + slot]) : -1;
    scount[r] = 0;
    if (ok && (slot == 0 || int(tab[P.slot_bin_off + slot - 1]) != sbin[r])) {
      int n = 1;
      while (slot + n < P.nslots && int(tab[P.slot_bin_off + slot + n]) == sbin[r]) ++n;
      scount[r] = n;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) sw[r][u] = ok ? tab[P.slot_w_off + slot * 16 + u] : 0.f;
  }
  // where the last stage puts its four outputs: X[64 m + rev3(lane)], natural order
  const int rev3 = ((lane & 3) << 4) | (lane & 12) | (lane >> 4);
  // Round 6: the swizzled strip positions of a lane are the same for every frame -- computed ONCE and held (made opaque, or the
  // compiler re-derives them per frame: ~100 of the ~320 vector instructions of a frame were this address arithmetic).
  int zpos[4][4];                                            // stage st exchanges elements base + k q (k = 0..3); stage 3 writes natural order
#pragma unroll
  for (int st = 0; st < 4; ++st) {
    const int q = 64 >> (2 * st), blk = lane / q, j = lane - blk * q, base = blk * 4 * q + j;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      zpos[st][k] = pz(base + k * q);
      asm volatile("" : "+v"(zpos[st][k]));
    }
  }
  int zout[4], unpos[4];
  const int (&upos)[4] = zpos[0];                           // (stage 0 writes elements lane + 64 k: the untangle's own bins)
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    zout[m] = pz(64 * m + rev3);
    unpos[m] = pz((256 - (lane + 64 * m)) & 255);
    asm volatile("" : "+v"(zout[m]), "+v"(unpos[m]));
  }

  // Frame g of the launch is frame g % nframes of utterance g / nframes.  The wave walks g = g0, g0 + stride, ...: the pair
  // (utterance, frame) is carried and advanced by (stride / nframes, stride % nframes) with a carry -- round 3 divided a
  // 64-bit index twice per frame (~100 of its ~450 instructions, tools/probe PMC + ISA).
  const int64_t total = int64_t(B) * nframes;
  const int64_t stride = int64_t(gridDim.x) * kFbankWaves * FW;
  const int sq = int(stride / nframes), sr = int(stride - int64_t(sq) * nframes);
  const int64_t f_first = (int64_t(blockIdx.x) * kFbankWaves + wave) * FW;
  int ub = int(f_first / nframes), ufr = int(f_first - int64_t(ub) * nframes);     // of the wave's CURRENT first frame
  // the sample pairs of a wave's NEXT frames are requested before the current ones are processed
  float2 vn[FW][4];
  // even frame length and shift, aligned buffer: every element is one aligned pair inside or outside the frame -- loads under a lane
  // mask, no branches (round 6: the general form below compiled to 25 branches per frame)
  const bool pair_ok = (nsamp % 2 == 0) && (P.frame_shift % 2 == 0) && (FL % 2 == 0) && (reinterpret_cast<uintptr_t>(pcm) % (2 * sizeof(S)) == 0);
  auto fetch = [&](int b0, int fr0) __attribute__((always_inline)) {
#pragma unroll
    for (int w = 0; w < FW; ++w) {
      int b = b0, fr = fr0 + w;
      if (fr >= nframes) { fr -= nframes; ++b; }             // (FW <= nframes: at most one carry)
      const bool live = b < B;
      const S* src = pcm + int64_t(b) * nsamp + int64_t(fr) * P.frame_shift;
      if (pair_ok) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int i = 2 * (lane + 64 * m);
          const bool in = live && i < FL;
          const S* at = in ? src + i : pcm;                   // (outside: any valid pair, dropped)
          float2 v;
          if constexpr (sizeof(S) == 4) {
            v = *reinterpret_cast<const float2*>(at);
          } else {
            const short2 q = *reinterpret_cast<const short2*>(at);
            v = make_float2(float(q.x), float(q.y));
          }
          vn[w][m] = in ? v : make_float2(0.f, 0.f);
        }
      } else {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int i = 2 * (lane + 64 * m);
          float2 v = make_float2(0.f, 0.f);
          if (live) {
            if (i < FL) v.x = float(src[i]);
            if (i + 1 < FL) v.y = float(src[i + 1]);
          }
          vn[w][m] = v;
        }
      }
    }
  };
  fetch(ub, ufr);
  for (int64_t f = f_first; f < total; f += stride) {
    // (utterance, frame) of the wave's next first frame
    int nb = ub + sq, nfr = ufr + sr;
    if (nfr >= nframes) { nfr -= nframes; ++nb; }
    // ---- DC removal (fbank.h:155-160)
    float2 v[FW][4];
    float mean[FW];
#pragma unroll
    for (int w = 0; w < FW; ++w) {
      float s = 0.f;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        v[w][m] = vn[w][m];
        s += v[w][m].x + v[w][m].y;
      }
      mean[w] = s;
    }
    fetch(nb, nfr);
#pragma unroll
    for (int w = 0; w < FW; ++w) mean[w] = fb_wave_sum(mean[w]);
    // ---- pre-emphasis 0.97 (fbank.h:122-127: y[i] = x[i] - 0.97 x[i-1], y[0] = x[0] - 0.97 x[0]), window
    //      (fbank.h:130-135).  x[2n-1] is the odd sample of element n-1: the lane below (lane 0: lane 63 of m-1)
    float2 a[FW][4];
#pragma unroll
    for (int w = 0; w < FW; ++w) {
      const float mu = mean[w] * inv_fl;   // (one rounding away from the quotient; the sum's order differs from the reference's serial one anyway)
      // (no range masks: samples past the frame were fetched as zeros and their window values ARE zero -- whatever the
      //  mean subtraction and the pre-emphasis make of them is multiplied away; round 3 spent 27 selects per frame on them)
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const float xe = v[w][m].x - mu;
        const float xo = v[w][m].y - mu;
        float prev = fb_dppx<0x138, 0xf, true>(xo);                           // wave_shr:1: odd sample of element n - 1
        if (m > 0) {
          const float top = v[w][m - 1].y - mu;                                // lane 0: lane 63 of the row of elements before
          const float wrap = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, top), 63));
          prev = lane == 0 ? wrap : prev;
        } else {
          prev = lane == 0 ? xe : prev;                                       // i = 0: its own value
        }
        a[w][m].x = (xe - 0.97f * prev) * wn[2 * m];
        a[w][m].y = (xo - 0.97f * xe) * wn[2 * m + 1];
      }
    }
    // ---- 256-point complex FFT, radix-4 DIF, 4 stages (stage st exchanges elements q = 64 >> 2 st apart)
#pragma unroll
    for (int st = 0; st < 4; ++st) {
#pragma unroll
      for (int w = 0; w < FW; ++w) {
        fb_f2* const z = reinterpret_cast<fb_f2*>(strip + w * kFbankStrip);
        fb_f2 a0, a1, a2, a3, o0, o1, o2, o3;
        if (st > 0) {
          a0 = z[zpos[st][0]]; a1 = z[zpos[st][1]]; a2 = z[zpos[st][2]]; a3 = z[zpos[st][3]];
        } else {
          a0 = fb_f2{a[w][0].x, a[w][0].y}; a1 = fb_f2{a[w][1].x, a[w][1].y};
          a2 = fb_f2{a[w][2].x, a[w][2].y}; a3 = fb_f2{a[w][3].x, a[w][3].y};
        }
        if (st < 3) {
          fb_bfly<true>(a0, a1, a2, a3, tws[st][0], tws[st][1], tws[st][2], o0, o1, o2, o3);
          z[zpos[st][0]] = o0; z[zpos[st][1]] = o1; z[zpos[st][2]] = o2; z[zpos[st][3]] = o3;
        } else {
          // positions 4 lane + m hold X[rev4(4 lane + m)] = X[64 m + rev3(lane)]: stored in natural order
          fb_bfly<false>(a0, a1, a2, a3, a0, a0, a0, o0, o1, o2, o3);
          z[zout[0]] = o0; z[zout[1]] = o1; z[zout[2]] = o2; z[zout[3]] = o3;
        }
      }
      wave_sync();
    }
    // ---- real-FFT untangle + power (fbank.h:173-175)
    float pw[FW][4];
#pragma unroll
    for (int w = 0; w < FW; ++w) {
      const fb_f2* const z = reinterpret_cast<const fb_f2*>(strip + w * kFbankStrip);
#pragma unroll
      for (int m = 0; m < 4; m += 2) {
        fb_f2 xa, xb;
        if (m == 0) fb_untangle2<false>(z[upos[0]], z[unpos[0]], twh[0], z[upos[1]], z[unpos[1]], twh[1], xa, xb);
        else fb_untangle2<true>(z[upos[2]], z[unpos[2]], twh[0], z[upos[3]], z[unpos[3]], twh[1], xa, xb);
        pw[w][m] = xa.x * xa.x + xa.y * xa.y;
        pw[w][m + 1] = xb.x * xb.x + xb.y * xb.y;
      }
    }
    wave_sync();                                          // every lane has read its Z values: the strips are free
#pragma unroll
    for (int w = 0; w < FW; ++w) {
      float* const pwr = strip + w * kFbankStrip;            // 256 power bins (+ 16 zeros of slack for the padded slots)
#pragma unroll
      for (int m = 0; m < 4; ++m) pwr[lane + 64 * m] = pw[w][m];
      if (lane < 16) pwr[256 + lane] = 0.f;
    }
    wave_sync();
    // ---- mel (fbank.h:179-186): one slot per lane and round, ascending k, then the slots of a filter in order
#pragma unroll
    for (int w = 0; w < FW; ++w) {
      const float* const pwr = strip + w * kFbankStrip;
      float* const melacc = strip + w * kFbankStrip + 512;
#pragma unroll
      for (int r = 0; r < ROUNDS; ++r) {
        float e = 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) e = fmaf(sw[r][u], pwr[sfirst[r] + u], e);
        melacc[lane + 64 * r] = e;                            // slot sums (slots past nslots: 0)
      }
    }
    wave_sync();
    // ---- log (fbank.h:187-190), store.  Slots are sorted by filter: a filter's slots are neighbours.
#pragma unroll
    for (int w = 0; w < FW; ++w) {
      int b = ub, fr = ufr + w;
      if (fr >= nframes) { fr -= nframes; ++b; }
      if (b < B) {
        const int64_t g = int64_t(b) * nframes + fr;
        const float* const melacc = strip + w * kFbankStrip + 512;
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
          const int slot = lane + 64 * r;
          if (scount[r] > 0) {
            float e = melacc[slot];
            for (int n = 1; n < scount[r]; ++n) e += melacc[slot + n];
#ifdef WEKWS_FBANK_FASTLOG
            feats[g * P.num_bins + sbin[r]] = __logf(fmaxf(e, FLT_EPSILON));
#else
            feats[g * P.num_bins + sbin[r]] = logf(fmaxf(e, FLT_EPSILON));
#endif
          }
        }
      }
    }
    wave_sync();
    ub = nb; ufr = nfr;
  }
}

// persistent waves: exactly one resident round (a second, partly filled round costs a whole wave lifetime).  The number
// of resident workgroups is a property of (kernel, device): looked up by the caller once per extractor (fbank_create).
template <typename S>
inline int fbank_resident_groups(const FbankParams& P) {
  const int rounds = (P.nslots + 63) / 64;
  if (rounds < 1 || rounds > 3) return 0;
  const size_t lds = size_t(kFbankWaves * kFbankFW * kFbankStrip) * sizeof(float);
  auto kern = rounds == 1 ? fbank_kernel<1, S> : rounds == 2 ? fbank_kernel<2, S> : fbank_kernel<3, S>;
  int per_cu = 0, dev = 0;
  hipDeviceProp_t prop;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 64 * kFbankWaves, lds) == hipSuccess &&
      hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && per_cu > 0)
    return per_cu * prop.multiProcessorCount;
  return 256 * 4;
}

template <typename S>
inline int launch_fbank(const FbankParams& P, const S* pcm, int B, int nsamp, int nframes, float* feats, int resident,
                        hipStream_t stream) {
  const int rounds = (P.nslots + 63) / 64;
  if (rounds < 1 || rounds > 3) return -4;
  const size_t lds = size_t(kFbankWaves * kFbankFW * kFbankStrip) * sizeof(float);
  const int64_t total = int64_t(B) * nframes;
  int64_t grid = (total + kFbankWaves * kFbankFW - 1) / (kFbankWaves * kFbankFW);
  auto kern = rounds == 1 ? fbank_kernel<1, S> : rounds == 2 ? fbank_kernel<2, S> : fbank_kernel<3, S>;
  if (resident > 0 && grid > resident) grid = resident;
  hipLaunchKernelGGL(kern, dim3(unsigned(grid)), dim3(64 * kFbankWaves), lds, stream, P, pcm, B, nsamp, nframes, feats);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace wekws
