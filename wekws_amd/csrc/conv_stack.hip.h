// Fused conv-backbone forward for gfx950 (MI355X): preprocessing Linear -> residual conv blocks ->
// classifier head, ONE kernel launch per (batch, <=112-frame tile).  Activations never leave the CU:
// the (channels x frames) tile of every utterance lives in LDS, the 1x1 / dense convolutions run on
// the exact-f32 matrix cores (v_mfma_f32_16x16x4_f32), the depthwise dilated taps, folded BatchNorm,
// ReLU, residual and sigmoid run on the VALU around them.
//
// Reference semantics implemented here (paths relative to the reference tree):
//   KWSModel.forward                       wekws/model/kws_model.py:65-76
//   LinearSubsampling1.forward             wekws/model/subsampling.py:53-57
//   TCN.forward / Block.forward            wekws/model/tcn.py:139-166, :35-61
//   DsCnnBlock.cnn / CnnBlock.cnn          wekws/model/tcn.py:101-114, :75-84
//   MDTC.forward / TCNStack / TCNBlock     wekws/model/mdtc.py:242-276, :181-198, :95-121
//   DSDilatedConv1d.forward                wekws/model/mdtc.py:55-59
//   Linear/Global/Last classifiers         wekws/model/classifier.py:63-67, :26-28, :38-40
//
// Work decomposition (wave64, 512-thread workgroups = 2 waves per SIMD):
//   - a workgroup owns U utterances (C=256/128: 1, C=64: 2, C=32: 4);
//   - GEMM view of a 1x1 conv:  D[o][t] = sum_c W[o][c] * a[c][t];  MFMA A operand = weights
//     (pre-packed on the host into per-lane fragment order, streamed from L2 with 16-byte loads),
//     B operand = activations read from LDS rows (channel-major, row stride == 16 mod 32 so that the
//     4-row x 16-frame fragment read is bank-conflict free), D = 4 consecutive output channels of one
//     frame per lane;
//   - each wave owns OW o-tiles (16 output channels each) x NT t-tiles (16 frames each) of one
//     utterance and keeps those accumulators in registers for a whole layer;
//   - the K dimension is produced in chunks of KC rows into a double-buffered LDS slab by all waves
//     (depthwise taps from the resident tile, causal halo from the streaming cache in global memory
//     or zeros), one barrier per chunk, so VALU/LDS production of chunk n+1 overlaps the MFMAs of
//     chunk n issued by the partner wave on the same SIMD.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <type_traits>

#include "nonfinite.hip.h"
#include "pk_safe.hip.h"

namespace wekws {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is needed once per (kernel, device).  A launcher keeps one static
// DynLdsGrant: the bytes already granted per device ordinal.  Thread-safe (two racing first calls both set the same
// attribute, which is idempotent) and correct for a process that drives several GPUs -- a plain `static bool` is neither.
constexpr int kMaxDevices = 64;
struct DynLdsGrant {
  std::atomic<int> bytes[kMaxDevices];
};
template <class K>
inline int grant_dynamic_lds(K kern, int bytes, DynLdsGrant& g) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return -3;
  if (g.bytes[dev].load(std::memory_order_acquire) >= bytes) return 0;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess)
    return -3;
  g.bytes[dev].store(bytes, std::memory_order_release);
  return 0;
}

enum : int { KIND_DS = 0, KIND_TCN = 1, KIND_MDTC = 2 };
enum : int { HEAD_LINEAR = 0, HEAD_GLOBAL = 1, HEAD_LAST = 2, HEAD_IDENTITY = 3 };

constexpr int kThreads = 512;
constexpr int kWaves = 8;

// One residual block.  Offsets are in floats from StackParams::w.
struct BlockDesc {
  int32_t dil;        // dilation
  int32_t pad;        // (ksize-1)*dil  == frames of left context == this block's cache slice length
  int32_t cache_off;  // offset of this block's slice inside the cache's last axis
  int32_t zadd;       // MDTC: 1 if the block closes a stack (output is added to the stack sum)
  uint32_t dw_w;      // DS/MDTC: folded depthwise weights [C][ksize]
  uint32_t dw_b;      // DS/MDTC: folded depthwise bias [C]
  uint32_t a1;        // packed MFMA A operand of the first GEMM (DS: pointwise, TCN: dense conv, MDTC: pointwise)
  uint32_t b1;        // its folded bias [C]
  uint32_t a2;        // MDTC: packed A of conv2
  uint32_t b2;        // MDTC: folded bias of conv2 [C]
  uint32_t dw_pk;     // DS/MDTC: taps + bias per channel padded to a multiple of 4 floats: [C][round_up(ksize+1,4)]
  uint32_t a1_16;     // same matrices split into fp16 hi/lo and packed for v_mfma_f32_16x16x32_f16
  uint32_t a2_16;     //   ([o-tile][k32][hi|lo][lane][8 halves], conv_stack_f16.hip.h)
  // block floating point of the split-fp16 kernels (conv_stack_f16.hip.h): the packed fp16 matrices hold W * s with
  // s a power of two that puts max|W| into [2^14, 2^15); inv_s* = 1 / s undoes it in the epilogue
  float inv_s1, inv_s2;
  // depthwise output bound: |dw(u) + b| <= dw_alpha * max|u| + dw_beta  (max_c sum_j |w[c][j]|, max_c |b[c]|)
  float dw_alpha, dw_beta;
  // MDTC mid tile ReLU(BN1(pointwise(a))): |mid| <= mid_alpha * max|a| + mid_beta  (largest row 1-norm of the folded
  // pointwise matrix, largest |folded bias|).  Chained behind the depthwise bound: a bound that is 2^18 too loose still
  // keeps the split's error below 2^-22 of the tile maximum, so bounds may compound WITHIN a block; across blocks the
  // exact maximum of the residual tile is re-measured.
  float mid_alpha, mid_beta;
  // ds256_mm: the depthwise taps enter the matrix cores scaled by dw_tap_s (power of two); dw_tap_inv = 1 / dw_tap_s
  float dw_tap_s, dw_tap_inv;
};

struct StackParams {
  const float* w;           // device weights (folded + packed)
  const BlockDesc* blocks;  // device array
  int32_t nblocks;
  int32_t idim;             // feature dim
  int32_t kpre;             // idim rounded up to 16
  int32_t ksize;
  int32_t odim;
  int32_t pre_relu;
  uint32_t pre_a, pre_b;    // packed A [C/16][kpre/16][64][4], bias [C]
  int32_t kpre16;           // idim rounded up to 32 (fp16 K step)
  uint32_t pre_a16;         // fp16 hi/lo packed A of the preprocessing Linear
  int32_t head;             // HEAD_*
  int32_t head_hidden;
  int32_t sigmoid;
  uint32_t head_w, head_b;  // LINEAR: Wc[odim][C], bc;  GLOBAL/LAST: W1[hh][C], b1
  uint32_t head_w2, head_b2;  // GLOBAL/LAST: W2[odim][hh], b2
  int32_t cache_len;        // P = sum of pads
  float pre_inv_s;          // 1 / power-of-two scale of the packed fp16 preprocessing matrix (pre_a16)
  float head_inv_s;         // same for a matrix-core classifier (head_a16: ds256_mm, dense_stack_f16)
};

struct CallArgs {
  const float* x;         // first frame of this tile, utterance 0
  int64_t xs_b;           // floats between utterances in x
  const float* in_cache;  // (B, C, P) or nullptr
  float* out_cache;       // (B, C, P) or nullptr
  float* y;               // first output row of this tile (per-frame heads) / (B, odim) (GLOBAL/LAST)
  int64_t ys_b;           // floats between utterances in y
  float* gsum;            // GLOBAL head, multi-tile: running per-channel sums (B, C) or nullptr
  int32_t B;
  int32_t T;              // frames in this tile (1..16*NT)
  int32_t T_total;        // frames of the whole call (GLOBAL mean divisor)
  int32_t first_tile, last_tile;
  int32_t head_slices;    // ds256_mm, CTC-sized heads: gridDim.y workgroups per utterance share the head's o-tiles (0/1: off)
  const NfCtx* nf;        // utterances with a non-finite input are re-computed in exact IEEE f32 (nonfinite.hip.h)
};

// Utterance b of this call has a NaN / Inf in its features or incoming cache: the reference's arithmetic instead of the kernel's.
// NOT inlined: one copy per translation unit, called from a cold branch with nothing live (arguments by value, in registers).
static __device__ __attribute__((noinline, unused)) void nf_repair_conv_call(const NfCtx* nf, const float* x, int64_t xs_b, const float* ic,
                                                                             float* oc, float* y, int64_t ys_b, float* gsum, int T,
                                                                             int T_total, int first_tile, int last_tile, int b) {
  nf_repair_conv(nf, x, xs_b, ic, oc, y, ys_b, gsum, T, T_total, first_tile, last_tile, b);
}
__device__ __forceinline__ void nf_repair_call(const CallArgs& A, int b) {
#ifdef WEKWS_NF_NOCALL                                       // (A/B builds only: the detection without the re-computation)
  return;
#endif
  nf_repair_conv_call(A.nf, A.x, A.xs_b, A.in_cache, A.out_cache, A.y, A.ys_b, A.gsum, A.T, A.T_total, A.first_tile, A.last_tile, b);
}

// PERSISTENT kernels (a workgroup walks over utterances): an utterance with a non-finite input is noted here and re-computed
// behind the loop, where nothing is live -- the hot loop gets one scalar compare per utterance.
struct NfList {
  int n, flag;
  int list[30];
};
__device__ __forceinline__ void nf_list_init(NfList& L) {
  if (threadIdx.x == 0) { L.n = 0; L.flag = 0; }
}
// Called by ALL threads of the workgroup, which all decided to come here from the same LDS words behind the same barrier; the
// barrier inside keeps those words until every wave has read them.
__device__ __forceinline__ void nf_list_note(NfList& L, int b) {
  __syncthreads();
  if (threadIdx.x == 0) {
    if (L.n < 30) L.list[L.n] = b;
    ++L.n;
    L.flag = 0;
  }
}
// Behind the loop.  `per_wg` utterances per loop step start at b = first, first + stride, ... (first = blockIdx.x * per_wg).
__device__ inline void nf_list_drain(NfList& L, const CallArgs& A, int idim, int cache_elems, int first, int stride, int per_wg = 1) {
  __syncthreads();
  const int nbad = __builtin_amdgcn_readfirstlane(L.n);
  if (nbad == 0) return;
  if (nbad <= 30) {
    for (int i = 0; i < nbad; ++i) nf_repair_call(A, __builtin_amdgcn_readfirstlane(L.list[i]));
    return;
  }
  // more than the list holds: look at every utterance of this workgroup again
  __shared__ unsigned nf_cell;
  for (int b0 = first; b0 < A.B; b0 += stride)
    for (int u = 0; u < per_wg && b0 + u < A.B; ++u) {
      const int bb = b0 + u;
      bool bad = nf_scan_rows(A.x + int64_t(bb) * A.xs_b, A.T, idim, idim, &nf_cell);
      if (!bad && A.in_cache) bad = nf_scan(A.in_cache + int64_t(bb) * cache_elems, cache_elems, &nf_cell);
      if (bad) nf_repair_call(A, bb);
    }
}

template <int KIND, int C, int NT>
struct Geom {
  static constexpr int U = (C >= 128) ? 1 : (128 / C);              // utterances per workgroup
  static constexpr int OT = C / 16;                                  // o-tiles per utterance
  static constexpr int OW = (OT * U >= 16) ? 2 : 1;                  // o-tiles per wave
  static constexpr int WO = OT / OW;                                 // waves along o
  static constexpr int WU = kWaves / WO;                             // waves along utterances
  static_assert(WO * WU == kWaves && WU == U, "wave decomposition");
  static constexpr int SS = (NT % 2) ? 16 * NT : 16 * NT + 16;       // LDS row stride (== 16 mod 32)
  static constexpr int KC = (C >= 64) ? 32 : 16;                     // K rows per produced chunk
  static constexpr int R = (KIND == KIND_MDTC) ? (C > 2 * KC ? C : 2 * KC) : 2 * KC;  // slab rows / utt
  static constexpr int H_FLOATS = U * C * SS;
  static constexpr int S_FLOATS = U * R * SS;
  static constexpr size_t LDS_BYTES = size_t(H_FLOATS + S_FLOATS) * 4;
};

// ---------------------------------------------------------------------------------------------
// MFMA building blocks.  A fragments are 16-byte loads of the host-packed image: for o-tile `ot`, 16-row
// K group `g`, lane l holds W[ot*16 + (l&15)][g*16 + s*4 + (l>>4)], s = 0..3 (one float per k-step).
// ---------------------------------------------------------------------------------------------
template <int OW, int NG>
__device__ __forceinline__ void load_a(float4 (&a)[NG][OW], const float4* __restrict__ ap, int ot_stride) {
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int ow = 0; ow < OW; ++ow) a[g][ow] = ap[ow * ot_stride + g * 64];
}

// acc += A(groups) x B(rows of the LDS slab).  bl: LDS pointer to (first K row + lane>>4, frame lane&15).
// The B fragments of k-step i+1 are requested before the MFMAs of k-step i are issued (one step = OW*NT MFMAs,
// >= 224 cycles of matrix pipe), so a wave that owns the pipe alone never waits on LDS latency.
template <int OW, int NT, int SS, int NG>
__device__ __forceinline__ void mfma_groups(f32x4 (&acc)[OW][NT], const float4 (&a)[NG][OW], const float* bl) {
  constexpr int STEPS = NG * 4;
  float b[2][NT];
#pragma unroll
  for (int tt = 0; tt < NT; ++tt) b[0][tt] = bl[tt * 16];
#pragma unroll
  for (int i = 0; i < STEPS; ++i) {
    const int g = i / 4, sidx = i % 4;
    if (i + 1 < STEPS) {
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) b[(i + 1) & 1][tt] = bl[((i + 1) * 4) * SS + tt * 16];
    }
    // pin: [reads of step i+1] then [MFMAs of step i]; without it the scheduler re-merges the two buffers and
    // re-issues each read two MFMAs before its use (an lgkmcnt stall per step when the wave runs alone)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ow = 0; ow < OW; ++ow) {
      const float av = sidx == 0 ? a[g][ow].x : sidx == 1 ? a[g][ow].y : sidx == 2 ? a[g][ow].z : a[g][ow].w;
#pragma unroll
      for (int tt = 0; tt < NT; ++tt)
        acc[ow][tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[i & 1][tt], acc[ow][tt], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// runtime group count (preprocessing GEMM: K = idim rounded up to 16)
template <int OW, int NT, int SS>
__device__ __forceinline__ void mfma_rows(f32x4 (&acc)[OW][NT], const float4* __restrict__ ap, int ot_stride,
                                          const float* bl, int nk16) {
  for (int g = 0; g < nk16; ++g) {
    float4 a[1][OW];
    load_a<OW, 1>(a, ap + g * 64, ot_stride);
    mfma_groups<OW, NT, SS, 1>(acc, a, bl + g * 16 * SS);
  }
}

template <int OW, int NT>
__device__ __forceinline__ void zero_acc(f32x4 (&acc)[OW][NT]) {
#pragma unroll
  for (int ow = 0; ow < OW; ++ow)
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) acc[ow][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
}

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + __expf(-v)); }

// The depthwise producers let a 16-lane group walk NT outputs per lane at stride d ("slide": the taps move over
// register-resident inputs); possible when d divides 16, i.e. d is a power of two <= 16 -- so lane tl's first frame
// (tl / d) NT d + tl % d is a shift and a mask.  (The division by a run-time d was ~40 vector instructions per block and
// wave: 7 % of an MDTC block's cycles went into the block top, per-phase stamps at steady clocks.)
__device__ __forceinline__ bool slide_ok(int d) { return d <= 16 && (d & (d - 1)) == 0; }
__device__ __forceinline__ int slide_base(int tl, int d, int nt) {
  const int sh = __builtin_ctz(unsigned(d));
  return (((tl >> sh) * nt) << sh) + (tl & (d - 1));
}
__device__ __forceinline__ float f4c(const float4& q, int r) { return r == 0 ? q.x : r == 1 ? q.y : r == 2 ? q.z : q.w; }

// ---------------------------------------------------------------------------------------------
// Classifier head on the resident tile (shared by the f32 and the split-fp16 kernels).  `slab` is free scratch.
// ---------------------------------------------------------------------------------------------
// The classifier of a small linear head, one value per thread, requested EARLY by the latency kernels (streaming steps) so
// that the head does not start with a trip to L2; conv_stack_head() takes it instead of staging the weights itself.
struct HeadPre {
  float w, b;
  bool valid;
};
template <int KIND, int C, int NT, int NTHR>
__device__ __forceinline__ HeadPre conv_stack_head_prefetch(const StackParams& P) {
  using G = Geom<KIND, C, NT>;
  HeadPre hp = {0.f, 0.f, false};
  const int K = P.odim, tid = threadIdx.x;
  if (P.head == HEAD_LINEAR && K * (C + 1) <= G::S_FLOATS && K * C <= NTHR) {
    hp.valid = true;
    if (tid < K * C) hp.w = P.w[P.head_w + tid];
    if (tid < K) hp.b = P.w[P.head_b + tid];
  }
  return hp;
}

// SSX: row stride of the resident tile when the caller lays it out itself (0: Geom's).  SWZ: row c starts at
// c * SS + (c & 3) (ds256_stream.hip.h: the skew spreads four-row groups over the LDS banks).
template <int KIND, int C, int NT, int NTHR = kThreads, int SSX = 0, bool SWZ = false>
__device__ __forceinline__ void conv_stack_head(const StackParams& P, const CallArgs& A, float* hbuf, float* slab, int b0,
                                                const HeadPre* pre = nullptr) {
  using G = Geom<KIND, C, NT>;
  constexpr int U = G::U, SS = SSX ? SSX : G::SS;
  const int tid = threadIdx.x;
  const int T = A.T;
  const float* __restrict__ W = P.w;
  const int K = P.odim;
  if (P.head == HEAD_LINEAR) {
    // y[t][k] = act(sum_c Wc[k][c] h[c][t] + bc[k])                         (classifier.py:63-67)
    // small heads: classifier weights staged in the (now free) slab, read back as LDS broadcasts
    const bool staged = K * (C + 1) <= G::S_FLOATS;
    if (staged) {
      if (pre && pre->valid) {
        if (tid < K * C) slab[tid] = pre->w;
        if (tid < K) slab[K * C + tid] = pre->b;
      } else {
        for (int e = tid; e < K * C; e += NTHR) slab[e] = W[P.head_w + e];
        for (int e = tid; e < K; e += NTHR) slab[K * C + e] = W[P.head_b + e];
      }
      __syncthreads();
    }
    const float* wsrc = staged ? slab : W + P.head_w;
    const float* bsrc = staged ? slab + K * C : W + P.head_b;
    const int nout = U * K * T;
    // few outputs (K = 1..2 keywords): split the channel sum over PARTS threads per output and combine through the
    // slab tail, so that all eight waves work instead of the first three.  (16 parts on a 10-frame streaming step
    // measured slower than 4: the second stage's serial sum grows faster than the first stage shrinks.)
    const int PARTS = (staged && nout * 4 <= NTHR && K * (C + 1) + 4 * nout <= G::S_FLOATS) ? 4
                    : (staged && nout * 2 <= NTHR && K * (C + 1) + 2 * nout <= G::S_FLOATS) ? 2 : 1;
    float* part = slab + K * (C + 1);
    for (int e0 = tid; e0 < nout * PARTS; e0 += NTHR) {
      const int e = e0 % nout, ps = e0 / nout;
      const int t = e % T;
      const int uk = e / T;
      const int u = uk / K, k = uk - u * K;
      const int c0 = ps * (C / PARTS), c1 = c0 + C / PARTS;
      const float* hc = hbuf + u * C * SS + t;
      const float* wk = wsrc + k * C;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll 4
      for (int c = c0; c < c1; c += 4) {
        const float4 w4 = *reinterpret_cast<const float4*>(wk + c);
        s0 = fmaf(w4.x, hc[(c + 0) * SS], s0);               // (c % 4 == 0: the skew of row c + i is i)
        s1 = fmaf(w4.y, hc[(c + 1) * SS + (SWZ ? 1 : 0)], s1);
        s2 = fmaf(w4.z, hc[(c + 2) * SS + (SWZ ? 2 : 0)], s2);
        s3 = fmaf(w4.w, hc[(c + 3) * SS + (SWZ ? 3 : 0)], s3);
      }
      const float sum = (s0 + s1) + (s2 + s3);
      if (PARTS > 1) {
        part[ps * nout + e] = sum;
      } else if (b0 + u < A.B) {
        float v = sum + bsrc[k];
        if (P.sigmoid) v = sigmoidf_(v);
        A.y[int64_t(b0 + u) * A.ys_b + int64_t(t) * K + k] = v;
      }
    }
    if (PARTS > 1) {
      __syncthreads();
      for (int e = tid; e < nout; e += NTHR) {
        const int t = e % T;
        const int uk = e / T;
        const int u = uk / K, k = uk - u * K;
        if (b0 + u >= A.B) continue;
        float v = bsrc[k];
        for (int ps = 0; ps < PARTS; ++ps) v += part[ps * nout + e];
        if (P.sigmoid) v = sigmoidf_(v);
        A.y[int64_t(b0 + u) * A.ys_b + int64_t(t) * K + k] = v;
      }
    }
  } else if (P.head == HEAD_IDENTITY) {
    for (int e = tid; e < U * T * C; e += NTHR) {
      const int c = e % C;
      const int ut = e / C;
      const int u = ut / T, t = ut - u * T;
      if (b0 + u >= A.B) continue;
      float v = hbuf[(u * C + c) * SS + (SWZ ? (c & 3) : 0) + t];
      if (P.sigmoid) v = sigmoidf_(v);
      A.y[int64_t(b0 + u) * A.ys_b + int64_t(t) * C + c] = v;
    }
  } else {
    // GLOBAL: m = mean_t h ; LAST: m = h[:, -1]  ->  W2 ReLU(W1 m + b1) + b2   (classifier.py:26-28, :38-40)
    float* mvec = slab;                 // [U][C]
    float* hid = slab + U * C;          // [U][head_hidden]
    const int HH = P.head_hidden;
    for (int e = tid; e < U * C; e += NTHR) {
      const int u = e / C, c = e - u * C;
      const float* hc = hbuf + (u * C + c) * SS + (SWZ ? (c & 3) : 0);
      float s;
      if (P.head == HEAD_GLOBAL) {
        s = 0.f;
        for (int t = 0; t < T; ++t) s += hc[t];
        if (A.gsum && (b0 + u) < A.B) {
          float* gp = A.gsum + int64_t(b0 + u) * C + c;
          if (!A.first_tile) s += *gp;
          if (!A.last_tile) *gp = s;
        }
        s = s / float(A.T_total);
      } else {
        s = hc[T - 1];
      }
      mvec[e] = s;
    }
    __syncthreads();
    if (A.last_tile) {
      for (int e = tid; e < U * HH; e += NTHR) {
        const int u = e / HH, j = e - u * HH;
        const float* w1 = W + P.head_w + j * C;
        float s = W[P.head_b + j];
        for (int c = 0; c < C; ++c) s = fmaf(w1[c], mvec[u * C + c], s);
        hid[e] = nf_relu(s);                                 // (a NaN carried in by the running sums of an earlier tile stays one)
      }
      __syncthreads();
      for (int e = tid; e < U * K; e += NTHR) {
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

// ---------------------------------------------------------------------------------------------
template <int KIND, int C, int NT, int KS>
__global__ __launch_bounds__(kThreads, 2) void conv_stack_kernel(const StackParams P, const CallArgs A) {
  using G = Geom<KIND, C, NT>;
  constexpr int U = G::U, OW = G::OW, SS = G::SS, KC = G::KC, R = G::R;
  constexpr int NG = KC / 16;                       // 16-row K groups per produced chunk
  constexpr int RP = (U * KC) / (kThreads / 16);    // slab rows each 16-lane group produces per chunk
  static_assert(RP * (kThreads / 16) == U * KC && RP >= 1, "producer decomposition");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const hbuf = lds;                 // [U][C][SS]   resident activations h_i
  float* const slab = lds + G::H_FLOATS;   // [U][R][SS]   GEMM B-operand rows

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform (SGPR)
  const int wo = wave % G::WO;             // which o-tile group
  const int wu = wave / G::WO;             // which utterance of the workgroup
  const int l15 = lane & 15, lq = lane >> 4;
  const int T = A.T;
  const int b0 = blockIdx.x * U;           // first utterance of this workgroup
  const float* __restrict__ W = P.w;
  const int Pc = P.cache_len;

  // producer mapping: 16 lanes share a slab row, each lane covers frames tl, tl+16, ...
  const int pg = tid >> 4, tl = tid & 15;

  // this wave's accumulator geometry
  const int o_base = wo * OW * 16;         // first output channel of the wave
  float* const h_w = hbuf + wu * C * SS;   // the wave's utterance tile
  const float* const slab_w = slab + wu * R * SS + lq * SS + l15;

  f32x4 acc[OW][NT];
  f32x4 zsum[KIND == KIND_MDTC ? OW : 1][KIND == KIND_MDTC ? NT : 1];
  if constexpr (KIND == KIND_MDTC) zero_acc(zsum);

  // ---- a NaN / Inf among the features or the incoming caches of this workgroup's utterances: the reference's IEEE arithmetic
  //      for all of them (nonfinite.hip.h).  This kernel has no operand maxima to ride on: one pass over its inputs (L2).
  {
    __shared__ unsigned nf_cell;
    bool bad = false;
    for (int u = 0; u < U && b0 + u < A.B; ++u) {
      bad = bad || nf_scan_rows(A.x + int64_t(b0 + u) * A.xs_b, T, P.idim, P.idim, &nf_cell);
      if (!bad && A.in_cache) bad = nf_scan(A.in_cache + int64_t(b0 + u) * C * Pc, int64_t(C) * Pc, &nf_cell);
    }
    if (bad) {
      for (int u = 0; u < U && b0 + u < A.B; ++u) nf_repair_call(A, b0 + u);
      return;
    }
  }

  // ============================ preprocessing: h0 = [ReLU](x Wpre^T + b) ============================
  {
    zero_acc(acc);
    const int ot_stride = (P.kpre / 16) * 64;
    const float4* ap = reinterpret_cast<const float4*>(W + P.pre_a) + (wo * OW) * ot_stride + lane;
    const float4 bias0 = *reinterpret_cast<const float4*>(W + P.pre_b + o_base + lq * 4);
    const float4 bias1 = *reinterpret_cast<const float4*>(W + P.pre_b + o_base + (OW - 1) * 16 + lq * 4);
    for (int k0 = 0; k0 < P.kpre; k0 += R) {
      const int rows = min(R, P.kpre - k0);
      __syncthreads();
      // slab[u][k][t] = x[b0+u][t][k0+k]  (zero beyond idim / T / B): one frame per wave per step, lanes along k
      for (int u = 0; u < U; ++u) {
        const bool ok = (b0 + u) < A.B;
        const float* xu = A.x + int64_t(b0 + u) * A.xs_b + k0;
        constexpr int FPW = (16 * NT + kWaves - 1) / kWaves;  // frames per wave
        for (int k = lane; k < rows; k += 64) {
          float xv[FPW];
#pragma unroll
          for (int i = 0; i < FPW; ++i) {   // all loads of the pass in flight together
            const int t = wave + i * kWaves;
            xv[i] = (ok && t < T && k0 + k < P.idim) ? xu[t * P.idim + k] : 0.f;
          }
#pragma unroll
          for (int i = 0; i < FPW; ++i) {
            const int t = wave + i * kWaves;
            if (t < 16 * NT) slab[(u * R + k) * SS + t] = xv[i];
          }
        }
      }
      __syncthreads();
      mfma_rows<OW, NT, SS>(acc, ap + (k0 / 16) * 64, ot_stride, slab_w, rows / 16);
    }
#pragma unroll
    for (int ow = 0; ow < OW; ++ow) {
      const int o = o_base + ow * 16 + lq * 4;
      const float4 bias = ow == 0 ? bias0 : bias1;
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        const int t = tt * 16 + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[ow][tt][r] + f4c(bias, r);
          if (P.pre_relu) v = fmaxf(v, 0.f);
          h_w[(o + r) * SS + t] = v;  // frames >= T hold finite don't-care values from here on
        }
      }
    }
    __syncthreads();
  }

  // ======================================= residual blocks =======================================
  for (int bi = 0; bi < P.nblocks; ++bi) {
    const BlockDesc bd = P.blocks[bi];
    const int d = bd.dil, pad = bd.pad;
    const int K1 = (KIND == KIND_TCN) ? C * KS : C;
    const int nch = K1 / KC;
    const int ot_stride1 = (K1 / 16) * 64;
    const float4* ap1 = reinterpret_cast<const float4*>(W + bd.a1) + (wo * OW) * ot_stride1 + lane;

    // register-resident depthwise taps (+bias) of the rows this thread produces next, fetched ahead of use
    float dww[KIND == KIND_TCN ? 1 : RP][KIND == KIND_TCN ? 1 : KS + 1];
    auto load_dw = [&](int n) __attribute__((always_inline)) {
      if constexpr (KIND != KIND_TCN) {
#pragma unroll
        for (int i = 0; i < RP; ++i) {
          const int item = pg + i * (kThreads / 16);
          const int c = n * KC + (item % KC);
#pragma unroll
          for (int j = 0; j < KS; ++j) dww[i][j] = W[bd.dw_w + c * KS + j];
          dww[i][KS] = W[bd.dw_b + c];
        }
      }
    };
    float4 a0[NG][OW], a1[NG][OW];
    load_dw(0);
    load_a<OW, NG>(a0, ap1, ot_stride1);
    // epilogue bias, requested now so its latency is long gone when the layer's MFMAs finish
    float4 ebias[OW];
#pragma unroll
    for (int ow = 0; ow < OW; ++ow)
      ebias[ow] = *reinterpret_cast<const float4*>(W + (KIND == KIND_MDTC ? bd.b2 : bd.b1) + o_base + ow * 16 + lq * 4);

    // ---- producer of K-chunk n into slab buffer `buf` (rows buf*KC .. buf*KC+KC-1 of every utterance).
    // 16 lanes share a slab row.  Lane l owns the NT output frames  f_i = base + i*d  (a run at stride = the
    // dilation; base = (l/d)*NT*d + l%d tiles [0, 16*NT) for every d that divides 16), so the KS taps of
    // consecutive outputs slide over the same inputs: NT+KS-1 LDS reads per row instead of NT*KS, all in
    // flight together, one latency exposure.  Left context (index < 0 relative to the tile start) comes from
    // the streaming cache in global memory when there is one, else it is zero (tcn.py:49-52); the read is
    // issued unconditionally at a clamped index and the result selected (branch-free).
    // The same lanes also hand the row's streaming cache over: new_cache = last `pad` frames of [cache | h]
    // (tcn.py:54, mdtc.py:112) -- spread over the layer instead of one write burst per layer.
    const bool slide = slide_ok(d);
    const int fbase = slide ? slide_base(tl, d, NT) : tl;   // first output frame of this lane
    const int fstep = slide ? d : 16;                             // distance between its output frames
    auto produce_impl = [&](int n, int buf, auto has_cache_tag) __attribute__((always_inline)) {
      constexpr bool HAS_CACHE = decltype(has_cache_tag)::value;
#pragma unroll
      for (int i = 0; i < RP; ++i) {
        const int item = pg + i * (kThreads / 16);
        const int u = item / KC, r = item % KC;
        float* dst = slab + (u * R + buf * KC + r) * SS;
        const bool uok = (b0 + u) < A.B;
        const int kk = n * KC + r;
        const int c = (KIND == KIND_TCN) ? kk / KS : kk;
        const int j0 = (KIND == KIND_TCN) ? kk % KS : 0;
        const int hoff = (u * C + c) * SS;
        const int64_t gbase = (int64_t(uok ? b0 + u : 0) * C + c) * Pc + bd.cache_off;
        // [cache | h] at frame idx (idx < 0: left context); both reads unconditional at clamped indices
#define fetch(idx_)                                                                      \
  ({                                                                                     \
    const int ix_ = (idx_);                                                              \
    float fv_ = hbuf[hoff + max(ix_, 0)];                                                \
    if constexpr (HAS_CACHE) {                                                           \
      const float fg_ = A.in_cache[gbase + pad + min(ix_, -1)];                          \
      fv_ = ix_ >= 0 ? fv_ : (uok ? fg_ : 0.f);                                          \
    } else {                                                                             \
      fv_ = ix_ >= 0 ? fv_ : 0.f;                                                        \
    }                                                                                    \
    fv_;                                                                                 \
  })
        if (A.out_cache && uok && j0 == 0) {
          for (int p = tl; p < pad; p += 16) {
            const int src = T + p - pad;  // index into h (negative: still inside the old cache)
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
        if constexpr (KIND == KIND_TCN) {
          // dense conv as GEMM over K' = (c, j):  row = h[c][t - (KS-1-j)*d]      (tcn.py:76-80)
          const int sh = (KS - 1 - j0) * d;
#pragma unroll
          for (int m = 0; m < NT; ++m) {
            const int t = tl + 16 * m;
            const float v = fetch(t - sh);
            dst[t] = (t < T) ? v : 0.f;
          }
        } else {
          // depthwise dilated conv + folded BN (+ReLU for DS-TCN)        (tcn.py:102-109, mdtc.py:55-58)
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
              dst[t] = (t < T) ? o : 0.f;
            }
          } else {
            // generic dilation (does not divide 16): one frame at a time
#pragma unroll 1
            for (int m = 0; m < NT; ++m) {
              const int t = tl + 16 * m;
              float o = dww[i][KS];
#pragma unroll
              for (int j = 0; j < KS; ++j) o = fmaf(dww[i][j], fetch(t - (KS - 1 - j) * d), o);
              if (KIND == KIND_DS) o = fmaxf(o, 0.f);
              dst[t] = (t < T) ? o : 0.f;
            }
          }
        }
      }
    };
#undef fetch
    const bool has_cache = A.in_cache != nullptr;
    auto produce = [&](int n, int buf) __attribute__((always_inline)) {
      if (has_cache) produce_impl(n, buf, std::true_type{});
      else produce_impl(n, buf, std::false_type{});
    };
    (void)fstep;

    // ---- GEMM 1 over K (= C, or C*KS for the dense conv): double-buffered chunks, one barrier per chunk.
    // Measured on MI355X (tools/ablate.sh): the f32 MFMA stream and the VALU/LDS producer do NOT overlap on a
    // SIMD even when they come from different waves (time = MFMA + producer, also with the two waves of a SIMD
    // staggered into different phases), so the producer is kept short instead of hidden.
    zero_acc(acc);
    produce(0, 0);
    load_dw(1);
    __syncthreads();
    // Two chunks per trip with ping-pong fragment registers (a0 / a1), no register copies, so the fragments
    // consumed by chunk n were requested a whole chunk earlier.  nch is even for every built shape.  Prefetches
    // are UNCONDITIONAL (chunk index clamped): a branch around a load makes the compiler's s_waitcnt placement
    // fall back to the conservative count at the join and drain the prefetch queue in front of the MFMAs.
    for (int n = 0; n < nch; n += 2) {
      produce(n + 1, 1);
      load_a<OW, NG>(a1, ap1 + (n + 1) * NG * 64, ot_stride1);
      load_dw(min(n + 2, nch - 1));
      __builtin_amdgcn_sched_barrier(0);  // keep the prefetch loads ahead of the MFMA block
      mfma_groups<OW, NT, SS, NG>(acc, a0, slab_w);
      __syncthreads();
      if (n + 2 < nch) produce(n + 2, 0);
      load_a<OW, NG>(a0, ap1 + min(n + 2, nch - 1) * NG * 64, ot_stride1);
      load_dw(min(n + 3, nch - 1));
      __builtin_amdgcn_sched_barrier(0);
      mfma_groups<OW, NT, SS, NG>(acc, a1, slab_w + KC * SS);
      __syncthreads();
    }

    if constexpr (KIND == KIND_MDTC) {
      // ---- mid = ReLU(BN1(pointwise))  -> slab rows [0, C) ; then conv2 (1x1) + BN2      (mdtc.py:113-116)
      constexpr int NG2 = C / 16;
      float4 a2[NG2][OW];
      load_a<OW, NG2>(a2, reinterpret_cast<const float4*>(W + bd.a2) + (wo * OW) * (NG2 * 64) + lane, NG2 * 64);
      float* mid_w = slab + wu * R * SS;
#pragma unroll
      for (int ow = 0; ow < OW; ++ow) {
        const int o = o_base + ow * 16 + lq * 4;
        const float4 bias = *reinterpret_cast<const float4*>(W + bd.b1 + o);
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
          const int t = tt * 16 + l15;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = acc[ow][tt][r] + f4c(bias, r);
            mid_w[(o + r) * SS + t] = fmaxf(v, 0.f);
          }
        }
      }
      __syncthreads();
      zero_acc(acc);
      mfma_groups<OW, NT, SS, NG2>(acc, a2, slab_w);
    }

    // ---- epilogue: bias (+ReLU) + residual, in place into h
#pragma unroll
    for (int ow = 0; ow < OW; ++ow) {
      const int o = o_base + ow * 16 + lq * 4;
      const float4 bias = ebias[ow];
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        const int t = tt * 16 + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[ow][tt][r] + f4c(bias, r);
          float* hp = h_w + (o + r) * SS + t;
          if constexpr (KIND == KIND_MDTC) {
            v = fmaxf(v + *hp, 0.f);        // ReLU(out + inputs)            (mdtc.py:117-118)
            if (bd.zadd) zsum[ow][tt][r] += v;  // sum of stack outputs       (mdtc.py:270-273)
          } else {
            v = fmaxf(v, 0.f) + *hp;        // y + x, nothing after the add   (tcn.py:60)
          }
          *hp = v;
        }
      }
    }
    __syncthreads();
  }

  if constexpr (KIND == KIND_MDTC) {
    // backbone output = sum of the stack outputs: overwrite the resident tile with it
#pragma unroll
    for (int ow = 0; ow < OW; ++ow) {
      const int o = o_base + ow * 16 + lq * 4;
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        const int t = tt * 16 + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (t < T) h_w[(o + r) * SS + t] = zsum[ow][tt][r];
      }
    }
    __syncthreads();
  }

  conv_stack_head<KIND, C, NT>(P, A, hbuf, slab, b0);
}

// Row softmax over the last axis (KWSModel.forward_softmax, kws_model.py:89): one wave per row, two passes over the
// row -- online (max, rescaled sum) with 16-byte loads through a 4-byte-aligned type (rows of an odd-width matrix are
// only dword aligned), then normalise in place.
static __global__ __attribute__((unused)) void softmax_rows_kernel(float* y, int64_t rows, int K) {
  const int64_t row = int64_t(blockIdx.x) * (blockDim.x / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  float* p = y + row * K;
  struct __attribute__((packed, aligned(4))) V4 { float v[4]; };
  const int K4 = K & ~3;
  float mx = -INFINITY, s = 0.f;
  auto take = [&](float v) __attribute__((always_inline)) {
    if (v > mx) { s *= __expf(mx - v); mx = v; }
    s += __expf(v - mx);
  };
  for (int k = lane * 4; k < K4; k += 256) {
    const V4 q = *reinterpret_cast<const V4*>(p + k);
#pragma unroll
    for (int j = 0; j < 4; ++j) take(q.v[j]);
  }
  if (K4 + lane < K) take(p[K4 + lane]);
  float gm = mx;
  for (int off = 32; off > 0; off >>= 1) gm = fmaxf(gm, __shfl_xor(gm, off));
  float gs = (mx == -INFINITY) ? 0.f : s * __expf(mx - gm);
  for (int off = 32; off > 0; off >>= 1) gs += __shfl_xor(gs, off);
  const float inv = 1.0f / gs;
  for (int k = lane * 4; k < K4; k += 256) {
    V4 q = *reinterpret_cast<const V4*>(p + k);
#pragma unroll
    for (int j = 0; j < 4; ++j) q.v[j] = __expf(q.v[j] - gm) * inv;
    *reinterpret_cast<V4*>(p + k) = q;
  }
  if (K4 + lane < K) p[K4 + lane] = __expf(p[K4 + lane] - gm) * inv;
}

// launcher implemented per KIND in conv_stack_{ds,tcn,mdtc}.hip
template <int KIND>
int launch_conv_stack(int C, int nt, const StackParams& P, const CallArgs& A, hipStream_t stream);
template <> int launch_conv_stack<KIND_DS>(int, int, const StackParams&, const CallArgs&, hipStream_t);
template <> int launch_conv_stack<KIND_TCN>(int, int, const StackParams&, const CallArgs&, hipStream_t);
template <> int launch_conv_stack<KIND_MDTC>(int, int, const StackParams&, const CallArgs&, hipStream_t);

template <int KIND, int C, int NT>
inline int launch_one(const StackParams& P, const CallArgs& A, hipStream_t stream) {
  using G = Geom<KIND, C, NT>;
  constexpr int KS = (KIND == KIND_MDTC) ? 5 : 8;  // the kernel sizes of the reference recipes (tcn.yaml / mdtc.yaml)
  if (P.ksize != KS) return -4;
  static DynLdsGrant grant;
  auto kern = conv_stack_kernel<KIND, C, NT, KS>;
  if (grant_dynamic_lds(kern, int(G::LDS_BYTES), grant)) return -3;
  const int grid = (A.B + G::U - 1) / G::U;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), G::LDS_BYTES, stream, P, A);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

#define WEKWS_DISPATCH_NT(KIND, CC)                                            \
  switch (nt) {                                                                \
    case 1: return launch_one<KIND, CC, 1>(P, A, stream);                      \
    case 2: return launch_one<KIND, CC, 2>(P, A, stream);                      \
    case 4: return launch_one<KIND, CC, 4>(P, A, stream);                      \
    case 7: return launch_one<KIND, CC, 7>(P, A, stream);                      \
    default: return -1;                                                        \
  }

// hidden dims with a compiled kernel: the reference recipes use 32 / 64 / 256 (128 for GRU);
// MDTC keeps a second full-width tile in LDS, which does not fit at C = 256.
#define WEKWS_DEFINE_LAUNCHER(KIND, WITH256)                                   \
  template <>                                                                  \
  int launch_conv_stack<KIND>(int C, int nt, const StackParams& P, const CallArgs& A, hipStream_t stream) { \
    switch (C) {                                                               \
      case 32: WEKWS_DISPATCH_NT(KIND, 32)                                     \
      case 64: WEKWS_DISPATCH_NT(KIND, 64)                                     \
      case 128: WEKWS_DISPATCH_NT(KIND, 128)                                   \
      case 256: if constexpr (WITH256) { WEKWS_DISPATCH_NT(KIND, 256) } else return -4; \
      default: return -4;                                                      \
    }                                                                          \
  }

}  // namespace wekws
