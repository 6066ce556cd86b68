// NON-FINITE INPUTS (round 6).  The reference's forward is plain IEEE fp32 arithmetic with torch.relu(nan) = nan
// (wekws/model/tcn.py:101-114, mdtc.py:95-121, kws_model.py:65-76): a NaN / +Inf / -Inf feature or cache element travels through
// the causal receptive field of its utterance -- NaN posteriors from that frame on, NaN / Inf columns in the returned cache --
// and touches no other utterance.  The specialised kernels of this directory cannot do that by themselves: their ReLUs are
// v_max_f32 (returns the operand that is a number), the block-floating operand scales come from maxima, and fp16 hi/lo splits
// turn Inf into NaN.  Until round 5 such an utterance came back with finite, meaningless scores.
//
// What happens now, in every kernel family:
//   * DETECTION rides on work the kernel does anyway: the per-utterance maximum of the feature tile / the incoming cache is
//     taken on the BIT PATTERNS of |v| (unsigned compare: for non-negative floats the integer order is the float order, and
//     NaN / Inf patterns -- 0x7f800000 and above -- sort above every finite value), so "max >= 0x7f800000" = "something is not
//     finite".  One v_and + v_max_u32 per input element instead of one v_max_f32; nothing per activation.
//   * An utterance whose inputs are finite takes the fast path, bit for bit as before.
//   * An utterance with a non-finite input is RE-COMPUTED by the workgroup that owns it with the routines of this file: the
//     reference's operator sequence in exact IEEE fp32 (v_fma_f32, ReLU as `v < 0 ? 0 : v`), straight from the host packer's
//     blob (the same blob the any-shape path of generic.hip.h runs on), activations in a global scratch slot.  No second
//     launch, no host round trip, nothing on the fast path but the compare.  Slow (a millisecond per utterance) and rare.
// The results are the reference's: same class (finite / NaN / +Inf / -Inf) at every position of y and of the returned cache,
// same finite values (tests/test_hip_nonfinite.py against goldens recorded from the live reference).
//
// Models that run zero-padded (wekws_hip.hip: pad_conv_shape, pad_gru_hidden -- widths / kernel sizes / hidden sizes no kernel
// is built for) carry exact-zero weights the caller's model does not have; 0 * NaN would spread the poison through taps and
// channels that do not exist.  For those models (NfCtx::skip_zero) a zero weight contributes nothing, which reproduces the
// unpadded model exactly.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/wekws_hip.h"

namespace wekws {

struct CallArgs;

struct NfCtx {
  wekws_hip_desc d;      // the shape the kernels run (padded shape for zero-padded models)
  const float* w;        // the packer's blob of that shape on the device (include/wekws_hip.h order; BatchNorm / CMVN folded)
  float* scratch;        // nslots x slot_floats
  unsigned* slots;       // [nslots] 0 = free, 1 = taken
  int32_t nslots;
  int32_t skip_zero;     // zero-padded model: a zero weight contributes nothing (see above)
  int64_t slot_floats;
  int32_t cache_len;     // conv: sum of paddings; fsmn: left_order - 1 + right_order
  int32_t pmax;          // longest padding of a block
  int32_t width;         // widest activation row
  int32_t tmax;          // frames per repair call the slot is sized for
  int32_t pre_diag;      // the preprocessing matrix is diagonal (NoSubsampling, subsampling.py:35-36, with CMVN folded in): the
                         // features ARE the hidden tile, channel by channel -- an Inf in one channel stays in that channel
};

// |v| as an unsigned integer: order-preserving for finite values, NaN / Inf on top
__device__ __forceinline__ unsigned nf_abs_bits(float v) { return __builtin_bit_cast(unsigned, v) & 0x7fffffffu; }
__device__ __forceinline__ bool nf_bad_bits(unsigned b) { return b >= 0x7f800000u; }
__device__ __forceinline__ bool nf_bad(float v) { return nf_bad_bits(nf_abs_bits(v)); }
// torch.relu: relu(NaN) = NaN, relu(-Inf) = 0, relu(+Inf) = +Inf
__device__ __forceinline__ float nf_relu(float v) { return v < 0.f ? 0.f : v; }
__device__ __forceinline__ float nf_sigmoid(float v) { return 1.0f / (1.0f + expf(-v)); }

// SANITISED loads (GRU kernels: sixteen streams share an MFMA tile and, per step, one operand scale -- an Inf in one stream's
// features would take the scale of its fifteen neighbours with it).  A non-finite input element enters the fast path as 0; the
// stream it belongs to is re-computed afterwards (gru_nf_fix_kernel), its neighbours never see it.
__device__ __forceinline__ float nf_clean(float v) { return nf_bad(v) ? 0.f : v; }
template <class V, int N>
__device__ __forceinline__ V nf_clean_vec(V v) {
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = nf_clean(v[i]);
  return v;
}

enum : int { NF_RELU = 1, NF_RES_AFTER = 2, NF_RES_BEFORE = 4, NF_SIGMOID = 8 };

// Y[t][n] = epilogue(sum_k W[n * w_ns + k * w_ks] X[t * xs + k] + bias[n]), t < Tn, n < N: one output per thread and pass.
// acc0 != nullptr: the sum starts from acc0[t * ys + n] and NO epilogue runs unless `last` (taps of a dense conv).
__device__ inline void nf_linear(const float* X, int64_t xs, const float* W, int64_t w_ns, int64_t w_ks, const float* bias,
                                 const float* R, int64_t rs, float* Y, int64_t ys, int Tn, int K, int N, int flags, bool skip0,
                                 bool accumulate = false, bool last = true) {
  const int nthr = blockDim.x;
  for (int e = threadIdx.x; e < Tn * N; e += nthr) {
    const int t = e / N, n = e - t * N;
    const float* xr = X + int64_t(t) * xs;
    const float* wr = W + int64_t(n) * w_ns;
    float acc = accumulate ? Y[int64_t(t) * ys + n] : 0.f;
#pragma unroll 1
    for (int k = 0; k < K; ++k) {                            // (not unrolled: the routine must stay small in registers, see NF_COLD)
      const float w = wr[int64_t(k) * w_ks];
      if (!(skip0 && w == 0.f)) acc = fmaf(w, xr[k], acc);
    }
    if (last) {
      if (bias) acc += bias[n];
      if ((flags & NF_RES_BEFORE) && R) acc += R[int64_t(t) * rs + n];
      if (flags & NF_RELU) acc = nf_relu(acc);
      if ((flags & NF_RES_AFTER) && R) acc += R[int64_t(t) * rs + n];
      if (flags & NF_SIGMOID) acc = nf_sigmoid(acc);
    }
    Y[int64_t(t) * ys + n] = acc;
  }
}

// preprocessing: LinearSubsampling1 (subsampling.py:53-57), or -- NoSubsampling -- the diagonal that carries the folded CMVN
__device__ inline void nf_preproc(const NfCtx* R, const float* xrow0, int64_t xs, const float* W, const float* bias, float* h, int64_t hs,
                                  int Tn) {
  const wekws_hip_desc& d = R->d;
  const int C = d.hdim;
  if (R->pre_diag) {
    for (int e = threadIdx.x; e < Tn * C; e += blockDim.x) {
      const int t = e / C, c = e - t * C;
      const float v = fmaf(W[int64_t(c) * d.idim + c], xrow0[int64_t(t) * xs + c], bias[c]);
      h[int64_t(t) * hs + c] = d.preproc_relu ? nf_relu(v) : v;
    }
    return;
  }
  nf_linear(xrow0, xs, W, d.idim, 1, bias, nullptr, 0, h, hs, Tn, d.idim, C, d.preproc_relu ? NF_RELU : 0, R->skip_zero != 0);
}

// One scratch slot for this workgroup (all threads call; returns the slot's base).  Holders never wait for anything, so a
// waiting workgroup always gets its turn.
__device__ inline float* nf_acquire(const NfCtx* R, int* slot_out) {
  __shared__ int nf_slot_s;
  __syncthreads();
  if (threadIdx.x == 0) {
    int got = -1;
    unsigned start = (blockIdx.x * 7u + blockIdx.y) % unsigned(R->nslots);
    while (got < 0) {
      for (int i = 0; i < R->nslots && got < 0; ++i) {
        const int s = int((start + unsigned(i)) % unsigned(R->nslots));
        if (atomicCAS(&R->slots[s], 0u, 1u) == 0u) got = s;
      }
      if (got < 0) __builtin_amdgcn_s_sleep(64);
    }
    __threadfence();
    nf_slot_s = got;
  }
  __syncthreads();
  *slot_out = nf_slot_s;
  return R->scratch + int64_t(nf_slot_s) * R->slot_floats;
}
__device__ inline void nf_release(const NfCtx* R, int slot) {
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) atomicExch(&R->slots[slot], 0u);
}

__device__ inline int nf_conv_nblocks(const wekws_hip_desc& d) {
  return d.backbone == WEKWS_HIP_BACKBONE_MDTC ? 1 + d.num_stack * d.stack_size : d.num_layers;
}
__device__ inline int nf_conv_dilation(const wekws_hip_desc& d, int i) {
  if (d.backbone == WEKWS_HIP_BACKBONE_MDTC) return i == 0 ? 1 : 1 << ((i - 1) % d.stack_size);
  return 1 << i;
}

// ---- conv backbones (DS-TCN, TCN, MDTC), one utterance `b` of one call tile: the operator sequence of generic_forward()
// (generic.hip.h:315-398) on (T, C) row-major activations in the slot.  Pointers are those of the kernel's CallArgs:
//   x        first frame of the tile, utterance 0 (+ b * xs_b);   y / ys_b likewise
//   in_cache / out_cache   (B, C, P) or nullptr;  gsum: GLOBAL head over several tiles (conv_stack_head's protocol)
__device__ inline void nf_repair_conv(const NfCtx* R, const float* x, int64_t xs_b, const float* in_cache, float* out_cache, float* y,
                                      int64_t ys_b, float* gsum, int T, int T_total, int first_tile, int last_tile, int b) {
  const wekws_hip_desc& d = R->d;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const bool s0 = R->skip_zero != 0;
  const int C = d.hdim, W = R->width, ks = d.kernel_size, Pc = R->cache_len;
  int slot;
  float* base = nf_acquire(R, &slot);
  const int64_t mat = int64_t(R->tmax) * W;
  float* h = base;
  float* o = base + mat;
  float* tm = base + 2 * mat;
  float* zs = base + 3 * mat;
  float* ub = base + 4 * mat;                                  // (pmax + T, C)
  const float* p = R->w;
  const float* xb = x + int64_t(b) * xs_b;
  const int act = d.activation == WEKWS_HIP_ACT_SIGMOID ? NF_SIGMOID : 0;

  nf_preproc(R, xb, d.idim, p, p + int64_t(C) * d.idim, h, C, T);
  p += int64_t(C) * d.idim + C;
  __syncthreads();
  const int nb = nf_conv_nblocks(d);
  int off = 0;
  bool zinit = true;
  for (int i = 0; i < nb; ++i) {
    const int dil = nf_conv_dilation(d, i), pad = (ks - 1) * dil;
    // u = [cache slice | h] (tcn.py:45-53, mdtc.py:98-104); the returned slice = its last `pad` rows
    for (int e = tid; e < (pad + T) * C; e += nthr) {
      const int c = e % C, tau = e / C;
      const int64_t ci = (int64_t(b) * C + c) * Pc + off;
      const float v = tau < pad ? (in_cache ? in_cache[ci + tau] : 0.f) : h[int64_t(tau - pad) * C + c];
      ub[e] = v;
      if (out_cache && tau >= T) out_cache[ci + (tau - T)] = v;
    }
    off += pad;
    __syncthreads();
    if (d.backbone == WEKWS_HIP_BACKBONE_TCN) {
      const float* wc = p; p += int64_t(C) * C * ks;           // W[o][c][j]
      const float* bc = p; p += C;
      for (int j = 0; j < ks; ++j) {
        const bool lastj = j == ks - 1;
        nf_linear(ub + int64_t(j) * dil * C, C, wc + j, int64_t(C) * ks, ks, bc, h, C, o, C, T, C, C, NF_RELU | NF_RES_AFTER, s0, j > 0,
                  lastj);
        __syncthreads();
      }
    } else {
      const float* wd = p; p += int64_t(C) * ks;
      const float* bd = p; p += C;
      const bool ds = d.backbone == WEKWS_HIP_BACKBONE_DS_TCN;
      for (int e = tid; e < T * C; e += nthr) {                // depthwise dilated conv (+ folded BN) [+ ReLU: DS-TCN only]
        const int c = e % C, t = e / C;
        float acc = bd[c];
#pragma unroll 1
        for (int j = 0; j < ks; ++j) {
          const float w = wd[int64_t(c) * ks + j];
          if (!(s0 && w == 0.f)) acc = fmaf(w, ub[int64_t(t + j * dil) * C + c], acc);
        }
        tm[e] = ds ? nf_relu(acc) : acc;
      }
      __syncthreads();
      if (ds) {
        const float* wp = p; p += int64_t(C) * C;
        const float* bp = p; p += C;
        nf_linear(tm, C, wp, C, 1, bp, h, C, o, C, T, C, C, NF_RELU | NF_RES_AFTER, s0);                  // tcn.py:101-114, :60
        __syncthreads();
      } else {                                                                                              // mdtc.py:95-121
        const float* w1 = p; p += int64_t(C) * C;
        const float* b1 = p; p += C;
        const float* w2 = p; p += int64_t(C) * C;
        const float* b2 = p; p += C;
        nf_linear(tm, C, w1, C, 1, b1, nullptr, 0, o, C, T, C, C, NF_RELU, s0);
        __syncthreads();
        nf_linear(o, C, w2, C, 1, b2, h, C, tm, C, T, C, C, NF_RELU | NF_RES_BEFORE, s0);
        __syncthreads();
        { float* t2 = tm; tm = o; o = t2; }                    // (the block's output is in `o`)
        if (i > 0 && (i - 1) % d.stack_size == d.stack_size - 1) {                                          // mdtc.py:270-273
          for (int e = tid; e < T * C; e += nthr) zs[e] = zinit ? o[e] : zs[e] + o[e];
          zinit = false;
          __syncthreads();
        }
      }
    }
    { float* t2 = h; h = o; o = t2; }
  }
  if (d.backbone == WEKWS_HIP_BACKBONE_MDTC) h = zs;

  float* yb = y + int64_t(b) * ys_b;
  if (d.head == WEKWS_HIP_HEAD_LINEAR) {
    nf_linear(h, C, p, C, 1, p + int64_t(d.odim) * C, nullptr, 0, yb, d.odim, T, C, d.odim, act, s0);
  } else if (d.head == WEKWS_HIP_HEAD_IDENTITY) {
    for (int e = tid; e < T * C; e += nthr) yb[e] = act ? nf_sigmoid(h[e]) : h[e];
  } else {
    const int HH = d.head_hidden;
    float* pooled = tm;
    for (int c = tid; c < C; c += nthr) {
      float s;
      if (d.head == WEKWS_HIP_HEAD_GLOBAL) {                   // classifier.py:27; several tiles: running sums in gsum
        s = 0.f;
        for (int t = 0; t < T; ++t) s += h[int64_t(t) * C + c];
        if (gsum) {
          float* gp = gsum + int64_t(b) * C + c;
          if (!first_tile) s += *gp;
          if (!last_tile) *gp = s;
        }
        s = s / float(T_total);
      } else {
        s = h[int64_t(T - 1) * C + c];                         // classifier.py:39
      }
      pooled[c] = s;
    }
    __syncthreads();
    if (last_tile) {
      float* hid = o;
      nf_linear(pooled, C, p, C, 1, p + int64_t(HH) * C, nullptr, 0, hid, HH, 1, C, HH, NF_RELU, s0);
      p += int64_t(HH) * C + HH;
      __syncthreads();
      nf_linear(hid, HH, p, HH, 1, p + int64_t(d.odim) * HH, nullptr, 0, yb, d.odim, 1, HH, d.odim, act, s0);
    }
  }
  nf_release(R, slot);
}

// ---- GRU (torch.nn.GRU built at kws_model.py:128-133; gate order r, z, n), one stream `b`, time-major through all layers:
// states (L, B, H) in / out, y (T, odim) per-frame linear head.
__device__ inline void nf_repair_gru(const NfCtx* R, const float* x, int64_t xs_b, const float* h0, float* hn, float* y, int64_t ys_b,
                                     int B, int T, int b) {
  const wekws_hip_desc& d = R->d;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const bool s0 = R->skip_zero != 0;
  const int H = d.hdim, L = d.num_layers;
  int slot;
  float* base = nf_acquire(R, &slot);
  float* hst = base;                 // [L][H]
  float* inp = hst + L * H;          // [H]
  float* gi = inp + H;               // [3H]
  float* gh = gi + 3 * H;            // [3H]
  for (int e = tid; e < L * H; e += nthr) {
    const int l = e / H, u = e - l * H;
    hst[e] = h0 ? h0[(int64_t(l) * B + b) * H + u] : 0.f;
  }
  __syncthreads();
  const float* wpre = R->w;
  const float* bpre = wpre + int64_t(H) * d.idim;
  const float* lay0 = bpre + H;
  const int64_t lstride = int64_t(6) * H * H + 6 * H;
  const float* wc = lay0 + lstride * L;
  const float* bcl = wc + int64_t(d.odim) * H;
  const int act = d.activation == WEKWS_HIP_ACT_SIGMOID ? NF_SIGMOID : 0;
  for (int t = 0; t < T; ++t) {
    nf_preproc(R, x + int64_t(b) * xs_b + int64_t(t) * d.idim, d.idim, wpre, bpre, inp, H, 1);
    __syncthreads();
    for (int l = 0; l < L; ++l) {
      const float* wih = lay0 + lstride * l;
      const float* whh = wih + int64_t(3) * H * H;
      const float* bih = whh + int64_t(3) * H * H;
      const float* bhh = bih + 3 * H;
      nf_linear(inp, H, wih, H, 1, bih, nullptr, 0, gi, 3 * H, 1, H, 3 * H, 0, s0);
      nf_linear(hst + l * H, H, whh, H, 1, bhh, nullptr, 0, gh, 3 * H, 1, H, 3 * H, 0, s0);
      __syncthreads();
      for (int u = tid; u < H; u += nthr) {
        const float r = 1.f / (1.f + expf(-(gi[u] + gh[u])));
        const float z = 1.f / (1.f + expf(-(gi[H + u] + gh[H + u])));
        const float c = tanhf(gi[2 * H + u] + r * gh[2 * H + u]);
        const float hv = (1.f - z) * c + z * hst[l * H + u];
        hst[l * H + u] = hv;
        inp[u] = hv;
      }
      __syncthreads();
    }
    nf_linear(inp, H, wc, H, 1, bcl, nullptr, 0, y + int64_t(b) * ys_b + int64_t(t) * d.odim, d.odim, 1, H, d.odim, act, s0);
    __syncthreads();
  }
  if (hn)
    for (int e = tid; e < L * H; e += nthr) {
      const int l = e / H, u = e - l * H;
      hn[(int64_t(l) * B + b) * H + u] = hst[e];
    }
  nf_release(R, slot);
}

// ---- FSMN (fsmn.py:462-495; cache (B, D, P, L), layer index innermost), one utterance `b` of one call tile
__device__ inline void nf_repair_fsmn(const NfCtx* R, const float* x, int64_t xs_b, const float* in_cache, float* out_cache, float* y,
                                      int64_t ys_b, int T, int b) {
  const wekws_hip_desc& d = R->d;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int C = d.hdim, W = R->width;
  const int A0 = d.aux[0], A1 = d.aux[1], D = d.num_stack, lo = d.kernel_size, ro = d.stack_size, P = lo - 1 + ro, L = d.num_layers;
  int slot;
  float* base = nf_acquire(R, &slot);
  const int64_t mat = int64_t(R->tmax) * W;
  float* h = base;
  float* o = base + mat;
  float* tm = base + 2 * mat;
  float* ub = base + 4 * mat;                                  // (P + T, D)
  const float* p = R->w;
  const int act = d.activation == WEKWS_HIP_ACT_SIGMOID ? NF_SIGMOID : 0;
  nf_linear(x + int64_t(b) * xs_b, d.idim, p, d.idim, 1, p + int64_t(A0) * d.idim, nullptr, 0, h, A0, T, d.idim, A0, 0, false);
  p += int64_t(A0) * d.idim + A0;
  __syncthreads();
  nf_linear(h, A0, p, A0, 1, p + int64_t(C) * A0, nullptr, 0, o, C, T, A0, C, NF_RELU, false);
  p += int64_t(C) * A0 + C;
  __syncthreads();
  { float* t2 = h; h = o; o = t2; }
  for (int l = 0; l < L; ++l) {
    const float* wproj = p; p += int64_t(D) * C;
    const float* taps = p; p += int64_t(D) * (lo + ro);
    const float* waff = p; p += int64_t(C) * D;
    const float* baff = p; p += C;
    nf_linear(h, C, wproj, C, 1, nullptr, nullptr, 0, tm, D, T, C, D, 0, false);
    __syncthreads();
    for (int e = tid; e < (P + T) * D; e += nthr) {
      const int c = e % D, tau = e / D;
      const int64_t ci = (int64_t(b) * D + c) * P * L + l;
      const float v = tau < P ? (in_cache ? in_cache[ci + int64_t(tau) * L] : 0.f) : tm[int64_t(tau - P) * D + c];
      ub[e] = v;
      if (out_cache && tau >= T) out_cache[ci + int64_t(tau - T) * L] = v;
    }
    __syncthreads();
    for (int e = tid; e < T * D; e += nthr) {                  // memory block (+ identity tap), fsmn.py:214-253
      const int c = e % D, t = e / D;
      float acc = 0.f;
#pragma unroll 1
      for (int j = 0; j < lo + ro; ++j) acc = fmaf(taps[int64_t(c) * (lo + ro) + j], ub[int64_t(t + j) * D + c], acc);
      tm[e] = acc;
    }
    __syncthreads();
    nf_linear(tm, D, waff, D, 1, baff, nullptr, 0, o, C, T, D, C, NF_RELU, false);
    __syncthreads();
    { float* t2 = h; h = o; o = t2; }
  }
  nf_linear(h, C, p, C, 1, p + int64_t(A1) * C, nullptr, 0, o, A1, T, C, A1, 0, false);
  p += int64_t(A1) * C + A1;
  __syncthreads();
  nf_linear(o, A1, p, A1, 1, p + int64_t(d.odim) * A1, nullptr, 0, y + int64_t(b) * ys_b, d.odim, T, A1, d.odim, act, false);
  nf_release(R, slot);
}

// ---- detection helpers -------------------------------------------------------------------------------------------------
// any non-finite value among n floats at p (all threads of the workgroup call; result is workgroup-uniform).  `cell`: one
// __shared__ unsigned the caller owns (zeroed here).
__device__ inline bool nf_scan(const float* p, int64_t n, unsigned* cell) {
  unsigned m = 0;
  for (int64_t e = threadIdx.x; e < n; e += blockDim.x) m = max(m, nf_abs_bits(p[e]));
  __syncthreads();
  if (threadIdx.x == 0) *cell = 0;
  __syncthreads();
  if (nf_bad_bits(m)) atomicMax(cell, m);
  __syncthreads();
  return nf_bad_bits(*cell);
}
// rows of `len` floats, `rows` of them, `stride` floats apart (the features of one utterance tile: T rows of idim)
__device__ inline bool nf_scan_rows(const float* p, int rows, int len, int64_t stride, unsigned* cell) {
  unsigned m = 0;
  for (int e = threadIdx.x; e < rows * len; e += blockDim.x) {
    const int r = e / len, k = e - r * len;
    m = max(m, nf_abs_bits(p[int64_t(r) * stride + k]));
  }
  __syncthreads();
  if (threadIdx.x == 0) *cell = 0;
  __syncthreads();
  if (nf_bad_bits(m)) atomicMax(cell, m);
  __syncthreads();
  return nf_bad_bits(*cell);
}

}  // namespace wekws
