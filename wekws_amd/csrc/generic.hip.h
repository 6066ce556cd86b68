// ANY-SHAPE path (round 5): every backbone / head / size the reference's init_model accepts (wekws/model/kws_model.py:97-214 takes
// any hidden_dim, kernel_size, num_layers ...) that the specialised kernels of this directory are NOT built for -- conv
// backbones wider than 256 channels (MDTC: 128), kernel sizes above 8 / 5, more residual blocks than the block-floating
// kernels track, GRU hidden sizes above 128 / more than 4 layers / pooled heads, FSMN layers that do not fit the LDS tile or
// whose weights fall outside the split-fp16 envelope -- and an FSMN for which precision F32 is requested.  Before round 5 those
// were WEKWS_HIP_EUNSUPPORTED (the reference runs them), resp. served with different rounding.
//
// One small family instead of one kernel per shape: activations live in HBM as (B, T, C) rows, every layer is a launch,
// all arithmetic is exact f32 (v_mfma_f32_16x16x4_f32 / v_fma_f32: every product exact, f32 accumulation, like the reference's fp32 math):
//   gen_gemm_kernel   Y = epilogue(X W^T + b): Linear / 1x1 conv / one tap of a dense conv; 64 x 64 tiles through LDS, f32 MFMA
//   gen_ctx_kernel    [cache | h] of a block as one (B, pad + T, C) buffer (tcn.py:45-53, mdtc.py:98-104, fsmn.py:228-236) and,
//                     from the same pass, the block's slice of the returned cache (its last `pad` rows)
//   gen_dw_kernel     depthwise dilated conv over that buffer (tcn.py:102-109, mdtc.py:55-58; the FSMN memory block
//                     fsmn.py:214-253 is the same sum with left_order + right_order taps and the identity folded into one of them)
//   gen_gru_cell_kernel   torch.nn.GRU's cell (gate order r, z, n) on gi = W_ih x + b_ih (all steps at once) and gh = W_hh h + b_hh
//   gen_mean_kernel / gen_last_kernel / gen_add_kernel    GlobalClassifier's mean (classifier.py:27), LastClassifier's row
//                     (classifier.py:39), MDTC's sum of stack outputs (mdtc.py:270-273)
// It is a correctness path, not a roofline one (20 .. 40 TFLOP/s: DS-TCN with 512 channels 162 k utt/s at B = 1024 x 98 frames):
// the recipes the reference ships all run on the specialised kernels.
// The weight blob is the host packer's (include/wekws_hip.h: BatchNorm and CMVN folded), uploaded as it is.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "../../include/wekws_hip.h"
#include "conv_stack.hip.h"

namespace wekws {

enum : int {
  GEN_RELU = 1,        // y = max(., 0)
  GEN_RES_AFTER = 2,   // y = act(.) + R          (tcn.py:60: residual after the ReLU)
  GEN_RES_BEFORE = 4,  // y = act(. + R)          (mdtc.py:117-118: residual before the ReLU)
  GEN_SIGMOID = 8,     // y = sigmoid(.)          (kws_model.py:196-199)
  GEN_ACCUM = 16,      // the product is added to what Y holds (further taps of a dense conv)
  GEN_PARTIAL = 32     // raw partial sums: no bias / epilogue (all taps of a dense conv but the last)
};

// Row (b, t) of X / R / Y sits at base + b * bs + t * rs; W[n][k] at W + n * w_ns + k * w_ks.
struct GenGemm {
  const float* X; int64_t x_bs, x_rs;
  const float* W; int64_t w_ns, w_ks;
  const float* bias;
  const float* R; int64_t r_bs, r_rs;
  float* Y; int64_t y_bs, y_rs;
  int Bn, Tn, K, N, flags;
};

constexpr int kGenTile = 64, kGenK = 16;

// 256 threads = 4 waves; workgroup tile 64 x 64, K in chunks of 16 through LDS; wave w multiplies rows 16 w .. 16 w + 15 by all
// 64 columns with v_mfma_f32_16x16x4_f32 (exact f32 products, f32 accumulate): A operand = lane's (row l % 16, k l / 16), B
// operand = (k l / 16, column l % 16), accumulator register i of lane l = (row 4 (l / 16) + i, column l % 16).
__global__ __launch_bounds__(256) void gen_gemm_kernel(const GenGemm g) {
  __shared__ float xs[kGenK][kGenTile + 4];
  __shared__ float ws[kGenK][kGenTile + 4];
  typedef float gen_f32x4 __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  const int64_t M = int64_t(g.Bn) * g.Tn;
  const int64_t m0 = int64_t(blockIdx.x) * kGenTile;
  const int n0 = blockIdx.y * kGenTile;
  // this thread's share of a staged tile: row (tid / 4) of the 64, four consecutive k
  const int lr = tid >> 2, lk = (tid & 3) * 4;
  const int64_t xm = m0 + lr;
  const float* xrow = nullptr;
  if (xm < M) { const int64_t b = xm / g.Tn, t = xm - b * g.Tn; xrow = g.X + b * g.x_bs + t * g.x_rs; }
  const int wn = n0 + lr;
  const float* wrow = wn < g.N ? g.W + int64_t(wn) * g.w_ns : nullptr;
  gen_f32x4 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = gen_f32x4{0.f, 0.f, 0.f, 0.f};
  // a thread's four k of a chunk are one 16-byte load where the operand's rows are 16-byte aligned runs (the usual case: row
  // strides and K multiples of four floats); the chunk after the current one is requested before the current one is multiplied
  const bool xvec = g.K % 4 == 0 && ((reinterpret_cast<uintptr_t>(g.X) | uintptr_t(g.x_bs * 4) | uintptr_t(g.x_rs * 4)) & 15) == 0;
  const bool wvec = g.K % 4 == 0 && g.w_ks == 1 && ((reinterpret_cast<uintptr_t>(g.W) | uintptr_t(g.w_ns * 4)) & 15) == 0;
  auto fetch = [&](const float* row, int64_t ks, bool vec, int k0) __attribute__((always_inline)) -> gen_f32x4 {
    gen_f32x4 v = {0.f, 0.f, 0.f, 0.f};
    const int k = k0 + lk;
    if (row && k < g.K) {
      if (vec) v = *reinterpret_cast<const gen_f32x4*>(row + k);
      else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (k + q < g.K) v[q] = row[int64_t(k + q) * ks];
      }
    }
    return v;
  };
  gen_f32x4 xq = fetch(xrow, 1, xvec, 0), wq = fetch(wrow, g.w_ks, wvec, 0);
  for (int k0 = 0; k0 < g.K; k0 += kGenK) {
#pragma unroll
    for (int q = 0; q < 4; ++q) { xs[lk + q][lr] = xq[q]; ws[lk + q][lr] = wq[q]; }
    __syncthreads();
    if (k0 + kGenK < g.K) { xq = fetch(xrow, 1, xvec, k0 + kGenK); wq = fetch(wrow, g.w_ks, wvec, k0 + kGenK); }
#pragma unroll
    for (int k4 = 0; k4 < kGenK; k4 += 4) {
      const float a = xs[k4 + lq][wave * 16 + l15];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, ws[k4 + lq][j * 16 + l15], acc[j], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t m = m0 + wave * 16 + lq * 4 + i;
    if (m >= M) continue;
    const int64_t b = m / g.Tn, t = m - b * g.Tn;
    float* yrow = g.Y + b * g.y_bs + t * g.y_rs;
    const float* rrow = g.R ? g.R + b * g.r_bs + t * g.r_rs : nullptr;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + j * 16 + l15;
      if (n >= g.N) continue;
      float v = acc[j][i];
      if (g.flags & GEN_ACCUM) v += yrow[n];
      if (!(g.flags & GEN_PARTIAL)) {
        if (g.bias) v += g.bias[n];
        if ((g.flags & GEN_RES_BEFORE) && rrow) v += rrow[n];
        if (g.flags & GEN_RELU) v = nf_relu(v);                // (torch.relu: a NaN stays a NaN -- this path is plain IEEE f32)
        if ((g.flags & GEN_RES_AFTER) && rrow) v += rrow[n];
        if (g.flags & GEN_SIGMOID) v = sigmoidf_(v);
      }
      yrow[n] = v;
    }
  }
}

// Where element (b, c, tau) of a streaming cache sits: conv backbones (B, C, P) with the block's slice at `off`
// (tcn.py:155-165, mdtc.py:250-275): bs = C P, cs = P, ts = 1; FSMN (B, D, P, L), layer index innermost (fsmn.py:495):
// bs = D P L, cs = P L, ts = L, off = layer.
struct GenCacheMap { int64_t bs, cs, ts, off; };

// u[b][tau][c] = tau < pad ? (cache ? cache(b, c, tau) : 0) : h[b][tau - pad][c];  out_cache(b, c, p) = u[b][T + p][c]
__global__ void gen_ctx_kernel(float* __restrict__ u, const float* __restrict__ h, int64_t h_bs, int64_t h_rs,
                               const float* __restrict__ cin, float* __restrict__ cout, GenCacheMap cm, int B, int T, int C, int pad) {
  const int64_t n = int64_t(B) * (pad + T) * C;
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
    const int c = int(i % C);
    const int64_t r = i / C;
    const int tau = int(r % (pad + T));
    const int64_t b = r / (pad + T);
    const float v = tau < pad ? (cin ? cin[b * cm.bs + c * cm.cs + tau * cm.ts + cm.off] : 0.f) : h[b * h_bs + int64_t(tau - pad) * h_rs + c];
    u[i] = v;
    if (cout && tau >= T) cout[b * cm.bs + c * cm.cs + int64_t(tau - T) * cm.ts + cm.off] = v;
  }
}

// out[b][t][c] = [ReLU](bias[c] + sum_j w[c][j] u[b][t + j dil][c]),  j = 0 the oldest tap (cross-correlation, like Conv1d)
__global__ void gen_dw_kernel(float* __restrict__ out, const float* __restrict__ u, const float* __restrict__ w,
                              const float* __restrict__ bias, int B, int T, int C, int ks, int dil, int relu) {
  const int pad = (ks - 1) * dil;
  const int64_t n = int64_t(B) * T * C;
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
    const int c = int(i % C);
    const int64_t r = i / C;
    const int t = int(r % T);
    const int64_t b = r / T;
    const float* up = u + (b * (pad + T) + t) * C + c;
    const float* wp = w + int64_t(c) * ks;
    float acc = bias ? bias[c] : 0.f;
    for (int j = 0; j < ks; ++j) acc = fmaf(wp[j], up[int64_t(j) * dil * C], acc);
    out[i] = relu ? nf_relu(acc) : acc;
  }
}

// mode 0: y += x;  1: y = x;  2: y = sigmoid(x)
__global__ void gen_add_kernel(float* __restrict__ y, const float* __restrict__ x, int64_t n, int mode) {
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x)
    y[i] = mode == 0 ? y[i] + x[i] : mode == 1 ? x[i] : sigmoidf_(x[i]);
}

// out[b][c] = mean_t h[b][t][c]  (classifier.py:27)  /  h[b][T - 1][c]  (classifier.py:39)
__global__ void gen_pool_kernel(float* __restrict__ out, const float* __restrict__ h, int B, int T, int C, int last) {
  const int64_t n = int64_t(B) * C;
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
    const int c = int(i % C);
    const int64_t b = i / C;
    const float* p = h + b * T * C + c;
    if (last) { out[i] = p[int64_t(T - 1) * C]; continue; }
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += p[int64_t(t) * C];
    out[i] = s / float(T);
  }
}

// One step of torch.nn.GRU for all streams: gi = W_ih x_t + b_ih (row (b, t) of a (B T, 3H) matrix), gh = W_hh h + b_hh (B, 3H):
//   r = s(gi_r + gh_r), z = s(gi_z + gh_z), n = tanh(gi_n + r gh_n), h' = (1 - z) n + z h   -> hst (B, H) and seq[b][t][:]
__global__ void gen_gru_cell_kernel(const float* __restrict__ gi, const float* __restrict__ gh, float* __restrict__ hst,
                                    float* __restrict__ seq, int B, int T, int H, int t) {
  const int64_t n = int64_t(B) * H;
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
    const int u = int(i % H);
    const int64_t b = i / H;
    const float* a = gi + (b * T + t) * 3 * H;
    const float* g = gh + b * 3 * H;
    const float r = 1.f / (1.f + expf(-(a[u] + g[u])));
    const float z = 1.f / (1.f + expf(-(a[H + u] + g[H + u])));
    const float c = tanhf(a[2 * H + u] + r * g[2 * H + u]);
    const float hp = hst[i];
    const float hn = (1.f - z) * c + z * hp;
    hst[i] = hn;
    seq[(b * T + t) * H + u] = hn;
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// y[r][c] = [ReLU](w[c * ld + c] x[r][c] + bias[c]): the diagonal preprocessing of NoSubsampling (subsampling.py:35-36: the features
// ARE the hidden tile; the diagonal carries a folded CMVN).  Channel by channel like the reference -- through the matrix product
// an Inf in one channel would meet the zeros of every other row (0 * Inf = NaN).
__global__ void gen_diag_kernel(float* __restrict__ y, const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                int64_t rows, int C, int ld, int relu) {
  const int64_t n = rows * C;
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
    const int c = int(i % C);
    const float v = fmaf(w[int64_t(c) * ld + c], x[i], bias[c]);
    y[i] = relu ? nf_relu(v) : v;
  }
}

struct GenericModel {
  bool pre_diag = false;         // the preprocessing matrix is diagonal (NoSubsampling)
  wekws_hip_desc d{};
  const float* w = nullptr;      // the packer's blob on the device
  int cache_len = 0;             // conv: sum of paddings; fsmn: left_order - 1 + right_order; gru: 0
};

inline int gen_nblocks(const wekws_hip_desc& d) {
  return d.backbone == WEKWS_HIP_BACKBONE_MDTC ? 1 + d.num_stack * d.stack_size : d.num_layers;
}
inline int gen_dilation(const wekws_hip_desc& d, int i) {
  if (d.backbone == WEKWS_HIP_BACKBONE_MDTC) return i == 0 ? 1 : 1 << ((i - 1) % d.stack_size);   // mdtc.py:151-156, :229-237
  return 1 << i;                                                                                   // tcn.py:131-137
}
inline int gen_cache_len(const wekws_hip_desc& d) {
  if (d.backbone == WEKWS_HIP_BACKBONE_GRU) return 0;
  if (d.backbone == WEKWS_HIP_BACKBONE_FSMN) return d.kernel_size - 1 + d.stack_size;
  int s = 0;
  for (int i = 0; i < gen_nblocks(d); ++i) s += (d.kernel_size - 1) * gen_dilation(d, i);
  return s;
}
// widest row any intermediate of the model has (floats)
inline int gen_width(const wekws_hip_desc& d) {
  int w = std::max(d.hdim, d.odim);
  w = std::max(w, d.head_hidden);
  if (d.backbone == WEKWS_HIP_BACKBONE_GRU) w = std::max(w, 3 * d.hdim);
  if (d.backbone == WEKWS_HIP_BACKBONE_FSMN) w = std::max(std::max(w, d.num_stack), std::max(d.aux[0], d.aux[1]));
  return w;
}
inline size_t gen_al(size_t v) { return (v + 255) / 256 * 256; }
// scratch of one forward: four (B T, width) matrices, the [cache | h] buffer, GRU step buffers
inline size_t gen_workspace_bytes(const GenericModel& m, int B, int T) {
  const wekws_hip_desc& d = m.d;
  const size_t rows = size_t(B) * T, w = gen_width(d);
  int pmax = 0;
  if (d.backbone == WEKWS_HIP_BACKBONE_FSMN) pmax = m.cache_len;
  else if (d.backbone != WEKWS_HIP_BACKBONE_GRU)
    for (int i = 0; i < gen_nblocks(d); ++i) pmax = std::max(pmax, (d.kernel_size - 1) * gen_dilation(d, i));
  const size_t cu = d.backbone == WEKWS_HIP_BACKBONE_FSMN ? d.num_stack : d.hdim;
  size_t n = 4 * gen_al(rows * w * 4) + gen_al(size_t(B) * (pmax + T) * cu * 4);
  if (d.backbone == WEKWS_HIP_BACKBONE_GRU) n += gen_al(size_t(B) * 3 * d.hdim * 4) + gen_al(size_t(B) * d.hdim * 4);
  return n;
}

inline int gen_grid(int64_t n) { return int(std::min<int64_t>((n + 255) / 256, 16384)); }

inline void gen_gemm(hipStream_t st, const float* X, int64_t x_bs, int64_t x_rs, const float* W, int64_t w_ns, int64_t w_ks,
                     const float* bias, const float* R, int64_t r_bs, int64_t r_rs, float* Y, int64_t y_bs, int64_t y_rs, int Bn,
                     int Tn, int K, int N, int flags) {
  const GenGemm g{X, x_bs, x_rs, W, w_ns, w_ks, bias, R, r_bs, r_rs, Y, y_bs, y_rs, Bn, Tn, K, N, flags};
  const int64_t M = int64_t(Bn) * Tn;
  hipLaunchKernelGGL(gen_gemm_kernel, dim3(unsigned((M + kGenTile - 1) / kGenTile), unsigned((N + kGenTile - 1) / kGenTile)), dim3(256), 0, st, g);
}
// dense rows: Y (M, N) = epilogue(X (M, K) W (N, K)^T + b)
inline void gen_linear(hipStream_t st, const float* X, const float* W, const float* bias, const float* R, float* Y, int64_t M, int K,
                       int N, int flags) {
  gen_gemm(st, X, 0, K, W, K, 1, bias, R, 0, N, Y, 0, N, 1, int(M), K, N, flags);
}

// The forward of wekws_hip_forward for a GenericModel (everything but the trailing softmax, which the caller applies).
// ws: gen_workspace_bytes(m, B, T) bytes of scratch.  Returns 0, or -3 if a launch failed.
inline int generic_forward(const GenericModel& m, const float* x, int B, int T, const float* in_cache, float* y, float* out_cache,
                           char* ws, hipStream_t st, hipError_t* launch_error = nullptr) {
  auto done = [&]() {                                       // (hipGetLastError resets the sticky error: read once, hand it back)
    const hipError_t e = hipGetLastError();
    if (launch_error) *launch_error = e;
    return e == hipSuccess ? 0 : -3;
  };
  const wekws_hip_desc& d = m.d;
  const int64_t rows = int64_t(B) * T;
  const int C = d.hdim, W = gen_width(d);
  const size_t mat = gen_al(size_t(rows) * W * 4);
  float* hA = reinterpret_cast<float*>(ws);
  float* hB = reinterpret_cast<float*>(ws + mat);
  float* tm = reinterpret_cast<float*>(ws + 2 * mat);
  float* zs = reinterpret_cast<float*>(ws + 3 * mat);
  float* ub = reinterpret_cast<float*>(ws + 4 * mat);
  const float* p = m.w;
  const int act = d.activation == WEKWS_HIP_ACT_SIGMOID ? GEN_SIGMOID : 0;
  float* h = hA;                                              // the current activation tile (rows, width of the layer)
  float* o = hB;
  auto swap = [&]() { std::swap(h, o); };

  if (d.backbone == WEKWS_HIP_BACKBONE_FSMN) {
    // fsmn.py:462-495 (preprocessing none, identity head: fsmn_ctc.yaml:36-56)
    const int A0 = d.aux[0], A1 = d.aux[1], D = d.num_stack, lo = d.kernel_size, ro = d.stack_size, P = lo - 1 + ro, L = d.num_layers;
    gen_linear(st, x, p, p + size_t(A0) * d.idim, nullptr, h, rows, d.idim, A0, 0);                   // in_linear1
    p += size_t(A0) * d.idim + A0;
    gen_linear(st, h, p, p + size_t(C) * A0, nullptr, o, rows, A0, C, GEN_RELU);                      // in_linear2 + ReLU
    p += size_t(C) * A0 + C;
    swap();
    for (int l = 0; l < L; ++l) {
      const float* wproj = p; p += size_t(D) * C;
      const float* taps = p; p += size_t(D) * (lo + ro);
      const float* waff = p; p += size_t(C) * D;
      const float* baff = p; p += C;
      gen_linear(st, h, wproj, nullptr, nullptr, tm, rows, C, D, 0);                                  // LinearTransform, no bias
      const GenCacheMap cm{int64_t(D) * P * L, int64_t(P) * L, L, l};
      hipLaunchKernelGGL(gen_ctx_kernel, dim3(gen_grid(int64_t(B) * (P + T) * D)), dim3(256), 0, st, ub, tm, int64_t(T) * D, int64_t(D),
                         in_cache, out_cache, cm, B, T, D, P);
      hipLaunchKernelGGL(gen_dw_kernel, dim3(gen_grid(rows * D)), dim3(256), 0, st, tm, ub, taps, static_cast<const float*>(nullptr), B, T,
                         D, lo + ro, 1, 0);                                                             // memory block (+ identity tap)
      gen_linear(st, tm, waff, baff, nullptr, o, rows, D, C, GEN_RELU);                               // AffineTransform + ReLU
      swap();
    }
    gen_linear(st, h, p, p + size_t(A1) * C, nullptr, o, rows, C, A1, 0);                             // out_linear1
    p += size_t(A1) * C + A1;
    gen_linear(st, o, p, p + size_t(d.odim) * A1, nullptr, y, rows, A1, d.odim, act);                 // out_linear2
    return done();
  }

  // ---- preprocessing: LinearSubsampling1 (subsampling.py:53-57) or the CMVN-only diagonal (preproc_relu = 0)
  if (m.pre_diag)
    hipLaunchKernelGGL(gen_diag_kernel, dim3(gen_grid(rows * C)), dim3(256), 0, st, h, x, p, p + size_t(C) * d.idim, rows, C, d.idim,
                       d.preproc_relu);
  else
    gen_linear(st, x, p, p + size_t(C) * d.idim, nullptr, h, rows, d.idim, C, d.preproc_relu ? GEN_RELU : 0);
  p += size_t(C) * d.idim + C;

  if (d.backbone == WEKWS_HIP_BACKBONE_GRU) {
    const int H = C, L = d.num_layers;
    float* gh = reinterpret_cast<float*>(reinterpret_cast<char*>(ub) + gen_al(size_t(B) * T * H * 4));   // (ub itself is unused here)
    float* hst = reinterpret_cast<float*>(reinterpret_cast<char*>(gh) + gen_al(size_t(B) * 3 * H * 4));
    for (int l = 0; l < L; ++l) {
      const float* wih = p; p += size_t(3) * H * H;
      const float* whh = p; p += size_t(3) * H * H;
      const float* bih = p; p += 3 * H;
      const float* bhh = p; p += 3 * H;
      gen_linear(st, h, wih, bih, nullptr, tm, rows, H, 3 * H, 0);                                    // gi for all steps
      if (in_cache) (void)hipMemcpyAsync(hst, in_cache + size_t(l) * B * H, size_t(B) * H * 4, hipMemcpyDeviceToDevice, st);
      else (void)hipMemsetAsync(hst, 0, size_t(B) * H * 4, st);
      for (int t = 0; t < T; ++t) {
        gen_linear(st, hst, whh, bhh, nullptr, gh, B, H, 3 * H, 0);
        hipLaunchKernelGGL(gen_gru_cell_kernel, dim3(gen_grid(int64_t(B) * H)), dim3(256), 0, st, tm, gh, hst, o, B, T, H, t);
      }
      if (out_cache) (void)hipMemcpyAsync(out_cache + size_t(l) * B * H, hst, size_t(B) * H * 4, hipMemcpyDeviceToDevice, st);
      swap();
    }
  } else {
    const int nb = gen_nblocks(d), ks = d.kernel_size, Pc = m.cache_len;
    int off = 0;
    bool zinit = true;
    for (int i = 0; i < nb; ++i) {
      const int dil = gen_dilation(d, i), pad = (ks - 1) * dil;
      const GenCacheMap cm{int64_t(C) * Pc, Pc, 1, off};
      hipLaunchKernelGGL(gen_ctx_kernel, dim3(gen_grid(int64_t(B) * (pad + T) * C)), dim3(256), 0, st, ub, h, int64_t(T) * C, int64_t(C),
                         in_cache, out_cache, cm, B, T, C, pad);
      off += pad;
      if (d.backbone == WEKWS_HIP_BACKBONE_DS_TCN) {
        const float* wd = p; p += size_t(C) * ks;
        const float* bd = p; p += C;
        const float* wp = p; p += size_t(C) * C;
        const float* bp = p; p += C;
        hipLaunchKernelGGL(gen_dw_kernel, dim3(gen_grid(rows * C)), dim3(256), 0, st, tm, ub, wd, bd, B, T, C, ks, dil, 1);
        gen_linear(st, tm, wp, bp, h, o, rows, C, C, GEN_RELU | GEN_RES_AFTER);                       // tcn.py:101-114, :60
      } else if (d.backbone == WEKWS_HIP_BACKBONE_TCN) {
        const float* wc = p; p += size_t(C) * C * ks;                                                  // W[o][c][j]
        const float* bc = p; p += C;
        for (int j = 0; j < ks; ++j) {                                                                 // tap j reads u rows t + j dil
          const bool lastj = j == ks - 1;
          gen_gemm(st, ub + int64_t(j) * dil * C, int64_t(pad + T) * C, C, wc + j, int64_t(C) * ks, ks, lastj ? bc : nullptr,
                   lastj ? h : nullptr, int64_t(T) * C, C, o, int64_t(T) * C, C, B, T, C, C,
                   (j ? GEN_ACCUM : 0) | (lastj ? (GEN_RELU | GEN_RES_AFTER) : GEN_PARTIAL));         // tcn.py:75-84, :60
        }
      } else {                                                                                          // MDTC, mdtc.py:95-121
        const float* wd = p; p += size_t(C) * ks;
        const float* bd = p; p += C;
        const float* w1 = p; p += size_t(C) * C;
        const float* b1 = p; p += C;
        const float* w2 = p; p += size_t(C) * C;
        const float* b2 = p; p += C;
        hipLaunchKernelGGL(gen_dw_kernel, dim3(gen_grid(rows * C)), dim3(256), 0, st, tm, ub, wd, bd, B, T, C, ks, dil, 0);
        gen_linear(st, tm, w1, b1, nullptr, o, rows, C, C, GEN_RELU);
        gen_linear(st, o, w2, b2, h, tm, rows, C, C, GEN_RELU | GEN_RES_BEFORE);
        std::swap(tm, o);                                                                               // (the block's output is in `o` again)
        if (i > 0 && (i - 1) % d.stack_size == d.stack_size - 1) {                                      // end of a stack: mdtc.py:270-273
          hipLaunchKernelGGL(gen_add_kernel, dim3(gen_grid(rows * C)), dim3(256), 0, st, zs, o, rows * C, zinit ? 1 : 0);
          zinit = false;
        }
      }
      swap();
    }
    if (d.backbone == WEKWS_HIP_BACKBONE_MDTC) h = zs;                                                  // the classifier sees the sum of the stack outputs
  }

  // ---- classifier (classifier.py:26-28, :38-40, :63-67) + activation (kws_model.py:196-210)
  if (d.head == WEKWS_HIP_HEAD_LINEAR) {
    gen_linear(st, h, p, p + size_t(d.odim) * C, nullptr, y, rows, C, d.odim, act);
  } else if (d.head == WEKWS_HIP_HEAD_IDENTITY) {
    hipLaunchKernelGGL(gen_add_kernel, dim3(gen_grid(rows * C)), dim3(256), 0, st, y, h, rows * C, act ? 2 : 1);
  } else {
    const int HH = d.head_hidden;
    float* pooled = tm;                                       // (tm and o are free here)
    hipLaunchKernelGGL(gen_pool_kernel, dim3(gen_grid(int64_t(B) * C)), dim3(256), 0, st, pooled, h, B, T, C, d.head == WEKWS_HIP_HEAD_LAST ? 1 : 0);
    float* hid = o;
    gen_linear(st, pooled, p, p + size_t(HH) * C, nullptr, hid, B, C, HH, GEN_RELU);
    p += size_t(HH) * C + HH;
    gen_linear(st, hid, p, p + size_t(d.odim) * HH, nullptr, y, B, HH, d.odim, act);
  }
  return done();
}

}  // namespace wekws
