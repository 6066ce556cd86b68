// libwekws_hip.so -- C ABI (include/wekws_hip.h) over the gfx950 kernels.
// Host side only: descriptor validation, weight-blob parsing, MFMA fragment pre-packing, upload,
// tiling of long inputs, kernel dispatch.  No torch, no STL types across the boundary.
#include "../../include/wekws_hip.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "conv_stack.hip.h"
#include "conv_stack_f16.hip.h"
#include "dense_stack_f16.hip.h"
#include "ds256_w16.hip.h"
#include "ds256_g16.hip.h"
#include "ds256_g32.hip.h"
#include "mdtc64_g4.hip.h"
#include "ds64_g4.hip.h"
#include "ds256_stream.hip.h"
#include "ds256_mm.hip.h"
#include "mdtc64_w16.hip.h"
#include "mdtc64_stream.hip.h"
#include "fbank.hip.h"
#include "fsmn_f16.hip.h"
#include "gru.hip.h"
#include "gru_f16.hip.h"
#include "gru_pipe.hip.h"
#include "generic.hip.h"
#include "route.h"
#include "mfcc.hip.h"
#include "splice.hip.h"
#include "topk.hip.h"
#include "det.hip.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess)                                                                      \
      return fail(e_ == hipErrorOutOfMemory ? WEKWS_HIP_ENOMEM : WEKWS_HIP_EDEVICE, "%s: %s", \
                  #expr, hipGetErrorString(e_));                                               \
  } while (0)

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// Makes `device` current for the scope and restores the caller's device afterwards: the library never changes the
// calling thread's current device behind its back (a Python __del__ may run at any time).
struct DeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit DeviceGuard(int device) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device) ok = hipSetDevice(device) == hipSuccess;
    else prev = -1;                       // already current: nothing to restore
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};

// s = 2^(14 - floor(log2 bound)): bound * s in [2^14, 2^15); *inv = 1 / s.
// (host twin of pow2_scale in conv_stack_f16.hip.h)
inline float pow2_scale_host(float bound, float* inv) {
  uint32_t bits;
  std::memcpy(&bits, &bound, 4);
  const uint32_t e = (bits >> 23) & 0xffu;
  uint32_t se = 268u - e;
  se = se > 253u ? 253u : se;
  const uint32_t sb = se << 23, ib = (254u - se) << 23;
  float s;
  std::memcpy(&s, &sb, 4);
  std::memcpy(inv, &ib, 4);
  return s;
}

// Host-side builder of the device weight image; every section starts 16-byte aligned.
struct Image {
  std::vector<float> data;
  // Largest spread (binades) between the row maxima, or between the column maxima, of any matrix packed for the fp16
  // matrix cores.  Block floating point gives every matrix ONE power-of-two scale: an element 2^-e below the matrix
  // maximum keeps 22 - max(0, e - 16) significand bits, so a row (or a K column) whose largest element sits more than
  // ~20 binades below the matrix maximum contributes with visibly less than fp32 precision (measured through the live
  // reference: tests/golden/make_hetero_golden.py).  wekws_hip_create routes such a model to the exact-f32 kernels.
  float spread_log2 = 0.f;
  void note_spread(const float* Wsrc, int O, int Ksrc, int ld) {
    std::vector<float> rmax(size_t(O), 0.f), cmax(size_t(Ksrc), 0.f);
    float wmax = 0.f;
    for (int o = 0; o < O; ++o)
      for (int k = 0; k < Ksrc; ++k) {
        const float a = std::fabs(Wsrc[size_t(o) * ld + k]);
        if (!std::isfinite(a)) continue;
        rmax[o] = a > rmax[o] ? a : rmax[o];
        cmax[k] = a > cmax[k] ? a : cmax[k];
        wmax = a > wmax ? a : wmax;
      }
    if (!(wmax > 0.f)) return;
    auto upd = [&](const std::vector<float>& v) {
      for (float x : v)
        if (x > 0.f) {                                       // (all-zero rows / columns: padding, pruned units)
          const float sp = std::log2(wmax / x);
          spread_log2 = sp > spread_log2 ? sp : spread_log2;
        }
    };
    upd(rmax);
    upd(cmax);
  }
  uint32_t reserve(size_t n) {
    size_t off = (data.size() + 3) / 4 * 4;
    data.resize(off + n, 0.f);
    return uint32_t(off);
  }
  uint32_t put(const float* src, size_t n) {
    uint32_t off = reserve(n);
    std::memcpy(data.data() + off, src, n * sizeof(float));
    return off;
  }
  // A operand of v_mfma_f32_16x16x4_f32 for D[o][t] = sum_k W[o][k] B[k][t]:
  // element (otile, g, lane, s) = W[otile*16 + (lane&15)][g*16 + s*4 + (lane>>4)], zero beyond the source.
  uint32_t put_packed_a(const float* Wsrc, int O, int Ksrc, int ld) {
    const int Op = round_up(O, 16), Kp = round_up(Ksrc, 16);
    uint32_t off = reserve(size_t(Op) * Kp);
    float* dst = data.data() + off;
    for (int ot = 0; ot < Op / 16; ++ot)
      for (int g = 0; g < Kp / 16; ++g)
        for (int lane = 0; lane < 64; ++lane)
          for (int s = 0; s < 4; ++s) {
            const int o = ot * 16 + (lane & 15), k = g * 16 + s * 4 + (lane >> 4);
            const float v = (o < O && k < Ksrc) ? Wsrc[size_t(o) * ld + k] : 0.f;
            dst[((size_t(ot) * (Kp / 16) + g) * 64 + lane) * 4 + s] = v;
          }
    return off;
  }
  // A operand of v_mfma_f32_16x16x32_f16, operands split into fp16 hi + lo (conv_stack_f16.hip.h):
  // [o-tile][k32][hi|lo][lane][8 halves], lane l holds W[otile*16 + (l&15)][k32*32 + 8*(l>>4) + e], e = 0..7.
  // Block floating point: the matrix is stored as W * s, s the power of two that puts max|W| into [2^14, 2^15) -- the top
  // of the fp16 range, where hi + lo carries 22 bits for 16 binades below the maximum; *inv_scale = 1 / s (exact) is what
  // the kernel's epilogue multiplies the accumulator with.
  uint32_t put_packed_a16(const float* Wsrc, int O, int Ksrc, int ld, float* inv_scale) {
    const int Op = round_up(O, 16), Kp = round_up(Ksrc, 32);
    const size_t halves = size_t(Op) * Kp * 2;
    uint32_t off = reserve(halves / 2);
    _Float16* dst = reinterpret_cast<_Float16*>(data.data() + off);
    if (inv_scale) note_spread(Wsrc, O, Ksrc, ld);
    float wmax = 0.f;
    for (int o = 0; o < O; ++o)
      for (int k = 0; k < Ksrc; ++k) {
        const float a = std::fabs(Wsrc[size_t(o) * ld + k]);
        if (std::isfinite(a) && a > wmax) wmax = a;
      }
    float inv_local = 1.f;
    const float sw = inv_scale ? pow2_scale_host(wmax, inv_scale) : (void(inv_local), 1.f);   // nullptr: stored unscaled
    for (int ot = 0; ot < Op / 16; ++ot)
      for (int ks = 0; ks < Kp / 32; ++ks)
        for (int lane = 0; lane < 64; ++lane)
          for (int e = 0; e < 8; ++e) {
            const int o = ot * 16 + (lane & 15), k = ks * 32 + 8 * (lane >> 4) + e;
            const float v = (o < O && k < Ksrc) ? Wsrc[size_t(o) * ld + k] * sw : 0.f;
            const _Float16 h = static_cast<_Float16>(v);
            const _Float16 l = static_cast<_Float16>(v - static_cast<float>(h));
            const size_t base = ((size_t(ot) * (Kp / 32) + ks) * 2) * 512;  // halves per (o-tile, k32, plane) = 64*8
            dst[base + lane * 8 + e] = h;
            dst[base + 512 + lane * 8 + e] = l;
          }
    return off;
  }
};

// Scratch device memory of a model, one grow-only buffer per HIP stream that has called it: calls on one stream are
// ordered by the stream, calls on different streams never share a buffer (long-input tile hand-over caches, the
// global head's running sums, the GRU's layer sequences).
struct StreamBuf {
  hipStream_t stream;
  char* ptr;
  size_t bytes;
  char* gran = nullptr;         // GRU wavefront: the granule buffer (gru_pipe.hip.h) -- holds nothing but {data, tag} granules
  size_t gran_bytes = 0;
  unsigned* ctl = nullptr;      // ... and its control words (launch epoch, acknowledgements): allocated ONCE per stream and never
                                // re-allocated -- tags must keep growing for as long as any granule buffer of the stream lives
  unsigned gran_layout = 0;     // how the last wavefront call carved `gran` (slots, layers): a call with another layout clears
                                // it first -- its tag words would otherwise overlay what were DATA words of the old layout
  unsigned* err_h = nullptr;    // one word of pinned, device-mapped HOST memory: a device-side wait that gave up leaves its
  unsigned* err_d = nullptr;    // code here (err_d = the device's address of it); read by the next call, no synchronisation
};

bool desc_conv(const wekws_hip_desc& d) {
  return d.backbone == WEKWS_HIP_BACKBONE_DS_TCN || d.backbone == WEKWS_HIP_BACKBONE_TCN ||
         d.backbone == WEKWS_HIP_BACKBONE_MDTC;
}

int n_blocks(const wekws_hip_desc& d) {
  if (d.backbone == WEKWS_HIP_BACKBONE_MDTC) return 1 + d.num_stack * d.stack_size;
  return d.num_layers;
}

// validates and returns the blob size (floats); 0 with g_err set if invalid
size_t blob_elems(const wekws_hip_desc& d) {
  if (d.abi_version != WEKWS_HIP_ABI_VERSION) { fail(WEKWS_HIP_EINVAL, "desc.abi_version %d != %d", d.abi_version, WEKWS_HIP_ABI_VERSION); return 0; }
  if (d.idim <= 0 || d.hdim <= 0 || d.odim <= 0) { fail(WEKWS_HIP_EINVAL, "idim/hdim/odim must be positive"); return 0; }
  if (d.backbone != WEKWS_HIP_BACKBONE_FSMN && (d.aux[0] || d.aux[1])) { fail(WEKWS_HIP_EINVAL, "desc.aux must be 0 for this backbone"); return 0; }
  if (d.precision < 0 || d.precision > WEKWS_HIP_PRECISION_F16) { fail(WEKWS_HIP_EINVAL, "desc.precision %d", d.precision); return 0; }
  if (d.activation < 0 || d.activation > WEKWS_HIP_ACT_SOFTMAX) { fail(WEKWS_HIP_EINVAL, "desc.activation %d", d.activation); return 0; }
  if (d.activation == WEKWS_HIP_ACT_SOFTMAX && (d.head == WEKWS_HIP_HEAD_GLOBAL || d.head == WEKWS_HIP_HEAD_LAST)) {
    fail(WEKWS_HIP_EINVAL, "softmax activation needs a per-frame head (forward_softmax is softmax over axis 2)");
    return 0;
  }
  const size_t C = d.hdim, K = d.odim, ks = d.kernel_size;
  if (d.backbone == WEKWS_HIP_BACKBONE_FSMN) {
    const size_t A1 = d.aux[0], A2 = d.aux[1], D = d.num_stack, ro = d.stack_size;
    if (d.num_layers <= 0 || d.num_stack <= 0 || d.kernel_size <= 0 || d.stack_size <= 0 || d.aux[0] <= 0 || d.aux[1] <= 0) {
      fail(WEKWS_HIP_EINVAL, "fsmn: num_layers/proj_dim/left_order/right_order/affine dims must be positive");
      return 0;
    }
    if (d.head != WEKWS_HIP_HEAD_IDENTITY || d.activation == WEKWS_HIP_ACT_SIGMOID || d.preproc_relu) {
      fail(WEKWS_HIP_EINVAL, "fsmn: preprocessing none, identity classifier and identity activation only");
      return 0;
    }
    return A1 * d.idim + A1 + C * A1 + C + size_t(d.num_layers) * (D * C + D * (ks + ro) + C * D + C) + A2 * C + A2 +
           K * A2 + K;
  }
  size_t n = C * d.idim + C;  // preprocessing
  switch (d.backbone) {
    case WEKWS_HIP_BACKBONE_DS_TCN:
      if (d.num_layers <= 0 || d.kernel_size <= 0) { fail(WEKWS_HIP_EINVAL, "tcn: num_layers/kernel_size"); return 0; }
      n += size_t(d.num_layers) * (C * ks + C + C * C + C);
      break;
    case WEKWS_HIP_BACKBONE_TCN:
      if (d.num_layers <= 0 || d.kernel_size <= 0) { fail(WEKWS_HIP_EINVAL, "tcn: num_layers/kernel_size"); return 0; }
      n += size_t(d.num_layers) * (C * C * ks + C);
      break;
    case WEKWS_HIP_BACKBONE_MDTC:
      if (d.num_stack <= 0 || d.stack_size <= 0 || d.kernel_size <= 0) { fail(WEKWS_HIP_EINVAL, "mdtc: num_stack/stack_size/kernel_size"); return 0; }
      n += size_t(n_blocks(d)) * (C * ks + C + 2 * (C * C + C));
      break;
    case WEKWS_HIP_BACKBONE_GRU:
      if (d.num_layers <= 0) { fail(WEKWS_HIP_EINVAL, "gru: num_layers"); return 0; }
      n += size_t(d.num_layers) * (2 * 3 * C * C + 2 * 3 * C);
      break;
    default:
      fail(WEKWS_HIP_EINVAL, "unknown backbone %d", d.backbone);
      return 0;
  }
  switch (d.head) {
    case WEKWS_HIP_HEAD_LINEAR: n += K * C + K; break;
    case WEKWS_HIP_HEAD_GLOBAL:
    case WEKWS_HIP_HEAD_LAST:
      if (d.head_hidden <= 0) { fail(WEKWS_HIP_EINVAL, "head_hidden must be positive"); return 0; }
      n += size_t(d.head_hidden) * C + d.head_hidden + K * size_t(d.head_hidden) + K;
      break;
    case WEKWS_HIP_HEAD_IDENTITY:
      if (d.odim != d.hdim) { fail(WEKWS_HIP_EINVAL, "identity head needs odim == hdim"); return 0; }
      break;
    default:
      fail(WEKWS_HIP_EINVAL, "unknown head %d", d.head);
      return 0;
  }
  return n;
}

}  // namespace

// (block-wise copy between two cache layouts (B, C, P): slice i of the destination -- d_off[i], len[i] -- comes from s_off[i] of the
// source; destination elements outside every slice, or in channels the source does not have, are zero)
struct CacheMap {
  int nb;
  int s_off[wekws::kAmaxMaxBlocks], d_off[wekws::kAmaxMaxBlocks], len[wekws::kAmaxMaxBlocks];
};
static __global__ void cache_remap_kernel(float* __restrict__ dst, const float* __restrict__ src, int B, int Cd, int Pd, int Cs, int Ps,
                                          const CacheMap m) {
  const int64_t n = int64_t(B) * Cd * Pd;
  for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < n; e += int64_t(gridDim.x) * blockDim.x) {
    const int p = int(e % Pd);
    const int64_t bc = e / Pd;
    const int c = int(bc % Cd), b = int(bc / Cd);
    float v = 0.f;
    if (c < Cs) {
      for (int i = 0; i < m.nb; ++i)
        if (p >= m.d_off[i] && p < m.d_off[i] + m.len[i]) v = src[(int64_t(b) * Cs + c) * Ps + m.s_off[i] + (p - m.d_off[i])];
    }
    dst[e] = v;
  }
}

// The same re-computation as its own launch, behind a kernel that is left exactly as it was (the register-resident kernels of
// ds64_g4.hip.h / mdtc64_g4.hip.h: one utterance per small workgroup, several workgroups per CU, at the register limit -- a
// detection branch inside them moved their register allocation into scratch).  One workgroup per utterance: a pass over its
// features (and incoming cache); what the kernel before wrote for an utterance with a NaN / Inf input is overwritten.
static __global__ __launch_bounds__(256) void conv_nf_fix_kernel(const wekws::CallArgs A, int idim, int cache_elems) {
  __shared__ unsigned cell;
  const int b = blockIdx.x;
  bool bad = wekws::nf_scan_rows(A.x + int64_t(b) * A.xs_b, A.T, idim, idim, &cell);
  if (!bad && A.in_cache) bad = wekws::nf_scan(A.in_cache + int64_t(b) * cache_elems, cache_elems, &cell);
  if (bad) wekws::nf_repair_call(A, b);
}

struct wekws_hip_model {
  wekws_hip_desc desc;
  int device = 0;
  float* d_w = nullptr;
  wekws::BlockDesc* d_blocks = nullptr;
  wekws::DenseBlock* d_dblocks = nullptr;
  wekws::StackParams sp{};
  wekws::DenseParams dp{};
  // kernel selection (defaults = the product choice; wekws_hip_set_option overrides, for A/B measurements and the tests
  // that keep every kernel family parity-green)
  bool mm_eligible = false; // DS-TCN h256 + per-frame linear head: the all-matrix-core kernel (ds256_mm.hip.h) can serve it
  bool mm_ok = false;     // ... and does: default for CTC-sized heads (odim > 16); WEKWS_HIP_OPT_MM forces it on / off
                          // (keyword heads: the 16-wave kernel is 12 % faster)
  bool mdtc16_eligible = false;
  bool ds_stream_eligible = false;     // ds256_stream.hip.h: C = 256, kernel size 8, dilations 1 / 2 / 4 / 8
  bool mdtc_stream_eligible = false;   // mdtc64_stream.hip.h: dilations 1 / 2 / 4 / 8, the two streams' caches fit into LDS
  bool mdtc16_ok = false; // MDTC h64: the 16-wave kernel (WEKWS_HIP_OPT_MDTC16 = 0: the generic 8-wave one)
  bool w16_ok = true;     // DS-TCN h256: the 16-wave kernel (WEKWS_HIP_OPT_W16 = 0: the generic 8-wave one)
  float spread_log2 = 0.f;  // Image::spread_log2 of the weights this model was created from
  bool out_of_envelope = false;  // DEFAULT / F16X3 request, but the weights are outside the split-fp16 envelope ...
  int gru_pipe = 1;       // GRU: 1 the layer wavefront (gru_pipe.hip.h) when every tile of streams gets its own slot, 2 always, 0 never
  bool auto_f32 = false;  // ... and therefore the F32 kernels run (WEKWS_HIP_OPT_ENVELOPE = 0 keeps the split-fp16 kernels)
  bool g16_ok = true;     // ... calls without an incoming cache: the register-resident kernel (ds256_g16.hip.h; WEKWS_HIP_OPT_G16 = 0: ds256_w16)
  bool g16_one_pass = false;   // ... with a grid of B workgroups instead of persistent ones (option value 2: A/B measurements)
  bool g16_ctx = true;    // ... and calls WITH an incoming cache: the kernel's context variant (option value 3: never, i.e. ds256_w16)
  int fsmn_slices = -1;   // FSMN / DS-TCN-CTC head slices per tile for small calls: -1 automatic, 0 / 1 off, n forces n
  bool stream_ok = true;  // DS-TCN h256 / MDTC h64, chunks of <= 16 frames: the kernel with the LDS-resident cache
                          // (WEKWS_HIP_OPT_STREAM = 0 keeps the batch kernel)
  bool dense_ok = false;  // plain TCN whose paddings fit the dense-stack kernel's halo
  wekws::GruParams gp{};
  wekws::GruF16Params gq{};
  wekws::FsmnParams fq{};
  int fsmn_max_nt = 0;
  int fsmn_cus = 256;     // compute units of the device (FSMN utterance packing, GRU pass splitting)
  // Conv backbones created with a hidden_dim / kernel_size no kernel is built for run as the next built shape (extra channels
  // and the extra OLDEST taps are zero everywhere, see pad_conv_shape); desc then describes the built shape and these keep
  // the caller's: its channel count, its cache length, and how its cache's per-block slices map into the wider ones.
  int user_hdim = 0;
  int user_cache_len = 0;
  CacheMap widen{}, narrow{};
  int cache_len = 0;
  // A shape no specialised kernel is built for (wider / deeper / longer kernels than the reference's recipes use), or an FSMN
  // that must run exact f32: the any-shape path of generic.hip.h on the packer's blob as it is (d_w); nothing else of this
  // struct is used then.
  wekws::RouteFlags rf{};   // what the model's shape admits (route.h); the bools above that say the same are kept for the options
  bool generic = false;
  wekws::GenericModel gm{};
  // Utterances with a NaN / Inf input leave the fast path and are re-computed in exact IEEE f32 (nonfinite.hip.h): the
  // descriptor + packer-order blob the kernels' shape corresponds to, and scratch slots, on the device.
  wekws::NfCtx nf_host{};
  wekws::NfCtx* nf_dev = nullptr;
  float* nf_w = nullptr;
  float* nf_scratch = nullptr;
  unsigned* nf_slots = nullptr;
  std::vector<StreamBuf> ws;       // per-stream workspaces (stream_workspace())
  std::mutex ws_mu;
};

// NoSubsampling (subsampling.py:35-36) arrives as a square preprocessing matrix that is diagonal (the packer folds a CMVN into it)
static bool pre_is_diagonal(const wekws_hip_desc& d, const float* blob) {
  if (d.backbone == WEKWS_HIP_BACKBONE_FSMN || d.idim != d.hdim) return false;
  for (int r = 0; r < d.hdim; ++r)
    for (int k = 0; k < d.idim; ++k)
      if (r != k && blob[size_t(r) * d.idim + k] != 0.f) return false;
  return true;
}

// (measurement aid: WEKWS_NF_FIX_ALL=1 runs the separate non-finite pass behind EVERY conv kernel, to price the extra launch)
static bool nf_fix_all() {
  static const bool v = [] { const char* e = std::getenv("WEKWS_NF_FIX_ALL"); return e && e[0] == '1'; }();
  return v;
}
static thread_local wekws::Route g_last_route{};             // the route of this thread's last conv launch (tests: hooks build)
static wekws::RouteOptions route_options(const wekws_hip_model* m) {
  wekws::RouteOptions o;
  o.w16_ok = m->w16_ok; o.g16_ok = m->g16_ok; o.g16_ctx = m->g16_ctx; o.g16_one_pass = m->g16_one_pass; o.stream_ok = m->stream_ok;
  o.mdtc16_ok = m->mdtc16_ok; o.mm_ok = m->mm_ok;
  o.f32 = m->desc.precision == WEKWS_HIP_PRECISION_F32 || m->auto_f32;
  o.split = m->desc.precision != WEKWS_HIP_PRECISION_F16;
  return o;
}

// The device-side context of nonfinite.hip.h for a model whose kernels run shape `d` on packer-order blob `blob` (host).
// tmax: most frames one kernel call covers.  Returns a WEKWS_HIP_* code.
static int nf_setup(wekws_hip_model* m, const wekws_hip_desc& d, const float* blob, size_t n_elems, int tmax) {
  wekws::NfCtx& c = m->nf_host;
  c = wekws::NfCtx{};
  c.d = d;
  c.nslots = 16;
  c.skip_zero = 0;
  c.tmax = tmax;
  int64_t slot = 0;
  if (d.backbone == WEKWS_HIP_BACKBONE_GRU) {
    c.width = d.hdim;
    slot = int64_t(d.num_layers) * d.hdim + 7 * int64_t(d.hdim);
  } else if (d.backbone == WEKWS_HIP_BACKBONE_FSMN) {
    c.cache_len = d.kernel_size - 1 + d.stack_size;
    c.pmax = c.cache_len;
    c.width = std::max(std::max(d.hdim, d.num_stack), std::max(d.aux[0], d.aux[1]));
    slot = 4 * int64_t(tmax) * c.width + int64_t(c.pmax + tmax) * d.num_stack;
  } else {
    int pmax = 0, sum = 0;
    for (int i = 0; i < n_blocks(d); ++i) {
      const int dil = d.backbone == WEKWS_HIP_BACKBONE_MDTC ? ((i == 0) ? 1 : (1 << ((i - 1) % d.stack_size))) : (1 << i);
      const int pad = (d.kernel_size - 1) * dil;
      pmax = std::max(pmax, pad);
      sum += pad;
    }
    c.cache_len = sum;
    c.pmax = pmax;
    c.width = std::max(d.hdim, d.head_hidden);
    slot = 4 * int64_t(tmax) * c.width + int64_t(pmax + tmax) * d.hdim;
  }
  c.slot_floats = (slot + 63) / 64 * 64;
  c.pre_diag = pre_is_diagonal(d, blob);
  hipError_t e = hipMalloc(&m->nf_w, n_elems * sizeof(float));
  if (e == hipSuccess) e = hipMemcpy(m->nf_w, blob, n_elems * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc(&m->nf_scratch, size_t(c.slot_floats) * c.nslots * sizeof(float));
  if (e == hipSuccess) e = hipMalloc(&m->nf_slots, c.nslots * sizeof(unsigned));
  if (e == hipSuccess) e = hipMemset(m->nf_slots, 0, c.nslots * sizeof(unsigned));
  if (e == hipSuccess) e = hipMalloc(&m->nf_dev, sizeof(wekws::NfCtx));
  c.w = m->nf_w;
  c.scratch = m->nf_scratch;
  c.slots = m->nf_slots;
  if (e == hipSuccess) e = hipMemcpy(m->nf_dev, &c, sizeof(c), hipMemcpyHostToDevice);
  if (e != hipSuccess)
    return fail(e == hipErrorOutOfMemory ? WEKWS_HIP_ENOMEM : WEKWS_HIP_EDEVICE, "non-finite path setup: %s", hipGetErrorString(e));
  return WEKWS_HIP_OK;
}
static void nf_teardown(wekws_hip_model* m) {
  if (m->nf_w) (void)hipFree(m->nf_w);
  if (m->nf_scratch) (void)hipFree(m->nf_scratch);
  if (m->nf_slots) (void)hipFree(m->nf_slots);
  if (m->nf_dev) (void)hipFree(m->nf_dev);
  m->nf_w = m->nf_scratch = nullptr;
  m->nf_slots = nullptr;
  m->nf_dev = nullptr;
}
// a zero-padded model (pad_conv_shape / pad_gru_hidden): zero weights contribute nothing on the non-finite path
static int nf_set_skip_zero(wekws_hip_model* m) {
  m->nf_host.skip_zero = 1;
  if (!m->nf_dev) return WEKWS_HIP_OK;
  DeviceGuard guard(m->device);
  if (hipMemcpy(m->nf_dev, &m->nf_host, sizeof(m->nf_host), hipMemcpyHostToDevice) != hipSuccess)
    return fail(WEKWS_HIP_EDEVICE, "non-finite path setup (padded model)");
  return WEKWS_HIP_OK;
}

// -> device pointer to at least `need` bytes owned by (model, stream); nullptr + error text on failure
static char* stream_workspace(wekws_hip_model* m, hipStream_t stream, size_t need, bool granules = false, unsigned layout = 0) {
  std::lock_guard<std::mutex> lk(m->ws_mu);
  StreamBuf* sb = nullptr;
  for (auto& e : m->ws) if (e.stream == stream) sb = &e;
  if (!sb) { m->ws.push_back(StreamBuf{stream, nullptr, 0}); sb = &m->ws.back(); }
  char*& ptr = granules ? sb->gran : sb->ptr;
  size_t& bytes = granules ? sb->gran_bytes : sb->bytes;
  if (bytes < need) {
    // Growing frees the old buffer behind a stream synchronisation -- which a stream that is being captured into a HIP
    // graph cannot do: such a call fails and names wekws_hip_reserve (no hidden synchronisation inside a capture).
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
      fail(WEKWS_HIP_EINVAL, "this call needs %zu bytes of workspace on a stream that is being captured: call "
                             "wekws_hip_reserve(model, B, T, stream) before the capture begins", need);
      return nullptr;
    }
    if (ptr) {
      (void)hipStreamSynchronize(stream);                     // earlier calls on this stream may still use the old buffer
      (void)hipFree(ptr);
      ptr = nullptr; bytes = 0;
    }
    const hipError_t e = hipMalloc(reinterpret_cast<void**>(&ptr), need);
    if (e != hipSuccess) {
      ptr = nullptr;
      fail(e == hipErrorOutOfMemory ? WEKWS_HIP_ENOMEM : WEKWS_HIP_EDEVICE, "workspace of %zu bytes: %s", need, hipGetErrorString(e));
      return nullptr;
    }
    bytes = need;
    // GRU wavefront (gru_pipe.hip.h): a granule buffer must never show a word that was not written as a tag -- it is
    // cleared once, here (tags start at 1 and only grow: the epoch lives in StreamBuf::ctl, which is never re-allocated)
    if (granules && hipMemsetAsync(ptr, 0, need, stream) != hipSuccess) {
      fail(WEKWS_HIP_EDEVICE, "workspace: hipMemsetAsync");
      return nullptr;
    }
    if (granules) sb->gran_layout = layout;
  }
  // ... and again whenever a call carves the buffer differently from the call before it (another number of slots): gate
  // granules carry their tag in every fourth word, state granules in every second, so a tag position of the new layout may
  // hold a float of the old one -- which after days of streaming could equal a live tag.  Stream-ordered, no synchronisation.
  if (granules && layout && sb->gran_layout != layout) {
    if (sb->gran_layout && hipMemsetAsync(ptr, 0, bytes, stream) != hipSuccess) {
      fail(WEKWS_HIP_EDEVICE, "workspace: hipMemsetAsync");
      return nullptr;
    }
    sb->gran_layout = layout;
  }
  return ptr;
}

// The control words of a stream's GRU wavefront launches (gru_pipe.hip.h): one small allocation per (model, stream), made
// on the first call (or by wekws_hip_reserve) and kept until the stream's workspace is released.
static unsigned* stream_ctl(wekws_hip_model* m, hipStream_t stream, unsigned** err_d = nullptr) {
  std::lock_guard<std::mutex> lk(m->ws_mu);
  StreamBuf* sb = nullptr;
  for (auto& e : m->ws) if (e.stream == stream) sb = &e;
  if (!sb) { m->ws.push_back(StreamBuf{stream, nullptr, 0}); sb = &m->ws.back(); }
  if (!sb->ctl) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
      fail(WEKWS_HIP_EINVAL, "first GRU call on a stream that is being captured: call wekws_hip_reserve(model, B, T, stream) before the capture begins");
      return nullptr;
    }
    if (hipMalloc(reinterpret_cast<void**>(&sb->ctl), wekws::kGruPipeCtlBytes) != hipSuccess ||
        hipMemsetAsync(sb->ctl, 0, wekws::kGruPipeCtlBytes, stream) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&sb->err_h), 64, hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer(reinterpret_cast<void**>(&sb->err_d), sb->err_h, 0) != hipSuccess) {
      if (sb->ctl) (void)hipFree(sb->ctl);
      if (sb->err_h) (void)hipHostFree(sb->err_h);
      sb->ctl = nullptr;
      sb->err_h = sb->err_d = nullptr;
      fail(WEKWS_HIP_ENOMEM, "GRU control words");
      return nullptr;
    }
    *static_cast<volatile unsigned*>(sb->err_h) = 0u;
  }
  if (err_d) *err_d = sb->err_d;
  return sb->ctl;
}

// Has a device-side wait of an earlier forward on this stream given up (gru_pipe.hip.h: give_up)?  Reads the stream's word of
// host memory -- no device call on the healthy path --; if set, clears it (host word now, the device's copy in stream order)
// and returns WEKWS_HIP_EDEVICE with the stage named.  The word is written by the kernel itself, so a caller that pipelines
// forwards hears of a failure on the first call AFTER the failed launch has run, at the latest from wekws_hip_forward_status.
static int stream_health(wekws_hip_model* m, hipStream_t stream) {
  unsigned* err_h = nullptr;
  unsigned* ctl = nullptr;
  {
    std::lock_guard<std::mutex> lk(m->ws_mu);
    for (auto& e : m->ws) if (e.stream == stream) { err_h = e.err_h; ctl = e.ctl; }
  }
  if (!err_h) return WEKWS_HIP_OK;
  const unsigned code = *static_cast<volatile unsigned*>(err_h);
  if (!code) return WEKWS_HIP_OK;
  // The host word is cleared only once the clear of the DEVICE word is really queued behind the launches that saw it (stream
  // order): launches already queued behind the failed one still find the device word set, end at once and set the host word
  // again -- the next call reports them too.  During a capture (or if the memset cannot be queued) both words stay: every call
  // keeps failing until a call outside the capture can clear them.
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  const bool capturing = hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
  if (ctl && !capturing && hipMemsetAsync(ctl + 2, 0, sizeof(unsigned), stream) == hipSuccess)
    *static_cast<volatile unsigned*>(err_h) = 0u;
  return fail(WEKWS_HIP_EDEVICE, "a bounded wait of the GRU wavefront gave up (code 0x%x: %s of stage %u): the outputs of the "
              "forwards issued on this stream since the last successful call -- including calls that returned OK while "
              "the failed launch was still queued -- are not valid", code,
              (code >> 8) == 1 ? "data" : "credit", code & 0xffu);
}

struct wekws_hip_fbank {
  wekws::FbankParams fp{};
  int device = 0;
  float* d_tables = nullptr;
  int resident_f32 = 0, resident_i16 = 0;   // workgroups of one resident round, per sample type (fbank.hip.h)
};

// FSMN: validate, zero-pad every channel count to a multiple of 32, pre-split + pre-pack the six kinds of dense
// layers as MFMA A operands (fsmn_f16.hip.h), upload.

// ---------------------------------------------------------------------------------------------------------------------
// Operand-channel balancing (round 3).  A matrix product W a is unchanged when column k of W is multiplied by c_k and
// element k of a by 1 / c_k.  Where a is produced by a per-channel stage the library owns -- the depthwise conv + folded
// BN (+ ReLU) in front of a pointwise conv, the (ReLU'd) rows of the matrix in front of another matrix -- the factor moves
// into that stage's weights at no cost, exactly (c_k a power of two > 0; ReLU is positively homogeneous).  The library
// uses the freedom to give every K column of such a matrix a maximum in [1, 2): the operand tile then carries every
// channel at the magnitude of its CONTRIBUTION, one block-floating scale per matrix / tile covers the whole K axis, and the
// chained operand bounds (dw_alpha, mid_alpha, FSMN's affine bounds: products of row 1-norms) stay tight.  Without it a
// trained model that parks a 2^16 factor in a BatchNorm in front of a pointwise conv breaks the MDTC / FSMN kernels at
// 1e-3 (tests/golden/cases.py "kcol" cases, measured) although fp32 arithmetic -- the reference -- does not care.
// What must NOT be rescaled: anything the caller sees -- the residual stream (the conv caches hold it), FSMN's projections
// (its cache), GRU states.  Matrices are walked from the output side so that a matrix's rows are rescaled (by its
// consumer's balancing) before its own columns are balanced.  Exact in every precision mode: the F32 kernels compute the
// same bits as without it.
// ---------------------------------------------------------------------------------------------------------------------
// c_k for column k of W[O][K] (leading dimension ld): the power of two that puts the column maximum into [1, 2); 1 for an
// all-zero (or non-finite) column
static std::vector<float> column_balance(const float* W, int O, int K, int ld) {
  std::vector<float> c(size_t(K), 1.f);
  for (int k = 0; k < K; ++k) {
    float mx = 0.f;
    for (int o = 0; o < O; ++o) {
      const float a = std::fabs(W[size_t(o) * ld + k]);
      if (std::isfinite(a) && a > mx) mx = a;
    }
    if (mx > 0.f) {
      int e = 0;
      (void)std::frexp(mx, &e);                              // mx = f 2^e, f in [0.5, 1)  ->  mx 2^(1 - e) in [1, 2)
      e = 1 - e;
      e = e > 100 ? 100 : e < -100 ? -100 : e;
      c[k] = std::ldexp(1.f, e);
    }
  }
  return c;
}
static void scale_columns(float* W, int O, int K, int ld, const std::vector<float>& c) {
  for (int o = 0; o < O; ++o)
    for (int k = 0; k < K; ++k) W[size_t(o) * ld + k] *= c[k];
}
// rows of the producing stage: W[K][n] (n values per channel) and optionally bias[K], multiplied by 1 / c_k
static void scale_rows_inv(float* W, int K, int n, float* bias, const std::vector<float>& c) {
  for (int k = 0; k < K; ++k) {
    const float ic = 1.f / c[k];
    for (int j = 0; j < n; ++j) W[size_t(k) * n + j] *= ic;
    if (bias) bias[k] *= ic;
  }
}
static void balance_operand_channels(const wekws_hip_desc& d, float* w) {
  const int C = d.hdim, ks = d.kernel_size;
  if (d.backbone == WEKWS_HIP_BACKBONE_DS_TCN) {
    float* p = w + size_t(C) * d.idim + C;
    for (int i = 0; i < d.num_layers; ++i) {                 // [wd C x ks][bd C][Wp C x C][bp C]
      float* wd = p; float* bd = wd + size_t(C) * ks; float* Wp = bd + C;
      const std::vector<float> c = column_balance(Wp, C, C, C);
      scale_columns(Wp, C, C, C, c);
      scale_rows_inv(wd, C, ks, bd, c);                      // a_k = ReLU(dw_k(u) + b_k): c_k > 0 commutes with the ReLU
      p = Wp + size_t(C) * C + C;
    }
  } else if (d.backbone == WEKWS_HIP_BACKBONE_MDTC) {
    float* p = w + size_t(C) * d.idim + C;
    for (int i = 0; i < n_blocks(d); ++i) {                  // [wd C x ks][bd C][W1 C x C][b1 C][W2 C x C][b2 C]
      float* wd = p; float* bd = wd + size_t(C) * ks; float* W1 = bd + C; float* b1 = W1 + size_t(C) * C;
      float* W2 = b1 + C;
      const std::vector<float> c2 = column_balance(W2, C, C, C);
      scale_columns(W2, C, C, C, c2);
      scale_rows_inv(W1, C, C, b1, c2);                      // mid_m = ReLU(W1[m] a + b1[m])
      const std::vector<float> c1 = column_balance(W1, C, C, C);
      scale_columns(W1, C, C, C, c1);
      scale_rows_inv(wd, C, ks, bd, c1);                     // a_k = BN(dw_k(u)) (linear)
      p = W2 + size_t(C) * C + C;
    }
  } else if (d.backbone == WEKWS_HIP_BACKBONE_FSMN) {
    const int I = d.idim, A1 = d.aux[0], A2 = d.aux[1], D = d.num_stack, K = d.odim, nt = d.kernel_size + d.stack_size;
    float* in1 = w; float* in1b = in1 + size_t(A1) * I; float* in2 = in1b + A1; float* in2b = in2 + size_t(C) * A1;
    float* lay = in2b + C;
    const size_t lstride = size_t(D) * C + size_t(D) * nt + size_t(C) * D + C;   // Wproj, taps, Waff, baff
    float* out1 = lay + lstride * d.num_layers; float* out1b = out1 + size_t(A2) * C;
    float* out2 = out1b + A2;
    auto waff = [&](int l) { return lay + lstride * l + size_t(D) * C + size_t(D) * nt; };
    {
      const std::vector<float> c = column_balance(out2, K, A2, A2);            // out_linear2 <- out_linear1 (linear)
      scale_columns(out2, K, A2, A2, c);
      scale_rows_inv(out1, A2, C, out1b, c);
    }
    {
      const std::vector<float> c = column_balance(out1, A2, C, C);             // out_linear1 <- ReLU(affine of the last layer)
      scale_columns(out1, A2, C, C, c);
      float* wa = waff(d.num_layers - 1);
      scale_rows_inv(wa, C, D, wa + size_t(C) * D, c);
    }
    for (int l = d.num_layers - 1; l >= 0; --l) {                               // Wproj(l) <- ReLU(affine(l-1)) | ReLU(in_linear2)
      float* wp = lay + lstride * l;
      const std::vector<float> c = column_balance(wp, D, C, C);
      scale_columns(wp, D, C, C, c);
      if (l > 0) {
        float* wa = waff(l - 1);
        scale_rows_inv(wa, C, D, wa + size_t(C) * D, c);
      } else {
        scale_rows_inv(in2, C, A1, in2b, c);
      }
      // (Waff(l)'s columns are fed by the memory block of Wproj(l)'s output, which is the layer's CACHE: not rescaled)
    }
    {
      const std::vector<float> c = column_balance(in2, C, A1, A1);             // in_linear2 <- in_linear1 (linear)
      scale_columns(in2, C, A1, A1, c);
      scale_rows_inv(in1, A1, I, in1b, c);
    }
  }
}

// A valid reference configuration without a specialised kernel (`why` names the limit it exceeds): the any-shape exact-f32
// path (generic.hip.h).  The reference's init_model takes any size (kws_model.py:114-170); before round 5 these were
// WEKWS_HIP_EUNSUPPORTED.
static int create_generic(const wekws_hip_desc& d, const float* blob, size_t n_elems, int device, wekws_hip_model** out) {
  if (desc_conv(d)) {
    // dilations are 1 << i and the cache length is their sum times (kernel_size - 1): a corrupt descriptor (hundreds of layers)
    // must not reach the shift or overflow the int the kernels index with
    const int depth = d.backbone == WEKWS_HIP_BACKBONE_MDTC ? d.stack_size : d.num_layers;
    if (depth > 24) return fail(WEKWS_HIP_EINVAL, "dilation 2^%d: %d layers per stack is beyond any receptive field", depth - 1, depth);
    int64_t sum = 0;
    for (int i = 0; i < n_blocks(d); ++i) {
      const int dil = d.backbone == WEKWS_HIP_BACKBONE_MDTC ? ((i == 0) ? 1 : (1 << ((i - 1) % d.stack_size))) : (1 << i);
      sum += int64_t(d.kernel_size - 1) * dil;
    }
    if (sum * std::max(1, d.hdim) > int64_t(INT32_MAX) / 4)
      return fail(WEKWS_HIP_EINVAL, "cache of %lld frames x %d channels per stream is out of range", (long long)sum, d.hdim);
  }
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail(WEKWS_HIP_EDEVICE, "device %d of %d", device, ndev);
  DeviceGuard guard(device);
  if (!guard.ok) return fail(WEKWS_HIP_EDEVICE, "hipSetDevice(%d)", device);
  wekws_hip_model* m = new (std::nothrow) wekws_hip_model();
  if (!m) return fail(WEKWS_HIP_ENOMEM, "host allocation");
  m->desc = d;
  m->device = device;
  m->generic = true;
  m->gru_pipe = 0;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&m->d_w), n_elems * sizeof(float));
  if (e == hipSuccess) e = hipMemcpy(m->d_w, blob, n_elems * sizeof(float), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    if (m->d_w) (void)hipFree(m->d_w);
    delete m;
    return fail(e == hipErrorOutOfMemory ? WEKWS_HIP_ENOMEM : WEKWS_HIP_EDEVICE, "weight upload: %s", hipGetErrorString(e));
  }
  m->gm.d = d;
  m->gm.w = m->d_w;
  m->gm.pre_diag = pre_is_diagonal(d, blob);
  m->gm.cache_len = m->cache_len = wekws::gen_cache_len(d);
  *out = m;
  return WEKWS_HIP_OK;
}

static int create_fsmn(const wekws_hip_desc& d, const float* blob_in, size_t n_elems, int device, wekws_hip_model** out) {
  // precision F32 is served with the reference's own arithmetic (exact f32 products): the any-shape path -- the block-floating
  // kernel below is the default / F16X3 / F16 one
  if (d.precision == WEKWS_HIP_PRECISION_F32) return create_generic(d, blob_in, n_elems, device, out);
  std::vector<float> balanced(blob_in, blob_in + n_elems);
  balance_operand_channels(d, balanced.data());
  const float* blob = balanced.data();
  // every precision request is served by the block-floating split-fp16 kernel (22-bit products, fp32 accumulate: the
  // accuracy of fp32 arithmetic at any operand scale, tests/test_hip_parity.py::test_scale_sweep); an exact-f32 FSMN
  // kernel is not built
  if (d.num_layers > wekws::kFsmnMaxLayers) return create_generic(d, blob_in, n_elems, device, out);   // (deeper than the kernel's table)
  const int I = d.idim, A1 = d.aux[0], A2 = d.aux[1], C = d.hdim, D = d.num_stack, K = d.odim;
  const int ntaps = d.kernel_size + d.stack_size;
  if (ntaps > wekws::kFsmnMaxTaps) return create_generic(d, blob_in, n_elems, device, out);            // (longer memory than the kernel's taps)
  wekws::FsmnParams q{};
  q.idim = I; q.odim = K; q.proj = D;
  q.kin = round_up(I, 32); q.a1p = round_up(A1, 32); q.linp = round_up(C, 32); q.dp = round_up(D, 32);
  q.a2p = round_up(A2, 32); q.op = round_up(K, 32);
  q.nlayers = d.num_layers; q.ntaps = ntaps; q.P = ntaps - 1; q.taps_ld = round_up(ntaps, 4);
  int max_nt = 0;
  for (int nt = 1; nt <= wekws::kFsmnTileFrames / 16; ++nt)
    if (wekws::FsmnLds::make(q, 16 * nt, 1).bytes() <= wekws::kFsmnLdsLimit) max_nt = nt;
  if (!max_nt) return create_generic(d, blob_in, n_elems, device, out);                                 // (layer widths beyond the 160 KiB LDS tile)

  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail(WEKWS_HIP_EDEVICE, "device %d of %d", device, ndev);
  DeviceGuard guard(device);
  if (!guard.ok) return fail(WEKWS_HIP_EDEVICE, "hipSetDevice(%d)", device);

  Image img;
  img.reserve(4);
  const float* p = blob;
  // a dense layer W[O][Ksrc] (+ bias[O]): A operand padded to (Op x Kp); bias padded with zeros to Op
  // block floating point (fsmn_f16.hip.h): matrix scale, and the output bound |W a + b| <= alpha max|a| + beta
  auto dense = [&](int O, int Ksrc, int Op, bool has_bias, uint32_t* a_off, uint32_t* b_off, wekws::FsmnDense* fd) {
    std::vector<float> wp(size_t(Op) * Ksrc, 0.f);
    std::memcpy(wp.data(), p, size_t(O) * Ksrc * sizeof(float));
    *a_off = img.put_packed_a16(wp.data(), Op, Ksrc, Ksrc, &fd->inv_s);
    float l1 = 0.f, bmax = 0.f;
    for (int o = 0; o < O; ++o) {
      double sum = 0.0;
      for (int k = 0; k < Ksrc; ++k) sum += std::fabs(double(p[size_t(o) * Ksrc + k]));
      l1 = float(sum) > l1 ? float(sum) : l1;
    }
    p += size_t(O) * Ksrc;
    if (has_bias) {
      std::vector<float> bp(Op, 0.f);
      std::memcpy(bp.data(), p, size_t(O) * sizeof(float));
      *b_off = img.put(bp.data(), Op);
      for (int o = 0; o < O; ++o) bmax = std::fabs(p[o]) > bmax ? std::fabs(p[o]) : bmax;
      p += O;
    }
    fd->alpha = l1 * 1.0001f;                                // (summation order / rounding of the device's accumulation)
    fd->beta = bmax;
  };
  dense(A1, I, q.a1p, true, &q.in1_a, &q.in1_b, &q.in1);
  dense(C, A1, q.linp, true, &q.in2_a, &q.in2_b, &q.in2);
  for (int l = 0; l < d.num_layers; ++l) {
    uint32_t none = 0;
    dense(D, C, q.dp, false, &q.layer[l].wp_a, &none, &q.layer[l].wp);
    std::vector<float> tp(size_t(q.dp) * q.taps_ld, 0.f);
    float tl1 = 0.f;
    for (int c = 0; c < D; ++c) {
      float sum = 0.f;
      for (int j = 0; j < ntaps; ++j) {
        tp[size_t(c) * q.taps_ld + j] = p[size_t(c) * ntaps + j];
        sum += std::fabs(p[size_t(c) * ntaps + j]);
      }
      tl1 = sum > tl1 ? sum : tl1;
    }
    q.layer[l].taps_l1 = tl1 * 1.00001f;
    q.layer[l].taps = img.put(tp.data(), tp.size());
    p += size_t(D) * ntaps;
    dense(C, D, q.linp, true, &q.layer[l].wa_a, &q.layer[l].wa_b, &q.layer[l].wa);
  }
  dense(A2, C, q.a2p, true, &q.out1_a, &q.out1_b, &q.out1);
  dense(K, A2, q.op, true, &q.out2_a, &q.out2_b, &q.out2);
  if (size_t(p - blob) != n_elems)
    return fail(WEKWS_HIP_EINVAL, "internal: blob walk consumed %zu of %zu floats", size_t(p - blob), n_elems);
  if (img.spread_log2 > WEKWS_HIP_F16X3_ENVELOPE_LOG2) {
    // a weight matrix spreads its row / column magnitudes beyond the envelope in which the split-fp16 kernel keeps fp32-level
    // accuracy: exact f32 instead (wekws_hip_effective_precision reports F32, wekws_hip_weight_spread_log2 the spread)
    const float spread = img.spread_log2;
    const int rc = create_generic(d, blob_in, n_elems, device, out);
    if (rc == WEKWS_HIP_OK) { (*out)->spread_log2 = spread; (*out)->out_of_envelope = true; }
    return rc;
  }

  wekws_hip_model* m = new (std::nothrow) wekws_hip_model();
  if (!m) return fail(WEKWS_HIP_ENOMEM, "host allocation");
  m->desc = d;
  m->device = device;
  m->cache_len = q.P;
  m->fsmn_max_nt = max_nt;
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) m->fsmn_cus = prop.multiProcessorCount;
  }
  hipError_t e = hipMalloc(&m->d_w, img.data.size() * sizeof(float));
  if (e == hipSuccess) e = hipMemcpy(m->d_w, img.data.data(), img.data.size() * sizeof(float), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    if (m->d_w) (void)hipFree(m->d_w);
    delete m;
    return fail(e == hipErrorOutOfMemory ? WEKWS_HIP_ENOMEM : WEKWS_HIP_EDEVICE, "weight upload: %s", hipGetErrorString(e));
  }
  q.w = m->d_w;
  m->fq = q;
  m->spread_log2 = img.spread_log2;
  if (const int rc = nf_setup(m, d, blob_in, n_elems, 16 * max_nt); rc != WEKWS_HIP_OK) {
    nf_teardown(m);
    (void)hipFree(m->d_w);
    delete m;
    return rc;
  }
  *out = m;
  return WEKWS_HIP_OK;
}

// Scratch bytes one wekws_hip_forward(m, B, T) takes from its stream's workspace (0: none) -- the single source for the
// forward paths below and for wekws_hip_reserve.
// GRU: does a call of T frames run the layer wavefront (gru_pipe.hip.h)?
static bool gru_pipe_call(const wekws_hip_model* m, int B, int T) {
  const wekws_hip_desc& d = m->desc;
  if (!(d.backbone == WEKWS_HIP_BACKBONE_GRU && m->gru_pipe && d.precision != WEKWS_HIP_PRECISION_F32 && !m->auto_f32 &&
        wekws::gru_pipe_supported(m->gq, T)))
    return false;
  wekws::GruPipeGeom g;
  if (!wekws::gru_pipe_geom(d.num_layers, B, T, m->fsmn_cus, &g)) return false;
  // many more tiles than resident slots: every workgroup serves several tiles one after the other (each round fills and drains
  // the pipeline), and beyond ~8 rounds the layer-major kernels (all CUs on every pass, two tiles per workgroup) win --
  // measured with the ring buffers, 2 layers: 1.65x at B = 2048 (2 rounds), 1.31x at 4096, 1.09x at 8192 (8 rounds), 0.96x at
  // B = 16384 (16 rounds); option value 2 runs the wavefront anyway
  return g.tiles <= 8 * g.slots || m->gru_pipe == 2;
}
// ... and the bytes of its granule buffer (a second per-stream buffer that holds nothing else)
static size_t granule_need(const wekws_hip_model* m, int B, int T) {
  if (B <= 0 || T <= 0 || !gru_pipe_call(m, B, T)) return 0;
  wekws::GruPipeBytes pb{};
  if (!wekws::gru_pipe_bytes(m->desc.num_layers, B, T, m->fsmn_cus, &pb)) return 0;
  return pb.granules(m->desc.num_layers);
}
static size_t workspace_need(const wekws_hip_model* m, int B, int T) {
  const wekws_hip_desc& d = m->desc;
  if (B <= 0 || T <= 0) return 0;
  if (m->generic) return wekws::gen_workspace_bytes(m->gm, B, T);     // (monotonic in B and T)
  if (d.backbone == WEKWS_HIP_BACKBONE_FSMN) {
    const int TILE = 16 * m->fsmn_max_nt;
    if (T <= TILE) return 0;
    return 2 * size_t(B) * d.num_stack * m->cache_len * d.num_layers * sizeof(float);
  }
  if (d.backbone == WEKWS_HIP_BACKBONE_GRU) {
    // (a hidden size below the built 128 runs zero-padded: widened copies of the caller's states, in + out)
    const size_t padded = m->user_hdim ? 2 * size_t(d.num_layers) * B * d.hdim * sizeof(float) : 0;
    if (d.precision == WEKWS_HIP_PRECISION_F32 || m->auto_f32 || !wekws::gru_f16_supported(m->gq)) return padded;
    size_t seq_b = 0, gi_b = 0, sc_b = 0;
    if (gru_pipe_call(m, B, T)) {                            // seq_in, seq_top | sc
      wekws::GruPipeBytes pb{};
      wekws::gru_pipe_bytes(d.num_layers, B, T, m->fsmn_cus, &pb);
      return pb.plain() + padded;
    }
    wekws::gru_f16_workspace_bytes(B, T, m->fsmn_cus, &seq_b, &gi_b, &sc_b);
    return 2 * ((seq_b + 255) / 256 * 256) + (gi_b + 255) / 256 * 256 + (sc_b + 255) / 256 * 256 + padded;
  }
  const size_t ce = size_t(B) * d.hdim * m->cache_len;
  const size_t padded = m->user_hdim ? 2 * ce : 0;          // the caller's caches, widened to the built channel count, in + out
  if (T <= WEKWS_HIP_TILE_FRAMES) return padded * sizeof(float);
  const size_t ge = d.head == WEKWS_HIP_HEAD_GLOBAL ? size_t(B) * d.hdim : 0;
  return (2 * ce + ge + padded) * sizeof(float);
}

// Conv backbones whose hidden_dim C is not one of the built widths (32 / 64 / 128 / 256) run as the next built width Cp with
// the extra channels ZERO everywhere: zero rows and columns in every matrix, zero taps and biases.  A zero channel stays
// zero through the whole network (ReLU(0) = 0, residual 0 + 0) and adds exact zeros to every sum it enters, so the
// posteriors are those of the C-channel model; maxima, and with them the block-floating scales, are unchanged.  Returns the
// widened blob in the documented order (include/wekws_hip.h); `d` must be a conv descriptor that passed blob_elems().
// GRU (torch.nn.GRU, kws_model.py:128-133) with a hidden size H below the built 128: the extra units have zero weights and
// biases in all three gates, so r = z = 1/2, n = tanh(0) = 0 and h' = (1 - z) n + z h stays 0 from a zero-padded h0 -- the
// real units never see them (zero columns).  Gate blocks [r | z | n] are padded one by one.
static std::vector<float> pad_gru_hidden(const wekws_hip_desc& d, const float* p, int Hp) {
  const int H = d.hdim, K = d.odim;
  std::vector<float> out;
  auto rows = [&](int R, int Rp, int cols, int colsp) {
    const size_t base = out.size();
    out.resize(base + size_t(Rp) * colsp, 0.f);
    for (int r = 0; r < R; ++r) std::memcpy(&out[base + size_t(r) * colsp], p + size_t(r) * cols, cols * sizeof(float));
    p += size_t(R) * cols;
  };
  rows(H, Hp, d.idim, d.idim);
  rows(H, Hp, 1, 1);
  for (int l = 0; l < d.num_layers; ++l) {
    for (int g = 0; g < 3; ++g) rows(H, Hp, H, Hp);          // W_ih
    for (int g = 0; g < 3; ++g) rows(H, Hp, H, Hp);          // W_hh
    for (int g = 0; g < 3; ++g) rows(H, Hp, 1, 1);           // b_ih
    for (int g = 0; g < 3; ++g) rows(H, Hp, 1, 1);           // b_hh
  }
  rows(K, K, H, Hp);
  rows(K, K, 1, 1);
  return out;
}

// Likewise a kernel size ks below the built one ksp: a causal dilated conv with ks taps IS the ksp-tap conv whose first
// (oldest) ksp - ks taps are zero; only the streaming cache differs (ksp - 1 instead of ks - 1 dilations per block: the
// extra, older frames meet zero taps) -- wekws_hip_forward copies the caller's slices into / out of the tails of the wider ones.
static std::vector<float> pad_conv_shape(const wekws_hip_desc& d, const float* p, int Cp, int ksp) {
  const int C = d.hdim, ks = d.kernel_size, K = d.odim;
  std::vector<float> out;
  auto taps = [&](int R, int Rp, int cols, int colsp) {      // [R][cols][ks] -> [Rp][colsp][ksp], taps right-aligned
    const size_t base = out.size();
    out.resize(base + size_t(Rp) * colsp * ksp, 0.f);
    for (int r = 0; r < R; ++r)
      for (int c = 0; c < cols; ++c)
        std::memcpy(&out[base + (size_t(r) * colsp + c) * ksp + (ksp - ks)], p + (size_t(r) * cols + c) * ks, ks * sizeof(float));
    p += size_t(R) * cols * ks;
  };
  auto rows = [&](int R, int Rp, int cols, int colsp, int inner) {     // [R][cols][inner] -> [Rp][colsp][inner], zero padded
    const size_t base = out.size();
    out.resize(base + size_t(Rp) * colsp * inner, 0.f);
    for (int r = 0; r < R; ++r)
      for (int c = 0; c < cols; ++c)
        std::memcpy(&out[base + (size_t(r) * colsp + c) * inner], p + (size_t(r) * cols + c) * inner, inner * sizeof(float));
    p += size_t(R) * cols * inner;
  };
  rows(C, Cp, d.idim, d.idim, 1);                            // preprocessing W [C][idim], b [C]
  rows(C, Cp, 1, 1, 1);
  for (int i = 0; i < n_blocks(d); ++i) {
    if (d.backbone == WEKWS_HIP_BACKBONE_TCN) {
      taps(C, Cp, C, Cp);                                    // dense conv [C][C][ks], b [C]
      rows(C, Cp, 1, 1, 1);
    } else {
      taps(C, Cp, 1, 1);                                     // depthwise taps [C][ks], bias [C]
      rows(C, Cp, 1, 1, 1);
      rows(C, Cp, C, Cp, 1);                                 // pointwise [C][C], b [C]
      rows(C, Cp, 1, 1, 1);
      if (d.backbone == WEKWS_HIP_BACKBONE_MDTC) {
        rows(C, Cp, C, Cp, 1);                               // conv2 [C][C], b [C]
        rows(C, Cp, 1, 1, 1);
      }
    }
  }
  if (d.head == WEKWS_HIP_HEAD_LINEAR) {
    rows(K, K, C, Cp, 1);                                    // Wc [K][C], bc [K]
    rows(K, K, 1, 1, 1);
  } else if (d.head == WEKWS_HIP_HEAD_GLOBAL || d.head == WEKWS_HIP_HEAD_LAST) {
    const int HH = d.head_hidden;
    rows(HH, HH, C, Cp, 1);                                  // W1 [HH][C], b1, W2 [K][HH], b2
    rows(HH, HH, 1, 1, 1);
    rows(K, K, HH, HH, 1);
    rows(K, K, 1, 1, 1);
  }
  return out;
}

// The frames of one FSMN call, cut into LDS tiles chained through ping-pong workspace caches
static int forward_fsmn(wekws_hip_model* m, const float* x, int B, int T, const float* in_cache, float* y,
                        float* out_cache, hipStream_t stream) {
  const wekws_hip_desc& d = m->desc;
  const int TILE = 16 * m->fsmn_max_nt;
  const int ntiles = (T + TILE - 1) / TILE;
  float* ws_cache[2] = {nullptr, nullptr};
  if (ntiles > 1) {
    const size_t ce = size_t(B) * d.num_stack * m->cache_len * d.num_layers;
    char* base = stream_workspace(m, stream, workspace_need(m, B, T));
    if (!base) return WEKWS_HIP_ENOMEM;
    ws_cache[0] = reinterpret_cast<float*>(base);
    ws_cache[1] = ws_cache[0] + ce;
  }
  for (int i = 0; i < ntiles; ++i) {
    const int t0 = i * TILE;
    const int Tt = (T - t0 < TILE) ? (T - t0) : TILE;
    wekws::FsmnArgs a{};
    a.x = x + size_t(t0) * d.idim;
    a.xs_b = int64_t(T) * d.idim;
    a.in_cache = (i == 0) ? in_cache : ws_cache[(i - 1) & 1];
    a.out_cache = (i == ntiles - 1) ? out_cache : ws_cache[i & 1];
    a.y = y + size_t(t0) * d.odim;
    a.ys_b = int64_t(T) * d.odim;
    a.B = B;
    a.T = Tt;
    a.nf = m->nf_dev;
    // short inputs: pack 2 or 4 utterances into one workgroup, as long as every CU still gets a workgroup
    const int nt = (Tt + 15) / 16;
    int u = 1;
    for (int cand = 4; cand >= 2; cand /= 2)
      if (nt * cand <= m->fsmn_max_nt && nt * cand <= 4 && B >= cand * m->fsmn_cus &&
          wekws::FsmnLds::make(m->fq, 16 * nt * cand, cand).bytes() <= wekws::kFsmnLdsLimit) { u = cand; break; }
    // few tiles on many CUs: split the vocabulary-sized last layer over up to 8 workgroups per tile (fsmn_f16.hip.h)
    a.head_slices = 1;
    if (const int groups = (B + u - 1) / u; d.odim >= 256 && groups * 2 <= m->fsmn_cus) {
      int sl = m->fsmn_cus / groups;
      sl = sl > 8 ? 8 : sl;
      if (m->fsmn_slices >= 0) sl = m->fsmn_slices > 0 ? m->fsmn_slices : 1;
      a.head_slices = sl;
    }
    const int rc = wekws::launch_fsmn_f16(nt, u, m->fq, a, stream);
    if (rc) return fail(rc, "fsmn launch failed (nt=%d u=%d): %s", nt, u, hipGetErrorString(hipGetLastError()));
  }
  return WEKWS_HIP_OK;
}

extern "C" {

#ifdef WEKWS_GRU_PIPE_STAMPS
// measurement build only (tools/probe/gru_stamps.py): the wall-clock stamps gru_pipe_kernel left behind
int wekws_hip_debug_gru_stamps(unsigned long long* dst, int n) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(wekws::gp_stamps), size_t(n) * 8) == hipSuccess ? 0 : -3;
}
#endif
const char* wekws_hip_last_error(void) { return g_err.c_str(); }
int wekws_hip_abi_version(void) { return WEKWS_HIP_ABI_VERSION; }

size_t wekws_hip_blob_elems(const wekws_hip_desc* desc) {
  if (!desc) { fail(WEKWS_HIP_EINVAL, "desc is NULL"); return 0; }
  return blob_elems(*desc);
}

int wekws_hip_create(const wekws_hip_desc* desc, const float* blob, size_t n_elems, int device,
                     wekws_hip_model** out) {
  if (!desc || !blob || !out) return fail(WEKWS_HIP_EINVAL, "NULL argument");
  *out = nullptr;
  const wekws_hip_desc& d = *desc;
  const size_t need = blob_elems(d);
  if (!need) return WEKWS_HIP_EINVAL;
  if (need != n_elems) return fail(WEKWS_HIP_EINVAL, "weight blob has %zu floats, descriptor needs %zu", n_elems, need);
  const int C = d.hdim, ks = d.kernel_size, K = d.odim;
  if (d.backbone == WEKWS_HIP_BACKBONE_FSMN) return create_fsmn(d, blob, n_elems, device, out);
  const float* const orig = blob;                           // (the any-shape path takes the packer's blob as it is)
  std::vector<float> balanced(blob, blob + n_elems);        // (exact power-of-two rescaling: see balance_operand_channels)
  balance_operand_channels(d, balanced.data());
  blob = balanced.data();
  // which shape the kernels run: as it is, zero-padded to the next built one, or the any-shape path -- a pure function of the
  // descriptor (route.h: conv_shape_plan; tests/test_route.py sweeps it on the CPU)
  const wekws::ShapePlan plan = desc_conv(d) ? wekws::conv_shape_plan(d, wekws::kAmaxMaxBlocks) : wekws::ShapePlan{wekws::SHAPE_AS_IS, C, ks, nullptr};
  if (plan.kind == wekws::SHAPE_GENERIC) return create_generic(d, orig, n_elems, device, out);
  {
    if (plan.kind == wekws::SHAPE_PADDED) {
      // any width up to 256 and any kernel size up to the built one (kws_model.py:114,142-157 take any): run as the next
      // built shape, zero-padded -- exact, see pad_conv_shape
      const int Cp = plan.C, ks_built = plan.ks;
      wekws_hip_desc dd = d;
      dd.hdim = Cp;
      dd.kernel_size = ks_built;
      const std::vector<float> wide = pad_conv_shape(d, blob, Cp, ks_built);
      if (wide.size() != blob_elems(dd)) return fail(WEKWS_HIP_EINVAL, "internal: widened blob has %zu floats, expected %zu", wide.size(), blob_elems(dd));
      const int rc = wekws_hip_create(&dd, wide.data(), wide.size(), device, out);
      if (rc != WEKWS_HIP_OK) return rc;
      wekws_hip_model* m = *out;
      m->user_hdim = C;
      if (const int rz = nf_set_skip_zero(m); rz != WEKWS_HIP_OK) { wekws_hip_destroy(m); *out = nullptr; return rz; }
      // the caller's cache: per block (ks - 1) dil frames, the tail of the built kernel's (ks_built - 1) dil
      int uoff = 0, boff = 0;
      m->widen.nb = m->narrow.nb = n_blocks(d);
      for (int i = 0; i < n_blocks(d); ++i) {
        const int dil = d.backbone == WEKWS_HIP_BACKBONE_MDTC ? ((i == 0) ? 1 : (1 << ((i - 1) % d.stack_size))) : (1 << i);
        const int ulen = (ks - 1) * dil, blen = (ks_built - 1) * dil;
        m->widen.s_off[i] = uoff; m->widen.d_off[i] = boff + (blen - ulen); m->widen.len[i] = ulen;
        m->narrow.s_off[i] = boff + (blen - ulen); m->narrow.d_off[i] = uoff; m->narrow.len[i] = ulen;
        uoff += ulen; boff += blen;
      }
      m->user_cache_len = uoff;
      return WEKWS_HIP_OK;
    }
  }
  if (!desc_conv(d)) {
    if (C < 128 && d.head == WEKWS_HIP_HEAD_LINEAR && d.num_layers <= wekws::kGruMaxLayers) {
      wekws_hip_desc dd = d;
      dd.hdim = 128;
      const std::vector<float> wide = pad_gru_hidden(d, blob, 128);
      if (wide.size() != blob_elems(dd)) return fail(WEKWS_HIP_EINVAL, "internal: widened blob has %zu floats, expected %zu", wide.size(), blob_elems(dd));
      const int rc = wekws_hip_create(&dd, wide.data(), wide.size(), device, out);
      if (rc != WEKWS_HIP_OK) return rc;
      wekws_hip_model* m = *out;
      m->user_hdim = C;
      if (const int rz = nf_set_skip_zero(m); rz != WEKWS_HIP_OK) { wekws_hip_destroy(m); *out = nullptr; return rz; }
      m->widen.nb = m->narrow.nb = 1;                        // states (L, B, H): one "slice" per row
      m->widen.s_off[0] = m->widen.d_off[0] = m->narrow.s_off[0] = m->narrow.d_off[0] = 0;
      m->widen.len[0] = m->narrow.len[0] = C;
      return WEKWS_HIP_OK;
    }
    // hidden sizes above the built 128, more layers than the kernels' tables, pooled / identity heads on a GRU
    if (C != 128 || d.num_layers > wekws::kGruMaxLayers || d.head != WEKWS_HIP_HEAD_LINEAR)
      return create_generic(d, orig, n_elems, device, out);
  }
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail(WEKWS_HIP_EDEVICE, "device %d of %d", device, ndev);
  DeviceGuard guard(device);
  if (!guard.ok) return fail(WEKWS_HIP_EDEVICE, "hipSetDevice(%d)", device);

  wekws_hip_model* m = new (std::nothrow) wekws_hip_model();
  if (!m) return fail(WEKWS_HIP_ENOMEM, "host allocation");
  m->desc = d;
  m->device = device;
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) m->fsmn_cus = prop.multiProcessorCount;
  }

  Image img;
  img.reserve(4);  // offset 0 is never a valid section
  const float* p = blob;
  wekws::StackParams& sp = m->sp;
  std::vector<wekws::BlockDesc> blocks;
  std::vector<wekws::DenseBlock> dblocks;

  // ---- preprocessing
  const uint32_t pre_a = img.put_packed_a(p, C, d.idim, d.idim);
  float pre_inv_s = 1.f;
  const uint32_t pre_a16 = desc_conv(d) ? img.put_packed_a16(p, C, d.idim, d.idim, &pre_inv_s) : 0;
  p += size_t(C) * d.idim;
  const uint32_t pre_b = img.put(p, C);
  p += C;

  if (desc_conv(d)) {
    sp.idim = d.idim;
    sp.kpre = round_up(d.idim, 16);
    sp.ksize = ks;
    sp.odim = K;
    sp.pre_relu = d.preproc_relu;
    sp.pre_a = pre_a;
    sp.pre_b = pre_b;
    sp.kpre16 = round_up(d.idim, 32);
    sp.pre_a16 = pre_a16;
    sp.pre_inv_s = pre_inv_s;
    sp.head_inv_s = 1.f;
    const int nb = n_blocks(d);
    int off = 0;
    for (int i = 0; i < nb; ++i) {
      wekws::BlockDesc b{};
      b.inv_s1 = b.inv_s2 = b.dw_tap_s = b.dw_tap_inv = 1.f;
      if (d.backbone == WEKWS_HIP_BACKBONE_MDTC) {
        b.dil = (i == 0) ? 1 : (1 << ((i - 1) % d.stack_size));          // mdtc.py:151-156, :229-237
        b.zadd = (i > 0 && (i - 1) % d.stack_size == d.stack_size - 1);  // mdtc.py:270-273
      } else {
        b.dil = 1 << i;                                                   // tcn.py:131-137
      }
      b.pad = (ks - 1) * b.dil;
      b.cache_off = off;
      off += b.pad;
      wekws::DenseBlock db{};
      db.dil = b.dil; db.pad = b.pad; db.cache_off = b.cache_off; db.zadd = b.zadd; db.inv_s1 = 1.f;
      if (d.backbone == WEKWS_HIP_BACKBONE_TCN) {
        b.a1 = img.put_packed_a(p, C, C * ks, C * ks);
        b.a1_16 = img.put_packed_a16(p, C, C * ks, C * ks, &b.inv_s1);
        {  // dense-stack kernel: K reordered to (tap, channel)
          std::vector<float> mw(size_t(C) * C * ks);
          for (int o = 0; o < C; ++o)
            for (int c = 0; c < C; ++c)
              for (int j = 0; j < ks; ++j) mw[(size_t(o) * ks + j) * C + c] = p[(size_t(o) * C + c) * ks + j];
          db.a1 = img.put_packed_a16(mw.data(), C, C * ks, C * ks, &db.inv_s1);
        }
        p += size_t(C) * C * ks;
        b.b1 = img.put(p, C);
        db.b1 = b.b1;
        p += C;
      } else {
        b.dw_w = img.put(p, size_t(C) * ks);
        b.dw_b = img.put(p + size_t(C) * ks, C);
        {  // taps + bias of a channel side by side, padded to whole float4s (one or three 16-byte loads per row)
          const int dwp = round_up(ks + 1, 4);
          std::vector<float> pk(size_t(C) * dwp, 0.f);
          for (int c = 0; c < C; ++c) {
            for (int j = 0; j < ks; ++j) pk[size_t(c) * dwp + j] = p[size_t(c) * ks + j];
            pk[size_t(c) * dwp + ks] = p[size_t(C) * ks + c];
          }
          b.dw_pk = img.put(pk.data(), pk.size());
          // block floating point: the depthwise output obeys |dw(u) + b| <= dw_alpha * max|u| + dw_beta; the taps enter the
          // matrix cores (ds256_mm) scaled to the top of the fp16 range
          float l1 = 0.f, bmax = 0.f, tmax = 0.f;
          for (int c = 0; c < C; ++c) {
            float sum = 0.f;
            for (int j = 0; j < ks; ++j) {
              const float a = std::fabs(p[size_t(c) * ks + j]);
              sum += a;
              if (std::isfinite(a) && a > tmax) tmax = a;
            }
            l1 = sum > l1 ? sum : l1;
            const float ab = std::fabs(p[size_t(C) * ks + c]);
            bmax = ab > bmax ? ab : bmax;
          }
          b.dw_alpha = l1 * 1.0000005f;                       // (rounding of the tap sum and of the device's FMA chain)
          b.dw_beta = bmax;
          b.dw_tap_s = pow2_scale_host(tmax, &b.dw_tap_inv);
        }
        p += size_t(C) * ks;
        p += C;
        b.a1 = img.put_packed_a(p, C, C, C);
        b.a1_16 = img.put_packed_a16(p, C, C, C, &b.inv_s1);
        {  // |W1 a + b1| <= mid_alpha * max|a| + mid_beta (MDTC mid tile)
          float l1 = 0.f, bmax = 0.f;
          for (int o = 0; o < C; ++o) {
            float sum = 0.f;
            for (int k = 0; k < C; ++k) sum += std::fabs(p[size_t(o) * C + k]);
            l1 = sum > l1 ? sum : l1;
            const float ab = std::fabs(p[size_t(C) * C + o]);
            bmax = ab > bmax ? ab : bmax;
          }
          b.mid_alpha = l1 * 1.0001f;
          b.mid_beta = bmax;
        }
        p += size_t(C) * C;
        b.b1 = img.put(p, C);
        p += C;
        if (d.backbone == WEKWS_HIP_BACKBONE_MDTC) {
          b.a2 = img.put_packed_a(p, C, C, C);
          b.a2_16 = img.put_packed_a16(p, C, C, C, &b.inv_s2);
          p += size_t(C) * C;
          b.b2 = img.put(p, C);
          p += C;
        }
      }
      blocks.push_back(b);
      dblocks.push_back(db);
    }
    m->cache_len = off;
    sp.cache_len = off;
    sp.nblocks = nb;
    sp.head = d.head;
    sp.head_hidden = d.head_hidden;
    sp.sigmoid = d.activation == WEKWS_HIP_ACT_SIGMOID;
    wekws::DenseParams& dp = m->dp;
    dp.head_inv_s = 1.f;
    if (d.head == WEKWS_HIP_HEAD_LINEAR) {
      if (K > 16) {  // wide (CTC) heads: rows padded to a multiple of 32 so that o-tiles come in pairs (ds256_mm.hip.h)
        const int Kp = round_up(K, 32);
        std::vector<float> wp(size_t(Kp) * C, 0.f);
        std::memcpy(wp.data(), p, size_t(K) * C * sizeof(float));
        dp.head_a16 = img.put_packed_a16(wp.data(), Kp, C, C, &dp.head_inv_s);
      } else {
        dp.head_a16 = img.put_packed_a16(p, K, C, C, &dp.head_inv_s);
      }
      sp.head_w = img.put(p, size_t(K) * C); p += size_t(K) * C;
      sp.head_b = img.put(p, K); p += K;
    } else if (d.head == WEKWS_HIP_HEAD_GLOBAL || d.head == WEKWS_HIP_HEAD_LAST) {
      const int HH = d.head_hidden;
      sp.head_w = img.put(p, size_t(HH) * C); p += size_t(HH) * C;
      sp.head_b = img.put(p, HH); p += HH;
      sp.head_w2 = img.put(p, size_t(K) * HH); p += size_t(K) * HH;
      sp.head_b2 = img.put(p, K); p += K;
    }
    // dense-stack kernel parameters (plain TCN): same scalars, its own block table
    dp.nblocks = nb; dp.idim = d.idim; dp.kpre16 = sp.kpre16; dp.ksize = ks; dp.odim = K; dp.pre_relu = d.preproc_relu;
    dp.pre_a16 = sp.pre_a16; dp.pre_b = sp.pre_b; dp.head = d.head; dp.head_hidden = d.head_hidden; dp.sigmoid = sp.sigmoid;
    dp.head_w = sp.head_w; dp.head_b = sp.head_b; dp.head_w2 = sp.head_w2; dp.head_b2 = sp.head_b2; dp.cache_len = off;
    dp.pre_inv_s = pre_inv_s;
    sp.head_inv_s = dp.head_inv_s;
    // what this shape can run on: one pure function of the descriptor (route.h), shared with the CPU tests
    m->rf = wekws::conv_route_flags(d, int(wekws::mdtc64_stream_lds_bytes(off)));
    if (m->rf.cache_len != off) { delete m; return fail(WEKWS_HIP_EINVAL, "internal: cache length %d vs %d", m->rf.cache_len, off); }
    m->dense_ok = m->rf.dense_ok;
    m->mdtc16_eligible = m->rf.mdtc16_eligible;
    m->ds_stream_eligible = m->rf.ds_stream_eligible;
    m->mdtc16_ok = m->mdtc16_eligible;
    m->mdtc_stream_eligible = m->rf.mdtc_stream_eligible;
    m->mm_eligible = m->rf.mm_eligible;
    // default: on for CTC-sized heads (its activation planes feed an MFMA classifier directly), off for keyword heads
    // (the 16-wave kernel is 12 % faster there)
    m->mm_ok = m->mm_eligible && K > 16;
  } else {
    wekws::GruParams& gp = m->gp;
    gp.idim = d.idim;
    gp.kpre = round_up(d.idim, 16);
    gp.odim = K;
    gp.pre_relu = d.preproc_relu;
    gp.pre_a = pre_a;
    gp.pre_b = pre_b;
    gp.nlayers = d.num_layers;
    gp.sigmoid = d.activation == WEKWS_HIP_ACT_SIGMOID;
    for (int l = 0; l < d.num_layers; ++l) {
      const float* wih = p; p += size_t(3) * C * C;
      const float* whh = p; p += size_t(3) * C * C;
      const float* bih = p; p += 3 * C;
      const float* bhh = p; p += 3 * C;
      gp.layer[l].a_ih = img.put_packed_a(wih, 3 * C, C, C);
      gp.layer[l].a_hh = img.put_packed_a(whh, 3 * C, C, C);
      m->gq.a_ih16[l] = img.put_packed_a16(wih, 3 * C, C, C, &m->gq.ih_inv_s[l]);
      m->gq.a_hh16[l] = img.put_packed_a16(whh, 3 * C, C, C, &m->gq.hh_inv_s[l]);
      gp.layer[l].b_ih = img.put(bih, 3 * C);
      gp.layer[l].b_hh = img.put(bhh, 3 * C);
    }
    m->gq.head_a16 = img.put_packed_a16(p, K, C, C, &m->gq.head_inv_s);
    gp.head_w = img.put(p, size_t(K) * C); p += size_t(K) * C;
    gp.head_b = img.put(p, K); p += K;
    m->gq.kpre16 = round_up(d.idim, 32);
    m->gq.pre_a16 = img.put_packed_a16(blob, C, d.idim, d.idim, &m->gq.pre_inv_s);
    {  // |Wpre x + b| <= pre_alpha * max|x| + pre_beta
      float l1 = 0.f, bmax = 0.f;
      for (int o = 0; o < C; ++o) {
        float sum = 0.f;
        for (int k = 0; k < d.idim; ++k) sum += std::fabs(blob[size_t(o) * d.idim + k]);
        l1 = sum > l1 ? sum : l1;
        const float ab = std::fabs(blob[size_t(C) * d.idim + o]);
        bmax = ab > bmax ? ab : bmax;
      }
      m->gq.pre_alpha = l1 * 1.00001f;
      m->gq.pre_beta = bmax;
    }
    m->cache_len = 0;
  }
  if (size_t(p - blob) != n_elems) {
    delete m;
    return fail(WEKWS_HIP_EINVAL, "internal: blob walk consumed %zu of %zu floats", size_t(p - blob), n_elems);
  }
  // the promise of DEFAULT / F16X3 is fp32-level accuracy: weights outside the envelope in which the split-fp16 kernels
  // keep it (Image::spread_log2) are served by the exact-f32 kernels instead (wekws_hip_effective_precision says so)
  m->spread_log2 = img.spread_log2;
  m->out_of_envelope = (d.precision == WEKWS_HIP_PRECISION_DEFAULT || d.precision == WEKWS_HIP_PRECISION_F16X3) &&
                       img.spread_log2 > WEKWS_HIP_F16X3_ENVELOPE_LOG2;
  m->auto_f32 = m->out_of_envelope;

  auto cleanup = [&]() {
    if (m->d_w) (void)hipFree(m->d_w);
    if (m->d_blocks) (void)hipFree(m->d_blocks);
    if (m->d_dblocks) (void)hipFree(m->d_dblocks);
    delete m;
  };
  hipError_t e = hipMalloc(&m->d_w, img.data.size() * sizeof(float));
  if (e == hipSuccess) e = hipMemcpy(m->d_w, img.data.data(), img.data.size() * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess && !blocks.empty()) {
    e = hipMalloc(&m->d_blocks, blocks.size() * sizeof(wekws::BlockDesc));
    if (e == hipSuccess)
      e = hipMemcpy(m->d_blocks, blocks.data(), blocks.size() * sizeof(wekws::BlockDesc), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(&m->d_dblocks, dblocks.size() * sizeof(wekws::DenseBlock));
    if (e == hipSuccess)
      e = hipMemcpy(m->d_dblocks, dblocks.data(), dblocks.size() * sizeof(wekws::DenseBlock), hipMemcpyHostToDevice);
  }
  if (e != hipSuccess) {
    cleanup();
    return fail(e == hipErrorOutOfMemory ? WEKWS_HIP_ENOMEM : WEKWS_HIP_EDEVICE, "weight upload: %s", hipGetErrorString(e));
  }
  sp.w = m->d_w;
  sp.blocks = m->d_blocks;
  m->dp.w = m->d_w;
  m->dp.blocks = m->d_dblocks;
  m->gp.w = m->d_w;
  m->gq.base = m->gp;
  if (const int rc = nf_setup(m, d, orig, n_elems, WEKWS_HIP_TILE_FRAMES); rc != WEKWS_HIP_OK) {
    nf_teardown(m);
    cleanup();
    return rc;
  }
  *out = m;
  return WEKWS_HIP_OK;
}

void wekws_hip_destroy(wekws_hip_model* m) {
  if (!m) return;
  DeviceGuard guard(m->device);
  if (m->d_w) (void)hipFree(m->d_w);
  if (m->d_blocks) (void)hipFree(m->d_blocks);
  if (m->d_dblocks) (void)hipFree(m->d_dblocks);
  nf_teardown(m);
  bool gave_up = false;
  for (auto& e : m->ws) {
    // (the device memory first: hipFree waits for the work that may still use it -- the caller's stream handles are not
    // touched, they may have been destroyed before the model -- so the health word is final when it is read)
    if (e.ptr) (void)hipFree(e.ptr);
    if (e.gran) (void)hipFree(e.gran);
    if (e.ctl) (void)hipFree(e.ctl);
    if (e.err_h) {
      gave_up = gave_up || *static_cast<volatile unsigned*>(e.err_h) != 0u;
      (void)hipHostFree(e.err_h);
    }
  }
  // (no return value to carry it: a failure nobody has asked about yet is at least left in wekws_hip_last_error())
  if (gave_up) {
    (void)fail(WEKWS_HIP_EDEVICE, "model destroyed with an unreported failure: a bounded wait of the GRU wavefront gave up");
    // LOUD: this is the one failure no later call can report (the last forward of a script that never asked for
    // wekws_hip_forward_status / release) -- the reference's Run would have thrown (keyword_spotting.cc:77-79)
    std::fprintf(stderr, "libwekws_hip: ERROR: %s -- the outputs of the last forward(s) on that stream are not valid\n", g_err.c_str());
  }
  delete m;
}

int wekws_hip_cache_dim(const wekws_hip_model* m) {
  if (!m) return 0;
  return m->desc.backbone == WEKWS_HIP_BACKBONE_FSMN ? m->desc.num_stack : m->user_hdim ? m->user_hdim : m->desc.hdim;
}
int wekws_hip_cache_len(const wekws_hip_model* m) { return !m ? 0 : m->user_hdim ? m->user_cache_len : m->cache_len; }

int wekws_hip_effective_precision(const wekws_hip_model* m) {
  if (!m) return fail(WEKWS_HIP_EINVAL, "NULL model");
  const wekws_hip_desc& d = m->desc;
  if (m->generic) return WEKWS_HIP_PRECISION_F32;                                     // the any-shape path: exact f32 products
  if (d.backbone == WEKWS_HIP_BACKBONE_FSMN) return WEKWS_HIP_PRECISION_F16X3;        // the block-floating kernel (fsmn_f16.hip.h)
  if (d.backbone == WEKWS_HIP_BACKBONE_GRU)
    return (d.precision == WEKWS_HIP_PRECISION_F32 || m->auto_f32 || !wekws::gru_f16_supported(m->gq))
               ? WEKWS_HIP_PRECISION_F32 : WEKWS_HIP_PRECISION_F16X3;
  if (d.precision == WEKWS_HIP_PRECISION_F32 || m->auto_f32) return WEKWS_HIP_PRECISION_F32;   // conv_stack.hip.h serves every shape
  if (d.precision == WEKWS_HIP_PRECISION_F16) {
    // one product per term only where a 16-wave kernel takes the `split` switch (wekws_hip_forward's dispatch)
    const bool ds16 = d.backbone == WEKWS_HIP_BACKBONE_DS_TCN && d.hdim == 256 && m->w16_ok && !m->mm_ok;
    const bool md16 = d.backbone == WEKWS_HIP_BACKBONE_MDTC && m->mdtc16_ok;
    if (ds16 || md16) return WEKWS_HIP_PRECISION_F16;
  }
  return WEKWS_HIP_PRECISION_F16X3;
}

float wekws_hip_weight_spread_log2(const wekws_hip_model* m) { return m ? m->spread_log2 : -1.f; }

size_t wekws_hip_cache_elems(const wekws_hip_model* m, int B) {
  if (!m || B <= 0) return 0;
  if (m->desc.backbone == WEKWS_HIP_BACKBONE_GRU) return size_t(m->desc.num_layers) * B * (m->user_hdim ? m->user_hdim : m->desc.hdim);
  if (m->desc.backbone == WEKWS_HIP_BACKBONE_FSMN) return size_t(B) * m->desc.num_stack * m->cache_len * m->desc.num_layers;
  if (m->user_hdim) return size_t(B) * m->user_hdim * m->user_cache_len;
  return size_t(B) * m->desc.hdim * m->cache_len;
}

size_t wekws_hip_output_elems(const wekws_hip_model* m, int B, int T) {
  if (!m || B <= 0 || T <= 0) return 0;
  if (m->desc.head == WEKWS_HIP_HEAD_GLOBAL || m->desc.head == WEKWS_HIP_HEAD_LAST) return size_t(B) * m->desc.odim;
  return size_t(B) * T * m->desc.odim;
}

int wekws_hip_set_option(wekws_hip_model* m, int option, int value) {
  if (!m) return fail(WEKWS_HIP_EINVAL, "NULL model");
  switch (option) {
    case WEKWS_HIP_OPT_W16: m->w16_ok = value != 0; break;
    case WEKWS_HIP_OPT_MDTC16: m->mdtc16_ok = m->mdtc16_eligible && value != 0; break;
    case WEKWS_HIP_OPT_STREAM: m->stream_ok = value != 0; break;
    case WEKWS_HIP_OPT_MM: m->mm_ok = m->mm_eligible && (value < 0 ? m->desc.odim > 16 : value != 0); break;
    case WEKWS_HIP_OPT_HEAD_SLICES: m->fsmn_slices = value; break;
    case WEKWS_HIP_OPT_G16: m->g16_ok = value != 0; m->g16_one_pass = value == 2; m->g16_ctx = value != 3; break;   // (2: one workgroup per utterance; 3: no context variants -- measurement aids)
    case WEKWS_HIP_OPT_ENVELOPE: m->auto_f32 = m->out_of_envelope && value != 0; break;
    case WEKWS_HIP_OPT_GRU_PIPE: m->gru_pipe = value < 0 ? 1 : value > 2 ? 2 : value; break;
    default: return fail(WEKWS_HIP_EINVAL, "unknown option %d", option);
  }
  return WEKWS_HIP_OK;
}

size_t wekws_hip_workspace_bytes(const wekws_hip_model* m, int B, int T) { return m ? workspace_need(m, B, T) + granule_need(m, B, T) : 0; }

// What a reservation for "calls of up to (B, T)" has to hold: workspace_need() is not monotonic -- a GRU chunk of <= 16
// frames spreads its streams over more, smaller workgroups (gru_f16_spw), so (256, 10) needs more scratch than (256, 20)
// and (128, 10) as much as (256, 10) -- so the maximum over the shapes where the geometry changes is taken: the frame
// counts {T, min(T, 16)} and the stream counts B, the packed-workgroup boundaries 2^k x (workgroups) below B, and the
// two-tiles-per-workgroup threshold.
static void reserve_need(const wekws_hip_model* m, int B, int T, size_t* plain, size_t* gran) {
  *plain = *gran = 0;
  const int ts[2] = {T, T < 16 ? T : 16};
  int bs[12], nb = 0;
  bs[nb++] = B;
  const int wgs = m->fsmn_cus < wekws::kGruMaxPackedWgs ? m->fsmn_cus : wekws::kGruMaxPackedWgs;
  for (int k = 1; k <= 16; k *= 2)
    if (k * wgs < B) bs[nb++] = k * wgs;
  if (16 * 256 < B) bs[nb++] = 16 * 256;
  if (m->desc.backbone == WEKWS_HIP_BACKBONE_GRU) {
    // the wavefront's geometry (gru_pipe_geom): most slots (= most rings) with one stream per tile
    wekws::GruPipeGeom g;
    if (wekws::gru_pipe_geom(m->desc.num_layers, 1 << 30, 1, m->fsmn_cus, &g)) {
      if (g.slots < B) bs[nb++] = g.slots;
      if (16 * g.slots + 1 <= B) bs[nb++] = 16 * g.slots + 1;
    }
  }
  for (int i = 0; i < nb; ++i)
    for (int j = 0; j < 2; ++j) {
      const size_t n = workspace_need(m, bs[i], ts[j]), g = granule_need(m, bs[i], ts[j]);
      *plain = n > *plain ? n : *plain;
      *gran = g > *gran ? g : *gran;
    }
}

int wekws_hip_reserve(wekws_hip_model* m, int B, int T, void* stream_) {
  if (!m) return fail(WEKWS_HIP_EINVAL, "NULL model");
  size_t need = 0, gran = 0;
  reserve_need(m, B, T, &need, &gran);
  if (!need && !gran) return WEKWS_HIP_OK;
  DeviceGuard guard(m->device);
  if (!guard.ok) return fail(WEKWS_HIP_EDEVICE, "hipSetDevice(%d)", m->device);
  if (need && !stream_workspace(m, static_cast<hipStream_t>(stream_), need)) return WEKWS_HIP_ENOMEM;
  if (gran && !stream_workspace(m, static_cast<hipStream_t>(stream_), gran, true)) return WEKWS_HIP_ENOMEM;
  if (m->desc.backbone == WEKWS_HIP_BACKBONE_GRU && m->gru_pipe && !stream_ctl(m, static_cast<hipStream_t>(stream_))) return WEKWS_HIP_ENOMEM;
  return WEKWS_HIP_OK;
}

int wekws_hip_release(wekws_hip_model* m, void* stream_) {
  if (!m) return fail(WEKWS_HIP_EINVAL, "NULL model");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  DeviceGuard guard(m->device);
  int lkrc = WEKWS_HIP_OK;
  std::lock_guard<std::mutex> lk(m->ws_mu);
  for (size_t i = 0; i < m->ws.size(); ++i)
    if (m->ws[i].stream == stream) {
      if (m->ws[i].ptr || m->ws[i].gran || m->ws[i].ctl) {
        (void)hipStreamSynchronize(stream);
        if (m->ws[i].err_h && *static_cast<volatile unsigned*>(m->ws[i].err_h)) {
          // an unreported failure must not vanish with the stream's buffers
          lkrc = fail(WEKWS_HIP_EDEVICE, "stream released with an unreported failure: a bounded wait of the GRU wavefront gave up");
        }
        if (m->ws[i].ptr) (void)hipFree(m->ws[i].ptr);
        if (m->ws[i].gran) (void)hipFree(m->ws[i].gran);
        if (m->ws[i].ctl) (void)hipFree(m->ws[i].ctl);
        if (m->ws[i].err_h) (void)hipHostFree(m->ws[i].err_h);
      }
      m->ws.erase(m->ws.begin() + i);
      break;
    }
  return lkrc;
}

int wekws_hip_forward_status(wekws_hip_model* m, void* stream_) {
  if (!m) return fail(WEKWS_HIP_EINVAL, "NULL model");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  DeviceGuard guard(m->device);
  if (!guard.ok) return fail(WEKWS_HIP_EDEVICE, "hipSetDevice(%d)", m->device);
  // the documented contract, for EVERY model: the stream is synchronised when this returns (the C++ runtime's Forward reads
  // its host buffer behind it) -- then the health word, which only streams with wavefront launches have
  const hipError_t e = hipStreamSynchronize(stream);
  if (e != hipSuccess) return fail(WEKWS_HIP_EDEVICE, "hipStreamSynchronize: %s", hipGetErrorString(e));
  return stream_health(m, stream);
}

// Test hook (not part of the ABI in include/wekws_hip.h): set the launch epoch of the stream's wavefront control block, so that
// a test can walk the 32-bit tag counter across its wrap (tests/test_hip_parity.py::test_gru_wavefront_epoch_wrap).
#ifdef WEKWS_TEST_HOOKS   // libwekws_hip_hooks.so only (make hooks): the product library exports nothing outside the header
extern "C" int wekws_hip_debug_set_gru_epoch(wekws_hip_model* m, void* stream_, unsigned epoch) {
  if (!m) return fail(WEKWS_HIP_EINVAL, "NULL model");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  DeviceGuard guard(m->device);
  if (!guard.ok) return fail(WEKWS_HIP_EDEVICE, "hipSetDevice(%d)", m->device);
  unsigned* ctl = stream_ctl(m, stream);
  if (!ctl) return WEKWS_HIP_ENOMEM;
  if (hipMemcpyAsync(ctl, &epoch, sizeof(epoch), hipMemcpyHostToDevice, stream) != hipSuccess ||
      hipStreamSynchronize(stream) != hipSuccess)
    return fail(WEKWS_HIP_EDEVICE, "setting the epoch: %s", hipGetErrorString(hipGetLastError()));
  return WEKWS_HIP_OK;
}
// A tenant that keeps CUs busy: `blocks` workgroups of 128 KB of LDS each (one per CU, like the wavefront's own), every one
// holding its CU for `ms` milliseconds of wall clock.  tests: a wavefront launch whose later workgroups find no CU for longer
// than its bounded waits must END (not hang) and be reported by the next call; a shorter squeeze must change nothing.
// ms < 0: the same occupancy with every SIMD BUSY (matrix + vector instructions, no memory traffic) for -ms milliseconds -- to
// tell a neighbour's compute / power from a neighbour's memory traffic (tools/probe/gru_neighbours.py)
__global__ void debug_hog_kernel(unsigned long long ticks, int busy) {
  extern __shared__ char hog_lds[];
  if (threadIdx.x == 0) hog_lds[0] = 1;
  const unsigned long long t0 = wall_clock64();              // 100 MHz
  if (!busy) {
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(127);
    return;
  }
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  typedef float f4 __attribute__((ext_vector_type(4)));
  h8 a = {1, 1, 1, 1, 1, 1, 1, 1}, b = a;
  f4 c0 = {0, 0, 0, 0}, c1 = c0;
  float v = float(threadIdx.x);
  while (wall_clock64() - t0 < ticks) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
      v = fmaf(v, 1.0001f, 0.5f);
    }
  }
  if (c0[0] + c1[0] + v == 12345.f) hog_lds[1] = 1;
}
// The routing functions of route.h, callable WITHOUT a device (tests/test_route.py, hooks library only).
//   desc: any conv descriptor wekws_hip_create accepts.  opts[9] (or NULL = the defaults for that precision): w16_ok, g16_ok, g16_ctx,
//   g16_one_pass, stream_ok, mdtc16_ok, mm_ok (-1: the default: CTC-sized heads), f32, split.  call[8]: B, T (of the tile), ntiles,
//   has_in, has_out, x16, cache16, cus.  out[14]: plan kind, built C, built ks, family, nt, split, ctx, fast, grid, threads, lds,
//   utts_per_wg, cache_len (built shape), max_pad.  why: the reason text for "any-shape path" / "no kernel".  Returns 0.
extern "C" int wekws_hip_debug_conv_route(const wekws_hip_desc* desc, const int* opts, const int* call, int* out, char* why, int why_len) {
  if (!desc || !call || !out || !desc_conv(*desc)) return WEKWS_HIP_EINVAL;
  for (int i = 0; i < 14; ++i) out[i] = 0;
  if (why && why_len > 0) why[0] = 0;
  auto say = [&](const char* t) { if (why && why_len > 0 && t) { std::strncpy(why, t, size_t(why_len) - 1); why[why_len - 1] = 0; } };
  const wekws::ShapePlan plan = wekws::conv_shape_plan(*desc, wekws::kAmaxMaxBlocks);
  out[0] = plan.kind; out[1] = plan.C; out[2] = plan.ks;
  if (plan.kind == wekws::SHAPE_GENERIC) { say(plan.why); return WEKWS_HIP_OK; }
  wekws_hip_desc d = *desc;
  d.hdim = plan.C; d.kernel_size = plan.ks;
  const wekws::RouteFlags f = wekws::conv_route_flags(d, 0);
  wekws::RouteFlags ff = wekws::conv_route_flags(d, int(wekws::mdtc64_stream_lds_bytes(f.cache_len)));
  wekws::RouteOptions o;
  o.f32 = d.precision == WEKWS_HIP_PRECISION_F32;
  o.split = d.precision != WEKWS_HIP_PRECISION_F16;
  o.mdtc16_ok = ff.mdtc16_eligible;
  o.mm_ok = ff.mm_eligible && d.odim > 16;
  if (opts) {
    o.w16_ok = opts[0]; o.g16_ok = opts[1]; o.g16_ctx = opts[2]; o.g16_one_pass = opts[3]; o.stream_ok = opts[4];
    o.mdtc16_ok = ff.mdtc16_eligible && opts[5]; o.mm_ok = ff.mm_eligible && (opts[6] < 0 ? d.odim > 16 : opts[6] != 0);
    o.f32 = opts[7]; o.split = opts[8];
  }
  wekws::RouteCall c{call[0], call[1], call[2], call[3], call[4], call[5], call[6], call[7]};
  const wekws::Route r = wekws::select_conv_route(d, ff, o, c, int(wekws::ds256_stream_lds_bytes(ff.cache_len)),
                                                  int(wekws::mdtc64_stream_lds_bytes(ff.cache_len)));
  out[3] = r.family; out[4] = r.nt; out[5] = r.split; out[6] = r.ctx; out[7] = r.fast; out[8] = r.grid; out[9] = r.threads; out[10] = r.lds_bytes;
  out[11] = r.utts_per_wg; out[12] = ff.cache_len; out[13] = ff.max_pad;
  if (r.family == wekws::ROUTE_NONE) say(r.why_not);
  else say(wekws::route_family_name(r.family));
  return WEKWS_HIP_OK;
}
// the route of the calling thread's last conv launch: out[9] = family, nt, split, ctx, fast, grid, threads, lds, utts_per_wg
extern "C" int wekws_hip_debug_last_route(int* out) {
  if (!out) return WEKWS_HIP_EINVAL;
  const wekws::Route& r = g_last_route;
  out[0] = r.family; out[1] = r.nt; out[2] = r.split; out[3] = r.ctx; out[4] = r.fast; out[5] = r.grid; out[6] = r.threads; out[7] = r.lds_bytes;
  out[8] = r.utts_per_wg;
  return WEKWS_HIP_OK;
}
extern "C" int wekws_hip_debug_hog(int device, int blocks, int ms, void* stream_) {
  DeviceGuard guard(device);
  const int busy = ms < 0;
  if (busy) ms = -ms;
  if (!guard.ok || blocks <= 0 || ms > 2000) return fail(WEKWS_HIP_EINVAL, "hog: device %d blocks %d ms %d", device, blocks, ms);
  static wekws::DynLdsGrant grant;
  if (wekws::grant_dynamic_lds(debug_hog_kernel, 128 * 1024, grant)) return fail(WEKWS_HIP_EDEVICE, "hog: LDS grant");
  hipLaunchKernelGGL(debug_hog_kernel, dim3(blocks), dim3(busy ? 512 : 64), 128 * 1024, static_cast<hipStream_t>(stream_), 100000ull * ms, busy);
  return hipGetLastError() == hipSuccess ? WEKWS_HIP_OK : fail(WEKWS_HIP_EDEVICE, "hog launch");
}
#endif

int wekws_hip_forward(wekws_hip_model* m, const float* x, int B, int T, const float* in_cache, float* y,
                      float* out_cache, int softmax, void* stream_) {
  if (!m || !x || !y) return fail(WEKWS_HIP_EINVAL, "NULL argument");
  if (B < 0 || T <= 0) return fail(WEKWS_HIP_EINVAL, "B=%d T=%d", B, T);
  if (B == 0) return WEKWS_HIP_OK;
  if (in_cache && in_cache == out_cache) return fail(WEKWS_HIP_EINVAL, "in_cache and out_cache must not alias");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  // the kernels go to the model's device whatever the calling thread's current device is (a default stream, NULL, means
  // that device's default stream); the caller's device is current again on return
  DeviceGuard guard(m->device);
  if (!guard.ok) return fail(WEKWS_HIP_EDEVICE, "hipSetDevice(%d)", m->device);
  const wekws_hip_desc& d = m->desc;
  const bool per_frame = d.head == WEKWS_HIP_HEAD_LINEAR || d.head == WEKWS_HIP_HEAD_IDENTITY;

  if (m->generic) {
    char* base = stream_workspace(m, stream, workspace_need(m, B, T));
    if (!base) return WEKWS_HIP_ENOMEM;
    hipError_t lerr = hipSuccess;
    const int rc = wekws::generic_forward(m->gm, x, B, T, in_cache, y, out_cache, base, stream, &lerr);
    if (rc) return fail(rc, "any-shape path: launch failed: %s", hipGetErrorString(lerr));
  } else if (d.backbone == WEKWS_HIP_BACKBONE_FSMN) {
    const int rc = forward_fsmn(m, x, B, T, in_cache, y, out_cache, stream);
    if (rc) return rc;
  } else if (d.backbone == WEKWS_HIP_BACKBONE_GRU) {
    const bool f16 = d.precision != WEKWS_HIP_PRECISION_F32 && !m->auto_f32 && wekws::gru_f16_supported(m->gq);
    // a wavefront launch of an EARLIER call on this stream that gave up is reported here, by the call that follows it (one
    // read of host memory; the reference's forward either returns correct values or raises -- keyword_spotting.cc:77-79)
    int rc = stream_health(m, stream);
    if (rc) return rc;
    float* user_h_out = nullptr;
    bool gru_nf_in_kernel = false;
    if (m->user_hdim) {                                      // zero-padded hidden size: widened copies of the caller's states
      const size_t need = workspace_need(m, B, T);
      char* base = stream_workspace(m, stream, need);
      if (!base) return WEKWS_HIP_ENOMEM;
      const size_t he = size_t(d.num_layers) * B * d.hdim;
      float* wide = reinterpret_cast<float*>(base + need) - 2 * he;       // (the tail of the workspace)
      const int rows = d.num_layers * B;
      const int grid = int(std::min<size_t>((he + 255) / 256, 4096));
      if (in_cache) {
        hipLaunchKernelGGL(cache_remap_kernel, dim3(grid), dim3(256), 0, stream, wide, in_cache, rows, 1, d.hdim, 1, m->user_hdim, m->widen);
        in_cache = wide;
      }
      if (out_cache) { user_h_out = out_cache; out_cache = wide + he; }
    }
    if (f16) {
      // workspace (layer sequences + gate pre-activations): one grow-only buffer per (model, stream) -- calls on the
      // same stream are ordered by the stream, calls on different streams never share a buffer
      size_t seq_b = 0, gi_b = 0, sc_b = 0;
      char* base = stream_workspace(m, stream, workspace_need(m, B, T));
      if (!base) return WEKWS_HIP_ENOMEM;
      if (gru_pipe_call(m, B, T)) {
        wekws::GruPipeBytes pb{};
        wekws::gru_pipe_bytes(d.num_layers, B, T, m->fsmn_cus, &pb);
        wekws::GruPipeGeom geo;
        wekws::gru_pipe_geom(d.num_layers, B, T, m->fsmn_cus, &geo);
        char* gran = stream_workspace(m, stream, pb.granules(d.num_layers), true, unsigned(geo.slots) << 8 | unsigned(d.num_layers));
        if (!gran) return WEKWS_HIP_ENOMEM;
        wekws::GruPipeWorkspace ws{};
        ws.ctl = stream_ctl(m, stream, &ws.err);
        if (!ws.ctl) return WEKWS_HIP_ENOMEM;
        ws.seq_in = base;
        ws.seq_top = ws.seq_in + pb.seq;
        ws.sc = reinterpret_cast<float*>(ws.seq_top + pb.seq);
        for (int l = 0; l < d.num_layers; ++l) { ws.gi[l] = gran; gran += pb.gi; }
        for (int l = 0; l + 1 < d.num_layers; ++l) { ws.hs[l] = gran; gran += pb.hs; }
        // Non-finite pass: the wavefront's own extra workgroups (one per slot) where there are CUs left for them to run BESIDE the
        // pipeline -- streaming chunks, small batches: no second launch, 2.68 -> 2.48 us per frame at B = 1 --; where the stage
        // workgroups fill the device they would only start behind it and scan 16 streams each (measured at B = 1024 x 98:
        // 0.186 ms against 0.176 with the separate launch, whose 1024 small workgroups scan in parallel): the launch stays.
        static const bool nf_in_kernel_off = [] { const char* e = std::getenv("WEKWS_GRU_NF_IN_KERNEL"); return e && e[0] == '0'; }();   // (A/B aid)
        if (!nf_in_kernel_off && (geo.stages + 1) * geo.slots <= m->fsmn_cus) {
          ws.nf = m->nf_dev;
          gru_nf_in_kernel = true;
        }
        rc = wekws::launch_gru_pipe(m->gq, ws, x, B, T, in_cache, y, out_cache, m->fsmn_cus, stream);
      } else {
        wekws::gru_f16_workspace_bytes(B, T, m->fsmn_cus, &seq_b, &gi_b, &sc_b);
        const size_t seq_al = (seq_b + 255) / 256 * 256, gi_al = (gi_b + 255) / 256 * 256;
        wekws::GruF16Workspace ws{{base, base + seq_al}, reinterpret_cast<float*>(base + 2 * seq_al),
                                  reinterpret_cast<float*>(base + 2 * seq_al + gi_al)};
        rc = wekws::launch_gru_f16(m->gq, ws, x, B, T, in_cache, y, out_cache, m->fsmn_cus, stream);
      }
    } else {
      rc = wekws::launch_gru(m->gp, x, B, T, in_cache, y, out_cache, stream);
    }
    if (rc) return fail(rc, "gru launch failed: %s", hipGetErrorString(hipGetLastError()));
    // streams with a NaN / Inf feature or state (their loads entered the kernels above as 0): the reference's arithmetic
    if (!gru_nf_in_kernel) {                                 // (the layer-major and exact-f32 kernels: their own launch behind them)
      hipLaunchKernelGGL(wekws::gru_nf_fix_kernel, dim3(B), dim3(256), 0, stream, m->nf_dev, x, B, T, in_cache, out_cache, y);
      if (hipGetLastError() != hipSuccess) return fail(WEKWS_HIP_EDEVICE, "gru non-finite pass: launch failed");
    }
    if (user_h_out) {
      const size_t ue = size_t(d.num_layers) * B * m->user_hdim;
      const int grid = int(std::min<size_t>((ue + 255) / 256, 4096));
      hipLaunchKernelGGL(cache_remap_kernel, dim3(grid), dim3(256), 0, stream, user_h_out, out_cache, d.num_layers * B, 1, m->user_hdim, 1,
                         d.hdim, m->narrow);
    }
  } else {
    const int TILE = WEKWS_HIP_TILE_FRAMES;
    const int ntiles = (T + TILE - 1) / TILE;
    const int C = d.hdim;
    float* ws_cache[2] = {nullptr, nullptr};
    float* gsum = nullptr;
    const size_t ce = size_t(B) * C * m->cache_len;
    float* user_out_cache = nullptr;                         // (widened models: where the caller wants the cache)
    if (ntiles > 1 || m->user_hdim) {
      char* base = stream_workspace(m, stream, workspace_need(m, B, T));
      if (!base) return WEKWS_HIP_ENOMEM;
      float* next = reinterpret_cast<float*>(base);
      if (ntiles > 1) {
        // long input: tiles hand the causal context over through ping-pong caches in the stream's workspace
        const size_t ge = d.head == WEKWS_HIP_HEAD_GLOBAL ? size_t(B) * C : 0;
        ws_cache[0] = next;
        ws_cache[1] = ws_cache[0] + ce;
        if (ge) gsum = ws_cache[1] + ce;
        next = ws_cache[1] + ce + ge;
      }
      if (m->user_hdim) {
        // the caller's caches have its own channel count and slice lengths: widened copies (zeros elsewhere) go to the kernels
        if (in_cache) {
          const int grid = int(std::min<size_t>((ce + 255) / 256, 4096));
          hipLaunchKernelGGL(cache_remap_kernel, dim3(grid), dim3(256), 0, stream, next, in_cache, B, C, m->cache_len, m->user_hdim,
                             m->user_cache_len, m->widen);
          in_cache = next;
        }
        if (out_cache) { user_out_cache = out_cache; out_cache = next + ce; }
      }
    }
    for (int i = 0; i < ntiles; ++i) {
      const int t0 = i * TILE;
      const int Tt = (T - t0 < TILE) ? (T - t0) : TILE;
      wekws::CallArgs a{};
      a.x = x + size_t(t0) * d.idim;
      a.xs_b = int64_t(T) * d.idim;
      a.in_cache = (i == 0) ? in_cache : ws_cache[(i - 1) & 1];
      a.out_cache = (i == ntiles - 1) ? out_cache : ws_cache[i & 1];
      a.y = per_frame ? y + size_t(t0) * d.odim : y;
      a.ys_b = per_frame ? int64_t(T) * d.odim : d.odim;
      a.gsum = gsum;
      a.B = B;
      a.T = Tt;
      a.T_total = T;
      a.first_tile = (i == 0);
      a.last_tile = (i == ntiles - 1);
      a.head_slices = 0;
      a.nf = m->nf_dev;
      if (m->mm_ok && d.odim >= 256 && B * 2 <= m->fsmn_cus) {       // CTC head, a handful of streams (ds256_mm.hip.h)
        const int sl = m->fsmn_cus / B;
        a.head_slices = m->fsmn_slices >= 0 ? m->fsmn_slices : (sl > 8 ? 8 : sl);
      }
      int rc;
      // ---- which kernel: one pure function of (shape flags, options, call) -- route.h; tests/test_route.py sweeps it on the CPU
      const wekws::RouteOptions ro = route_options(m);
      wekws::RouteCall rcall{};
      rcall.B = B; rcall.T = Tt; rcall.ntiles = ntiles;
      rcall.has_in = a.in_cache != nullptr; rcall.has_out = a.out_cache != nullptr;
      rcall.x16 = reinterpret_cast<uintptr_t>(a.x) % 16 == 0 && a.xs_b % 4 == 0;
      // (the streaming kernels move whole caches with 16-byte accesses: both of the CALL's cache pointers must be 16-byte aligned)
      rcall.cache16 = (reinterpret_cast<uintptr_t>(in_cache) | reinterpret_cast<uintptr_t>(out_cache)) % 16 == 0;
      rcall.cus = m->fsmn_cus;
      if (ntiles == 1) { rcall.has_in = in_cache != nullptr; rcall.has_out = out_cache != nullptr; }
      const wekws::Route route = wekws::select_conv_route(d, m->rf, ro, rcall, int(wekws::ds256_stream_lds_bytes(m->cache_len)),
                                                          int(wekws::mdtc64_stream_lds_bytes(m->cache_len)));
      if (route.family == wekws::ROUTE_NONE) return fail(WEKWS_HIP_EUNSUPPORTED, "no kernel for this call: %s", route.why_not ? route.why_not : "?");
      g_last_route = route;
      const bool split = route.split != 0;
      const int nt = route.nt;
      const int grid_cus = m->g16_one_pass ? (1 << 30) : m->fsmn_cus;   // persistent kernels: the largest grid
      switch (route.family) {
        case wekws::ROUTE_DS256_STREAM: rc = wekws::launch_ds256_stream(split, m->sp, a, stream); break;
        case wekws::ROUTE_DS256_G32: rc = wekws::launch_ds256_g32(nt, m->sp, a, stream, grid_cus); break;
        case wekws::ROUTE_DS256_MM: rc = wekws::launch_ds256_mm(nt, m->sp, a, m->dp.head_a16, stream); break;
        case wekws::ROUTE_DS256_G16: rc = wekws::launch_ds256_g16(nt, split, m->sp, a, stream, grid_cus); break;
        case wekws::ROUTE_DS256_W16: rc = wekws::launch_ds256_w16(nt, split, m->sp, a, stream); break;
        case wekws::ROUTE_DS64_G4: rc = wekws::launch_ds64_g4(nt, split, m->sp, a, stream); break;
        case wekws::ROUTE_MDTC64_STREAM: rc = wekws::launch_mdtc64_stream(split, m->sp, a, stream); break;
        case wekws::ROUTE_MDTC64_G4: rc = wekws::launch_mdtc64_g4(nt, split, m->sp, a, stream); break;
        case wekws::ROUTE_MDTC64_W16: rc = wekws::launch_mdtc64_w16(nt, split, m->sp, a, stream); break;
        case wekws::ROUTE_MDTC32_G4: rc = wekws::launch_mdtc32_g4(nt, split, m->sp, a, stream); break;
        case wekws::ROUTE_DENSE_F16: rc = wekws::launch_dense_stack_f16<wekws::KIND_TCN>(C, nt, m->dp, a, stream); break;
        case wekws::ROUTE_CONV_F16:
          rc = d.backbone == WEKWS_HIP_BACKBONE_DS_TCN ? wekws::launch_conv_stack_f16<wekws::KIND_DS>(C, nt, m->sp, a, stream)
               : d.backbone == WEKWS_HIP_BACKBONE_TCN  ? wekws::launch_conv_stack_f16<wekws::KIND_TCN>(C, nt, m->sp, a, stream)
                                                        : wekws::launch_conv_stack_f16<wekws::KIND_MDTC>(C, nt, m->sp, a, stream);
          break;
        default:
          rc = d.backbone == WEKWS_HIP_BACKBONE_DS_TCN ? wekws::launch_conv_stack<wekws::KIND_DS>(C, nt, m->sp, a, stream)
               : d.backbone == WEKWS_HIP_BACKBONE_TCN  ? wekws::launch_conv_stack<wekws::KIND_TCN>(C, nt, m->sp, a, stream)
                                                        : wekws::launch_conv_stack<wekws::KIND_MDTC>(C, nt, m->sp, a, stream);
          break;
      }
      // (a launcher that refuses what the route chose: the two have drifted apart -- an internal error, never a silent fall-through)
      if (rc == -4)
        return fail(WEKWS_HIP_EUNSUPPORTED, "internal: kernel family %s refused the call the route chose for it (C=%d nt=%d T=%d cache %d/%d)",
                    wekws::route_family_name(route.family), C, nt, Tt, rcall.has_in, rcall.has_out);
      if (rc) return fail(rc, "conv-stack launch failed (C=%d nt=%d): %s", C, nt, hipGetErrorString(hipGetLastError()));
      if (nf_fix_all()) {                                    // (measurement aid only: the non-finite pass as its own launch)
        hipLaunchKernelGGL(conv_nf_fix_kernel, dim3(B), dim3(256), 0, stream, a, d.idim, C * m->cache_len);
        if (hipGetLastError() != hipSuccess) return fail(WEKWS_HIP_EDEVICE, "non-finite pass: launch failed");
      }
    }
    if (user_out_cache) {
      const size_t ue = size_t(B) * m->user_hdim * m->user_cache_len;
      const int grid = int(std::min<size_t>((ue + 255) / 256, 4096));
      hipLaunchKernelGGL(cache_remap_kernel, dim3(grid), dim3(256), 0, stream, user_out_cache, out_cache, B, m->user_hdim, m->user_cache_len,
                         C, m->cache_len, m->narrow);
    }
  }
  if (softmax || d.activation == WEKWS_HIP_ACT_SOFTMAX) {
    const int64_t rows = per_frame ? int64_t(B) * T : B;
    const int K = d.odim;
    hipLaunchKernelGGL(wekws::softmax_rows_kernel, dim3(unsigned((rows + 3) / 4)), dim3(256), 0, stream, y, rows, K);
    if (hipGetLastError() != hipSuccess) return fail(WEKWS_HIP_EDEVICE, "softmax launch failed");
  }
  return WEKWS_HIP_OK;
}

// --------------------------------------------- fbank ---------------------------------------------
int wekws_hip_fbank_create(const wekws_hip_fbank_cfg* cfg, int device, wekws_hip_fbank** out) {
  if (!cfg || !out) return fail(WEKWS_HIP_EINVAL, "NULL argument");
  *out = nullptr;
  if (cfg->num_bins <= 0 || cfg->num_bins > wekws::kFbankMaxBins || cfg->sample_rate <= 0 || cfg->frame_length <= 0 ||
      cfg->frame_shift <= 0 || cfg->frame_length > wekws::kFbankMaxFft)
    return fail(WEKWS_HIP_EINVAL, "fbank cfg out of range");
  if (cfg->window != WEKWS_HIP_WINDOW_HAMMING && cfg->window != WEKWS_HIP_WINDOW_POVEY)
    return fail(WEKWS_HIP_EINVAL, "fbank window %d", cfg->window);
  if (cfg->frame_length <= 64)
    // (the reference would transform 64 points or fewer; frames that short -- 4 ms at 16 kHz -- have no recipe, and the
    // mel slots of a 512-point spectrum sampled every 8th bin or sparser are not laid out for it)
    return fail(WEKWS_HIP_EUNSUPPORTED, "fbank frame_length %d: frames of 65 .. 512 samples are built", cfg->frame_length);
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail(WEKWS_HIP_EDEVICE, "device %d of %d", device, ndev);
  DeviceGuard guard(device);
  if (!guard.ok) return fail(WEKWS_HIP_EDEVICE, "hipSetDevice(%d)", device);
  wekws_hip_fbank* f = new (std::nothrow) wekws_hip_fbank();
  if (!f) return fail(WEKWS_HIP_ENOMEM, "host allocation");
  f->device = device;
  std::vector<float> tables;
  const int empty = wekws::fbank_build_tables(cfg->num_bins, cfg->sample_rate, cfg->frame_length, cfg->frame_shift, cfg->window,
                                              &f->fp, &tables);
  if (empty >= 0) {                                           // (the reference's constructor CHECK-fails: fbank.h:81)
    delete f;
    return fail(WEKWS_HIP_EINVAL, "fbank: mel filter %d of %d covers no FFT bin (sample_rate %d, frame_length %d): fewer bins", empty,
                cfg->num_bins, cfg->sample_rate, cfg->frame_length);
  }
  hipError_t e = hipMalloc(&f->d_tables, tables.size() * sizeof(float));
  if (e == hipSuccess) e = hipMemcpy(f->d_tables, tables.data(), tables.size() * sizeof(float), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    if (f->d_tables) (void)hipFree(f->d_tables);
    delete f;
    return fail(WEKWS_HIP_EDEVICE, "fbank table upload: %s", hipGetErrorString(e));
  }
  f->fp.tables = f->d_tables;
  f->resident_f32 = wekws::fbank_resident_groups<float>(f->fp);
  f->resident_i16 = wekws::fbank_resident_groups<int16_t>(f->fp);
  *out = f;
  return WEKWS_HIP_OK;
}

void wekws_hip_fbank_destroy(wekws_hip_fbank* f) {
  if (!f) return;
  DeviceGuard guard(f->device);
  if (f->d_tables) (void)hipFree(f->d_tables);
  delete f;
}

int wekws_hip_fbank_num_frames(const wekws_hip_fbank* f, int nsamp) {
  if (!f || nsamp < f->fp.frame_length) return 0;
  return 1 + (nsamp - f->fp.frame_length) / f->fp.frame_shift;  // fbank.h:141-142
}

int wekws_hip_fbank_compute(wekws_hip_fbank* f, const float* pcm, int B, int nsamp, float* feats, void* stream_) {
  if (!f || !pcm || !feats) return fail(WEKWS_HIP_EINVAL, "NULL argument");
  if (B < 0 || nsamp < 0) return fail(WEKWS_HIP_EINVAL, "B=%d nsamp=%d", B, nsamp);
  const int nf = wekws_hip_fbank_num_frames(f, nsamp);
  if (B == 0 || nf == 0) return WEKWS_HIP_OK;
  DeviceGuard guard(f->device);
  const int rc = wekws::launch_fbank<float>(f->fp, pcm, B, nsamp, nf, feats, f->resident_f32, static_cast<hipStream_t>(stream_));
  if (rc) return fail(rc, "fbank launch failed: %s", hipGetErrorString(hipGetLastError()));
  return WEKWS_HIP_OK;
}

int wekws_hip_fbank_compute_i16(wekws_hip_fbank* f, const int16_t* pcm, int B, int nsamp, float* feats, void* stream_) {
  if (!f || !pcm || !feats) return fail(WEKWS_HIP_EINVAL, "NULL argument");
  if (B < 0 || nsamp < 0) return fail(WEKWS_HIP_EINVAL, "B=%d nsamp=%d", B, nsamp);
  const int nf = wekws_hip_fbank_num_frames(f, nsamp);
  if (B == 0 || nf == 0) return WEKWS_HIP_OK;
  DeviceGuard guard(f->device);
  const int rc = wekws::launch_fbank<int16_t>(f->fp, pcm, B, nsamp, nf, feats, f->resident_i16, static_cast<hipStream_t>(stream_));
  if (rc) return fail(rc, "fbank launch failed: %s", hipGetErrorString(hipGetLastError()));
  return WEKWS_HIP_OK;
}

// --------------------------------------------- context expansion + frame skip ---------------------------------------------
int wekws_hip_splice_frames(int T, int right, int skip) {
  if (skip <= 0 || right < 0 || T <= 0) return 0;
  // init_dataset.py:50  feats_ctx[:, :T - right]  -- a NEGATIVE bound (an utterance shorter than its right context) is Python's
  // "all but the last right - T": 2 T - right frames survive -- then :64-65 keeps every skip-th
  const int kept = T >= right ? T - right : (2 * T > right ? 2 * T - right : 0);
  return (kept + skip - 1) / skip;
}

int wekws_hip_splice(const float* feats, int B, int T, int F, int left, int right, int skip, float* out, void* stream_) {
  if (!feats || !out) return fail(WEKWS_HIP_EINVAL, "NULL argument");
  if (B < 0 || T < 0 || F <= 0 || left < 0 || right < 0 || skip <= 0)
    return fail(WEKWS_HIP_EINVAL, "B=%d T=%d F=%d left=%d right=%d skip=%d", B, T, F, left, right, skip);
  // init_dataset.py:45-48: the left-margin loop reads feats_ctx[:, left] -- the reference raises IndexError for left >= T (any B)
  if (left >= 1 && left >= T)
    return fail(WEKWS_HIP_EINVAL, "splice: left context %d >= T = %d (the reference's left-margin loop raises IndexError)", left, T);
  const int To = wekws_hip_splice_frames(T, right, skip);
  if (B == 0 || To == 0) return WEKWS_HIP_OK;
  if ((int64_t(B) * To * (left + right + 1) * F + 255) / 256 > 0x7fffffffLL) return fail(WEKWS_HIP_EINVAL, "splice: too many elements for one launch");
  const int rc = wekws::launch_splice(feats, B, T, F, left, right, skip, To, out, static_cast<hipStream_t>(stream_));
  if (rc) return fail(rc, "splice launch failed: %s", hipGetErrorString(hipGetLastError()));
  return WEKWS_HIP_OK;
}

// --------------------------------------------- MFCC tail (DCT + lifter) ---------------------------------------------
int wekws_hip_dct_lifter(const float* logmel, int64_t rows, int num_bins, int num_ceps, float cepstral_lifter, float* out,
                         void* stream_) {
  if (!logmel || !out) return fail(WEKWS_HIP_EINVAL, "NULL argument");
  if (rows < 0 || num_bins <= 0 || num_bins > wekws::kMfccMaxBins || num_ceps <= 0 || num_ceps > num_bins || cepstral_lifter < 0.f)
    return fail(WEKWS_HIP_EINVAL, "rows=%lld num_bins=%d num_ceps=%d lifter=%g (need 0 < num_ceps <= num_bins <= %d)",
                (long long)rows, num_bins, num_ceps, double(cepstral_lifter), wekws::kMfccMaxBins);
  if (rows == 0) return WEKWS_HIP_OK;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
    return fail(WEKWS_HIP_EDEVICE, "no HIP device");
  const int rc = wekws::launch_dct_lifter(logmel, rows, num_bins, num_ceps, cepstral_lifter, out, cus, static_cast<hipStream_t>(stream_));
  if (rc) return fail(rc, "dct_lifter launch failed: %s", hipGetErrorString(hipGetLastError()));
  return WEKWS_HIP_OK;
}

// --------------------------------------------- CTC first beam prune ---------------------------------------------
int wekws_hip_softmax_topk(const float* logits, int64_t rows, int K, int k, float* probs, int32_t* idx, void* stream_) {
  if (!logits || !probs || !idx) return fail(WEKWS_HIP_EINVAL, "NULL argument");
  if (rows < 0 || K <= 0 || k < 1 || k > wekws::kTopkMax) return fail(WEKWS_HIP_EINVAL, "rows=%lld K=%d k=%d (k must be 1..%d)", (long long)rows, K, k, wekws::kTopkMax);
  if (rows == 0) return WEKWS_HIP_OK;
  if ((rows + 3) / 4 > 0x7fffffffLL) return fail(WEKWS_HIP_EINVAL, "softmax_topk: too many rows for one launch");
  const int rc = wekws::launch_softmax_topk(logits, rows, K, k, probs, idx, static_cast<hipStream_t>(stream_));
  if (rc) return fail(rc, "softmax_topk launch failed: %s", hipGetErrorString(hipGetLastError()));
  return WEKWS_HIP_OK;
}

// --------------------------------------------- DET scoring ---------------------------------------------
int wekws_hip_score_maxpool(const float* scores, int B, int T, int K, const int32_t* lengths, float* max_out,
                            int32_t* argmax_out, void* stream_) {
  if (!scores || !max_out) return fail(WEKWS_HIP_EINVAL, "NULL argument");
  if (B < 0 || T <= 0 || K <= 0) return fail(WEKWS_HIP_EINVAL, "B=%d T=%d K=%d", B, T, K);
  if (B == 0) return WEKWS_HIP_OK;
  if ((int64_t(B) * K + 3) / 4 > 0x7fffffffLL) return fail(WEKWS_HIP_EINVAL, "score_maxpool: too many rows for one launch");
  const int rc = wekws::launch_det_maxpool(scores, B, T, K, lengths, max_out, argmax_out, static_cast<hipStream_t>(stream_));
  if (rc) return fail(rc, "det_maxpool launch failed: %s", hipGetErrorString(hipGetLastError()));
  return WEKWS_HIP_OK;
}

static int det_false_alarms(bool text6, const float* scores, int B, int T, int K, int keyword, const int32_t* lengths,
                            const double* thresholds, int n_thr, int window_shift, int32_t* alarms, void* stream_) {
  if (!scores || !thresholds || !alarms) return fail(WEKWS_HIP_EINVAL, "NULL argument");
  if (B < 0 || T <= 0 || K <= 0 || keyword < 0 || keyword >= K || n_thr <= 0 || window_shift <= 0)
    return fail(WEKWS_HIP_EINVAL, "B=%d T=%d K=%d keyword=%d n_thr=%d window_shift=%d", B, T, K, keyword, n_thr, window_shift);
  if (B == 0) return WEKWS_HIP_OK;
  if ((int64_t(B) * n_thr + 255) / 256 > 0x7fffffffLL) return fail(WEKWS_HIP_EINVAL, "det_false_alarms: too many items for one launch");
  const int rc = wekws::launch_det_alarms(text6, scores, B, T, K, keyword, lengths, thresholds, n_thr, window_shift, alarms,
                                          static_cast<hipStream_t>(stream_));
  if (rc) return fail(rc, "det_alarm launch failed: %s", hipGetErrorString(hipGetLastError()));
  return WEKWS_HIP_OK;
}
int wekws_hip_det_false_alarms(const float* scores, int B, int T, int K, int keyword, const int32_t* lengths,
                               const double* thresholds, int n_thr, int window_shift, int32_t* alarms, void* stream_) {
  return det_false_alarms(false, scores, B, T, K, keyword, lengths, thresholds, n_thr, window_shift, alarms, stream_);
}
int wekws_hip_det_false_alarms_text(const float* scores, int B, int T, int K, int keyword, const int32_t* lengths,
                                    const double* thresholds, int n_thr, int window_shift, int32_t* alarms, void* stream_) {
  return det_false_alarms(true, scores, B, T, K, keyword, lengths, thresholds, n_thr, window_shift, alarms, stream_);
}

}  // extern "C"
