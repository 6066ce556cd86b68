// Probe (round 6): the interleave rows of overlap_probe.hip again, with the vector instructions the depthwise phase really issues --
// v_fmac_f32_dpp row_shr (KIND 1) and the split mix of g16_split_pair (KIND 2: max, mul, cvt_pk, cvt back, sub, cvt_pk) -- next to
// plain v_fma_f32 (KIND 0), and with the B-fragment traffic of the matrix phase (two ds_read_b128 per three MFMAs, LDS = 1).
// Every wave runs groups of [1 MFMA ; N vector instructions]; prints timer ticks per group and SIMD (an MFMA alone: ~10.4 ticks).
//   hipcc -O3 -w --offload-arch=gfx950 tools/probe/overlap_probe2.hip -o build/var/overlap_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__device__ __forceinline__ void vop(float (&v)[8], int i, float c1, float c2) {
  if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i & 7]) : "v"(c1), "v"(c2));
  else if constexpr (KIND == 1) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v[i & 7]) : "v"(v[(i + 3) & 7]), "v"(c1));
  else {                                                       // one sixth of a split per slot, roughly: alternate the forms
    const int k = i % 6;
    if (k == 0) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[i & 7]) : "v"(c2));
    else if (k == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i & 7]) : "v"(c1));
    else if (k == 2 || k == 5) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(v[i & 7]) : "v"(v[(i + 1) & 7]), "v"(v[(i + 2) & 7]));
    else if (k == 3) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(v[i & 7]) : "v"(v[(i + 1) & 7]));
    else asm volatile("v_sub_f32 %0, %1, %2" : "=v"(v[i & 7]) : "v"(v[(i + 1) & 7]), "v"(v[(i + 2) & 7]));
  }
}

template <int KIND, int N, int LDS>
__global__ __launch_bounds__(1024) void inter_k(int iters, long long* cyc, float* sink) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  const int wave = threadIdx.x >> 6;
  float s = threadIdx.x * 1e-3f;
  const float c1 = 1.0001f, c2 = 0.5f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = s + i;
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i * 1e-4f;
  f16x8 a, b, b2;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(s + e); b[e] = (_Float16)(s - e); b2[e] = b[e]; }
  f32x4 c4[8];
  for (int i = 0; i < 8; ++i) c4[i] = f32x4{0, 0, 0, 0};
  const f16x8* lp = reinterpret_cast<const f16x8*>(lds) + (threadIdx.x & 255);
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 9; ++g) {
      if (LDS && g % 3 == 0) { b = lp[(g / 3) * 256]; b2 = lp[(g / 3) * 256 + 768]; }
      __builtin_amdgcn_sched_barrier(0);
      c4[g & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, (g & 1) ? b2 : b, c4[g & 7], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < N; ++i) vop<KIND>(v, g * N + i, c1, c2);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  for (int i = 0; i < 8; ++i) s += v[i] + c4[i][0];
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
  sink[blockIdx.x * 1024 + threadIdx.x] = s;
}

static long long* dc;
static float* ds;
template <int KIND, int N, int LDS>
void run(int threads) {
  const int iters = 500;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((inter_k<KIND, N, LDS>), dim3(256), dim3(threads), 0, 0, iters, dc, ds);
    hipDeviceSynchronize();
  }
  std::vector<long long> c(256 * 16);
  hipMemcpy(c.data(), dc, c.size() * 8, hipMemcpyDeviceToHost);
  const int nw = threads / 64;
  double m = 0;
  for (int b = 0; b < 256; ++b) for (int w = 0; w < nw; ++w) m += c[b * 16 + w];
  m /= 256.0 * nw;
  static const char* kn[] = {"v_fma_f32", "v_fmac_f32_dpp", "split mix"};
  printf("%-15s lds=%d %d waves/SIMD  N=%d per MFMA: %6.2f ticks per group and SIMD\n", kn[KIND], LDS, nw / 4, N, m / (iters * 9.0) / (nw / 4));
}
template <int KIND, int LDS>
void run_all() {
  for (int threads : {512, 1024}) {
    run<KIND, 0, LDS>(threads); run<KIND, 1, LDS>(threads); run<KIND, 2, LDS>(threads); run<KIND, 3, LDS>(threads);
    run<KIND, 4, LDS>(threads); run<KIND, 6, LDS>(threads);
  }
}
int main() {
  hipMalloc(&dc, 256 * 16 * 8);
  hipMalloc(&ds, 256 * 1024 * 4);
  run_all<0, 0>(); run_all<1, 0>(); run_all<2, 0>();
  run_all<0, 1>(); run_all<1, 1>(); run_all<2, 1>();
  return 0;
}
