// Probe: (1) lane layout of v_mfma_f32_16x16x32_f16 operands, (2) issue rate of f32 vs f16 MFMA from one wave per SIMD,
// (3) whether a co-resident VALU-only wave slows the MFMA wave down (and vice versa).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <vector>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ void layout_k(const _Float16* A /*16x32 row-major [i][k]*/, const _Float16* B /*32x16 [k][j]*/, float* D) {
  const int l = threadIdx.x;
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = A[(l & 15) * 32 + (l >> 4) * 8 + e]; b[e] = B[((l >> 4) * 8 + e) * 16 + (l & 15)]; }
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}

// mode bit0: waves 0-3 run MFMA (kind = f32 / f16); bit1: waves 4-7 run a VALU fma chain.  out: cycles per wave
template <int KIND>
__global__ __launch_bounds__(512) void rate_k(int mode, int iters, long long* cyc, float* sink) {
  const int wave = threadIdx.x >> 6;
  long long t0 = 0, t1 = 0;
  float s = threadIdx.x * 1e-3f;
  if (wave < 4) {
    if (mode & 1) {
      f32x4 c[8];
      for (int i = 0; i < 8; ++i) c[i] = f32x4{0, 0, 0, 0};
      f16x8 a, b;
      for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(s + e); b[e] = (_Float16)(s - e); }
      float af = s, bf = s + 1.f;
      __syncthreads();
      t0 = __builtin_readcyclecounter();
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (KIND == 0) c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, c[i], 0, 0, 0);
          else c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[i], 0, 0, 0);
        }
      }
      t1 = __builtin_readcyclecounter();
      for (int i = 0; i < 8; ++i) s += c[i][0];
    } else { __syncthreads(); }
  } else {
    if (mode & 2) {
      if (mode & 4) __builtin_amdgcn_s_setprio(3);
      float v[8];
      for (int i = 0; i < 8; ++i) v[i] = s + i;
      __syncthreads();
      t0 = __builtin_readcyclecounter();
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = fmaf(v[i], 1.0001f, 0.5f);
      }
      t1 = __builtin_readcyclecounter();
      for (int i = 0; i < 8; ++i) s += v[i];
    } else { __syncthreads(); }
  }
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
  sink[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
  // ---- layout
  std::vector<_Float16> A(16 * 32), B(32 * 16);
  for (int i = 0; i < 16; ++i) for (int k = 0; k < 32; ++k) A[i * 32 + k] = (_Float16)((i * 7 + k * 3) % 11 - 5);
  for (int k = 0; k < 32; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = (_Float16)((k * 5 + j * 2) % 13 - 6);
  _Float16 *dA, *dB; float* dD;
  hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dD, 256 * 4);
  hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(layout_k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  std::vector<float> D(256); hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
  double err = 0;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double r = 0; for (int k = 0; k < 32; ++k) r += (double)A[i * 32 + k] * (double)B[k * 16 + j]; err = fmax(err, fabs(r - D[i * 16 + j])); }
  printf("layout check (A[i=l&15][k=8*(l>>4)+e], B[k=8*(l>>4)+e][j=l&15], D[4*(l>>4)+r][l&15]): max err %g\n", err);
  // ---- rates
  long long* dc; float* ds; hipMalloc(&dc, 256 * 8 * 8); hipMalloc(&ds, 256 * 512 * 4);
  const int iters = 2000;
  for (int kind = 0; kind < 2; ++kind) for (int mode : {1, 2, 3, 7}) {
    for (int rep = 0; rep < 2; ++rep) {
      if (kind == 0) hipLaunchKernelGGL(rate_k<0>, dim3(256), dim3(512), 0, 0, mode, iters, dc, ds);
      else hipLaunchKernelGGL(rate_k<1>, dim3(256), dim3(512), 0, 0, mode, iters, dc, ds);
      hipDeviceSynchronize();
    }
    std::vector<long long> c(256 * 8); hipMemcpy(c.data(), dc, c.size() * 8, hipMemcpyDeviceToHost);
    double m = 0, v = 0; for (int b = 0; b < 256; ++b) { for (int w = 0; w < 4; ++w) m += c[b * 8 + w]; for (int w = 4; w < 8; ++w) v += c[b * 8 + w]; }
    m /= 1024; v /= 1024;
    printf("%s mode=%d  mfma wave: %.1f cyc per MFMA   valu wave: %.2f cyc per v_fma\n", kind ? "f16 16x16x32" : "f32 16x16x4 ", mode,
           m / (iters * 8.0), v / (iters * 8.0));
  }
  return 0;
}
