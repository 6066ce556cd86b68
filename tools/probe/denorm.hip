#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ void k(float* out) {
  const int l = threadIdx.x;
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)0.f; b[e] = (_Float16)0.f; }
  // A[i][k]: put a subnormal fp16 (2^-20 ~ 9.54e-7) at k=0 for every row; B[0][j] = 1 -> D[i][j] = 2^-20 if not flushed
  if ((l >> 4) == 0) { a[0] = (_Float16)9.5367431640625e-07f; b[0] = (_Float16)1.0f; a[1] = (_Float16)3.0e-5f; b[1] = (_Float16)0.5f; }
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  if (l == 0) { out[0] = c[0]; out[1] = (float)a[0]; out[2] = (float)a[1]; }
  // split accuracy check: x = hi + lo
  float x = 0.0123456789f * (l + 1);
  _Float16 hi = (_Float16)x; _Float16 lo = (_Float16)(x - (float)hi);
  out[4 + l] = x - ((float)hi + (float)lo);
}
int main() {
  float* d; hipMalloc(&d, 4 * 128);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[128]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  printf("D = %.9g (expect 2^-20 + 1.5e-5 = %.9g if subnormals honoured; 0 or 1.5e-5 if flushed)  a0=%g a1=%g\n", h[0], 9.5367431640625e-07 + 0.5 * (double)(float)(_Float16)3.0e-5f, h[1], h[2]);
  double m = 0; for (int i = 0; i < 64; ++i) m = fmax(m, fabs(h[4 + i]) / (0.0123456789 * (i + 1)));
  printf("max relative residual of hi+lo split: %.3g\n", m);
  return 0;
}
