// Probe (round 6, second version): the EXACT instruction sequence of the failing ds64_g4 build's head -- four register moves, seven
// v_pk_fma_f32 "step 1" (op_sel_hi:[1,0,0], seven destinations), then the seven "step 2" of which the first is
// v_pk_fma_f32 d, a, b, d op_sel:[0,1,0] -- beside waves of the same SIMD that run MFMA chains.  In the failing build lanes 48..63 of
// that one instruction's LOW result came back without the product (the value of step 1), rarely; tools/probe/d64_dump.py.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/pk_opsel_probe2.hip -o /tmp/pk_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256, 4) void probe(const float* in, unsigned* bad, int iters, int mfma_waves) {
  __shared__ float lds[7168];
  const int tid = threadIdx.x, wave = tid >> 6;
  for (int i = tid; i < 7168; i += 256) lds[i] = in[(blockIdx.x * 7 + i) & 4095];
  __syncthreads();
  unsigned nlo = 0, nhi = 0, nlo3 = 0;
  if (wave < mfma_waves) {
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(tid * 1e-3f + e); b[e] = (_Float16)(tid * 2e-3f - e); }
    f32x4 c[7] = {};
    for (int it = 0; it < iters * 2; ++it) {
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[i], 0, 0, 0);
        c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, c[i], 0, 0, 0);
        c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, a, c[i], 0, 0, 0);
      }
    }
    float s = 0.f;
    for (int i = 0; i < 7; ++i) s += c[i][0];
    if (s == 12345.f) nhi = 1;
  } else {
    for (int it = 0; it < iters; ++it) {
      const int base = (tid * 13 + it * 7) % 7000;
      f2 w_x = f2{lds[base], lds[base + 1]}, w_y = f2{lds[base + 2], lds[base + 3]};   // (w0.x, w1.x), (w0.y, w1.y)
      f2 h[7], acc[7];
      for (int t = 0; t < 7; ++t) h[t] = f2{lds[base + 4 + 2 * t], lds[base + 5 + 2 * t]};   // (hv[t][0], hv[t][1])
      asm volatile(
          "v_pk_fma_f32 %0, %7, %9, 0 op_sel_hi:[1,0,0]\n\t"
          "v_pk_fma_f32 %1, %7, %10, 0 op_sel_hi:[1,0,0]\n\t"
          "v_pk_fma_f32 %2, %7, %11, 0 op_sel_hi:[1,0,0]\n\t"
          "v_pk_fma_f32 %3, %7, %12, 0 op_sel_hi:[1,0,0]\n\t"
          "v_pk_fma_f32 %4, %7, %13, 0 op_sel_hi:[1,0,0]\n\t"
          "v_pk_fma_f32 %5, %7, %14, 0 op_sel_hi:[1,0,0]\n\t"
          "v_pk_fma_f32 %6, %7, %15, 0 op_sel_hi:[1,0,0]\n\t"
          "v_pk_fma_f32 %0, %8, %9, %0 op_sel:[0,1,0]\n\t"
          "v_pk_fma_f32 %1, %8, %10, %1 op_sel:[0,1,0]\n\t"
          "v_pk_fma_f32 %2, %8, %11, %2 op_sel:[0,1,0]\n\t"
          "v_pk_fma_f32 %3, %8, %12, %3 op_sel:[0,1,0]\n\t"
          "v_pk_fma_f32 %4, %8, %13, %4 op_sel:[0,1,0]\n\t"
          "v_pk_fma_f32 %5, %8, %14, %5 op_sel:[0,1,0]\n\t"
          "v_pk_fma_f32 %6, %8, %15, %6 op_sel:[0,1,0]\n\t"
          "s_nop 1"
          : "=&v"(acc[0]), "=&v"(acc[1]), "=&v"(acc[2]), "=&v"(acc[3]), "=&v"(acc[4]), "=&v"(acc[5]), "=&v"(acc[6])
          : "v"(w_x), "v"(w_y), "v"(h[0]), "v"(h[1]), "v"(h[2]), "v"(h[3]), "v"(h[4]), "v"(h[5]), "v"(h[6]));
      for (int t = 0; t < 7; ++t) {
        const float rx = fmaf(w_y.x, h[t].y, w_x.x * h[t].x), ry = fmaf(w_y.y, h[t].y, w_x.y * h[t].x);
        if (__float_as_uint(acc[t].x) != __float_as_uint(rx)) { nlo += 1; if ((tid & 63) >= 48) nlo3 += 1; }
        if (__float_as_uint(acc[t].y) != __float_as_uint(ry)) nhi += 1;
      }
    }
  }
  if (nlo) atomicAdd(bad, nlo);
  if (nhi) atomicAdd(bad + 1, nhi);
  if (nlo3) atomicAdd(bad + 2, nlo3);
}

int main() {
  std::vector<float> h(4096);
  for (int i = 0; i < 4096; ++i) h[i] = float((i * 2654435761u) >> 8 & 0xffff) / 6553.6f - 5.f;
  float* d; unsigned* bad;
  (void)hipMalloc(&d, 4096 * 4); (void)hipMalloc(&bad, 16);
  (void)hipMemcpy(d, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  for (int mw = 0; mw <= 3; ++mw)
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipMemset(bad, 0, 16);
      probe<<<4096, 256>>>(d, bad, 1000, mw);
      unsigned r[4];
      (void)hipMemcpy(r, bad, 16, hipMemcpyDeviceToHost);
      printf("%d MFMA waves per workgroup, rep %d: low-half mismatches %u (of them lanes 48..63: %u), high-half %u (%s)\n", mw, rep, r[0], r[2], r[1],
             hipGetErrorString(hipGetLastError()));
    }
  return 0;
}
