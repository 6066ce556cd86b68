// Issue cost of individual VALU instructions (cycles per wave64 instruction) with 1, 2 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ void k(int iters, long long* cyc, float* sink) {
  float v[8]; f32x2 p[8]; 
  for (int i = 0; i < 8; ++i) { v[i] = threadIdx.x * 1e-3f + i; p[i] = f32x2{v[i], v[i] + 1.f}; }
  unsigned u[8]; for (int i = 0; i < 8; ++i) u[i] = threadIdx.x + i;
  __syncthreads();
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == 0) v[i] = fmaf(v[i], 1.0001f, 0.5f);
      else if (OP == 1) p[i] = __builtin_elementwise_fma(p[i], f32x2{1.0001f, 1.0002f}, f32x2{0.5f, 0.25f});
      else if (OP == 2) { _Float16 h = (_Float16)v[i]; v[i] = v[i] + (float)h; }            // cvt_f16 + cvt_f32 + add
      else if (OP == 3) { asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(u[i]) : "v"(v[i])); }
      else if (OP == 4) { asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(v[i]), "v"(v[(i + 1) & 7])); }
      else if (OP == 5) { asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(u[i]) : "v"(u[i]), "v"(u[(i + 1) & 7])); }
      else if (OP == 6) { asm volatile("v_max_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(v[(i + 1) & 7])); }
      else if (OP == 7) { asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(v[i]) : "v"(u[i])); }
      else if (OP == 8) { asm volatile("v_lshl_add_u64 %0, %1, 2, %2" : "=v"(*(unsigned long long*)&p[i]) : "v"(*(unsigned long long*)&p[i]), "v"(*(unsigned long long*)&p[(i + 1) & 7])); }
      else if (OP == 9) { asm volatile("v_add_u32 %0, %1, %2" : "=v"(u[i]) : "v"(u[i]), "v"(u[(i + 1) & 7])); }
      // round 3: DPP row shifts (independent accumulators / one dependent chain), and the split's mix instructions
      else if (OP == 10) { asm volatile("v_fmac_f32_dpp %0, %1, %2 row_shr:3 row_mask:0xf bank_mask:0xf" : "+v"(v[i]) : "v"(p[i].x), "v"(p[i].y)); }
      else if (OP == 11) { asm volatile("v_fmac_f32_dpp %0, %1, %2 row_shr:3 row_mask:0xf bank_mask:0xf" : "+v"(v[0]) : "v"(p[i].x), "v"(p[i].y)); }
      else if (OP == 12) { asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[0]) : "v"(p[i].x), "v"(p[i].y)); }
      else if (OP == 13) { asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(u[i]) : "v"(v[i]), "v"(p[i].y)); }
      else if (OP == 14) { asm volatile("v_mov_b32_dpp %0, %1 row_shr:3 row_mask:0xf bank_mask:0xf" : "+v"(v[i]) : "v"(p[i].x)); }
      else if (OP == 15) { asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[i]) : "v"(p[i].x), "v"(p[i].y)); }
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0; for (int i = 0; i < 8; ++i) s += v[i] + p[i].x + p[i].y + u[i];
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(const char* name, long long* dc, float* ds) {
  const int iters = 2000;
  printf("%-28s", name);
  for (int threads : {256, 512, 1024}) {
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, iters, dc, ds);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, iters, dc, ds);
    hipDeviceSynchronize();
    std::vector<long long> c(256 * 16); hipMemcpy(c.data(), dc, c.size() * 8, hipMemcpyDeviceToHost);
    double m = 0; int nw = threads / 64; for (int b = 0; b < 256; ++b) for (int w = 0; w < nw; ++w) m += c[b * 16 + w];
    m /= 256.0 * nw;
    printf("  %dw/SIMD: %5.2f cyc/instr (SIMD: %4.2f)", nw / 4, m / (iters * 8.0), m / (iters * 8.0) / (nw / 4));
  }
  printf("\n");
}
int main() {
  long long* dc; float* ds; hipMalloc(&dc, 256 * 16 * 8); hipMalloc(&ds, 256 * 1024 * 4);
  run<0>("v_fma_f32", dc, ds); run<1>("v_pk_fma_f32 (2 fma)", dc, ds); run<2>("cvt16+cvt32+add (3 ops)", dc, ds);
  run<3>("v_cvt_f16_f32", dc, ds); run<4>("v_cvt_pkrtz_f16_f32", dc, ds); run<5>("v_cndmask_b32", dc, ds);
  run<6>("v_max_f32", dc, ds); run<7>("v_cvt_f32_f16", dc, ds); run<8>("v_lshl_add_u64", dc, ds); run<9>("v_add_u32", dc, ds);
  run<15>("v_fmac_f32 (8 chains)", dc, ds); run<10>("v_fmac_f32_dpp (8 chains)", dc, ds);
  run<12>("v_fmac_f32 (1 chain)", dc, ds); run<11>("v_fmac_f32_dpp (1 chain)", dc, ds);
  run<13>("v_fma_mixlo_f16", dc, ds); run<14>("v_mov_b32_dpp", dc, ds);
  return 0;
}
