// Probe: lane semantics of DPP row_shr / row_shl / row_ror (v_mov_b32_dpp and v_fmac_f32_dpp), bound_ctrl off.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* out) {
  const int l = threadIdx.x;
  float xc = 100.f + l, xp = 200.f + l, w = 1.f;
  float a = -1.f, b = -1.f, c = -1.f, o = 0.f, o2 = 0.f;
  asm volatile("s_nop 4\n\tv_mov_b32_dpp %0, %1 row_shr:3 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(xc));
  asm volatile("s_nop 4\n\tv_mov_b32_dpp %0, %1 row_shl:13 row_mask:0xf bank_mask:0xf" : "+v"(b) : "v"(xp));
  asm volatile("s_nop 4\n\tv_mov_b32_dpp %0, %1 row_ror:3 row_mask:0xf bank_mask:0xf" : "+v"(c) : "v"(xc));
  asm volatile("s_nop 4\n\tv_fmac_f32_dpp %0, %1, %2 row_shr:3 row_mask:0xf bank_mask:0xf" : "+v"(o) : "v"(xc), "v"(w));
  asm volatile("s_nop 4\n\tv_fmac_f32_dpp %0, %1, %2 row_shl:13 row_mask:0xf bank_mask:0xf" : "+v"(o2) : "v"(xp), "v"(w));
  out[l] = a; out[64 + l] = b; out[128 + l] = c; out[192 + l] = o; out[256 + l] = o2;
}
int main() {
  float* d; hipMalloc(&d, 320 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[320]; hipMemcpy(h, d, 1280, hipMemcpyDeviceToHost);
  const char* n[5] = {"mov row_shr:3 (old=-1)", "mov row_shl:13 (old=-1)", "mov row_ror:3", "fmac row_shr:3 (acc 0)", "fmac row_shl:13 (acc 0)"};
  for (int r = 0; r < 5; ++r) { printf("%-26s", n[r]); for (int i = 0; i < 18; ++i) printf(" %g", h[r * 64 + i]); printf("\n"); }
  return 0;
}
