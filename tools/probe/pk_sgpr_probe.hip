// Probe (round 3): does a packed-f32 instruction honour op_sel_hi on an SGPR-pair operand?  The compiler emits
//   v_pk_mul_f32 v[a:b], s[n:n+1], v[c:d] op_sel_hi:[0,1]        (lo = s_n * v_c, hi = s_n * v_d: scalar broadcast)
// for float2 * uniform scalar; this checks what the hardware computes when s[n+1] holds something else.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/pk_sgpr_probe.hip -o build/probe_bin/pk_sgpr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void k(float* out, float s0, float s1) {
  f32x2 v = {3.f, 5.f}, r0, r1, r2, r3;
  unsigned long long sp = (static_cast<unsigned long long>(__float_as_uint(s1)) << 32) | __float_as_uint(s0);
  asm volatile("" : "+s"(sp));
  asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r0) : "s"(sp), "v"(v));
  asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r1) : "v"(v), "s"(sp));
  asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r2) : "s"(sp), "v"(v));
  f32x2 z = {100.f, 200.f};
  asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(r3) : "v"(v), "s"(sp), "v"(z));
  if (threadIdx.x == 0) {
    out[0] = r0.x; out[1] = r0.y; out[2] = r1.x; out[3] = r1.y; out[4] = r2.x; out[5] = r2.y; out[6] = r3.x; out[7] = r3.y;
  }
}
int main() {
  float* d; hipMalloc(&d, 64);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 2.f, 7.f);
  float h[8]; hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
  printf("v = (3, 5), s = (2, 7)\n");
  printf("pk_mul s,v op_sel_hi:[0,1]  -> (%g, %g)   broadcast would be (6, 10)\n", h[0], h[1]);
  printf("pk_mul v,s op_sel_hi:[1,0]  -> (%g, %g)   broadcast would be (6, 10)\n", h[2], h[3]);
  printf("pk_mul s,v (no op_sel)      -> (%g, %g)   pairwise is (6, 35)\n", h[4], h[5]);
  printf("pk_fma v,s,-z op_sel_hi:[1,0,1] -> (%g, %g)   broadcast would be (-94, -190)\n", h[6], h[7]);
  return 0;
}
