// Probe (round 6, seventh version: constant operands with operand selects on the remaining registers): WHICH packed-f32 forms return wrong lanes 48..63 beside MFMA waves on gfx950?
// Every form is ONE instruction  d = a * b + c  (or mul / add) between s_nop 7 fences, own destination, operands long settled.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/pk_opsel_probe3.hip -o /tmp/pk_probe3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define NFORM 11
#define FORMS(X) \
  X(0, "v_pk_fma_f32 %0, %1, 0.5, %3 op_sel:[0,0,1] op_sel_hi:[1,0,0] neg_hi:[0,0,1]", rfma(ax, 0.5f, cy), rfma(ay, 0.5f, -cx)) \
  X(1, "v_pk_fma_f32 %0, %1, 0.5, %3 op_sel:[0,0,1] op_sel_hi:[1,0,1]", rfma(ax, 0.5f, cy), rfma(ay, 0.5f, cy)) \
  X(2, "v_pk_fma_f32 %0, %1, 0.5, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]", rfma(ax, 0.5f, cx), rfma(ay, 0.5f, cy)) \
  X(3, "v_pk_fma_f32 %0, %1, 0.5, %3 op_sel:[1,0,0] op_sel_hi:[1,0,1]", rfma(ay, 0.5f, cx), rfma(ay, 0.5f, cy)) \
  X(4, "v_pk_fma_f32 %0, 0.5, %2, %3 op_sel:[0,0,1] op_sel_hi:[0,1,1]", rfma(0.5f, bx, cy), rfma(0.5f, by, cy)) \
  X(5, "v_pk_fma_f32 %0, 0.5, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,1,1]", rfma(0.5f, by, cx), rfma(0.5f, by, cy)) \
  X(6, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,1]", rfma(ax, bx, cy), rfma(ay, by, cy)) \
  X(7, "v_pk_fma_f32 %0, %1, %2, 0 op_sel:[0,1,0] op_sel_hi:[1,1,0]", rmul(ax, by), rmul(ay, by)) \
  X(8, "v_pk_mul_f32 %0, %1, 2.0 op_sel:[1,0] op_sel_hi:[1,0]", rmul(ay, 2.0f), rmul(ay, 2.0f)) \
  X(9, "v_pk_add_f32 %0, 1.0, %2 op_sel:[0,1] op_sel_hi:[0,1]", radd(1.0f, by), radd(1.0f, by)) \
  X(10, "v_pk_add_f32 %0, %1, 1.0 op_sel:[1,0] op_sel_hi:[1,0]", radd(ay, 1.0f), radd(ay, 1.0f)) \

// the expected values come from PLAIN (VOP3, one value per lane) instructions written out, so that the compiler cannot turn the
// reference itself into one of the packed forms under test
__device__ __forceinline__ float rfma(float a, float b, float c) { float d; asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ float rmul(float a, float b) { float d; asm volatile("v_mul_f32_e64 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ float radd(float a, float b) { float d; asm volatile("v_add_f32_e64 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }

__global__ __launch_bounds__(256, 4) void probe(const float* in, unsigned* bad, int iters, int mfma_waves) {
  __shared__ float lds[7168];
  const int tid = threadIdx.x, wave = tid >> 6;
  for (int i = tid; i < 7168; i += 256) lds[i] = in[(blockIdx.x * 7 + i) & 4095];
  __syncthreads();
  if (wave < mfma_waves) {
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(tid * 1e-3f + e); b[e] = (_Float16)(tid * 2e-3f - e); }
    f32x4 c[7] = {};
    for (int it = 0; it < iters * 6; ++it) {
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[i], 0, 0, 0);
        c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, c[i], 0, 0, 0);
        c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, a, c[i], 0, 0, 0);
      }
    }
    float s = 0.f;
    for (int i = 0; i < 7; ++i) s += c[i][0];
    if (s == 12345.f) atomicAdd(bad + 4 * NFORM + 4, 1u);
    return;
  }
  unsigned lo[NFORM] = {}, hi[NFORM] = {}, lo3[NFORM] = {}, loc[NFORM] = {};
  for (int it = 0; it < iters; ++it) {
    const int base = (tid * 13 + it * 7) % 7000;
    float ax = lds[base], ay = lds[base + 1], bx = lds[base + 2], by = lds[base + 3], cx = lds[base + 4], cy = lds[base + 5];
    asm volatile("" : "+v"(ax), "+v"(ay), "+v"(bx), "+v"(by), "+v"(cx), "+v"(cy));
    const f2 a = f2{ax, ay}, b = f2{bx, by}, c = f2{cx, cy};
#define RUN(id, text, elo, ehi)                                                                         \
    {                                                                                                   \
      f2 d;                                                                                             \
      asm volatile("s_nop 7\n\t" text "\n\ts_nop 7" : "=&v"(d) : "v"(a), "v"(b), "v"(c));                \
      const float rl = (elo), rh = (ehi);                                                               \
      if (__float_as_uint(d.x) != __float_as_uint(rl)) {                                                \
        lo[id]++;                                                                                       \
        if ((tid & 63) >= 48) lo3[id]++;                                                                \
        if (__float_as_uint(d.x) == __float_as_uint(cx)) loc[id]++;                                    \
      }                                                                                                 \
      if (__float_as_uint(d.y) != __float_as_uint(rh)) hi[id]++;                                        \
    }
    FORMS(RUN)
  }
  for (int f = 0; f < NFORM; ++f) {
    if (lo[f]) atomicAdd(bad + 4 * f, lo[f]);
    if (hi[f]) atomicAdd(bad + 4 * f + 1, hi[f]);
    if (lo3[f]) atomicAdd(bad + 4 * f + 2, lo3[f]);
    if (loc[f]) atomicAdd(bad + 4 * f + 3, loc[f]);
  }
}

int main() {
  std::vector<float> h(4096);
  for (int i = 0; i < 4096; ++i) h[i] = float((i * 2654435761u) >> 8 & 0xffff) / 6553.6f - 5.f;
  float* d; unsigned* bad;
  (void)hipMalloc(&d, 4096 * 4); (void)hipMalloc(&bad, 4 * 4 * (NFORM + 32));
  (void)hipMemcpy(d, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  static const char* names[NFORM] = {
#define NAME(id, text, elo, ehi) text,
      FORMS(NAME)};
  for (int mw = 0; mw <= 2; mw += 2) {
    (void)hipMemset(bad, 0, 4 * 4 * (NFORM + 32));
    probe<<<4096, 256>>>(d, bad, 100, mw);
    static unsigned r[4 * (NFORM + 32)];
    (void)hipMemcpy(r, bad, 4 * 4 * NFORM, hipMemcpyDeviceToHost);
    printf("== %d MFMA waves per workgroup (%s)\n", mw, hipGetErrorString(hipGetLastError()));
    for (int f = 0; f < NFORM; ++f)
      printf("  low wrong %9u (lanes 48..63: %9u; == c.lo, product missing: %9u)  high wrong %9u   %s\n", r[4 * f], r[4 * f + 2], r[4 * f + 3], r[4 * f + 1], names[f]);
  }
  return 0;
}
