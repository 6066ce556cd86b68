// Probe (round 6, third version): WHICH packed-f32 forms return wrong lanes 48..63 beside MFMA waves on gfx950?
// Every form is ONE instruction  d = a * b + c  (or mul / add) between s_nop 7 fences, own destination, operands long settled.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/pk_opsel_probe3.hip -o /tmp/pk_probe3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define NFORM 12
// form id, asm text, expected lo, expected hi
#define FORMS(X)                                                                                                  \
  X(0, "v_pk_fma_f32 %0, %1, %2, %3", fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y))                                     \
  X(1, "v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]", fmaf(a.x, b.x, c.x), fmaf(a.y, b.x, c.y))                   \
  X(2, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]", fmaf(a.x, b.y, c.x), fmaf(a.y, b.y, c.y))                      \
  X(3, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]", fmaf(a.y, b.x, c.x), fmaf(a.y, b.y, c.y))                      \
  X(4, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]", fmaf(a.x, b.x, c.y), fmaf(a.y, b.y, c.y))                      \
  X(5, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]", fmaf(a.x, b.y, c.x), fmaf(a.y, b.x, c.y))    \
  X(6, "v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]", a.x * b.y, a.y * b.y)                                                \
  X(7, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1]", a.x + b.y, a.y + b.y)                                                \
  X(8, "v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]", fmaf(a.x, b.x, c.x), fmaf(a.x, b.y, c.y))                   \
  X(9, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]", fmaf(a.y, -b.y, c.x), fmaf(a.y, b.x, c.y)) \
  X(10, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_hi:[0,1,0]", fmaf(a.x, b.y, c.x), fmaf(a.y, -b.y, c.y)) \
  X(11, "v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]", a.x * b.x, a.x * b.y)

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
    if (s == 12345.f) atomicAdd(bad + 100, 1u);
    return;
  }
  unsigned lo[NFORM] = {}, hi[NFORM] = {}, lo3[NFORM] = {}, loc[NFORM] = {};
  for (int it = 0; it < iters; ++it) {
    const int base = (tid * 13 + it * 7) % 7000;
    const f2 a = f2{lds[base], lds[base + 1]}, b = f2{lds[base + 2], lds[base + 3]}, c = f2{lds[base + 4], lds[base + 5]};
#define RUN(id, text, elo, ehi)                                                                         \
    {                                                                                                   \
      f2 d;                                                                                             \
      asm volatile("s_nop 7\n\t" text "\n\ts_nop 7" : "=&v"(d) : "v"(a), "v"(b), "v"(c));                \
      const float rl = (elo), rh = (ehi);                                                               \
      if (__float_as_uint(d.x) != __float_as_uint(rl)) {                                                \
        lo[id]++;                                                                                       \
        if ((tid & 63) >= 48) lo3[id]++;                                                                \
        if (__float_as_uint(d.x) == __float_as_uint(c.x)) loc[id]++;                                    \
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
  (void)hipMalloc(&d, 4096 * 4); (void)hipMalloc(&bad, 4 * 128);
  (void)hipMemcpy(d, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  static const char* names[NFORM] = {
#define NAME(id, text, elo, ehi) text,
      FORMS(NAME)};
  for (int mw = 0; mw <= 2; mw += 2) {
    (void)hipMemset(bad, 0, 4 * 128);
    probe<<<4096, 256>>>(d, bad, 300, mw);
    unsigned r[128];
    (void)hipMemcpy(r, bad, 4 * 128, hipMemcpyDeviceToHost);
    printf("== %d MFMA waves per workgroup (%s)\n", mw, hipGetErrorString(hipGetLastError()));
    for (int f = 0; f < NFORM; ++f)
      printf("  low wrong %9u (lanes 48..63: %9u; == c.lo, product missing: %9u)  high wrong %9u   %s\n", r[4 * f], r[4 * f + 2], r[4 * f + 3], r[4 * f + 1], names[f]);
  }
  return 0;
}
