// Probe (round 6, eighth version): WHICH neighbours trigger the packed-f32 hazard of wekws_amd/csrc/pk_safe.hip.h?  Waves 2, 3 of every
// workgroup run  v_pk_fma_f32 d, a, b, c op_sel:[0,1,0]  (the known-bad form) against a plain-VOP3 reference; waves 0, 1 run one kind
// of neighbour work: nothing, plain vector FMAs, LDS traffic, DPP FMAs, transcendental ops, or MFMAs of several shapes / types.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/pk_opsel_probe8.hip -o /tmp/pk_probe8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float rfma(float a, float b, float c) { float d; asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }

__global__ __launch_bounds__(256, 4) void probe(const float* in, unsigned* bad, int iters, int kind) {
  __shared__ float lds[7168];
  const int tid = threadIdx.x, wave = tid >> 6;
  for (int i = tid; i < 7168; i += 256) lds[i] = in[(blockIdx.x * 7 + i) & 4095];
  __syncthreads();
  if (wave < 2) {
    float s = tid * 1e-3f;
    const int n = iters * 60;
    if (kind == 1) {                                             // plain vector FMAs
      float v[8];
      for (int i = 0; i < 8; ++i) v[i] = s + i;
      for (int it = 0; it < n * 4; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(1.0001f), "v"(0.5f));
      for (int i = 0; i < 8; ++i) s += v[i];
    } else if (kind == 2) {                                      // LDS reads + writes
      float* p = lds + (tid & 127) * 4;
      for (int it = 0; it < n * 4; ++it) { f32x4 q = *reinterpret_cast<volatile f32x4*>(p); q[0] += 1.f; *reinterpret_cast<volatile f32x4*>(p + 512 * 4) = q; }
    } else if (kind == 3) {                                      // DPP FMAs
      float v[8];
      for (int i = 0; i < 8; ++i) v[i] = s + i;
      for (int it = 0; it < n * 4; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v[i]) : "v"(v[(i + 3) & 7]), "v"(1e-6f));
      for (int i = 0; i < 8; ++i) s += v[i];
    } else if (kind == 4) {                                      // transcendental
      float v[8];
      for (int i = 0; i < 8; ++i) v[i] = s + i;
      for (int it = 0; it < n * 2; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %0\n\ts_nop 0\n\tv_log_f32 %0, %0" : "+v"(v[i]));
      for (int i = 0; i < 8; ++i) s += v[i];
    } else if (kind >= 5) {
      f16x8 a, b;
      for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(s + e); b[e] = (_Float16)(s - e); }
      if (kind == 5) {                                           // f16 16x16x32
        f32x4 c[7] = {};
        for (int it = 0; it < n; ++it)
#pragma unroll
          for (int i = 0; i < 7; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[i], 0, 0, 0);
        for (int i = 0; i < 7; ++i) s += c[i][0];
      } else if (kind == 6) {                                    // exact f32 16x16x4
        f32x4 c[7] = {};
        for (int it = 0; it < n; ++it)
#pragma unroll
          for (int i = 0; i < 7; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(s, s + 1.f, c[i], 0, 0, 0);
        for (int i = 0; i < 7; ++i) s += c[i][0];
      } else if (kind == 7) {                                    // f16 32x32x16
        f32x16 c[3] = {};
        for (int it = 0; it < n / 2; ++it)
#pragma unroll
          for (int i = 0; i < 3; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[i], 0, 0, 0);
        for (int i = 0; i < 3; ++i) s += c[i][0];
      } else if (kind == 8) {                                    // i8 16x16x64
        i32x4 c[7] = {};
        i32x4 ai = {tid, tid * 3, tid * 5, tid * 7};
        for (int it = 0; it < n; ++it)
#pragma unroll
          for (int i = 0; i < 7; ++i) c[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ai, ai, c[i], 0, 0, 0);
        for (int i = 0; i < 7; ++i) s += float(c[i][0]);
      } else if (kind == 9) {                                    // ONE MFMA every ~100 cycles (a sparse matrix stream)
        f32x4 c = {};
        for (int it = 0; it < n; ++it) { c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); __builtin_amdgcn_s_sleep(1); }
        s += c[0];
      }
    }
    if (s == 12345.f) atomicAdd(bad + 3, 1u);
    return;
  }
  unsigned nlo = 0, nlo3 = 0, nhi = 0;
  for (int it = 0; it < iters; ++it) {
    const int base = (tid * 13 + it * 7) % 7000;
    float ax = lds[base], ay = lds[base + 1], bx = lds[base + 2], by = lds[base + 3], cx = lds[base + 4], cy = lds[base + 5];
    asm volatile("" : "+v"(ax), "+v"(ay), "+v"(bx), "+v"(by), "+v"(cx), "+v"(cy));
    const f2 a = f2{ax, ay}, b = f2{bx, by}, c = f2{cx, cy};
    f2 d;
    asm volatile("s_nop 7\n\tv_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]\n\ts_nop 7" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
    if (__float_as_uint(d.x) != __float_as_uint(rfma(ax, by, cx))) { ++nlo; if ((tid & 63) >= 48) ++nlo3; }
    if (__float_as_uint(d.y) != __float_as_uint(rfma(ay, by, cy))) ++nhi;
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
  static const char* names[] = {"nothing", "plain v_fma_f32 chains", "LDS reads + writes", "v_fmac_f32_dpp chains", "v_exp_f32 / v_log_f32",
                                "v_mfma_f32_16x16x32_f16", "v_mfma_f32_16x16x4_f32", "v_mfma_f32_32x32x16_f16", "v_mfma_i32_16x16x64_i8",
                                "one f16 MFMA per ~100 cycles"};
  for (int kind = 0; kind < 10; ++kind) {
    (void)hipMemset(bad, 0, 16);
    probe<<<4096, 256>>>(d, bad, 300, kind);
    unsigned r[4];
    (void)hipMemcpy(r, bad, 16, hipMemcpyDeviceToHost);
    printf("neighbours: %-30s low wrong %9u of %u (lanes 48..63: %9u)  high wrong %u  (%s)\n", names[kind], r[0], 4096u * 128u * 300u, r[2], r[1],
           hipGetErrorString(hipGetLastError()));
  }
  return 0;
}
