// Probe (round 6): does a v_pk_fma_f32 whose LOW result takes the HIGH half of a source (op_sel:[0,1,0]) ever return a wrong low
// half on gfx950 when several workgroups share a CU and other waves keep the matrix pipe / LDS busy?  (ds64_g4's rare wrong
// posteriors -- profiles/r06_experiments.txt -- always sat in the one output whose chain the compiler had lowered to that form.)
//   hipcc -O3 --offload-arch=gfx950 tools/probe/pk_opsel_probe.hip -o build/probe_bin/pk_opsel_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256, 4) void probe(const float* in, unsigned* bad, int iters, int mode) {
  __shared__ float lds[7168];
  const int tid = threadIdx.x, wave = tid >> 6;
  for (int i = tid; i < 7168; i += 256) lds[i] = in[(blockIdx.x * 7 + i) & 4095];
  __syncthreads();
  unsigned nbad = 0;
  if (mode == 1 && wave < 2) {                                   // two waves of the workgroup keep the matrix pipe busy
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(tid * 1e-3f + e); b[e] = (_Float16)(tid * 2e-3f - e); }
    f32x4 c[4] = {};
    for (int it = 0; it < iters * 4; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[i], 0, 0, 0);
    }
    if (c[0][0] + c[1][0] + c[2][0] + c[3][0] == 12345.f) nbad = 1u << 31;
  } else {
    for (int it = 0; it < iters; ++it) {
      const int base = (tid * 13 + it * 7) % 7000;
      float w0x = lds[base], w1x = lds[base + 1], w0y = lds[base + 2], w1y = lds[base + 3];
      f2 h = f2{lds[base + 4], lds[base + 5]}, acc, ref;
      f2 wx = f2{w0x, w1x}, wy = f2{w0y, w1y};
      // acc = wx * h.lo  (both halves), then acc += wy * h.hi with op_sel:[0,1,0] -- the sequence of the failing build
      asm volatile("v_pk_fma_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]\n\t"
                   "s_nop 1\n\t"
                   "v_pk_fma_f32 %0, %3, %2, %0 op_sel:[0,1,0]\n\t"
                   "s_nop 1"
                   : "=&v"(acc) : "v"(wx), "v"(h), "v"(wy));
      ref.x = fmaf(w0y, h.y, w0x * h.x);
      ref.y = fmaf(w1y, h.y, w1x * h.x);
      if (__float_as_uint(acc.x) != __float_as_uint(ref.x)) nbad += 1;
      if (__float_as_uint(acc.y) != __float_as_uint(ref.y)) nbad += 1 << 16;
    }
  }
  if (nbad) atomicAdd(bad + (nbad >> 31 ? 2 : 0), nbad & 0xffff), atomicAdd(bad + 1, (nbad >> 16) & 0x7fff);
}

int main() {
  std::vector<float> h(4096);
  for (int i = 0; i < 4096; ++i) h[i] = float((i * 2654435761u) >> 8 & 0xffff) / 6553.6f - 5.f;
  float* d; unsigned* bad;
  hipMalloc(&d, 4096 * 4); hipMalloc(&bad, 16);
  hipMemcpy(d, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  for (int mode = 0; mode < 2; ++mode)
    for (int rep = 0; rep < 3; ++rep) {
      hipMemset(bad, 0, 16);
      probe<<<4096, 256>>>(d, bad, 2000, mode);
      unsigned r[4];
      hipMemcpy(r, bad, 16, hipMemcpyDeviceToHost);
      printf("mode %d rep %d: low-half mismatches %u, high-half mismatches %u (%s)\n", mode, rep, r[0], r[1], hipGetErrorString(hipGetLastError()));
    }
  return 0;
}
