// Are naturally aligned 16-byte stores / loads torn?  Producer workgroups keep rewriting 16-byte items {k, k, k, k} (k counts up);
// consumer workgroups on the same XCD and on other XCDs keep reading them (sc1, L2-served) and count items whose four dwords
// differ.  Both store policies the GRU wavefront uses: default (the consumer shares the L2) and write-through (sc1).
//   hipcc --offload-arch=gfx950 -O3 tools/probe/tear16.hip -o /tmp/tear16 && /tmp/tear16
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-result"
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void tear(unsigned* buf, unsigned long long* out, int iters, int sc1_store) {
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 1 << 20, 0x00020000);
  const int role = blockIdx.x >= 64;                       // blocks 0..63 produce, 64..255 consume
  const int off = ((blockIdx.x & 63) * 256 + threadIdx.x) * 16;   // consumers b, b+64, b+128 read producer (b & 63)'s items
  if (!role) {
    for (int k = 1; k <= iters; ++k) {
      const u32x4 v = {unsigned(k), unsigned(k), unsigned(k), unsigned(k)};
      if (sc1_store) __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 16);
      else __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 0);
    }
  } else {
    unsigned long long torn = 0, seen = 0;
    unsigned last = 0;
    for (int k = 0; k < iters; ++k) {
      u32x4 v;                                              // (volatile: a plain builtin load is hoisted out of the loop)
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(off), "s"(rs) : "memory");
      torn += (v[0] != v[1]) | (v[1] != v[2]) | (v[2] != v[3]);
      seen += v[0] != last;
      last = v[0];
    }
    atomicAdd(out, torn);
    atomicAdd(out + 1, seen);
  }
}
int main() {
  unsigned* buf; unsigned long long* out;
  hipMalloc(&buf, 1 << 20); hipMalloc(&out, 16);
  for (int sc1 = 0; sc1 < 2; ++sc1) {
    hipMemset(buf, 0, 1 << 20); hipMemset(out, 0, 16);
    hipLaunchKernelGGL(tear, dim3(256), dim3(256), 0, 0, buf, out, 400000, sc1);
    hipDeviceSynchronize();
    unsigned long long h[2]; hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    printf("store %s: %llu torn 16-byte items of %llu reads (%llu value changes observed)\n", sc1 ? "sc1" : "default", h[0],
           192ull * 256 * 400000, h[1]);
  }
  return 0;
}
