// Probe (round 3): what does the WRITE_SIZE counter report for store patterns of KNOWN size?
//   full      110,100,480 bytes written as contiguous, 16-byte aligned float4 stores (whole 128-byte lines)
//   rows_x4x3 the streaming cache's shape: 1024 x 256 rows of 105 floats (420 bytes, only dword aligned), every row written
//             as 15 runs of 7 floats with dwordx4 + dwordx3 stores (what ds256_g16 hands over)
//   rows_dw   the same rows written with 4-byte stores, 16 consecutive floats per 16 lanes (what ds256_w16 hands over)
// All three write exactly the same number of bytes.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/write_calib.hip -o build/probe_bin/write_calib
//   rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/prof_wcal -o pmc -- build/probe_bin/write_calib
#include <hip/hip_runtime.h>
#include <cstdio>
namespace wekws {                                            // (tools/prof_summary.py lists kernels of this namespace)
constexpr int ROWS = 1024 * 256, P = 105;
__global__ void full(float4* d, size_t n4) {
  for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n4; i += size_t(gridDim.x) * blockDim.x) d[i] = float4{1.f, 2.f, 3.f, 4.f};
}
struct __attribute__((packed, aligned(4))) V4 { float v[4]; };
struct __attribute__((packed, aligned(4))) V3 { float v[3]; };
__global__ void rows_x4x3(float* d) {   // thread = (row, run of 7)
  const size_t e = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
  if (e >= size_t(ROWS) * 15) return;
  const size_t row = e / 15; const int run = int(e % 15);
  float* p = d + row * P + run * 7;
  *reinterpret_cast<V4*>(p) = V4{{1.f, 2.f, 3.f, 4.f}};
  *reinterpret_cast<V3*>(p + 4) = V3{{5.f, 6.f, 7.f}};
}
__global__ void rows_dw(float* d) {     // 16 lanes = 16 consecutive floats of a row; 4 rows per 64-lane instruction
  const size_t e = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
  const size_t row = e / 112; const int c = int(e % 112);
  if (row < size_t(ROWS) && c < P) d[row * P + c] = float(c);
}
}  // namespace wekws
using namespace wekws;
int main() {
  const size_t bytes = size_t(ROWS) * P * 4;
  float* d; hipMalloc(&d, bytes + 4096);
  for (int rep = 0; rep < 5; ++rep) {
    hipLaunchKernelGGL(full, dim3(4096), dim3(256), 0, 0, reinterpret_cast<float4*>(d), bytes / 16);
    hipLaunchKernelGGL(rows_x4x3, dim3((size_t(ROWS) * 15 + 255) / 256), dim3(256), 0, 0, d);
    hipLaunchKernelGGL(rows_dw, dim3((size_t(ROWS) * 112 + 255) / 256), dim3(256), 0, 0, d);
  }
  hipDeviceSynchronize();
  printf("each kernel writes %zu bytes = %.1f KiB\n", bytes, bytes / 1024.0);
  return 0;
}
