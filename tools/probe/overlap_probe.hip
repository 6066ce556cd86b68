// Probe (round 3): how much vector / LDS work runs beside a saturated fp16 MFMA stream on one SIMD, for the two MFMA
// shapes (16x16x32: 16-cycle issue, 32x32x16: 32-cycle issue), in the two arrangements the headline kernel could use:
//   roles      -- 16-wave workgroup, waves 0..7 only multiply, waves 8..15 only run the producer-like vector/LDS mix
//   interleave -- all 16 waves run  [1 MFMA ; N vector instructions]  groups
// Prints cycles per MFMA (per SIMD) and vector instructions per 1000 SIMD cycles for every combination.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/overlap_probe.hip -o build/probe_bin/overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define VFMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c1), "v"(c2))

// ---- roles.  mode bit0: waves 0..7 multiply; bit1: waves 8..15 run the vector mix; bit2: vector waves at s_setprio 3;
//      bit3: MFMA waves at s_setprio 3.   VK = 0: independent v_fma chain, 1: producer-like mix (LDS reads, fma, split, LDS 2-byte writes)
template <int MK, int VK>
__global__ __launch_bounds__(1024) void roles_k(int mode, int iters, long long* cyc, float* sink) {
  __shared__ float lds[16 * 1024];
  const int wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16 * 1024; i += 1024) lds[i] = i * 1e-4f;
  long long t0 = 0, t1 = 0;
  float s = threadIdx.x * 1e-3f;
  __syncthreads();
  if (wave < 8) {
    if (mode & 1) {
      if (mode & 8) __builtin_amdgcn_s_setprio(3);
      f16x8 a, b;
      for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(s + e); b[e] = (_Float16)(s - e); }
      if constexpr (MK == 0) {
        f32x4 c[8];
        for (int i = 0; i < 8; ++i) c[i] = f32x4{0, 0, 0, 0};
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
          for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[i], 0, 0, 0);
        }
        t1 = __builtin_readcyclecounter();
        for (int i = 0; i < 8; ++i) s += c[i][0];
      } else {
        f32x16 c[4];
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) c[i][e] = 0.f;
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
          for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[i], 0, 0, 0);
        }
        t1 = __builtin_readcyclecounter();
        for (int i = 0; i < 4; ++i) s += c[i][0];
      }
    }
  } else {
    if (mode & 2) {
      if (mode & 4) __builtin_amdgcn_s_setprio(3);
      const float c1 = 1.0001f, c2 = 0.5f;
      float v[8];
      for (int i = 0; i < 8; ++i) v[i] = s + i;
      const int lane = threadIdx.x & 63;
      float* rp = lds + (wave - 8) * 2048 + lane;                       // conflict-free 4-byte reads
      _Float16* wp = reinterpret_cast<_Float16*>(lds + (wave - 8) * 2048) + lane * 8;   // 2-byte stores, 16 B apart (plane order)
      t0 = __builtin_readcyclecounter();
      for (int it = 0; it < iters; ++it) {
        if constexpr (VK == 0) {
#pragma unroll
          for (int i = 0; i < 8; ++i) VFMA(v[i]);
        } else {
          // one "output": 2 LDS reads, 8 fma, max, 3-instruction split, 2 two-byte LDS stores  (= 16 issue slots)
          const float x0 = rp[(it & 7) * 64], x1 = rp[((it + 3) & 7) * 64 + 512];
          float o = x0;
#pragma unroll
          for (int i = 0; i < 8; ++i) o = fmaf(v[i], i & 1 ? x1 : x0, o);
          o = fmaxf(o, 0.f);
          unsigned hr; float d;
          asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hr) : "v"(o), "v"(c1));
          asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(d) : "v"(o), "v"(c1), "v"(hr));
          const _Float16 l = static_cast<_Float16>(d);
          wp[(it & 7) * 1024] = __builtin_bit_cast(_Float16, static_cast<unsigned short>(hr & 0xffffu));
          wp[(it & 7) * 1024 + 512 * 1] = l;
          v[it & 7] += d * 1e-9f;
        }
      }
      t1 = __builtin_readcyclecounter();
      for (int i = 0; i < 8; ++i) s += v[i];
    }
  }
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
  sink[blockIdx.x * 1024 + threadIdx.x] = s + lds[threadIdx.x];
}

// ---- interleave: every wave runs groups of [1 MFMA ; N v_fma] (independent chains), WAVES waves per workgroup
template <int MK, int N>
__global__ __launch_bounds__(1024) void inter_k(int iters, long long* cyc, float* sink) {
  const int wave = threadIdx.x >> 6;
  float s = threadIdx.x * 1e-3f;
  const float c1 = 1.0001f, c2 = 0.5f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = s + i;
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(s + e); b[e] = (_Float16)(s - e); }
  f32x4 c4[8];
  f32x16 c16[4];
  for (int i = 0; i < 8; ++i) c4[i] = f32x4{0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) c16[i][e] = 0.f;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      if constexpr (MK == 0) c4[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c4[g], 0, 0, 0);
      else c16[g & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c16[g & 3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < N; ++i) VFMA(v[(g * N + i) & 7]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  for (int i = 0; i < 8; ++i) s += v[i] + c4[i][0];
  for (int i = 0; i < 4; ++i) s += c16[i][0];
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
  sink[blockIdx.x * 1024 + threadIdx.x] = s;
}

static long long* dc;
static float* ds;
static std::vector<long long> fetch() {
  std::vector<long long> c(256 * 16);
  hipMemcpy(c.data(), dc, c.size() * 8, hipMemcpyDeviceToHost);
  return c;
}

template <int MK, int VK>
void run_roles(const char* name) {
  const int iters = 1000;
  for (int mode : {1, 2, 3, 7, 11}) {
    for (int rep = 0; rep < 2; ++rep) {
      hipMemset(dc, 0, 256 * 16 * 8);
      hipLaunchKernelGGL((roles_k<MK, VK>), dim3(256), dim3(1024), 0, 0, mode, iters, dc, ds);
      hipDeviceSynchronize();
    }
    auto c = fetch();
    double m = 0, v = 0;
    for (int b = 0; b < 256; ++b) { for (int w = 0; w < 8; ++w) m += c[b * 16 + w]; for (int w = 8; w < 16; ++w) v += c[b * 16 + w]; }
    m /= 2048; v /= 2048;
    const int nm = MK == 0 ? 8 : 4, nv = VK == 0 ? 8 : 16;   // per iteration and wave
    // two waves of each role per SIMD
    printf("roles %-28s mode=%2d  SIMD cycles per MFMA: %6.1f   vector issue slots per 1000 SIMD cycles: %6.1f (%.1f cyc/slot/wave)\n",
           name, mode, m ? m / (iters * nm) / 2 : 0., v ? 1000. * 2 * iters * nv / v : 0., v / (iters * nv));
  }
}
template <int MK, int N>
void run_inter(const char* name, int threads) {
  const int iters = 500;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((inter_k<MK, N>), dim3(256), dim3(threads), 0, 0, iters, dc, ds);
    hipDeviceSynchronize();
  }
  auto c = fetch();
  const int nw = threads / 64;
  double m = 0;
  for (int b = 0; b < 256; ++b) for (int w = 0; w < nw; ++w) m += c[b * 16 + w];
  m /= 256.0 * nw;
  const double per_group_simd = m / (iters * 8.0) / (nw / 4);
  printf("interleave %-14s %2d waves/SIMD  N=%d vector per MFMA: %6.1f SIMD cycles per group (MFMA alone %d)\n", name, nw / 4, N,
         per_group_simd, MK == 0 ? 16 : 32);
}
template <int MK>
void run_inter_all(const char* name) {
  for (int threads : {256, 512, 1024}) {
    run_inter<MK, 0>(name, threads); run_inter<MK, 1>(name, threads); run_inter<MK, 2>(name, threads);
    run_inter<MK, 3>(name, threads); run_inter<MK, 4>(name, threads); run_inter<MK, 5>(name, threads);
    run_inter<MK, 6>(name, threads); run_inter<MK, 8>(name, threads); run_inter<MK, 10>(name, threads);
  }
}

// ---- wall clock: pure MFMA, 4 waves per SIMD, operands from memory (random), TFLOP/s and the tick rate of s_memtime
template <int MK>
__global__ __launch_bounds__(1024) void wall_k(int iters, const _Float16* src, long long* cyc, float* sink) {
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = src[(threadIdx.x * 8 + e) & 8191]; b[e] = src[(threadIdx.x * 8 + e + 4096) & 8191]; }
  f32x4 c4[8];
  f32x16 c16[4];
  for (int i = 0; i < 8; ++i) c4[i] = f32x4{0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) c16[i][e] = 0.f;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      if constexpr (MK == 0) c4[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c4[g], 0, 0, 0);
      else c16[g & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c16[g & 3], 0, 0, 0);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += c4[i][0];
  for (int i = 0; i < 4; ++i) s += c16[i][0];
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
  sink[blockIdx.x * 1024 + threadIdx.x] = s;
}
template <int MK>
void run_wall(const char* name, const _Float16* dsrc) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((wall_k<MK>), dim3(256), dim3(1024), 0, 0, iters, dsrc, dc, ds);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((wall_k<MK>), dim3(256), dim3(1024), 0, 0, iters, dsrc, dc, ds);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  auto c = fetch();
  double m = 0;
  for (int b = 0; b < 256; ++b) for (int w = 0; w < 16; ++w) m += c[b * 16 + w];
  m /= 4096;
  const double flop = 256.0 * 16 * iters * 8 * (MK == 0 ? 16384.0 : 32768.0);
  printf("wall %-10s 4 waves/SIMD, random operands: %.3f ms  %.1f TFLOP/s   ticks per wave %.0f -> tick rate %.3f GHz, %.2f ticks per MFMA per SIMD\n",
         name, ms, flop / ms * 1e-9, m, m / ms * 1e-6, m / (iters * 8.0) / 4);
}

int main() {
  {
    std::vector<_Float16> h(8192);
    unsigned x = 12345;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (_Float16)(((x >> 8) & 0xffff) / 32768.0f - 1.0f); }
    _Float16* dsrc;
    hipMalloc(&dsrc, 8192 * 2);
    hipMemcpy(dsrc, h.data(), 8192 * 2, hipMemcpyHostToDevice);
    hipMalloc(&dc, 256 * 16 * 8);
    hipMalloc(&ds, 256 * 1024 * 4);
    run_wall<0>("16x16x32", dsrc);
    run_wall<1>("32x32x16", dsrc);
    run_wall<0>("16x16x32", dsrc);
    run_wall<1>("32x32x16", dsrc);
  }
  hipMalloc(&dc, 256 * 16 * 8);
  hipMalloc(&ds, 256 * 1024 * 4);
  run_roles<0, 0>("16x16x32 + v_fma chain");
  run_roles<1, 0>("32x32x16 + v_fma chain");
  run_roles<0, 1>("16x16x32 + producer mix");
  run_roles<1, 1>("32x32x16 + producer mix");
  run_inter_all<0>("16x16x32");
  run_inter_all<1>("32x32x16");
  return 0;
}
