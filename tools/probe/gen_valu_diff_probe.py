#!/usr/bin/env python3
"""Generates tools/probe/valu_diff_probe.hip: a DIFFERENTIAL probe of the vector-instruction forms the library uses (the modifier-bearing
shapes of build/isa/*.s: DPP controls, SDWA selects, op_sel on mixed-precision FMAs, packed 16-bit ops, permutes, conversions, ...).
Every form runs between s_nop fences on the same inputs twice -- in waves that have their SIMDs to themselves, and in waves that share
them with MFMA waves -- and the two runs' results must be bit-identical (no expected values needed: the instruction is its own
reference).  Written after the packed-f32 finding of round 6 (wekws_amd/csrc/pk_safe.hip.h) to look for relatives of that hazard."""
FORMS = [
    # DPP (the depthwise taps, the wave reductions)
    "v_fmac_f32_dpp %[d], %[a], %[b] row_shr:1 row_mask:0xf bank_mask:0xf",
    "v_fmac_f32_dpp %[d], %[a], %[b] row_shr:7 row_mask:0xf bank_mask:0xf",
    "v_fmac_f32_dpp %[d], %[a], %[b] row_shr:15 row_mask:0xf bank_mask:0xf",
    "v_fmac_f32_dpp %[d], %[a], %[b] row_shl:1 row_mask:0xf bank_mask:0xf",
    "v_fmac_f32_dpp %[d], %[a], %[b] row_shl:9 row_mask:0xf bank_mask:0xf",
    "v_max_u32_dpp %[d], %[a], %[b] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf",
    "v_max_u32_dpp %[d], %[a], %[b] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf",
    "v_max_u32_dpp %[d], %[a], %[b] row_half_mirror row_mask:0xf bank_mask:0xf",
    "v_max_u32_dpp %[d], %[a], %[b] row_mirror row_mask:0xf bank_mask:0xf",
    "v_max_u32_dpp %[d], %[a], %[b] row_bcast:15 row_mask:0xa bank_mask:0xf",
    "v_max_u32_dpp %[d], %[a], %[b] row_bcast:31 row_mask:0xc bank_mask:0xf",
    "v_add_f32_dpp %[d], %[a], %[b] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1",
    "v_add_f32_dpp %[d], %[a], %[b] row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1",
    "v_mov_b32_dpp %[d], %[a] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1",
    "v_mov_b32_dpp %[d], %[a] row_bcast:15 row_mask:0xa bank_mask:0xf",
    # conversions and the hi / lo split
    "v_cvt_pk_f16_f32 %[d], %[a], %[b]",
    "v_cvt_f32_f16_e32 %[d], %[a]",
    "v_cvt_f32_f16_sdwa %[d], %[a] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1",
    "v_cvt_f32_f16_sdwa %[d], -%[a] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1",
    "v_cvt_f32_f16_e64 %[d], -%[a]",
    "v_cvt_f16_f32_e32 %[d], %[a]",
    "v_cvt_u32_f32_e32 %[d], %[a]",
    "v_cvt_f32_i32_sdwa %[d], sext(%[a]) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1",
    "v_cvt_f32_i32_sdwa %[d], sext(%[a]) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0",
    "v_fma_mixlo_f16 %[d], %[a], %[b], 0",
    "v_fma_mix_f32 %[d], %[a], %[b], -%[c] op_sel_hi:[0,0,1]",
    "v_fma_mixlo_f16 %[d], %[a], %[b], -%[c] op_sel_hi:[0,0,1]",
    "v_fma_mixhi_f16 %[d], %[a], %[b], -%[c] op_sel_hi:[0,0,1]",
    "v_fma_mix_f32 %[d], %[a], %[b], %[c] op_sel:[0,0,1] op_sel_hi:[0,0,1]",
    # SDWA integer forms, permutes, bit fields
    "v_add_u32_sdwa %[d], %[a], %[b] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1",
    "v_and_b32_sdwa %[d], %[a], %[b] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD",
    "v_lshlrev_b32_sdwa %[d], %[a], %[b] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0",
    "v_mul_u32_u24_sdwa %[d], %[a], %[b] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD",
    "v_mul_lo_u16_sdwa %[d], %[a], %[b] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD",
    "v_sub_u16_sdwa %[d], %[a], %[b] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD",
    "v_perm_b32 %[d], %[a], %[b], %[c]",
    "v_alignbit_b32 %[d], %[a], %[b], 16",
    "v_bfe_u32 %[d], %[a], 23, 8",
    "v_max3_f32 %[d], %[a], |%[b]|, |%[c]|",
    "v_max3_u32 %[d], %[a], %[b], %[c]",
    # packed 16-bit integer ops
    "v_pk_mul_lo_u16 %[d], %[a], %[b] op_sel_hi:[1,0]",
    "v_pk_mad_u16 %[d], %[a], %[b], %[c] op_sel_hi:[1,0,1]",
    "v_pk_lshrrev_b16 %[d], 3, %[a] op_sel_hi:[0,1]",
    "v_pk_sub_i16 %[d], %[a], %[b]",
    "v_pk_add_u16 %[d], %[a], %[b] op_sel:[0,1] op_sel_hi:[1,0]",
    "v_pk_fma_f16 %[d], %[a], %[b], %[c] op_sel:[0,1,0]",
    "v_pk_mul_f16 %[d], %[a], %[b] op_sel:[0,1] op_sel_hi:[1,0]",
    "v_pk_max_f16 %[d], %[a], %[b] op_sel:[0,1]",
    # plain and transcendental
    "v_fma_f32 %[d], %[a], %[b], %[c]",
    "v_exp_f32_e32 %[d], %[a]",
    "v_rcp_f32_e32 %[d], %[a]",
    "v_log_f32_e64 %[d], |%[a]|",
    "v_ldexp_f32 %[d], %[a], 3",
    "v_med3_f32 %[d], %[a], %[b], %[c]",
    # 64-bit / packed f32 (known: the op_sel:[0,1] forms differ; listed as the positive control)
    "v_pk_fma_f32 %[D], %[A], %[B], %[C]",
    "v_pk_fma_f32 %[D], %[A], %[B], %[C] op_sel_hi:[1,0,1]",
    "v_pk_fma_f32 %[D], %[B], %[A], %[C] op_sel:[1,0,0]",
    "v_pk_mul_f32 %[D], %[A], %[B] op_sel_hi:[0,1]",
    "v_pk_add_f32 %[D], %[A], %[B] op_sel:[1,0] op_sel_hi:[0,1] neg_hi:[1,0]",
    "v_pk_mov_b32 %[D], %[A], %[B] op_sel:[1,0]",
    "v_lshl_add_u64 %[D], %[A], 2, %[B]",
    "v_mad_u64_u32 %[D], vcc, %[a], %[b], %[C]",
    "v_pk_fma_f32 %[D], %[A], %[B], %[C] op_sel:[0,1,0]",   # <- positive control: must differ
]
HDR = r'''// GENERATED by tools/probe/gen_valu_diff_probe.py -- do not edit.  See that file.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/valu_diff_probe.hip -o /tmp/valu_diff && /tmp/valu_diff
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define NFORM %d
static const char* kNames[NFORM] = {
%s};

__global__ __launch_bounds__(256, 4) void probe(const unsigned* in, unsigned* out, int iters, int mfma_waves) {
  const int tid = threadIdx.x, wave = tid >> 6;
  if (wave < 2) {                                                // waves 0, 1: MFMA chains (mode 1) or nothing (mode 0)
    if (mfma_waves) {
      f16x8 a, b;
      for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(tid * 1e-3f + e); b[e] = (_Float16)(tid * 2e-3f - e); }
      f32x4 c[7] = {};
      for (int it = 0; it < iters * 40; ++it) {
#pragma unroll
        for (int i = 0; i < 7; ++i) {
          c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[i], 0, 0, 0);
          c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, c[i], 0, 0, 0);
          c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, a, c[i], 0, 0, 0);
        }
      }
      float s = 0.f;
      for (int i = 0; i < 7; ++i) s += c[i][0];
      if (s == 12345.f) out[0] = 1;
    }
    return;
  }
  const int gid = blockIdx.x * 128 + (tid - 128);                // the workgroup's waves 2, 3 run the forms in both modes
  unsigned sum[NFORM];
  for (int f = 0; f < NFORM; ++f) sum[f] = 0u;
  for (int it = 0; it < iters; ++it) {
    const unsigned* p = in + ((gid * 7 + it * 131) & 4095);
    unsigned a = p[0], b = p[1], c = p[2];
    unsigned long long A = (unsigned long long)p[3] << 32 | p[4], B = (unsigned long long)p[5] << 32 | p[6], C = (unsigned long long)p[7] << 32 | p[8];
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(A), "+v"(B), "+v"(C));
%s  }
  for (int f = 0; f < NFORM; ++f) out[16 + f * (gridDim.x * 128) + gid] = sum[f];
}

int main() {
  const int nwg = 2048, n = nwg * 128;
  std::vector<unsigned> h(4096 + 16);
  unsigned x = 12345u;
  for (auto& v : h) {                                            // floats of moderate size in most words (sign, exponent 120..134, random mantissa)
    x = x * 1664525u + 1013904223u;
    v = (x & 0x807fffffu) | ((120u + (x >> 9) %% 15u) << 23);
  }
  unsigned *din, *dout;
  (void)hipMalloc(&din, h.size() * 4);
  (void)hipMalloc(&dout, (16 + size_t(NFORM) * n) * 4);
  (void)hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  std::vector<unsigned> r0(16 + size_t(NFORM) * n), r1(r0.size());
  int total = 0;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipMemset(dout, 0, r0.size() * 4);
    probe<<<nwg, 256>>>(din, dout, 200, 0);
    (void)hipMemcpy(r0.data(), dout, r0.size() * 4, hipMemcpyDeviceToHost);
    (void)hipMemset(dout, 0, r0.size() * 4);
    probe<<<nwg, 256>>>(din, dout, 200, 1);
    (void)hipMemcpy(r1.data(), dout, r0.size() * 4, hipMemcpyDeviceToHost);
    if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); return 1; }
    for (int f = 0; f < NFORM; ++f) {
      int nd = 0, hi_lanes = 0;
      for (int g = 0; g < n; ++g)
        if (r0[16 + size_t(f) * n + g] != r1[16 + size_t(f) * n + g]) { ++nd; if ((g & 63) >= 48) ++hi_lanes; }
      if (nd) { printf("rep %%d DIFFERS beside MFMA waves: %%7d of %%d threads (lanes 48..63: %%d)   %%s\n", rep, nd, n, hi_lanes, kNames[f]); ++total; }
    }
  }
  printf("%%d forms x 3 repetitions: %%d (form, repetition) pairs differ\n", NFORM, total);
  return 0;
}
'''
body = ""
for i, t in enumerate(FORMS):
    wide = "%[D]" in t
    body += "    {\n"
    body += "      unsigned d = c; unsigned long long D = C;\n"
    body += f'      asm volatile("s_nop 7\\n\\t{t}\\n\\ts_nop 7" : [d] "+v"(d), [D] "+v"(D) : [a] "v"(a), [b] "v"(b), [c] "v"(c), [A] "v"(A), [B] "v"(B), [C] "v"(C) : "vcc");\n'
    body += f"      sum[{i}] = sum[{i}] * 31u + " + ("unsigned(D) + 7u * unsigned(D >> 32)" if wide else "d") + ";\n"
    body += "    }\n"
names = "".join('    "' + t.replace('"', '\\"') + '",\n' for t in FORMS)
open(__file__.replace("gen_valu_diff_probe.py", "valu_diff_probe.hip"), "w").write(HDR % (len(FORMS), names, body))
print(len(FORMS), "forms")
