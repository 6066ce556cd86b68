// Packed-f32 (VOP3P) instructions on gfx950: one operand-select pattern that must not be used, and helpers that do not use it.
//
// Found in round 6 (the cause of the "rare wrong posteriors" of ds64_g4 that rounds 4-6 had only been able to move around by
// re-arranging code -- profiles/r06_experiments.txt).  Measured on MI355X with tools/probe/pk_opsel_probe4.hip / probe5 (every
// op_sel / op_sel_hi combination of v_pk_fma_f32, v_pk_mul_f32, v_pk_add_f32, 4096 workgroups x 100 iterations):
//
//   a packed-f32 instruction whose LOW result is formed from  the LOW half of its first vector-register source and the HIGH half of
//   its second vector-register source  -- op_sel:[0,1,..] on three registers; op_sel:[0,.,1] when src1 is a constant or a scalar
//   register (those do not count: probe5, probe7); the third register's select plays no role, nor does op_sel_hi -- returns its low
//   result in LANES 48..63 as if that second source's high half were zero (the product missing from an FMA) whenever other waves of
//   the same SIMD are issuing MFMAs at that moment.  Never without MFMA neighbours, never in lanes 0..47, never in the high result,
//   never for the mirror pattern (first source high, second low), never with a single register source.  2 x 10^5 wrong values per
//   5 x 10^7 in the probe; in ds64_g4 (four workgroups per CU, the head's FMA chain of one workgroup beside the matrix phase of another)
//   ~2 % of the utterances of a large batch, all in the one output whose chain the compiler had lowered to
//   v_pk_fma_f32 d, w, h, d op_sel:[0,1,0]; in the fbank kernel beside an MDTC forward on another stream 1 .. 6 % of the frames.
//
// The compiler (ROCm 7.2) emits that form freely -- its SLP vectoriser pairs the two outputs' chains (p0, p1) += (w0, w1) * h[1] and
// takes h[1] as the high half of the register pair (h[0], h[1]) -- and a kernel without MFMAs of its own meets it as soon as a
// tenant on another stream puts MFMA waves on its SIMD.  So: (1) the code below spells the commutative operands the safe way round
// (the selected-high operand first) in inline assembly, where no canonicalisation can turn them back; (2) tests/test_isa_hazard.py
// disassembles the built library and fails on any packed-f32 instruction with that select pattern.
#pragma once
#include <hip/hip_runtime.h>

namespace wekws {

typedef float pk_f32x2 __attribute__((ext_vector_type(2)));
typedef float pk_f32x4 __attribute__((ext_vector_type(4)));

// (p.x, p.y) += (wx.x, wx.y) h[0] + (wy.x, wy.y) h[1] + (wz.x, wz.y) h[2] + (ww.x, ww.y) h[3]  -- two outputs' partial head sums over
// a lane's four channels, each lane half its own fmaf chain in this order (the scalar source it replaces: p = fmaf(w.c, h[c], p)).
// h[1] / h[3] are the HIGH halves of the pairs (h[0], h[1]) / (h[2], h[3]): those steps take the pair as src0 (high half selected for
// both results) and the weights as src1.  A packed result must not be read by the very next vector instruction (gfx940+ forwarding
// hazard; the compiler cannot look into the block): s_nop 0 between the dependent steps and at both ends.
__device__ __forceinline__ void head_fma4(pk_f32x2& p, pk_f32x2 wx, pk_f32x2 wy, pk_f32x2 wz, pk_f32x2 ww, const pk_f32x4 h) {
  const pk_f32x2 h01 = __builtin_shufflevector(h, h, 0, 1), h23 = __builtin_shufflevector(h, h, 2, 3);
  asm volatile(
      "s_nop 0\n\t"
      "v_pk_fma_f32 %0, %1, %5, %0 op_sel_hi:[1,0,1]\n\ts_nop 0\n\t"        // (wx.lo h.lo, wx.hi h.lo)
      "v_pk_fma_f32 %0, %5, %2, %0 op_sel:[1,0,0]\n\ts_nop 0\n\t"            // (h.hi wy.lo, h.hi wy.hi)
      "v_pk_fma_f32 %0, %3, %6, %0 op_sel_hi:[1,0,1]\n\ts_nop 0\n\t"
      "v_pk_fma_f32 %0, %6, %4, %0 op_sel:[1,0,0]\n\ts_nop 0"
      : "+v"(p)
      : "v"(wx), "v"(wy), "v"(wz), "v"(ww), "v"(h01), "v"(h23));
}
// the weight pairs of the two outputs, made once per head
struct HeadPairs {
  pk_f32x2 x, y, z, w;
  __device__ __forceinline__ HeadPairs(const float4 w0, const float4 w1) : x{w0.x, w1.x}, y{w0.y, w1.y}, z{w0.z, w1.z}, w{w0.w, w1.w} {}
};

}  // namespace wekws
