// WHICH KERNEL RUNS A CALL -- one pure function (round 6; VERDICT r5 "make routing testable and smaller").
//
// Until round 5 the choice lived in two places: a 70-line conditional chain in wekws_hip_forward and the "return -4" conditions
// inside every launcher, tried one after the other.  All three defects the round-5 fuzz found were in that layer (a fifth
// DS-TCN h256 block whose hand-over the 16-wave kernel did not cover, hidden_dim 32 taken for a built width, ...), in code that
// could only be exercised with a GPU.  Now:
//   * conv_route_flags()  what a conv model can run on, from its (built-shape) descriptor alone -- wekws_hip_create stores it;
//   * select_conv_route() (flags, options, call) -> {family, tile count, split, context variant, persistent grid, LDS bytes},
//     with the invariants of the choice checked (built widths, LDS within the CU's 160 KiB, the hand-over covering the
//     longest padding, alignment preconditions);
//   * wekws_hip_forward switches on the family; a launcher that refuses what the route chose is an internal error, not a
//     fall-through.
// Plain C++ (no HIP): tests/test_route.py sweeps the fuzz generator's configurations through it on the CPU via the hooks
// library (wekws_hip_debug_conv_route); the GPU suite (every family parity-green against the oracle) runs through the same
// switch.
#pragma once
#include <stdint.h>

#include "../../include/wekws_hip.h"

namespace wekws {

enum RouteFamily : int {
  ROUTE_NONE = 0,
  ROUTE_DS256_STREAM,      // ds256_stream.hip.h   chunks <= 16 frames, the stream's cache in LDS
  ROUTE_DS256_G32,         // ds256_g32.hip.h      exact f32, tile in registers, persistent
  ROUTE_DS256_MM,          // ds256_mm.hip.h       CTC-sized heads: classifier on the matrix cores
  ROUTE_DS256_G16,         // ds256_g16.hip.h      tile in registers (16 waves); ctx: with an incoming cache
  ROUTE_DS256_W16,         // ds256_w16.hip.h      LDS tile, 16 waves
  ROUTE_DS64_G4,           // ds64_g4.hip.h        one utterance per 4-wave workgroup; ctx
  ROUTE_MDTC64_STREAM,     // mdtc64_stream.hip.h  chunks <= 16 frames, two streams per workgroup
  ROUTE_MDTC64_G4,         // mdtc64_g4.hip.h <64> one utterance per 4-wave workgroup; ctx
  ROUTE_MDTC64_W16,        // mdtc64_w16.hip.h     LDS tile, 16 waves
  ROUTE_MDTC32_G4,         // mdtc64_g4.hip.h <32> one utterance per 2-wave workgroup; ctx
  ROUTE_DENSE_F16,         // dense_stack_f16.hip.h plain TCN on the matrix cores
  ROUTE_CONV_F16,          // conv_stack_f16.hip.h  LDS tile, 8 waves, any built width, split fp16
  ROUTE_CONV_F32,          // conv_stack.hip.h      LDS tile, 8 waves, any built width, exact f32
  ROUTE_FAMILIES
};

inline const char* route_family_name(int f) {
  static const char* const n[] = {"none", "ds256_stream", "ds256_g32", "ds256_mm", "ds256_g16", "ds256_w16", "ds64_g4", "mdtc64_stream",
                                  "mdtc64_g4", "mdtc64_w16", "mdtc32_g4", "dense_stack_f16", "conv_stack_f16", "conv_stack"};
  return f >= 0 && f < ROUTE_FAMILIES ? n[f] : "?";
}

// What the model's SHAPE admits (computed once, wekws_hip_create) ...
struct RouteFlags {
  int32_t cache_len, max_pad, kpre16;
  int32_t dils_1248;              // every block's dilation is 1, 2, 4 or 8 and its padding (kernel_size - 1) x dilation
  int32_t ds_stream_eligible;     // DS-TCN, C = 256, kernel size 8, dils_1248
  int32_t mdtc16_eligible;        // MDTC, C = 64, kernel size 5
  int32_t mdtc_stream_eligible;   // ... dils_1248, features <= 128 dims, both streams' caches fit the LDS
  int32_t mm_eligible;            // DS-TCN h256 with a per-frame linear head and paddings <= 56
  int32_t dense_ok;               // plain TCN whose paddings fit the dense-stack kernel's halo
};
// ... what the options say (wekws_hip_set_option; defaults = the product's choice) ...
struct RouteOptions {
  int32_t w16_ok = 1, g16_ok = 1, g16_ctx = 1, g16_one_pass = 0, stream_ok = 1, mdtc16_ok = 1, mm_ok = 0, f32 = 0 /* precision F32 or the
            weights outside the split-fp16 envelope */, split = 1 /* F16X3: three products; F16: one */;
};
// ... and the call (one tile of it)
struct RouteCall {
  int32_t B, T, ntiles;           // T: frames of THIS tile (<= WEKWS_HIP_TILE_FRAMES)
  int32_t has_in, has_out;        // caches
  int32_t x16;                    // features: 16-byte aligned rows in whole 4-float units (x % 16 == 0, row stride % 4 == 0)
  int32_t cache16;                // both cache pointers 16-byte aligned
  int32_t cus;
};
struct Route {
  int32_t family, nt, split, ctx, fast, grid, threads, lds_bytes, utts_per_wg;
  const char* why_not;            // set when family == ROUTE_NONE: the invariant that failed
};

inline int route_blocks(const wekws_hip_desc& d) {
  return d.backbone == WEKWS_HIP_BACKBONE_MDTC ? 1 + d.num_stack * d.stack_size : d.num_layers;
}
inline int route_dilation(const wekws_hip_desc& d, int i) {
  if (d.backbone == WEKWS_HIP_BACKBONE_MDTC) return i == 0 ? 1 : 1 << ((i - 1) % d.stack_size);   // mdtc.py:151-156, :229-237
  return 1 << i;                                                                                   // tcn.py:131-137
}
inline int route_round_up(int v, int m) { return (v + m - 1) / m * m; }

inline RouteFlags conv_route_flags(const wekws_hip_desc& d, int mdtc_stream_lds_bytes) {
  RouteFlags f{};
  const int C = d.hdim, ks = d.kernel_size, nb = route_blocks(d);
  f.kpre16 = route_round_up(d.idim, 32);
  f.dils_1248 = 1;
  for (int i = 0; i < nb; ++i) {
    const int dil = route_dilation(d, i), pad = (ks - 1) * dil;
    f.cache_len += pad;
    f.max_pad = pad > f.max_pad ? pad : f.max_pad;
    if (!(dil == 1 || dil == 2 || dil == 4 || dil == 8)) f.dils_1248 = 0;
  }
  f.dense_ok = d.backbone == WEKWS_HIP_BACKBONE_TCN && f.max_pad <= 56 && C <= 128;
  f.mdtc16_eligible = d.backbone == WEKWS_HIP_BACKBONE_MDTC && C == 64 && ks == 5;
  f.ds_stream_eligible = d.backbone == WEKWS_HIP_BACKBONE_DS_TCN && C == 256 && ks == 8 && f.dils_1248;
  f.mdtc_stream_eligible = f.mdtc16_eligible && f.kpre16 <= 128 && (64 * f.cache_len) % 4 == 0 && mdtc_stream_lds_bytes <= 158 * 1024 &&
                           f.dils_1248;
  f.mm_eligible = d.backbone == WEKWS_HIP_BACKBONE_DS_TCN && C == 256 && ks == 8 && f.max_pad <= 56 && d.head == WEKWS_HIP_HEAD_LINEAR;
  return f;
}

// The shape a conv model RUNS as (wekws_hip_create): as it is, zero-padded to the next built width / kernel size (exact: see
// pad_conv_shape in wekws_hip.hip), or on the any-shape path of generic.hip.h.
enum : int { SHAPE_AS_IS = 0, SHAPE_PADDED = 1, SHAPE_GENERIC = 2 };
struct ShapePlan {
  int32_t kind, C, ks;            // SHAPE_PADDED: the built width / kernel size it runs as
  const char* why;                // SHAPE_GENERIC: the limit it exceeds
};
inline ShapePlan conv_shape_plan(const wekws_hip_desc& d, int max_blocks) {
  ShapePlan p{SHAPE_AS_IS, d.hdim, d.kernel_size, nullptr};
  const int C = d.hdim, ks = d.kernel_size;
  const bool mdtc = d.backbone == WEKWS_HIP_BACKBONE_MDTC;
  const int ks_built = mdtc ? 5 : 8;                          // the kernel sizes of the reference recipes ({ds_tcn,tcn}.yaml: 8; mdtc*.yaml: 5)
  auto generic = [&](const char* why) { p.kind = SHAPE_GENERIC; p.why = why; return p; };
  // built widths: 64 / 128 / 256, and 32 for MDTC (mdtc_small.yaml; DS-TCN / TCN with 32 channels run as 64: round-5 defect 2)
  const bool odd_c = C != 64 && C != 128 && C != 256 && !(C == 32 && mdtc);
  if (odd_c || (ks >= 1 && ks < ks_built)) {
    if (C > 256) return generic("wider than any built kernel (256 channels)");
    if (ks > ks_built) return generic("kernel size above the built one");
    const int Cp = !odd_c ? C : (C < 32 && mdtc) ? 32 : C < 64 ? 64 : C < 128 ? 128 : 256;
    if (mdtc && Cp > 128) return generic("MDTC wider than 128 channels does not fit the LDS tile");
    if (route_blocks(d) > max_blocks) return generic("more residual blocks than the cache maps hold");
    if (odd_c && d.head == WEKWS_HIP_HEAD_IDENTITY) return generic("identity head on a padded width (y is the tile itself)");
    p.kind = SHAPE_PADDED; p.C = Cp; p.ks = ks_built;
    return p;
  }
  if (mdtc && C == 256) return generic("MDTC with 256 channels does not fit the LDS tile");
  if (ks != ks_built) return generic("kernel size above the built one");
  if (d.precision != WEKWS_HIP_PRECISION_F32 && route_blocks(d) > max_blocks) return generic("more residual blocks than the split-fp16 kernels track maxima for");
  return p;
}

// frame tiles of 16 columns a call of T frames takes: 1, 2, 4 or 7
inline int route_nt(int T) {
  const int nt16 = (T + 15) / 16;
  return nt16 <= 1 ? 1 : nt16 <= 2 ? 2 : nt16 <= 4 ? 4 : 7;
}

// d: the descriptor of the shape the kernels RUN (after zero-padding to a built width / kernel size).
// ds_stream_lds / mdtc_stream_lds: LDS bytes of the two streaming-step kernels for this model's cache (their own headers'
// formulas, handed in so that this file stays plain C++).
inline Route select_conv_route(const wekws_hip_desc& d, const RouteFlags& f, const RouteOptions& o, const RouteCall& c, int ds_stream_lds,
                               int mdtc_stream_lds) {
  Route r{};
  const int C = d.hdim, ks = d.kernel_size, K = d.odim;
  const int nt = route_nt(c.T);
  const bool f16 = !o.f32;
  const bool has_in = c.has_in != 0;
  const bool linear2 = d.head == WEKWS_HIP_HEAD_LINEAR && K <= 2;
  const bool x_items = d.idim % 8 == 0 && c.x16;                 // whole aligned 8-float feature items
  const bool pooled = d.head == WEKWS_HIP_HEAD_GLOBAL || d.head == WEKWS_HIP_HEAD_LAST;
  auto fail = [&](const char* why) { r = Route{}; r.why_not = why; return r; };
  auto done = [&](int family, int nt_, bool ctx, bool fast, int grid, int threads, int lds, int upw) {
    r.family = family; r.nt = nt_; r.split = o.split; r.ctx = ctx; r.fast = fast; r.grid = grid; r.threads = threads; r.lds_bytes = lds;
    r.utts_per_wg = upw;
    return r;
  };
  if (c.B <= 0 || c.T <= 0 || c.T > WEKWS_HIP_TILE_FRAMES) return fail("tile of 1 .. 112 frames");
  if (!(C == 32 || C == 64 || C == 128 || C == 256)) return fail("hidden width is not a built one (32 / 64 / 128 / 256): wekws_hip_create pads");
  const int ks_built = d.backbone == WEKWS_HIP_BACKBONE_MDTC ? 5 : 8;
  if (ks != ks_built) return fail("kernel size is not the built one: wekws_hip_create pads or takes the any-shape path");

  // LDS of the tile kernels (conv_stack.hip.h Geom, ds256_w16.hip.h W16Geom, conv_stack_f16.hip.h)
  const int SS = (nt % 2) ? 16 * nt : 16 * nt + 16;
  const int U = C >= 128 ? 1 : 128 / C;
  auto conv_f32_lds = [&]() {
    const int KC = C >= 64 ? 32 : 16;
    const int R = d.backbone == WEKWS_HIP_BACKBONE_MDTC ? (C > 2 * KC ? C : 2 * KC) : 2 * KC;
    return (U * C * SS + U * R * SS) * 4;
  };
  const int tt = 16 * nt;
  auto w16_lds = [](int ntk) { return 1280 * 16 * ntk + 4096; };   // W16Geom<NT>::LDS_BYTES: 4 operand planes of 64 TT bytes + 256 rows of TT + 4 floats

  switch (d.backbone) {
    case WEKWS_HIP_BACKBONE_DS_TCN: {
      const bool strm = f16 && f.ds_stream_eligible && o.w16_ok && !o.mm_ok && o.stream_ok && c.ntiles == 1 && c.T <= 16 &&
                        (c.has_in || c.has_out) && c.cache16 && ds_stream_lds <= 160 * 1024;
      if (strm) {
        if (256 * f.cache_len > 7 * 4 * 1024) return fail("ds256_stream: a stream's cache is more than seven 16-byte items per thread");
        return done(ROUTE_DS256_STREAM, 1, true, false, c.B, 1024, ds_stream_lds, 1);
      }
      const bool reg_ok = C == 256 && o.w16_ok && o.g16_ok && f.ds_stream_eligible;   // the register-resident kernels' model side
      const bool fast = linear2 && f.kpre16 <= 64 && x_items;
      if (!f16) {
        if (reg_ok && !has_in && fast) return done(ROUTE_DS256_G32, nt, false, true, c.B < c.cus || o.g16_one_pass ? c.B : c.cus, 1024, w16_lds(nt), 1);
        if (conv_f32_lds() > 160 * 1024) return fail("conv_stack: tile beyond the LDS");
        return done(ROUTE_CONV_F32, nt, has_in, false, (c.B + U - 1) / U, 512, conv_f32_lds(), U);
      }
      if (o.mm_ok) {
        if (!f.mm_eligible) return fail("ds256_mm on a model it is not built for");
        return done(ROUTE_DS256_MM, nt, has_in, false, c.B, 1024, 0, 1);
      }
      if (reg_ok && (!has_in || (o.g16_ctx && nt >= 2))) {
        const int ntk = has_in && nt < 4 ? 4 : nt;                   // the context tile is one 16-lane row: >= 4 tiles
        if (!has_in) return done(ROUTE_DS256_G16, ntk, false, fast, fast && !o.g16_one_pass && c.B > c.cus ? c.cus : c.B, 1024, w16_lds(ntk), 1);
        if (fast) return done(ROUTE_DS256_G16, ntk, true, true, !o.g16_one_pass && c.B > c.cus ? c.cus : c.B, 1024, w16_lds(ntk), 1);
        // (other heads / feature layouts with an incoming cache: the LDS-tile kernel below)
      }
      if (C == 256 && o.w16_ok) {
        // the hand-over of ds256_w16 walks a block's slice in passes of 64 columns: any padding is covered (round-5 defect 1)
        return done(ROUTE_DS256_W16, nt, has_in, false, c.B, 1024, w16_lds(nt), 1);
      }
      if (C == 64 && o.g16_ok && d.num_layers <= 4 && f.dils_1248 && (!has_in || (o.g16_ctx && nt >= 2)) && linear2 && f.kpre16 <= 96 && x_items) {
        if (has_in && !(nt <= 4 || nt == 7)) return fail("ds64_g4 context variant: 4 or 7 tiles");
        return done(ROUTE_DS64_G4, has_in && nt <= 4 ? 4 : nt, has_in, true, c.B, 256, 2 * (64 / 8) * tt * 16, 1);
      }
      return done(ROUTE_CONV_F16, nt, has_in, false, (c.B + U - 1) / U, 512, 0, U);
    }
    case WEKWS_HIP_BACKBONE_TCN:
      if (!f16) return done(ROUTE_CONV_F32, nt, has_in, false, (c.B + U - 1) / U, 512, conv_f32_lds(), U);
      if (f.dense_ok) return done(ROUTE_DENSE_F16, nt, has_in, false, (c.B + U - 1) / U, 512, 0, U);
      return done(ROUTE_CONV_F16, nt, has_in, false, (c.B + U - 1) / U, 512, 0, U);
    case WEKWS_HIP_BACKBONE_MDTC: {
      const bool m16 = f16 && o.mdtc16_ok && f.mdtc16_eligible;
      if (m16 && f.mdtc_stream_eligible && o.stream_ok && c.ntiles == 1 && c.T <= 16 && (c.has_in || c.has_out) && c.cache16 && x_items)
        return done(ROUTE_MDTC64_STREAM, 1, true, false, (c.B + 1) / 2, 1024, mdtc_stream_lds, 2);
      const bool head_ok = linear2 || (pooled && d.head_hidden <= 448);
      if (m16 && o.g16_ok && f.mdtc_stream_eligible && (!has_in || (o.g16_ctx && nt >= 2 && (c.B > 2 || nt < 7))) && head_ok && f.kpre16 <= 96 &&
          x_items && (!has_in || linear2))
        return done(ROUTE_MDTC64_G4, has_in && nt <= 4 ? 4 : nt, has_in, true, c.B, 256, 2 * (64 / 8) * tt * 16, 1);
      if (m16) return done(ROUTE_MDTC64_W16, nt, has_in, false, (c.B + 1) / 2, 1024, 0, 2);
      if (f16 && C == 32 && o.g16_ok && d.stack_size <= 4 && f.dils_1248 && (!has_in || (o.g16_ctx && nt >= 2)) && head_ok && f.kpre16 <= 64 && x_items &&
          (!has_in || linear2))
        return done(ROUTE_MDTC32_G4, has_in && nt <= 4 ? 4 : nt, has_in, true, c.B, 128, 2 * (32 / 8) * tt * 16, 1);
      if (f16) {
        if (C > 128) return fail("MDTC wider than 128 channels does not fit the LDS tile: any-shape path");
        return done(ROUTE_CONV_F16, nt, has_in, false, (c.B + U - 1) / U, 512, 0, U);
      }
      if (conv_f32_lds() > 160 * 1024) return fail("conv_stack: tile beyond the LDS");
      return done(ROUTE_CONV_F32, nt, has_in, false, (c.B + U - 1) / U, 512, conv_f32_lds(), U);
    }
    default:
      return fail("not a conv backbone");
  }
}

}  // namespace wekws
