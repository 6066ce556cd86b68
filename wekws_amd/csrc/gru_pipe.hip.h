// GRU forward as a WAVEFRONT over the layers: one launch, the passes of gru_f16.hip.h as pipeline STAGES on different CUs.
// Reference semantics as gru.hip.h (torch.nn.GRU as built at wekws/model/kws_model.py:128-133; LinearSubsampling1
// subsampling.py:53-57; LinearClassifier classifier.py:63-67); arithmetic, operand values and block floating point exactly
// those of gru_f16.hip.h -- every column sees the same instructions in the same order, so the results are bit-identical.
//
// gru_f16.hip.h walks the network layer by layer: P, I0, R0, I1, R1, H.  Only the recurrences R are serial in time, but they
// ran one after the other (L x T layer-steps of ~1.1 us on 16 streams per CU) while most CUs idled.  Layer l's step t only
// needs layer l-1's step t, so here a tile of streams is worked on by 2 L workgroups AT THE SAME TIME:
//
//   stage 0        PI    in0[t] = [ReLU](Wpre x[t] + b);  gi0[t] = W_ih0 in0[t] + b      (no recurrence: runs ahead)
//   stage 2 l + 1  R_l   h_l[t] = cell(gi_l[t], h_l[t-1])                                (the serial chain; last layer: + head)
//   stage 2 l      I_l   gi_l[t] = W_ih,l h_{l-1}[t] + b     (l >= 1)                    (per step, behind R_{l-1})
//
// so the chain is T + (a few steps of lag) instead of L T, each stage keeps ITS matrices in registers for the whole launch, and
// a workgroup (= one CU: 8 waves x up to 256 registers) serves several tiles one after the other ("rounds") when there are
// more tiles than CUs / (2 L) ("slots").
//
// Hand-over between the workgroups of a slot (MI355X: a CU's L1 never sees other CUs' stores, per-XCD L2s are not coherent
// with each other, and a write-through store is acknowledged only after 1.5 .. 2 us under load -- measured: a version with
// progress flags behind `s_waitcnt vmcnt(0)` spent more time waiting for acknowledgements than computing):
//   * everything that crosses workgroups travels as GRANULES that carry a 32-bit tag next to the data, written by 16-byte
//     buffer stores and read with sc1 loads (L2-served, never a stale L1 line): hidden states as two 8-byte {hi | lo fp16
//     pair, tag} granules per store, gate pre-activations as one 16-byte {v0, v1, v2, tag} granule per store.  The second form
//     leans on naturally aligned 16-byte stores and loads being single transactions (tools/probe/tear16.hip: no torn item in
//     3.9e10 concurrent reads, both store policies); vector stores are ISSUE-bound on this part (~60 cycles per
//     wave-instruction on a CU), and with 8-byte gate granules the six stores per wave and step were the first stage's time.  The data IS the flag: a consumer
//     requests a step's granules ahead of time, checks the tags when it needs the values (all equal to the step's tag: xor / or, one compare)
//     and re-requests until they are there.  Nobody waits for a store to complete,
//     there is no fence, no flag and no poll on the fast path, and a producer never waits for its consumer inside a round.
//   * where the stores go: a workgroup announces the XCD it runs on (HW_REG_XCC_ID) in a control word; a producer that finds its
//     consumer on ITS OWN XCD within a few microseconds stores with the default policy -- the two share one L2, which then
//     has the data when the store is acknowledged (~0.4 us), and the consumer's L2-served loads see it --; otherwise (other
//     XCD, or the consumer not resident yet) it stores write-through.  vmcnt is ONE in-order counter for loads and stores,
//     so whatever a wave requests behind its own stores waits for their acknowledgement: with write-through stores the
//     recurrence's step became the acknowledgement latency.  (Observed placement: block b runs on XCD b % 8, so with the
//     slots padded to a multiple of 8 a slot's stages share an XCD; nothing relies on it.)
//   * the buffers are RINGS of kGruPipeRing (16) steps per slot: a slot's steps are numbered g = round * T + t, step g lives at
//     position g mod 16 and carries tag = (launch epoch + 1) + g / 16, the tag of its lap.  (One position per step of the call was
//     504 MB written + 524 MB fetched per launch at B = 1024 x 98 frames -- PMC -- i.e. 5 TB/s of HBM traffic for values that live
//     a few microseconds; the rings are 80 MB for 64 slots and stay in the 256 MB memory-side cache -- sc1 loads are served
//     from the fabric, not from L2, so the byte counts at the L2 boundary did not change, only where they are served from:
//     4.8 -> 5.7 M utt/s.  Group-scope loads -- sc0 -- for a producer on the same XCD were tried: they hit stale L1 lines.)
//     The epoch is a word in the control block that the LAST workgroup to start advances by the launch's laps (one relaxed
//     agent-scope counter; everybody reads the epoch before counting itself in) -- nothing depends on a per-launch kernel
//     argument, so a captured graph replays correctly.  Granule buffers hold nothing but granules (every tag word was written by
//     some launch, or is the zero the allocation was cleared to; tags start at 1), so a stale or foreign word cannot pass for
//     the current tag (compared for equality: the 32-bit epoch may wrap; 0 is never a tag).
//   * a position is written again 16 steps later, when its consumer is through with it: a 64-bit CREDIT word per (slot, stage)
//     = {tag0 of the launch, steps the consumer has finished reading}, published every fourth step (behind the consumer's
//     barrier: every wave has the step in registers), read by the producer only when its cached copy does not cover the
//     step it is about to write, never for a launch's first 16 steps (what the ring holds then belongs to launches that are
//     finished, in stream order) and never in launches of <= 16 steps (the streaming chunks).  A word of another launch
//     never matches tag0.  A consumer that has to wait for data publishes first: the producer it waits for may be waiting
//     for exactly that credit (the time-packed first stage writes a whole lap at once).
// No deadlock, also with other launches on the device (round 5; the workgroup numbering in the kernel says why): a consumer's
// producers have lower workgroup indices ON ITS XCD, so whenever it is resident they are resident or done (in-order dispatch
// per XCD); a producer waits for its consumer's credits, and the numbering keeps a launch's resident workgroups to complete
// slots plus at most one incomplete slot per XCD -- complete slots finish on their own and free the CUs the incomplete ones
// wait for.  Every wait is bounded anyway (~0.1 s; error word in device AND host memory: the launch then ends with garbage
// instead of hanging the GPU, and the next library call on the stream returns WEKWS_HIP_EDEVICE).
#pragma once
#include "gru_f16.hip.h"

// Policy of the granule loads.  What ships (WEKWS_GRU_PIPE_L2 undefined or 0): sc1 loads, served from the fabric.
// -DWEKWS_GRU_PIPE_L2=n builds the L2-resident hand-over (round 5): a step's FIRST request is a plain load behind an L1
// invalidate (n = 1: buffer_inv sc0, n = 2: buffer_inv sc1) -- it is served by this XCD's L2, which holds the line dirty when
// the producer runs on the same XCD and stored with the default policy (the same-XCD fast path of the stores) -- and every
// RE-request (the tag check failed: producer late, or on another XCD, where this L2 may hold a stale clean copy for ever) is
// an sc1 load as before.  So placement never decides correctness, only where the first request is served from.
#ifndef WEKWS_GRU_PIPE_L2
#define WEKWS_GRU_PIPE_L2 0
#endif
#if WEKWS_GRU_PIPE_L2 == 1
#define GP_LDPOL ""
#define GP_LDINV "buffer_inv sc0\n\t"
#elif WEKWS_GRU_PIPE_L2 == 2
#define GP_LDPOL ""
#define GP_LDINV "buffer_inv sc1\n\t"
#else
#define GP_LDPOL "sc1"
#define GP_LDINV ""
#endif
#define GP_RETRY_POL "sc1"
// pause in front of a re-request of granules that were not there yet (units of 64 clocks)
#ifndef GP_DATA_SLEEP
#define GP_DATA_SLEEP 2
#endif
namespace wekws {

constexpr int kGruPipeStages = 2 * kGruMaxLayers;
constexpr int kGruPipeMaxSlots = 128;
// Granule buffers are RINGS of kGruPipeRing steps per slot (round 4: with one buffer position per step of the call, B = 1024 x
// 98 frames wrote 504 MB and fetched 524 MB per launch -- PMC -- i.e. 5 TB/s of HBM traffic for intermediates that live a few
// microseconds; a ring of 16 steps is 1.25 MB per slot and stays in L2 / the memory-side cache).
constexpr int kGruPipeRingLog = 4, kGruPipeRing = 1 << kGruPipeRingLog;
// control words (their own allocation per stream, zero when made, never re-allocated): [0] epoch, [1] workgroups done, [2] error,
// [16 + 2 (slot * stages + stage)] 64-bit credits: {tag0, steps of this launch the stage's consumer has finished reading},
// [16 + 2 slots * stages + slot * stages + stage] where the workgroup runs: tag0 << 4 | XCD
// [16 + 3 slots * stages + slot * stages + stage] round 6: "this recurrence stage has written the outputs of round r": tag0 + r
//   (read only by the slot's non-finite workgroup, and only when a stream of the tile has a NaN / Inf input: below)
constexpr int kGruPipeDoneWord = 16 + 3 * kGruPipeMaxSlots * kGruPipeStages;
constexpr int kGruPipeCtlWords = 16 + 4 * kGruPipeMaxSlots * kGruPipeStages;
constexpr size_t kGruPipeCtlBytes = (size_t(kGruPipeCtlWords) * 4 + 255) / 256 * 256;
// Bound of ONE wait, in re-requests.  The first kGruPipeFastSpins are back to back (~0.5 .. 1 us each: a load's round trip);
// from then on every re-request sleeps ~3 us first (s_sleep 100 = 6400 clocks), so the bound is 0.1 .. 0.2 s of EXECUTED
// waiting whatever the memory system's latency is (round 5's first cut counted 2^16 back-to-back re-requests -- ~30 ms: a
// launch squeezed for 30 ms by another tenant gave up, measured) -- a count, not a wall-clock reading, so that a queue the
// scheduler pre-empts in favour of another process does not time out while its waves are saved; and a waiting workgroup
// stops hammering the fabric with requests after the first ~50 us.  A wait that gives up marks the whole launch dead
// (every later wait of every workgroup falls through at once: the launch ends within a few of these bounds), and the host
// hears of it on its next call (GruPipeWorkspace::err).
constexpr unsigned kGruPipeFastSpins = 64;
constexpr unsigned kGruPipeSpinLimit = 1u << 15;
constexpr int kGruPipeGiStep = 8 * 4 * 1024;                  // bytes of one step of gate granules: [wave][item][lane][16]
constexpr int kGruPipeHStep = 16 * 16 * 64;                   // bytes of one step of state granules: [k-octet][stream][8][8]
constexpr int kGruPipePlanes = 128 * 1024;                    // staging buffers of stage 0; 16 steps of planes of a time-packed tile
constexpr int kGruPipeLds = kGruPipePlanes + 1024;            // dynamic LDS of a workgroup (one per CU anyway): the planes, and
                                                              // behind them the per-column scales of a time-packed tile

struct GruPipeWorkspace {
  unsigned* ctl;                // control words
  unsigned* err;                // device address of a word in HOST (pinned, mapped) memory: a wait that gave up leaves its code
                                // here too, where the next library call on the stream finds it without synchronising anything
  char* seq_in;                 // [slot][T] time-packed tiles only: preprocessing output planes (stage 0 internal)
  char* seq_top;                // [slot][T] last layer's output planes (read back by its own workgroup's head pass)
  float* sc;                    // [slot][T][16] time-packed tiles only: 1 / scale of the preprocessing output
  char* gi[kGruMaxLayers];      // granules [slot][kGruPipeRing][kGruPipeGiStep]: gate pre-activations of layer l
  char* hs[kGruMaxLayers];      // granules [region][T][kGruPipeHStep]: output sequence of layer l (l < L - 1)
  const NfCtx* nf;              // round 6: streams with a NaN / Inf input are re-computed by the slot's non-finite workgroup
};

static __device__ __attribute__((noinline, unused)) void nf_repair_gru_call(const NfCtx* nf, const float* x, int64_t xs_b, const float* h0,
                                                                            float* hn, float* y, int64_t ys_b, int B, int T, int b) {
  nf_repair_gru(nf, x, xs_b, h0, hn, y, ys_b, B, T, b);
}

// measurement build (tools/probe/gru_stamps.py): 100 MHz wall-clock stamps of slot 0's stages, row k, step t
#ifdef WEKWS_GRU_PIPE_STAMPS
__device__ unsigned long long gp_stamps[16 * 1024];
#define GP_STAMP(k, t) do { if (slot == 0 && tid == 0 && (t) < 1024) gp_stamps[(k) * 1024 + (t)] = wall_clock64(); } while (0)
#else
#define GP_STAMP(k, t) do {} while (0)
#endif

typedef unsigned gp_u32x4 __attribute__((ext_vector_type(4)));
constexpr int kGpSc1 = 16;                                    // buffer aux bit: sc1 (write-through store / L2-served load)

__device__ __forceinline__ unsigned gp_ld_ctl(const unsigned* p) {
  return unsigned(__builtin_amdgcn_readfirstlane(int(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))));
}
__device__ __forceinline__ unsigned long long gp_ld_ctl64(const unsigned long long* p) {
  const unsigned long long v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return (unsigned long long)unsigned(__builtin_amdgcn_readfirstlane(int(unsigned(v)))) |
         ((unsigned long long)unsigned(__builtin_amdgcn_readfirstlane(int(unsigned(v >> 32)))) << 32);
}
// Barrier of the per-step loops: orders LDS traffic only.  __syncthreads() is `s_waitcnt vmcnt(0) lgkmcnt(0)` + s_barrier on
// this target (the workgroup-scope fence covers global memory): every step then waited for the gate values requested two
// steps ahead AND for the acknowledgement of the step's own stores.  Nothing in these loops hands global data to another
// wave of the workgroup.
__device__ __forceinline__ void gp_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// Granule loads the compiler does not count: hipcc merges its vmcnt bookkeeping at loop headers and ends up with
// `s_waitcnt vmcnt(0)` in front of every tag check -- the step then waits for the OTHER buffer's requests and for its own
// stores.  These loads are inline assembly (invisible to that bookkeeping; its own waits only get more conservative), and
// the wait in front of their first use is spelled out: vmcnt is one in-order counter, so "at most the N newest requests
// outstanding" is exact.  The wait statement names every destination as read-write, so nothing touches them before it.
typedef unsigned gp_desc __attribute__((ext_vector_type(4)));
__device__ __forceinline__ gp_desc gp_make_desc(const void* base, unsigned bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  return gp_desc{unsigned(__builtin_amdgcn_readfirstlane(int(unsigned(a)))),
                 unsigned(__builtin_amdgcn_readfirstlane(int(unsigned(a >> 32) & 0xffffu))),
                 unsigned(__builtin_amdgcn_readfirstlane(int(bytes))), 0x00020000u};
}
// four 16-byte items at voff + {0, 1, 2, 3} KiB
template <bool RETRY = false>
__device__ __forceinline__ void gp_ld4(gp_u32x4 (&g)[4], int voff, gp_desc rs) {
  if constexpr (RETRY)
    asm volatile(
        "s_nop 4\n\t"
        "buffer_load_dwordx4 %0, %4, %5, 0 offen " GP_RETRY_POL "\n\t"
        "buffer_load_dwordx4 %1, %4, %5, 0 offen offset:1024 " GP_RETRY_POL "\n\t"
        "buffer_load_dwordx4 %2, %4, %5, 0 offen offset:2048 " GP_RETRY_POL "\n\t"
        "buffer_load_dwordx4 %3, %4, %5, 0 offen offset:3072 " GP_RETRY_POL
        : "=&v"(g[0]), "=&v"(g[1]), "=&v"(g[2]), "=&v"(g[3])
        : "v"(voff), "s"(rs)
        : "memory");
  else
    asm volatile(
        GP_LDINV "s_nop 4\n\t"
        "buffer_load_dwordx4 %0, %4, %5, 0 offen " GP_LDPOL "\n\t"
        "buffer_load_dwordx4 %1, %4, %5, 0 offen offset:1024 " GP_LDPOL "\n\t"
        "buffer_load_dwordx4 %2, %4, %5, 0 offen offset:2048 " GP_LDPOL "\n\t"
        "buffer_load_dwordx4 %3, %4, %5, 0 offen offset:3072 " GP_LDPOL
        : "=&v"(g[0]), "=&v"(g[1]), "=&v"(g[2]), "=&v"(g[3])
        : "v"(voff), "s"(rs)
        : "memory");
}
template <int N>
__device__ __forceinline__ void gp_wait4(gp_u32x4 (&g)[4]) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]) : "i"(N) : "memory");
}
// The same three with a wave-uniform switch INSIDE the statement (the last layer's early tag check): `if (on) wait` written as
// two statements in the arms of a branch is what made the compiler move in-flight registers (see the step loop).
template <int N>
__device__ __forceinline__ void gp_wait4_if(gp_u32x4 (&g)[4], unsigned on) {           // on ? vmcnt(N) : nothing
  asm volatile("s_cmp_eq_u32 %4, 0\n\ts_cbranch_scc1 .Lgpw%=\n\ts_waitcnt vmcnt(%5)\n.Lgpw%=:"
               : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]) : "s"(__builtin_amdgcn_readfirstlane(int(on))), "i"(N) : "memory", "scc");
}
__device__ __forceinline__ void gp_ld4_if(gp_u32x4 (&g)[4], int voff, gp_desc rs, unsigned on) {   // on ? request again : nothing
  asm volatile(
      "s_cmp_eq_u32 %6, 0\n\ts_cbranch_scc1 .Lgpl%=\n\t"
      "s_nop 4\n\t"
      "buffer_load_dwordx4 %0, %4, %5, 0 offen " GP_RETRY_POL "\n\t"
      "buffer_load_dwordx4 %1, %4, %5, 0 offen offset:1024 " GP_RETRY_POL "\n\t"
      "buffer_load_dwordx4 %2, %4, %5, 0 offen offset:2048 " GP_RETRY_POL "\n\t"
      "buffer_load_dwordx4 %3, %4, %5, 0 offen offset:3072 " GP_RETRY_POL "\n.Lgpl%=:"
      : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3])
      : "v"(voff), "s"(rs), "s"(__builtin_amdgcn_readfirstlane(int(on)))
      : "memory", "scc");
}
template <int N>
__device__ __forceinline__ void gp_wait4_sel(gp_u32x4 (&g)[4], unsigned all) {         // all ? vmcnt(0) : vmcnt(N)
  asm volatile("s_cmp_eq_u32 %4, 0\n\ts_cbranch_scc1 .Lgpa%=\n\ts_waitcnt vmcnt(0)\n\ts_branch .Lgpb%=\n"
               ".Lgpa%=:\n\ts_waitcnt vmcnt(%5)\n.Lgpb%=:"
               : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]) : "s"(__builtin_amdgcn_readfirstlane(int(all))), "i"(N) : "memory", "scc");
}
// two 16-byte items at voff, voff + 16
template <bool RETRY = false>
__device__ __forceinline__ void gp_ld2(gp_u32x4 (&g)[2], int voff, gp_desc rs) {
  if constexpr (RETRY)
    asm volatile(
        "s_nop 4\n\t"
        "buffer_load_dwordx4 %0, %2, %3, 0 offen " GP_RETRY_POL "\n\t"
        "buffer_load_dwordx4 %1, %2, %3, 0 offen offset:16 " GP_RETRY_POL
        : "=&v"(g[0]), "=&v"(g[1])
        : "v"(voff), "s"(rs)
        : "memory");
  else
    asm volatile(
        GP_LDINV "s_nop 4\n\t"
        "buffer_load_dwordx4 %0, %2, %3, 0 offen " GP_LDPOL "\n\t"
        "buffer_load_dwordx4 %1, %2, %3, 0 offen offset:16 " GP_LDPOL
        : "=&v"(g[0]), "=&v"(g[1])
        : "v"(voff), "s"(rs)
        : "memory");
}
template <int N>
__device__ __forceinline__ void gp_wait2(gp_u32x4 (&g)[2]) {
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(g[0]), "+v"(g[1]) : "i"(N) : "memory");
}
// One step of features for this lane (K steps 0 and 1, 2 x 16 bytes each), not counted by the compiler either
__device__ __forceinline__ void gp_ldx4(f32x4 (&v)[4], const float* p0, const float* p1) {
  asm volatile(
      "global_load_dwordx4 %0, %4, off\n\t"
      "global_load_dwordx4 %1, %4, off offset:16\n\t"
      "global_load_dwordx4 %2, %5, off\n\t"
      "global_load_dwordx4 %3, %5, off offset:16"
      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
      : "v"(p0), "v"(p1)
      : "memory");
}
template <int N>
__device__ __forceinline__ void gp_waitx16(f32x4 (&a)[4], f32x4 (&b)[4], f32x4 (&c)[4], f32x4 (&d)[4]) {
  asm volatile("s_waitcnt vmcnt(%16)"
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]),
                 "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3])
               : "i"(N)
               : "memory");
}
// one step's four: exact ? vmcnt(N) : vmcnt(0), one statement (see gp_wait4_sel)
template <int N>
__device__ __forceinline__ void gp_waitx4_sel(f32x4 (&a)[4], unsigned exact) {
  asm volatile("s_cmp_eq_u32 %4, 0\n\ts_cbranch_scc1 .Lgpx%=\n\ts_waitcnt vmcnt(%5)\n\ts_branch .Lgpy%=\n"
               ".Lgpx%=:\n\ts_waitcnt vmcnt(0)\n.Lgpy%=:"
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "s"(__builtin_amdgcn_readfirstlane(int(exact))), "i"(N) : "memory", "scc");
}
// 16-byte store: default policy (the consumer shares this XCD's L2) or write-through
__device__ __forceinline__ void gp_st16(gp_u32x4 v, __amdgpu_buffer_rsrc_t rs, int off, bool same_xcd) {
  if (same_xcd) __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 0);
  else __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, kGpSc1);
}
// the twelve gate values of a lane and step (r0..3, z0..3, n0..3) as four granules {v, v, v, tag} at off + {0, 1, 2, 3} KiB
__device__ __forceinline__ void gp_st_gates(const f32x4 (&v)[3], unsigned tag, __amdgpu_buffer_rsrc_t rs, int off, bool same_xcd) {
  gp_st16(gp_u32x4{__float_as_uint(v[0][0]), __float_as_uint(v[0][1]), __float_as_uint(v[0][2]), tag}, rs, off, same_xcd);
  gp_st16(gp_u32x4{__float_as_uint(v[0][3]), __float_as_uint(v[1][0]), __float_as_uint(v[1][1]), tag}, rs, off + 1024, same_xcd);
  gp_st16(gp_u32x4{__float_as_uint(v[1][2]), __float_as_uint(v[1][3]), __float_as_uint(v[2][0]), tag}, rs, off + 2048, same_xcd);
  gp_st16(gp_u32x4{__float_as_uint(v[2][1]), __float_as_uint(v[2][2]), __float_as_uint(v[2][3]), tag}, rs, off + 3072, same_xcd);
}
// two granules {a, tag}, {b, tag}
__device__ __forceinline__ gp_u32x4 gp_pair(unsigned a, unsigned b, unsigned tag) { return gp_u32x4{a, tag, b, tag}; }

// NKP: K steps of the preprocessing product (idim <= 32 NKP).  PK: every tile has at most 8 streams (the launches that serve
// streaming chunks: spw <= 8) -- the first stage then consists of its time-packed path alone.  One kernel with both paths kept
// a spilled value of the chunked path's 250 registers in the packed path's prologue, reloaded from scratch between the
// chunk's two products (+0.5 us on a 10-frame chunk).
template <int NKP, bool PK>
__global__ __launch_bounds__(kThreads) void gru_pipe_kernel(const GruF16Params Q, const GruPipeWorkspace WS,
                                                            const float* __restrict__ x, int B, int T,
                                                            const float* __restrict__ h0, float* __restrict__ y,
                                                            float* __restrict__ hn, int tiles, int slots, int slots_p,
                                                            int spw) {
  constexpr int NN = 1;
  using G = GruF16Geom<NN>;
  constexpr int MB = G::MB, H = kGruH, PH = G::PLANE_H;
  constexpr int KSB = 4 * MB * 16;                            // bytes per K step inside a plane
  constexpr int OTS = (H / 32) * 128;                         // uint4 per o-tile of an H-deep matrix (4 K steps)
  constexpr int SEQ = G::SEQ_STEP, GIS = kGruPipeGiStep, HSS = kGruPipeHStep;
  const GruParams& P = Q.base;
  extern __shared__ __attribute__((aligned(16))) char gp_lds[];

  // Workgroup -> (slot, stage), GROUP-major: eight slots x all stages are 8 x stages consecutive workgroups, stage-major inside
  // the group.  (i) Workgroup b runs on XCD b % 8 (static round-robin), so a slot's stages share an XCD -- and its L2 -- as
  // before.  (ii) Every XCD dispatches ITS workgroups in index order, and in this numbering that order is slot after slot,
  // producers first: at any moment the resident workgroups of a launch are complete slots plus at most one slot per XCD whose
  // later stages are still waiting for a CU.  Complete slots depend on nobody and finish; what they free goes to the next
  // workgroups in order.  So several wavefront launches sharing the device (streams, models, processes) cannot starve each
  // other into a deadlock as long as launches x (stages - 1) < CUs per XCD -- round 4's stage-major numbering put ALL
  // producers first, and two launches that each got their producer half resident waited for consumers that had no CU
  // (their ring credits never came).
  const int nstg = 2 * Q.base.nlayers;
  // Round 6: behind the stage workgroups come `slots` NON-FINITE workgroups, one per slot (index = slots_p * stages + slot: the
  // same XCD -- and L2 -- as the slot's stages).  Each looks at the features and incoming states of its slot's tiles while the
  // pipeline runs; if they are all finite -- the normal case -- it is gone long before the pipeline is; if not, it waits for the
  // slot's recurrence stages to finish the round and re-computes those streams (nonfinite.hip.h).  The stages never wait for it.
  const bool nf_wg = WS.nf != nullptr && int(blockIdx.x) >= slots_p * nstg;
  const int bx = nf_wg ? 0 : int(blockIdx.x);
  const int grp = bx / (8 * nstg), gr = bx - grp * (8 * nstg);
  const int stage = nf_wg ? nstg : gr >> 3, slot = nf_wg ? int(blockIdx.x) - slots_p * nstg : grp * 8 + (gr & 7);
  if (slot >= slots) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  const float* __restrict__ W = P.w;
  const int L = P.nlayers, K = P.odim, idim = P.idim;
  const int u0 = wave * 16 + lq * 4;                        // first of this lane's 4 hidden units
  const int frag = (lq * MB + l15) * 16;                    // this lane's B-fragment item of stream tile 0, K step 0
  const int wr_off = (((u0 >> 3) * MB + l15) * 8 + (u0 & 7)) * 2;   // this lane's 4 units inside an H-wide plane
  const int gvo = wave * 4096 + lane * 16;                  // this lane's first gate granule inside a step (+ item * 1024)
  const int rounds = (tiles + slots - 1) / slots;
  unsigned* const ctl = WS.ctl;
  constexpr int RING = kGruPipeRing, RLOG = kGruPipeRingLog;
  unsigned long long* const cred = reinterpret_cast<unsigned long long*>(ctl + 16);
  unsigned long long* const cred_out = cred + slot * kGruPipeStages + (stage > 0 ? stage - 1 : 0);   // what this stage has finished reading
  const unsigned long long* const cred_in = cred + slot * kGruPipeStages + stage;                    // what its consumer has finished reading
  const unsigned tag0 = gp_ld_ctl(ctl) + 1u;                // tag of this launch's first lap of the rings
  // (the epoch load must have RETURNED before this workgroup counts itself in below: the statement consumes the value -- the
  // wave waits for it here -- and, as a memory clobber, keeps the compiler from moving the atomic above it; the memory
  // pipeline issues in program order)
  asm volatile("" :: "s"(tag0) : "memory");
  // The LAST workgroup to get here advances the epoch by the tags this launch uses.  Every workgroup has read the epoch
  // (above: the value is back before the atomic is issued) before it counts itself in, so the one that counts last knows
  // that nobody will read it again in this launch -- and no atomic round trip sits at the END of the launch, where a
  // 10-frame chunk would pay ~1 us for it.
  if (tid == 0) {
    const unsigned nwg = unsigned((2 * P.nlayers + (WS.nf ? 1 : 0)) * slots);
    if (__hip_atomic_fetch_add(ctl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nwg - 1u) {
      __hip_atomic_store(ctl + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned laps = unsigned((rounds * T - 1) >> RLOG) + 1u;   // a slot's steps are numbered g = round * T + t
      laps = laps < unsigned(rounds) ? unsigned(rounds) : laps;    // (the "round done" words are tag0 + round: unique across launches too)
      __hip_atomic_store(ctl, tag0 - 1u + laps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  unsigned* const done_w = ctl + kGruPipeDoneWord + slot * kGruPipeStages;   // [stage] of this slot
  auto round_tag = [&](int r) __attribute__((always_inline)) -> unsigned {
    const unsigned v = tag0 + unsigned(r);
    return v ? v : 0x80000000u;
  };
  if (nf_wg) {
    __shared__ unsigned nfx_mask, nfx_ok;
    int round = 0;
#pragma unroll 1
    for (int tile = slot; tile < tiles; tile += slots, ++round) {
      const int b0 = tile * spw, nb = min(B, b0 + spw) - b0;
      if (tid == 0) { nfx_mask = 0u; nfx_ok = 1u; }
      __syncthreads();
      const int per = T * idim;
      unsigned mine = 0u;
      for (int e = tid; e < nb * per; e += kThreads) {
        const int sidx = e / per;
        if (nf_bad(x[int64_t(b0) * per + e])) mine |= 1u << sidx;
      }
      if (h0)
        for (int e = tid; e < L * nb * H; e += kThreads) {
          const int lyr = e / (nb * H), rem = e - lyr * (nb * H), sidx = rem / H;
          if (nf_bad(h0[(int64_t(lyr) * B + b0 + sidx) * H + (rem - sidx * H)])) mine |= 1u << sidx;
        }
      if (mine) atomicOr(&nfx_mask, mine);
      __syncthreads();
      const unsigned mask = unsigned(__builtin_amdgcn_readfirstlane(int(nfx_mask)));
      if (!mask) continue;
      // a stream of this tile has a NaN / Inf input: wait (bounded, like every wait of this kernel) until the slot's recurrence
      // stages have written their outputs of this round -- garbage for that stream, its loads entered as 0 --, then write the
      // reference's.  Same L2 as the stages (same XCD): their stores are there when their "round done" word is.
      if (tid == 0) {
        for (int lyr = 0; lyr < L && nfx_ok; ++lyr) {
          unsigned spins = 0;
          while (gp_ld_ctl(done_w + 2 * lyr + 1) != round_tag(round)) {
            __builtin_amdgcn_s_sleep(100);
            if (++spins > kGruPipeSpinLimit || gp_ld_ctl(ctl + 2) != 0u) { nfx_ok = 0u; break; }
          }
        }
        if (!nfx_ok && WS.err) __hip_atomic_store(WS.err, 0x300u + unsigned(nstg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      __syncthreads();
      if (!__builtin_amdgcn_readfirstlane(int(nfx_ok))) return;
      for (int sidx = 0; sidx < nb; ++sidx)
        if (mask >> sidx & 1u) nf_repair_gru_call(WS.nf, x, int64_t(T) * idim, h0, hn, y, int64_t(T) * K, B, T, b0 + sidx);
    }
    return;
  }
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
  unsigned* const where = ctl + 16 + (2 * kGruPipeMaxSlots + slot) * kGruPipeStages;   // [stage] of this slot
  if (tid == 0) __hip_atomic_store(where + stage, (tag0 << 4) | 8u | xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // does this stage's consumer run on this XCD?  asked after the weights have been requested; gives up after ~20 us (the
  // consumer may not be resident yet: write-through stores are right wherever it turns up)
  auto peer_here = [&](int st) __attribute__((always_inline)) -> bool {
    for (int i = 0; i < 48; ++i) {
      const unsigned w = gp_ld_ctl(where + st);
      if ((w & 8u) && (w >> 4) == (tag0 & 0x0fffffffu)) return (w & 7u) == xcc;   // (bit 3: written at all)
      __builtin_amdgcn_s_sleep(8);
    }
    return false;
  };
  auto consumer_here = [&]() __attribute__((always_inline)) -> bool { return peer_here(stage + 1); };
  // Step g (= round * T + t) of a slot lives at ring position g mod RING and carries the tag of its lap; a position is
  // written again RING steps later, when the consumer has said that it is through with it: a CREDIT word per (slot, stage)
  // = {tag0 of the launch, steps finished}, published every few steps.  A word of another launch never matches tag0; the
  // first RING steps of a launch need no credit (what the ring holds then is of earlier launches: finished, in stream order).
  // (the epoch is a 32-bit counter that wraps -- after ~30 h of back-to-back single-chunk launches --: tags are compared for
  // EQUALITY, never for order, and 0 -- what a cleared buffer holds -- is never a tag)
  auto lap_tag = [&](int g) __attribute__((always_inline)) -> unsigned {
    const unsigned r = tag0 + unsigned(g >> RLOG);
    return r ? r : 0x80000000u;
  };
  auto ring_pos = [&](int g) __attribute__((always_inline)) -> int { return g & (RING - 1); };
  // One more re-request of a bounded wait: true = stop waiting (the launch is dead: this wait ran out, or another workgroup's
  // did -- looked up every 64th time; the results are garbage then and the error word says so).
  unsigned dead = 0;
  auto give_up = [&](unsigned& spins, unsigned what) __attribute__((always_inline)) -> bool {
    if (dead) return true;
    ++spins;
    if (spins > kGruPipeFastSpins) __builtin_amdgcn_s_sleep(100);
    if ((spins & 63u) == 0u) {
      if (const unsigned seen = gp_ld_ctl(ctl + 2); seen != 0u) {
        // another workgroup's wait ran out -- in THIS launch, or in an earlier one whose device word could not be cleared yet (the
        // host clears it when it reports the failure; during a graph capture it cannot).  Either way this launch's outputs are
        // not valid: the host word says so again, so the call after this one reports it too (launches queued behind a
        // failed one died silently once the first report had cleared the host word).
        dead = 1u;
        if (WS.err) __hip_atomic_store(WS.err, seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return true;
      }
    }
    if (spins <= kGruPipeSpinLimit) return false;
    dead = 1u;
    __hip_atomic_store(ctl + 2, what + unsigned(stage), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (WS.err) __hip_atomic_store(WS.err, what + unsigned(stage), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return true;
  };
  unsigned credit = 0;                                        // steps the consumer is known to have finished
  auto wait_credit = [&](int g) __attribute__((always_inline)) {   // before step g is written
    const unsigned need = unsigned(g - RING + 1);
    if (g < RING || credit >= need) return;
    unsigned spins = 0;
    for (;;) {
      const unsigned long long c = gp_ld_ctl64(cred_in);
      if (unsigned(c >> 32) == tag0) {
        credit = unsigned(c);
        if (credit >= need) break;
      }
      __builtin_amdgcn_s_sleep(8);
      if (give_up(spins, 0x200u)) break;
    }
  };
  // this stage has finished reading its input up to (not including) step g; the store goes where the producer's loads look
  const auto rs_cred = __builtin_amdgcn_make_buffer_rsrc(cred_out, 0, 8, 0x00020000);
  // (a launch of at most RING steps per slot -- the streaming chunks -- needs no credits at all)
  const bool credits = rounds * T > RING;
  auto publish = [&](int g, bool same_xcd) __attribute__((always_inline)) {
    if (credits && tid == 0) {
      typedef unsigned gp_u32x2 __attribute__((ext_vector_type(2)));
      const gp_u32x2 v = {unsigned(g), tag0};
      if (same_xcd) __builtin_amdgcn_raw_buffer_store_b64(v, rs_cred, 0, 0, 0);
      else __builtin_amdgcn_raw_buffer_store_b64(v, rs_cred, 0, 0, kGpSc1);
    }
  };

  __shared__ AmaxCell gp_cell;
  // bound of layer l's hidden state over a stream tile: max(1, max|h0[l]|) (gru_f16.hip.h); all threads, three barriers
  auto h_bound = [&](int l, int b0, int bend) __attribute__((always_inline)) -> float {
    __syncthreads();
    if (tid == 0) gp_cell.v = 0u;
    __syncthreads();
    float m = 0.f;
    if (h0)
      for (int e = tid; e < MB * H; e += kThreads) {
        const int sidx = b0 + e / H;
        if (sidx < bend) m = fmaxf(m, fabsf(nf_clean(h0[(int64_t(l) * B + sidx) * H + (e % H)])));
      }
    amax_publish(&gp_cell, m);
    __syncthreads();
    return fmaxf(1.f, amax_read(&gp_cell));
  };
  if (stage == 0) {
    // ============ stage PI: in0[t] = [ReLU](Wpre x[t] + b) (subsampling.py:53-57), gi0[t] = W_ih0 in0[t] + b ============
    const bool xvec = (idim % 4 == 0) && (reinterpret_cast<uintptr_t>(x) % 16 == 0);
    // this lane's 8 features of K step ks of (stream s, step t); zeros outside
    auto load_x8 = [&](int s, int t, int ks, bool ok) __attribute__((always_inline)) -> gru_f32x8 {
      gru_f32x8 v = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const int k0 = ks * 32 + lq * 8;
      if (ok && k0 < idim) {
        const float* src = x + (int64_t(s) * T + t) * idim + k0;
        if (xvec && k0 + 8 <= idim) {
          const f32x4 lo4 = *reinterpret_cast<const f32x4*>(src), hi4 = *reinterpret_cast<const f32x4*>(src + 4);
          v = gru_f32x8{lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (k0 + j < idim) v[j] = src[j];
        }
      }
      return nf_clean_vec<gru_f32x8, 8>(v);                  // (a NaN / Inf feature enters as 0: nonfinite.hip.h)
    };
    // A chunk of a few streams is served in ONE time-packed pass, and the whole launch waits for it: its features are the
    // first thing this stage asks for -- in front of 100 KB of weight fragments and of the handshake (measured at B = 1,
    // T = 10: layer 0 had its first gate values 4.2 us into the launch, 2.4 us after it was ready for them).
    gru_f32x8 xfirst[NKP];
    {
      const int b0 = slot * spw, nb = min(B, b0 + spw) - b0;
      const int psh = nb <= 1 ? 0 : nb <= 2 ? 1 : nb <= 4 ? 2 : 3;
      const int pcs = l15 & ((1 << psh) - 1), pdt = l15 >> psh;
      const bool cv = slot < tiles && nb <= 8 && pdt < T && pcs < nb;
#pragma unroll
      for (int ks = 0; ks < NKP; ++ks) xfirst[ks] = load_x8(b0 + pcs, pdt, ks, cv);
    }
    // (... and the consumer's "where" word is requested here and looked at in front of the first store: by then it has
    // normally been written, and the chunk does not wait for a control word's round trip between its two products)
    const unsigned where_early = __hip_atomic_load(where + stage + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    F16Frag a[NKP];
    {
      const int nkp = Q.kpre16 / 32;                          // K steps the packed matrix really has (<= NKP: the launcher)
      const uint4* ap = reinterpret_cast<const uint4*>(W + Q.pre_a16) + size_t(wave) * nkp * 128 + lane;
#pragma unroll
      for (int ks = 0; ks < NKP; ++ks) {
        a[ks].h = a[ks].l = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
        if (ks < nkp) a[ks] = load_frag(ap + ks * 128);
      }
    }
    const f32x4 bpre = *reinterpret_cast<const f32x4*>(W + P.pre_b + u0);
    const GruLayer gl = P.layer[0];
    F16Frag wi[3][4];
    {
      const uint4* aih = reinterpret_cast<const uint4*>(W + Q.a_ih16[0]) + lane;
#pragma unroll
      for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) wi[g][ks] = load_frag(aih + (g * 8 + wave) * OTS + ks * 128);
    }
    f32x4 bias[3];
    bias[0] = *reinterpret_cast<const f32x4*>(W + gl.b_ih + u0) + *reinterpret_cast<const f32x4*>(W + gl.b_hh + u0);
    bias[1] = *reinterpret_cast<const f32x4*>(W + gl.b_ih + H + u0) + *reinterpret_cast<const f32x4*>(W + gl.b_hh + H + u0);
    bias[2] = *reinterpret_cast<const f32x4*>(W + gl.b_ih + 2 * H + u0);
    const float ih_inv = Q.ih_inv_s[0];
    // (the handshake is asked for in front of the first store: a packed tile has its products to do first)
    bool near = false, near_known = false;
    auto ask_near = [&]() __attribute__((always_inline)) {
      if (near_known) return;
      const unsigned w = unsigned(__builtin_amdgcn_readfirstlane(int(where_early)));
      near = ((w & 8u) && (w >> 4) == (tag0 & 0x0fffffffu)) ? (w & 7u) == xcc : consumer_here();
      near_known = true;
      GP_STAMP(15, stage);
      GP_STAMP(14, near ? 100 + stage : 200 + stage);
    };

    int round = 0;
#pragma unroll 1
    for (int tile = slot; tile < tiles; tile += slots, ++round) {
      const int b0 = tile * spw, bend = min(B, b0 + spw), nb = bend - b0;
      const int gb0 = round * T;                                // number of this tile's step 0 in the slot's stream of steps
      const auto rs_g = __builtin_amdgcn_make_buffer_rsrc(WS.gi[0] + size_t(slot) * RING * GIS, 0, RING * GIS, 0x00020000);
      const bool packed = PK || nb <= 8;
      if (packed) {
        // TIME-PACKED tile (gru_f16.hip.h): the 16 MFMA columns are (step, stream) pairs; P for all steps into the workspace
        // (this workgroup's own stores and loads), then I0
        const int psh = nb <= 1 ? 0 : nb <= 2 ? 1 : nb <= 4 ? 2 : 3;
        const int TP = 16 >> psh;
        const int pcs = l15 & ((1 << psh) - 1), pdt = l15 >> psh;
        // (the planes between P and I0: in LDS when all T steps fit -- the streaming chunks --, else in the workspace)
        // (... and with them the columns' scales: read back from the workspace they cost the chunk a store's acknowledgement
        // plus a load's round trip between the two products)
        const bool in_lds = size_t(T) * SEQ <= size_t(kGruPipePlanes);
        char* const seq0 = in_lds ? gp_lds : WS.seq_in + size_t(slot) * T * SEQ;
        float* const sc = in_lds ? reinterpret_cast<float*>(gp_lds + kGruPipePlanes) : WS.sc + size_t(slot) * T * 16;
        for (int t0 = 0; t0 < T; t0 += TP) {
          const int t = t0 + pdt;
          const bool cv = t < T && pcs < nb;                    // this column exists
          gru_f32x8 xr[NKP];
          float ax = 0.f;
#pragma unroll
          for (int ks = 0; ks < NKP; ++ks) {
            xr[ks] = (round == 0 && t0 == 0) ? xfirst[ks] : load_x8(b0 + pcs, t, ks, cv);
#pragma unroll
            for (int j = 0; j < 8; ++j) ax = fmaxf(ax, fabsf(xr[ks][j]));
          }
          ax = fmaxf(ax, __shfl_xor(ax, 16));
          ax = fmaxf(ax, __shfl_xor(ax, 32));
          float cx, inv_s0;
          const float sx = pow2_scale(ax, &cx);
          const float s0 = pow2_scale(fmaf(Q.pre_alpha, ax, Q.pre_beta), &inv_s0);
          cx *= Q.pre_inv_s;
          if (wave == 0 && lq == 0 && cv) sc[t * 16 + pcs] = inv_s0;
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < NKP; ++ks) {
            const gru_f32x8 xs = xr[ks] * sx;
            const f16x8 bh = __builtin_convertvector(xs, f16x8);
            const f16x8 bl = __builtin_convertvector(xs - __builtin_convertvector(bh, gru_f32x8), f16x8);
            gru_mfma1(acc, a[ks], bh, bl);
          }
          f32x4 v = acc * cx + bpre;
          if (P.pre_relu) v = __builtin_elementwise_max(v, f32x4{0.f, 0.f, 0.f, 0.f});
          f16x4 vh, vl;
          gru_split4(v * s0, vh, vl);
          if (cv) {
            char* dst = seq0 + size_t(t) * SEQ + (((u0 >> 3) * MB + pcs) * 8 + (u0 & 7)) * 2;
            *reinterpret_cast<f16x4*>(dst) = vh;
            *reinterpret_cast<f16x4*>(dst + PH) = vl;
          }
        }
        __threadfence_block();
        __syncthreads();
        ask_near();
        for (int t0 = 0; t0 < T; t0 += TP) {
          const int t = t0 + pdt, tc = min(t, T - 1);
          const bool cv = t < T && pcs < nb;
          const char* p = seq0 + size_t(tc) * SEQ + (lq * MB + pcs) * 16;   // this column's fragment, K step 0
          f32x4 acc[3];
#pragma unroll
          for (int g = 0; g < 3; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            const f16x8 bh = *reinterpret_cast<const f16x8*>(p + ks * KSB);
            const f16x8 bl = *reinterpret_cast<const f16x8*>(p + PH + ks * KSB);
#pragma unroll
            for (int g = 0; g < 3; ++g) gru_mfma1(acc[g], wi[g][ks], bh, bl);
          }
          const float cin = sc[tc * 16 + pcs] * ih_inv;
          wait_credit(gb0 + min(t0 + TP, T) - 1);
          if (cv) {                                             // into the slot the recurrence's lane (stream pcs, same lq) reads
            const f32x4 v[3] = {acc[0] * cin + bias[0], acc[1] * cin + bias[1], acc[2] * cin + bias[2]};
            gp_st_gates(v, lap_tag(gb0 + t), rs_g, ring_pos(gb0 + t) * GIS + wave * 4096 + (lq * 16 + pcs) * 16, near);
          }
        }
      } else if constexpr (!PK) {
        ask_near();
        // CS steps at a time through LDS staging buffers (operand planes): while I0 multiplies chunk c, P makes chunk c + 1 in
        // the other buffer -- one barrier per chunk
        constexpr int CS = G::CS, CHUNK = CS * SEQ;
        // the features of chunk c + 1 are requested in front of I0(c): the requests are then OLDER than I0's stores in the
        // wave's in-order memory queue, and P(c + 1) does not wait for those stores' acknowledgements
        const int sx_ = b0 + l15;
        // NKP == 2 (the launcher: idim <= 64 in whole octets, 16-byte aligned rows): the requests are inline assembly with a
        // counted wait -- behind them the wave issues I0's 16 stores, and hipcc's own bookkeeping would wait for those too
        // (vmcnt(0) at the head of P).  Otherwise: plain loads.
        static_assert(CS == 4, "one preparing wave pair per step of a chunk");
        constexpr bool XF = NKP == 2;
        const float* xp0 = x + (int64_t(min(sx_, B - 1)) * T) * idim + lq * 8;               // K step 0: features 8 lq ..
        const float* xp1 = xp0 + ((32 + lq * 8 < idim) ? 32 : 0);                          // K step 1 (clamped when outside)
        const bool xv0 = sx_ < bend && lq * 8 < idim, xv1 = sx_ < bend && 32 + lq * 8 < idim;
        float inv_c[CS], inv_n[CS];
#pragma unroll
        for (int dt = 0; dt < CS; ++dt) inv_n[dt] = 1.f;
        // gi0 of step t0 + dt from the planes of chunk buffer `buf`
        auto i_step = [&](int t0, int dt, const char* buf) __attribute__((always_inline)) {
          const int t = t0 + dt;
          if (t < T) {
            f32x4 acc[3];
#pragma unroll
            for (int g = 0; g < 3; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
            const char* p = buf + dt * SEQ + frag;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
              const f16x8 bh = *reinterpret_cast<const f16x8*>(p + ks * KSB);
              const f16x8 bl = *reinterpret_cast<const f16x8*>(p + PH + ks * KSB);
#pragma unroll
              for (int g = 0; g < 3; ++g) gru_mfma1(acc[g], wi[g][ks], bh, bl);
            }
            const float cin = inv_c[dt] * ih_inv;
            const f32x4 v[3] = {acc[0] * cin + bias[0], acc[1] * cin + bias[1], acc[2] * cin + bias[2]};
            gp_st_gates(v, lap_tag(gb0 + t), rs_g, ring_pos(gb0 + t) * GIS + gvo, near);
          }
        };
        if constexpr (XF) {
          // ---- round 5: the feature operand is made ONCE per step, not by every wave.  Stamps at B = 1024 x 98 frames (round 5,
          // tools/probe/gru_stamps.py) showed THIS stage -- which has no recurrence and was meant to run ahead -- as the slowest
          // of the pipeline: 8.3 us per 4-step chunk = 2.07 us per step with everything downstream in lock-step behind it (the
          // recurrences need ~1.3).  Its 168 MFMAs per wave and chunk are 2.6 us; the rest was eight waves each loading the
          // WHOLE feature step (128 wave-loads per chunk next to the 128 gate stores, on a CU whose vector-memory path issues one
          // wave-instruction per ~30 .. 60 cycles), each taking the tile maximum, scaling, splitting and converting the same
          // B fragments (~100 vector instructions per wave and step, 8 x redundant).  Now wave w prepares K step w / 4 of step
          // w % 4 of the chunk AFTER the next -- both K steps loaded for the maximum (32 wave-loads per chunk), one converted --
          // into fragment-ordered planes in LDS (hi | lo, 1 KiB each: a fragment is one ds_read_b128 at lane x 16) together
          // with the step's three scale constants; P then reads fragments and constants like I0 reads its planes.  Same values in
          // the same lanes, same instructions on them: bit-identical.  LDS: 64 KB of in0 planes (as before) + 32 KB + 128 B.
          constexpr int XST = 4096;                               // bytes of one step of feature planes: [K step][hi | lo][lane][16]
          char* const xpl = gp_lds + 2 * CHUNK;                   // [2 chunks][CS steps][XST]
          float* const xsc = reinterpret_cast<float*>(xpl + 2 * CS * XST);   // [2][CS][4]: cx, s0, 1 / s0, -
          static_assert(2 * CHUNK + 2 * CS * XST + 2 * CS * 16 <= kGruPipeLds, "feature planes behind the staging buffers");
          const int pdt = wave & 3;
          const bool pk1 = wave >= 4;                             // this wave converts K step 1 (else 0) of step pdt
          f32x4 xq1[4];
          auto x_req = [&](int t) __attribute__((always_inline)) {
            const int64_t to = int64_t(min(t, T - 1)) * idim;
            gp_ldx4(xq1, xp0 + to, xp1 + to);
          };
          // the registers hold step t0c + pdt: maximum over the tile, scales, this wave's K step as fragments -> planes `xb`
          auto x_prep = [&](int t0c, int xb) __attribute__((always_inline)) {
            const int t = t0c + pdt;
            if (t < T) {
              const gru_f32x8 z8 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
              // (a NaN / Inf feature enters as 0: nonfinite.hip.h)
              const gru_f32x8 k0 = xv0 ? nf_clean_vec<gru_f32x8, 8>(gru_f32x8{xq1[0][0], xq1[0][1], xq1[0][2], xq1[0][3], xq1[1][0], xq1[1][1], xq1[1][2], xq1[1][3]}) : z8;
              const gru_f32x8 k1 = xv1 ? nf_clean_vec<gru_f32x8, 8>(gru_f32x8{xq1[2][0], xq1[2][1], xq1[2][2], xq1[2][3], xq1[3][0], xq1[3][1], xq1[3][2], xq1[3][3]}) : z8;
              float ax = 0.f;                                     // max|x[t]| over the tile
#pragma unroll
              for (int j = 0; j < 8; ++j) ax = fmaxf(ax, fmaxf(fabsf(k0[j]), fabsf(k1[j])));
              ax = __uint_as_float(unsigned(__builtin_amdgcn_readlane(int(wave_umax63(__float_as_uint(ax))), 63)));
              float cx, inv_s0;
              const float sx = pow2_scale(ax, &cx);
              const float s0 = pow2_scale(fmaf(Q.pre_alpha, ax, Q.pre_beta), &inv_s0);
              cx *= Q.pre_inv_s;
              const gru_f32x8 xs = (pk1 ? k1 : k0) * sx;
              const f16x8 bh = __builtin_convertvector(xs, f16x8);
              const f16x8 bl = __builtin_convertvector(xs - __builtin_convertvector(bh, gru_f32x8), f16x8);
              char* dst = xpl + (xb * CS + pdt) * XST + (pk1 ? 2048 : 0) + lane * 16;
              *reinterpret_cast<f16x8*>(dst) = bh;
              *reinterpret_cast<f16x8*>(dst + 1024) = bl;
              if (!pk1 && lane == 0) *reinterpret_cast<f32x4*>(xsc + (xb * CS + pdt) * 4) = f32x4{cx, s0, inv_s0, 0.f};
            }
          };
          // in0[t0 + dt] = [ReLU](Wpre x + b) from feature planes `xb`, scaled, split, into the in0 planes `buf`
          auto p_step = [&](int t0, int dt, char* buf, int xb, float (&inv)[CS]) __attribute__((always_inline)) {
            const int t = t0 + dt;
            inv[dt] = 1.f;
            if (t < T) {
              const f32x4 sc = *reinterpret_cast<const f32x4*>(xsc + (xb * CS + dt) * 4);
              const char* xf = xpl + (xb * CS + dt) * XST + lane * 16;
              f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
              for (int ks = 0; ks < NKP; ++ks) {
                const f16x8 bh = *reinterpret_cast<const f16x8*>(xf + ks * 2048);
                const f16x8 bl = *reinterpret_cast<const f16x8*>(xf + ks * 2048 + 1024);
                gru_mfma1(acc, a[ks], bh, bl);
              }
              f32x4 v = acc * sc[0] + bpre;
              if (P.pre_relu) v = __builtin_elementwise_max(v, f32x4{0.f, 0.f, 0.f, 0.f});
              f16x4 vh, vl;
              gru_split4(v * sc[1], vh, vl);
              char* dst = buf + dt * SEQ + wr_off;
              *reinterpret_cast<f16x4*>(dst) = vh;
              *reinterpret_cast<f16x4*>(dst + PH) = vl;
              inv[dt] = sc[2];
            }
          };
          // prologue: feature planes of chunks 0 and 1, in0 planes of chunk 0
          x_req(pdt);
          gp_waitx4_sel<0>(xq1, 0u);
          x_prep(0, 0);
          if (CS < T) x_req(CS + pdt);
          gp_barrier();                                           // feature planes of chunk 0
#pragma unroll
          for (int dt = 0; dt < CS; ++dt) p_step(0, dt, gp_lds, 0, inv_c);
          if (CS < T) {
            gp_waitx4_sel<0>(xq1, 0u);
            x_prep(CS, 1);
            if (2 * CS < T) x_req(2 * CS + pdt);
          }
          gp_barrier();                                           // in0 planes of chunk 0, feature planes of chunk 1
          for (int t0 = 0, c = 0; t0 < T; t0 += CS, ++c) {
            const char* const buf = gp_lds + (c & 1) * CHUNK;
            char* const nbuf = gp_lds + ((c + 1) & 1) * CHUNK;
            const bool more = t0 + CS < T;                        // (then this chunk is a full one: all its stores are issued)
            GP_STAMP(1, t0);
            wait_credit(gb0 + min(t0 + CS, T) - 1);               // (a poll's loads only add to what is behind the requests)
            // Waves w and w + 4 share a SIMD: one of the two does its products first and its preprocessing second, the other
            // the other way round -- P is vector work, I0 matrix work and stores.
            const bool i_first = wave >= 4;
            if (i_first) {
#pragma unroll
              for (int dt = 0; dt < CS; ++dt) i_step(t0, dt, buf);
            }
            if (more) {
#pragma unroll
              for (int dt = 0; dt < CS; ++dt) p_step(t0 + CS, dt, nbuf, (c + 1) & 1, inv_n);
              // the features of chunk c + 2 were requested one iteration ago: behind the request this wave has issued the 16
              // gate stores of one full chunk (I0 of chunk c - 1 or c, whichever comes first in this wave's order) and nothing
              // else, whatever the order -- vmcnt(16) is exact; the first iteration's request comes from the prologue with
              // fewer behind it and waits for everything.  Their planes are the ones P(c) read one iteration ago.
              if (t0 + 2 * CS < T) {
                gp_waitx4_sel<16>(xq1, c > 0 ? 1u : 0u);
                x_prep(t0 + 2 * CS, c & 1);
                if (t0 + 3 * CS < T) x_req(t0 + 3 * CS + pdt);
              }
            }
            if (!i_first) {
#pragma unroll
              for (int dt = 0; dt < CS; ++dt) i_step(t0, dt, buf);
            }
            GP_STAMP(2, t0);
#pragma unroll
            for (int dt = 0; dt < CS; ++dt) inv_c[dt] = inv_n[dt];
            gp_barrier();
          }
          gp_waitx4_sel<0>(xq1, 0u);                              // (nothing may still be landing when the registers are re-used)
        } else {
          // any other feature layout: every wave loads and converts the step itself (plain loads, the compiler's own waits)
          gru_f32x8 xr[CS][NKP];
          auto x_chunk = [&](int t0) __attribute__((always_inline)) {
#pragma unroll
            for (int dt = 0; dt < CS; ++dt)
#pragma unroll
              for (int ks = 0; ks < NKP; ++ks) xr[dt][ks] = load_x8(sx_, t0 + dt, ks, sx_ < bend && t0 + dt < T);
          };
          auto p_step = [&](int t0, int dt, char* buf, float (&inv)[CS]) __attribute__((always_inline)) {
            const int t = t0 + dt;
            inv[dt] = 1.f;
            if (t < T) {
              float ax = 0.f;                                   // max|x[t]| over the tile: every wave holds the whole step
#pragma unroll
              for (int ks = 0; ks < NKP; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) ax = fmaxf(ax, fabsf(xr[dt][ks][j]));
              ax = __uint_as_float(unsigned(__builtin_amdgcn_readlane(int(wave_umax63(__float_as_uint(ax))), 63)));
              float cx, inv_s0;
              const float sx = pow2_scale(ax, &cx);
              const float s0 = pow2_scale(fmaf(Q.pre_alpha, ax, Q.pre_beta), &inv_s0);
              cx *= Q.pre_inv_s;
              inv[dt] = inv_s0;
              f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
              for (int ks = 0; ks < NKP; ++ks) {
                const gru_f32x8 xs = xr[dt][ks] * sx;
                const f16x8 bh = __builtin_convertvector(xs, f16x8);
                const f16x8 bl = __builtin_convertvector(xs - __builtin_convertvector(bh, gru_f32x8), f16x8);
                gru_mfma1(acc, a[ks], bh, bl);
              }
              f32x4 v = acc * cx + bpre;
              if (P.pre_relu) v = __builtin_elementwise_max(v, f32x4{0.f, 0.f, 0.f, 0.f});
              f16x4 vh, vl;
              gru_split4(v * s0, vh, vl);
              char* dst = buf + dt * SEQ + wr_off;
              *reinterpret_cast<f16x4*>(dst) = vh;
              *reinterpret_cast<f16x4*>(dst + PH) = vl;
            }
          };
          x_chunk(0);
#pragma unroll
          for (int dt = 0; dt < CS; ++dt) p_step(0, dt, gp_lds, inv_c);
          gp_barrier();
          for (int t0 = 0, c = 0; t0 < T; t0 += CS, ++c) {
            const char* const buf = gp_lds + (c & 1) * CHUNK;
            GP_STAMP(1, t0);
            wait_credit(gb0 + min(t0 + CS, T) - 1);
            if (t0 + CS < T) x_chunk(t0 + CS);
#pragma unroll
            for (int dt = 0; dt < CS; ++dt) i_step(t0, dt, buf);
            GP_STAMP(2, t0);
            if (t0 + CS < T) {
#pragma unroll
              for (int dt = 0; dt < CS; ++dt) p_step(t0 + CS, dt, gp_lds + ((c + 1) & 1) * CHUNK, inv_n);
            }
#pragma unroll
            for (int dt = 0; dt < CS; ++dt) inv_c[dt] = inv_n[dt];
            gp_barrier();
          }
        }
      }
      __syncthreads();                                        // (staging buffers free for the next tile)
    }
    return;
  }

  const int l = stage >> 1;
  if (stage & 1) {
    // ============================ stage R_l: the recurrence of layer l (last layer: + head) ============================
    const GruLayer gl = P.layer[l];
    const bool last = l == L - 1;
    F16Frag wh[3][4];
    {
      const uint4* ahh = reinterpret_cast<const uint4*>(W + Q.a_hh16[l]) + lane;
#pragma unroll
      for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) wh[g][ks] = load_frag(ahh + (g * 8 + wave) * OTS + ks * 128);
    }
    const f32x4 b_hn = *reinterpret_cast<const f32x4*>(W + gl.b_hh + 2 * H + u0);
    const bool near = !last && consumer_here();
    const bool near_up = peer_here(stage - 1);                // (credits go the other way: to the producer of this stage's input)
    // (Two waves share a SIMD, and a step is 36 MFMAs followed by ~25 transcendental and ~60 plain VALU instructions per
    // wave; a priority for one of the two -- its cell math under the other's products -- measured nothing.)
    // Last layer with a keyword-sized head (K <= 16: one o-tile): y(t - 1) = [sigmoid](Wc h(t - 1) + bc) needs exactly the B
    // fragments of h(t - 1) every wave reads for its recurrent product anyway -- wave t mod 8 adds the head's 12 MFMAs to
    // step t.  No separate head pass, no copy of the sequence in global memory, no trip to L2 behind the last step.
    const bool head_in = last && K <= 16;
    F16Frag ahd[4];
    f32x4 bc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) ahd[ks].h = ahd[ks].l = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
    if (head_in) {
      const uint4* ap = reinterpret_cast<const uint4*>(W + Q.head_a16) + lane;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) ahd[ks] = load_frag(ap + ks * 128);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (lq * 4 + r < K) bc[r] = W[P.head_b + lq * 4 + r];
    }
    GP_STAMP(15, stage);
    GP_STAMP(14, near ? 100 + stage : 200 + stage);

    int round = 0;
#pragma unroll 1
    for (int tile = slot; tile < tiles; tile += slots, ++round) {
      const int b0 = tile * spw, bend = min(B, b0 + spw), nb = bend - b0;
      const int gb0 = round * T;
      const float hb = h_bound(l, b0, bend);
      float chh;
      const float shl = pow2_scale(hb, &chh);
      chh *= Q.hh_inv_s[l];
      char* const sout = WS.seq_top + size_t(slot) * T * SEQ;
      const auto rs_o = __builtin_amdgcn_make_buffer_rsrc(WS.hs[last ? 0 : l] + size_t(last ? 0 : slot) * RING * HSS, 0, last ? 0 : RING * HSS, 0x00020000);
      float chd;
      (void)pow2_scale(hb, &chd);
      chd *= Q.head_inv_s;
      // y of one step from its accumulator (this lane: outputs 4 lq .. 4 lq + 3 of stream l15)
      auto put_y = [&](const f32x4& acc, int t) __attribute__((always_inline)) {
        const int s = b0 + l15;
        if (s < bend) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (lq * 4 + r < K) {
              float v = fmaf(acc[r], chd, bc[r]);
              if (P.sigmoid) v = sigmoidf_(v);
              y[(int64_t(s) * T + t) * K + lq * 4 + r] = v;
            }
        }
      };
      f32x4 hreg;
      {
        const int s = b0 + l15;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (h0 && s < bend) v = nf_clean_vec<f32x4, 4>(*reinterpret_cast<const f32x4*>(h0 + (int64_t(l) * B + s) * H + u0));
        hreg = v;
        f16x4 vh, vl;
        gru_split4(v * shl, vh, vl);
        char* dst = gp_lds + wr_off;
        *reinterpret_cast<f16x4*>(dst) = vh;
        *reinterpret_cast<f16x4*>(dst + PH) = vl;
      }
      // gate pre-activations of this lane as granule pairs [gate][half], requested two steps ahead into two buffers that
      // take turns (no copies: whatever is requested behind this wave's stores returns behind their acknowledgement)
      gp_u32x4 ga[4], gb[4];
      const gp_desc ds_g = gp_make_desc(WS.gi[l] + size_t(slot) * RING * GIS, unsigned(RING) * GIS);
      auto load_g = [&](gp_u32x4 (&gg)[4], int t) __attribute__((always_inline)) { gp_ld4(gg, ring_pos(gb0 + min(t, T - 1)) * GIS + gvo, ds_g); };
      auto reload_g = [&](gp_u32x4 (&gg)[4], int t) __attribute__((always_inline)) { gp_ld4<true>(gg, ring_pos(gb0 + min(t, T - 1)) * GIS + gvo, ds_g); };
      // a time-packed first stage writes the columns of real streams only: the other lanes' granules never arrive
      const bool live = l > 0 || nb > 8 || l15 < nb;
      auto tags_ok = [&](const gp_u32x4 (&gg)[4], unsigned tag) __attribute__((always_inline)) -> bool {
        const unsigned d = (gg[0][3] ^ tag) | (gg[1][3] ^ tag) | (gg[2][3] ^ tag) | (gg[3][3] ^ tag);
        return !__builtin_amdgcn_ballot_w64(live && d != 0u);    // all four granules carry this step's tag
      };
      load_g(ga, 0);
      load_g(gb, 1);
      __syncthreads();
      unsigned spins = 0;
      // The LAST layer's gate values come from a stage that is barely one step ahead when a single chunk is being served
      // (B = 1, T = 10: the request two steps ahead returned the previous launch's granules on EVERY step, and the second
      // request cost the step its round trip behind the products: 1.48 us per step against 1.16 in layer 0).  It therefore
      // looks at the tags BEFORE the products and, if they are old, asks again at once -- the round trip runs under the
      // MFMAs.  Other layers keep the late look: their state stores sit in front of the requests in vmcnt's order, and an
      // early wait would expose the acknowledgement the products hide.
      const unsigned early = last ? 1u : 0u;
      auto step = [&](int t, gp_u32x4 (&g0)[4]) __attribute__((always_inline)) {
        GP_STAMP(4 * l + 3, t);
        const char* hbp = gp_lds + (t & 1) * 2 * PH + frag;           // image of h(t-1); requested in front of the look
        f16x8 hh[4], hl[4];                                           // (K steps 0 and 1: registers for more there are not)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          hh[ks] = *reinterpret_cast<const f16x8*>(hbp + ks * KSB);
          hl[ks] = *reinterpret_cast<const f16x8*>(hbp + PH + ks * KSB);
        }
        const unsigned tag = lap_tag(gb0 + t);
        const int g_off = ring_pos(gb0 + t) * GIS + gvo;
        gp_wait4_if<4>(g0, early);
        const unsigned again = (early && !tags_ok(g0, tag)) ? 1u : 0u;
        gp_ld4_if(g0, g_off, ds_g, again);
        char* const hw = gp_lds + ((t + 1) & 1) * 2 * PH;             // image of h(t)
        f32x4 acc[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
        const bool my_head = head_in && t > 0 && (t & 7) == wave;     // (wave-uniform)
        f32x4 acch = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 2; ks < 4; ++ks) {
          hh[ks] = *reinterpret_cast<const f16x8*>(hbp + ks * KSB);
          hl[ks] = *reinterpret_cast<const f16x8*>(hbp + PH + ks * KSB);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
          for (int g = 0; g < 3; ++g) gru_mfma1(acc[g], wh[g][ks], hh[ks], hl[ks]);
          if (my_head) gru_mfma1(acch, ahd[ks], hh[ks], hl[ks]);
        }
        if (my_head) put_y(acch, t - 1);
        GP_STAMP(4 * l + 4, t);
        // This step's gate values were requested at the end of step t - 2; behind them this wave has issued the stores of
        // step t - 1 and four more requests (step t + 1; past the end, the last step again) -- which alone may still be out.
        // ONE wait statement for every step: two (vmcnt(6) / vmcnt(0) in the arms of a branch) made the compiler COPY the
        // registers -- before the wait, i.e. before the data had landed -- on the last step (wrong h_n in one tile in ~20).
        gp_wait4_sel<4>(g0, again);
        // are they there?  (upstream runs ahead: normally yes)
        if (!tags_ok(g0, tag)) {
          // about to wait for the producer: it may be waiting for THIS stage's credit (a producer that writes a ring's worth of
          // steps at once -- the time-packed first stage -- needs every step before this one acknowledged)
          publish(gb0 + t, near_up);
          spins = 0;
          do {
            __builtin_amdgcn_s_sleep(GP_DATA_SLEEP);
            reload_g(g0, t);
            gp_wait4<0>(g0);
            if (give_up(spins, 0x100u)) break;
          } while (!tags_ok(g0, tag));
        }
        GP_STAMP(4 * l + 5, t);
        // every wave is through with the steps before this one (the barrier that ended step t - 1): tell the producer every
        // fourth step.  Here -- in front of the cell math -- the store is older than the next request, so the counted waits
        // stay exact, and it is acknowledged long before the next of them.
        if ((t & 3) == 0 && t > 0) publish(gb0 + t, near_up);
        // cell math, register-local (PyTorch formulation, gate order r, z, n)
        f32x4 hv;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float gr = __uint_as_float(g0[r / 3][r % 3]);                 // value i = 4 gate + r sits in granule i / 3
          const float gz = __uint_as_float(g0[(4 + r) / 3][(4 + r) % 3]);
          const float gn = __uint_as_float(g0[(8 + r) / 3][(8 + r) % 3]);
          const float rg = gru_sigmoid(fmaf(acc[0][r], chh, gr));
          const float zg = gru_sigmoid(fmaf(acc[1][r], chh, gz));
          const float ng = gru_tanh(gn + rg * fmaf(acc[2][r], chh, b_hn[r]));
          hv[r] = ng + zg * (hreg[r] - ng);                     // (1 - z) n + z h
        }
        hreg = hv;
        f16x4 vh, vl;
        gru_split4(hv * shl, vh, vl);
        {
          char* dst = hw + wr_off;
          *reinterpret_cast<f16x4*>(dst) = vh;
          *reinterpret_cast<f16x4*>(dst + PH) = vl;
        }
        if (last && !head_in) {                                 // read back by this workgroup's own head pass
          char* gd = sout + size_t(t) * SEQ + wr_off;
          *reinterpret_cast<f16x4*>(gd) = vh;
          *reinterpret_cast<f16x4*>(gd + PH) = vl;
        } else if (!last) {                                     // four state granules {hi | lo << 16, tag}
          const gp_u32x4 hl4 = __builtin_bit_cast(gp_u32x4, __builtin_shufflevector(vh, vl, 0, 1, 2, 3, 4, 5, 6, 7));
          const unsigned p0 = __builtin_amdgcn_perm(hl4[2], hl4[0], 0x05040100u), p1 = __builtin_amdgcn_perm(hl4[2], hl4[0], 0x07060302u);
          const unsigned p2 = __builtin_amdgcn_perm(hl4[3], hl4[1], 0x05040100u), p3 = __builtin_amdgcn_perm(hl4[3], hl4[1], 0x07060302u);
          const int vo = ring_pos(gb0 + t) * HSS + (((u0 >> 3) * 16 + l15) * 8 + (u0 & 7)) * 8;
          wait_credit(gb0 + t);
          gp_st16(gp_pair(p0, p1, tag), rs_o, vo, near);
          gp_st16(gp_pair(p2, p3, tag), rs_o, vo + 16, near);
        }
        load_g(g0, t + 2);   // (clamped to the last step: every step has four requests behind its own)
        gp_barrier();      // h(t) is complete and every wave is done with h(t-1)
      };
      for (int t = 0; t < T; t += 2) {
        step(t, ga);
        if (t + 1 < T) step(t + 1, gb);
      }
      gp_wait4<0>(ga);     // nothing of this tile may still be on its way into registers
      gp_wait4<0>(gb);
      if (head_in && wave == (T & 7)) {                        // y(T - 1) from the image the last step left behind its barrier
        const char* hbp = gp_lds + (T & 1) * 2 * PH + frag;
        f32x4 acch = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const f16x8 hh = *reinterpret_cast<const f16x8*>(hbp + ks * KSB);
          const f16x8 hl = *reinterpret_cast<const f16x8*>(hbp + PH + ks * KSB);
          gru_mfma1(acch, ahd[ks], hh, hl);
        }
        put_y(acch, T - 1);
      }
      if (hn) {
        const int s = b0 + l15;
        if (s < bend) *reinterpret_cast<f32x4*>(hn + (int64_t(l) * B + s) * H + u0) = hreg;
      }
      publish(gb0 + T, near_up);                                // every gate value of this tile has been read
      if (last && !head_in) {
        // ================= head: y[t] = [sigmoid](Wc h_top[t] + bc), waves take steps round-robin =================
        __threadfence_block();
        __syncthreads();
        const bool packed = nb <= 8;
        const int psh = nb <= 1 ? 0 : nb <= 2 ? 1 : nb <= 4 ? 2 : 3;
        const int TP = 16 >> psh;
        const int pcs = l15 & ((1 << psh) - 1), pdt = l15 >> psh;
        const int head_tiles = (K + 15) / 16;
        for (int ot = 0; ot < head_tiles; ++ot) {
          const uint4* ahd = reinterpret_cast<const uint4*>(W + Q.head_a16) + size_t(ot) * OTS + lane;
          F16Frag a[4];
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) a[ks] = load_frag(ahd + ks * 128);
          const int k0 = ot * 16 + lq * 4;
          f32x4 bc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (k0 + r < K) bc[r] = W[P.head_b + k0 + r];
          if (packed) {
            for (int t0 = wave * TP; t0 < T; t0 += (kThreads / 64) * TP) {
              const int t = t0 + pdt, tc = min(t, T - 1);
              const bool cv = t < T && pcs < nb;
              const char* p = sout + size_t(tc) * SEQ + (lq * MB + pcs) * 16;
              f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
              for (int ks = 0; ks < 4; ++ks) {
                const f16x8 bh = *reinterpret_cast<const f16x8*>(p + ks * KSB);
                const f16x8 bl = *reinterpret_cast<const f16x8*>(p + PH + ks * KSB);
                gru_mfma1(acc, a[ks], bh, bl);
              }
              if (cv) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                  if (k0 + r < K) {
                    float v = fmaf(acc[r], chd, bc[r]);
                    if (P.sigmoid) v = sigmoidf_(v);
                    y[(int64_t(b0 + pcs) * T + t) * K + k0 + r] = v;
                  }
              }
            }
            continue;
          }
          for (int t = wave; t < T; t += kThreads / 64) {
            const char* p = sout + size_t(t) * SEQ + frag;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
              const f16x8 bh = *reinterpret_cast<const f16x8*>(p + ks * KSB);
              const f16x8 bl = *reinterpret_cast<const f16x8*>(p + PH + ks * KSB);
              gru_mfma1(acc, a[ks], bh, bl);
            }
            const int s = b0 + l15;
            if (s < bend) {
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (k0 + r < K) {
                  float v = fmaf(acc[r], chd, bc[r]);
                  if (P.sigmoid) v = sigmoidf_(v);
                  y[(int64_t(s) * T + t) * K + k0 + r] = v;
                }
            }
          }
        }
      }
      // (round 6) this stage's outputs of the round -- h_n, and y from the last layer -- are written: say so for the slot's
      // non-finite workgroup.  Every thread's stores have been acknowledged (vmcnt(0)) before the barrier, the word follows it;
      // the reader sits on this XCD, behind the same L2.  One store per round; nobody here waits for anything.
      __builtin_amdgcn_s_waitcnt(0x0F70);
      __syncthreads();
      if (WS.nf && tid == 0 && !dead) __hip_atomic_store(done_w + stage, round_tag(round), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else {
    // ===================== stage I_l (l >= 1): gi_l[t] = W_ih,l h_{l-1}[t] + b, every wave on its own =====================
    const GruLayer gl = P.layer[l];
    F16Frag wi[3][4];
    {
      const uint4* aih = reinterpret_cast<const uint4*>(W + Q.a_ih16[l]) + lane;
#pragma unroll
      for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) wi[g][ks] = load_frag(aih + (g * 8 + wave) * OTS + ks * 128);
    }
    f32x4 bias[3];
    bias[0] = *reinterpret_cast<const f32x4*>(W + gl.b_ih + u0) + *reinterpret_cast<const f32x4*>(W + gl.b_hh + u0);
    bias[1] = *reinterpret_cast<const f32x4*>(W + gl.b_ih + H + u0) + *reinterpret_cast<const f32x4*>(W + gl.b_hh + H + u0);
    bias[2] = *reinterpret_cast<const f32x4*>(W + gl.b_ih + 2 * H + u0);
    const bool near = consumer_here();
    const bool near_up = peer_here(stage - 1);
    GP_STAMP(15, stage);
    GP_STAMP(14, near ? 100 + stage : 200 + stage);

    int round = 0;
#pragma unroll 1
    for (int tile = slot; tile < tiles; tile += slots, ++round) {
      const int b0 = tile * spw, bend = min(B, b0 + spw);
      const int gb0 = round * T;
      float inv_in;
      (void)pow2_scale(h_bound(l - 1, b0, bend), &inv_in);    // scale of the previous layer's planes
      const float cin = inv_in * Q.ih_inv_s[l];
      const auto rs_g = __builtin_amdgcn_make_buffer_rsrc(WS.gi[l] + size_t(slot) * RING * GIS, 0, RING * GIS, 0x00020000);
      // One step of the previous layer's state is 16 KB of granules; EVERY wave needs all of it as its B operand.  Eight waves
      // pulling it through the CU's 64 B/clk vector-memory path was 128 KB per step (measured: ~0.9 us of the stage's step), so
      // each wave fetches ONE EIGHTH -- k-octets 2 w and 2 w + 1: lane = (octet, stream, half), four granules = 32 bytes --,
      // checks its tags, and writes the values as fp16 hi / lo operand planes into LDS (two buffers taking turns, the layout
      // of the recurrence's own state image); one barrier per step, then every wave reads its fragments from LDS.
      const int po = 2 * wave + (lane >> 5), ps = (lane >> 1) & 15, phf = lane & 1;
      const int pvo = (po * 16 + ps) * 64 + phf * 32;         // this lane's 32 bytes inside a step of granules
      const int pwr = ((po * MB + ps) * 8 + phf * 4) * 2;     // ... and its 4 halves inside a plane
      gp_u32x4 raw[2];
      const gp_desc ds_i = gp_make_desc(WS.hs[l - 1] + size_t(slot) * RING * HSS, unsigned(RING) * HSS);
      auto load_h = [&](int t) __attribute__((always_inline)) { gp_ld2(raw, ring_pos(gb0 + t) * HSS + pvo, ds_i); };
      auto reload_h = [&](int t) __attribute__((always_inline)) { gp_ld2<true>(raw, ring_pos(gb0 + t) * HSS + pvo, ds_i); };
      unsigned spins = 0;
      // waits until this wave's share of step t is there, then writes it into plane buffer `buf`; `behind` = this wave has
      // issued the step's four gate stores behind the request (they may stay out)
      auto take = [&](int t, char* buf, bool behind) __attribute__((always_inline)) {
        const unsigned tag = lap_tag(gb0 + t);
        if (behind) gp_wait2<4>(raw);
        else gp_wait2<0>(raw);
        auto stale = [&]() __attribute__((always_inline)) -> bool {
          const unsigned d = (raw[0][1] ^ tag) | (raw[0][3] ^ tag) | (raw[1][1] ^ tag) | (raw[1][3] ^ tag);
          return __builtin_amdgcn_ballot_w64(d != 0u) != 0;
        };
        if (stale()) {
          publish(gb0 + t, near_up);                            // (as in the recurrence: never wait without having said so)
          spins = 0;
          do {
            __builtin_amdgcn_s_sleep(GP_DATA_SLEEP);
            reload_h(t);
            gp_wait2<0>(raw);
            if (give_up(spins, 0x100u)) break;
          } while (stale());
        }
        gp_u32x4 hl;                                            // granule data: hi | lo << 16 of one unit
        hl[0] = __builtin_amdgcn_perm(raw[0][2], raw[0][0], 0x05040100u);
        hl[1] = __builtin_amdgcn_perm(raw[1][2], raw[1][0], 0x05040100u);
        hl[2] = __builtin_amdgcn_perm(raw[0][2], raw[0][0], 0x07060302u);
        hl[3] = __builtin_amdgcn_perm(raw[1][2], raw[1][0], 0x07060302u);
        *reinterpret_cast<unsigned long long*>(buf + pwr) = (unsigned long long)hl[0] | ((unsigned long long)hl[1] << 32);
        *reinterpret_cast<unsigned long long*>(buf + PH + pwr) = (unsigned long long)hl[2] | ((unsigned long long)hl[3] << 32);
      };
      load_h(0);
      take(0, gp_lds, false);
      if (T > 1) load_h(1);
      __syncthreads();
      for (int t = 0; t < T; ++t) {
        GP_STAMP(11, t);
        // every wave has taken the steps <= t (the barrier that ended the last iteration): tell the producer every fourth
        // step -- here, so that the store's acknowledgement comes back under the products
        if ((t & 3) == 3) publish(gb0 + t + 1, near_up);
        const char* p = gp_lds + (t & 1) * 2 * PH + frag;
        f32x4 acc[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const f16x8 bh = *reinterpret_cast<const f16x8*>(p + ks * KSB);
          const f16x8 bl = *reinterpret_cast<const f16x8*>(p + PH + ks * KSB);
#pragma unroll
          for (int g = 0; g < 3; ++g) gru_mfma1(acc[g], wi[g][ks], bh, bl);
        }
        GP_STAMP(12, t);
        const f32x4 v[3] = {acc[0] * cin + bias[0], acc[1] * cin + bias[1], acc[2] * cin + bias[2]};
        wait_credit(gb0 + t);
        gp_st_gates(v, lap_tag(gb0 + t), rs_g, ring_pos(gb0 + t) * GIS + gvo, near);
        if (t + 1 < T) {
          take(t + 1, gp_lds + ((t + 1) & 1) * 2 * PH, true);
          if (t + 2 < T) load_h(t + 2);                       // in flight behind the next step's products
        }
        GP_STAMP(13, t);
        gp_barrier();                                         // the planes of step t + 1 are complete; those of step t are free
      }
      __syncthreads();                                        // every wave has read the whole tile
      publish(gb0 + T, near_up);
    }
  }
}

// ---- geometry of one call: stream slots per workgroup, tiles, resident slots ----
struct GruPipeGeom {
  int stages, spw, tiles, slots, slots_p;
};
inline bool gru_pipe_geom(int nlayers, int B, int T, int cus, GruPipeGeom* g) {
  g->stages = 2 * nlayers;
  int smax = cus / g->stages;
  smax = smax > kGruPipeMaxSlots ? kGruPipeMaxSlots : smax;
  if (smax < 1 || nlayers > kGruMaxLayers) return false;
  // streaming chunks (T <= 16): fewer streams per tile while every tile still gets its own slot -- a workgroup's time does
  // not depend on how many of its 16 MFMA columns are real, and tiles of <= 8 streams run the first stage time-packed (all
  // steps in one or two MFMA tiles).  Longer inputs: full tiles -- a time-packed first stage makes ALL steps before the
  // recurrence sees the first one (measured at B = 256 x 98 frames: 0.67x of the layer-major kernels)
  int spw = T <= 16 ? 1 : 16;
  while (spw < 16 && (B + spw - 1) / spw > smax) spw *= 2;
  g->spw = spw;
  g->tiles = (B + spw - 1) / spw;
  g->slots = g->tiles < smax ? g->tiles : smax;
  g->slots_p = (g->slots + 7) / 8 * 8;                       // block b runs on XCD b % 8: a slot's stages share an XCD
  return true;
}
// bytes of one call: the plain workspace behind the control words (seq_in, seq_top, sc: each per slot) and the granule
// workspace (gi per layer, state granules per layer below the top: a ring per slot each)
struct GruPipeBytes {
  size_t seq, sc, gi, hs;
  size_t plain() const { return 2 * seq + sc; }
  size_t granules(int nlayers) const { return size_t(nlayers) * gi + size_t(nlayers - 1) * hs; }
};
inline bool gru_pipe_bytes(int nlayers, int B, int T, int cus, GruPipeBytes* b) {
  GruPipeGeom g;
  if (!gru_pipe_geom(nlayers, B, T, cus, &g)) return false;
  auto al = [](size_t v) { return (v + 255) / 256 * 256; };
  b->seq = al(size_t(g.slots) * T * GruF16Geom<1>::SEQ_STEP);
  b->sc = al(size_t(g.slots) * T * 16 * sizeof(float));
  b->gi = al(size_t(g.slots) * kGruPipeRing * kGruPipeGiStep);
  b->hs = al(size_t(g.slots) * kGruPipeRing * kGruPipeHStep);
  return true;
}
// (a slot's steps of one launch are numbered in an int)
inline bool gru_pipe_supported(const GruF16Params& Q, int T) {
  return gru_f16_supported(Q) && T < (1 << 24);
}

inline int launch_gru_pipe(const GruF16Params& Q, const GruPipeWorkspace& ws, const float* x, int B, int T, const float* h0,
                           float* y, float* hn, int cus, hipStream_t stream) {
  if (!gru_pipe_supported(Q, T)) return -4;
  GruPipeGeom g;
  if (!gru_pipe_geom(Q.base.nlayers, B, T, cus, &g)) return -4;
  using G = GruF16Geom<1>;
  static DynLdsGrant grant[4];
  // <2>: at most two K steps of features in whole, 16-byte aligned octets (the 40-d / 64-d front ends); <4>: anything else
  const bool k2 = Q.kpre16 <= 64 && Q.base.idim % 8 == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0;
  const bool pk = g.spw <= 8;
  auto kern = k2 ? (pk ? gru_pipe_kernel<2, true> : gru_pipe_kernel<2, false>) : (pk ? gru_pipe_kernel<4, true> : gru_pipe_kernel<4, false>);
  static_assert(kGruPipeLds >= int(G::LDS_BYTES), "staging buffers");
  if (grant_dynamic_lds(kern, kGruPipeLds, grant[(k2 ? 0 : 2) + (pk ? 1 : 0)])) return -3;
  // (+ one non-finite workgroup per slot behind the stage workgroups when the caller handed a context over: gru_pipe_kernel)
  hipLaunchKernelGGL(kern, dim3(g.stages * g.slots_p + (ws.nf ? g.slots : 0)), dim3(kThreads), kGruPipeLds, stream, Q, ws, x, B, T, h0,
                     y, hn, g.tiles, g.slots, g.slots_p, g.spw);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace wekws
