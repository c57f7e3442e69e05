// nsr_kernels.hip -- MI355X (gfx950 / CDNA4) kernels of the NeRF volumetric render path.
//
// What this file replaces (reference gyhandy/Neural-Sim-NeRF, all pure-PyTorch op chains):
//   RN = optimization/utils/run_nerf_noscale.py, RH = optimization/utils/run_nerf_helpers.py
//   render_rays RN:390-501, batchify_rays RN:43-55, run_network RN:26-40, raw2outputs RN:343-387,
//   Embedder RH:18-48, NeRF MLP RH:99-122, get_rays RH:156-165, sample_pdf RH:199-243.
//
// Design:
//   * one persistent 256-thread workgroup per CU (4 waves, ONE wave per SIMD, up to 512 VGPR+AGPR each);
//   * a work item = 2 rays = 1 coarse network pass (2x64 points) + 3 fine passes (2x192 points);
//   * a network pass evaluates 128 points (32 per wave).  Activations never leave registers: every layer is
//       H_out^T[256 x 32pts] = W[256 x K] * H_in^T[K x 32pts]
//     on v_mfma_f32_32x32x2_f32 with the WEIGHTS as the A operand and the activations as the B operand, so
//     the C/D fragment of layer L (lane = point, register = feature) is, register for register, the B
//     operand of layer L+1.  The host packer permutes each weight matrix's K order to match (kappa below);
//   * the weights (2.3 MiB fp32 per network, L2-resident) are streamed through a 6 x 16 KiB LDS ring by
//     global_load_lds_dwordx4 (LDS-DMA), NS-1 slabs ahead, one s_barrier per 64 MFMAs; the packed global
//     image IS the LDS image (lane-linear), read back with conflict-free ds_read_b128;
//   * per-ray state (z values, raw network outputs, weights, cdf) lives in LDS; the transmittance scan and the
//     cdf scan accumulate sequentially in fp64 exactly like torch-CPU cumprod/cumsum do;
//   * arithmetic that feeds comparisons (cdf, searchsorted, inverse-CDF samples, sort) is IEEE fp32 op by op:
//     this file is compiled with -ffp-contract=off and uses fmaf only where the reference has a GEMM.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "nsr_device.h"

// Translation units (r06).  This file holds the device code of every fused kernel; libnsr.so compiles it FOUR times, each
// unit DEFINING its own subset of the __global__ kernels and only declaring the others (their host-side launch stubs are
// ordinary functions, resolved at link time; no device function is called across units, so no relocatable device code):
//   nsr_api.hip        NSR_UNIT_F32  the host API, the stage kernels, the x32 fp32 kernels (k_render, k_render_vjp, k_run_network)
//   nsr_fused_h2.hip   NSR_UNIT_H2   the f16x2 kernels (the default arithmetic): k_render_h2*, k_render_vjp_h2*
//   nsr_fused_b3.hip   NSR_UNIT_B3   the bf16x3 kernels: k_render_b3*, k_render_vjp_b3*
//   nsr_fused_x16.hip  NSR_UNIT_X16  the x16 fp32 kernels: k_render16, k_render16p, k_render_vjp16, k_render_vjp16p
// A unit that defines none of the NSR_UNIT_* macros (libnsr_probe.so) gets everything.
#if !defined(NSR_UNIT_F32) && !defined(NSR_UNIT_H2) && !defined(NSR_UNIT_B3) && !defined(NSR_UNIT_X16)
#define NSR_UNIT_F32 1
#define NSR_UNIT_H2 1
#define NSR_UNIT_B3 1
#define NSR_UNIT_X16 1
#endif
#ifndef NSR_UNIT_F32
#define NSR_UNIT_F32 0
#endif
#ifndef NSR_UNIT_H2
#define NSR_UNIT_H2 0
#endif
#ifndef NSR_UNIT_B3
#define NSR_UNIT_B3 0
#endif
#ifndef NSR_UNIT_X16
#define NSR_UNIT_X16 0
#endif
#define NSR_CAT_(a, b) a##b
#define NSR_CAT(a, b) NSR_CAT_(a, b)

namespace nsr {

// Opaque copies: the per-item phases of the persistent kernels index LDS and the argument block with expressions that
// are invariant across the main loop (tid-derived LDS addresses, argument-block fields).  Left alone, LICM hoists ~80
// of them above the loop, where they stay live across the MFMA passes -- which have no registers to spare -- and get
// spilled to scratch (34 VGPRs + 258 SGPRs in k_render16).  Re-deriving them from an opaque copy inside the loop body
// costs one or two VALU/SALU instructions per use and keeps the passes free of spill code.
__device__ __forceinline__ int opaque_v(int x) { asm volatile("" : "+v"(x)); return x; }
template <typename T>
__device__ __forceinline__ const T* opaque_s(const T* p) { asm volatile("" : "+s"(p)); return p; }

// A 64-bit value that is wave-uniform by construction (read back from an LDS broadcast slot): say so.  Everything
// derived from it (ray index, camera / ray addresses) then lives in scalar registers, and the ray staging loads become
// scalar-cache loads instead of queueing behind the LDS-DMA weight stream in the vector memory pipeline.
__device__ __forceinline__ long long uniform64(long long v) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)v >> 32));
  return (long long)(((unsigned long long)hi << 32) | lo);
}

// ------------------------------------------------------------------------------------------------------
// ring (LDS weight stream)
// ------------------------------------------------------------------------------------------------------
constexpr int kRing16 = 3;          // ring slots of the x16 kernels

struct Ring {
  char* smem;            // LDS base (generic pointer)
  const char* base;      // packed networks (global), consecutive at `stride` bytes: coarse | fine | fine^T (backward)
  long long stride;
  int lane_off;          // wave*4096 + lane*16
  int wave_lds;          // wave*4096
  int pslab;             // producer: next slab index within the pass stream
  int pslot;             // producer: next ring slot
  int pphase;            // producer: pass index within the item (0 = coarse net)
  int ppi;               // passes per item: 1 (coarse only), 4 (coarse + 3 fine) or 7 (+ 3 backward)
  int cslot;             // consumer: slot of the slab being consumed
  int pnet_off;          // producer: byte offset of the current net within `base` (buffer-descriptor form)
  int pn0, pn1;          // producer: passes [0,pn0) of a cycle stream net 0, [pn0,pn1) net 1, the rest net 2
  int dd;                // 1: data-driven sequence -- the consumer names the network of the FOLLOWING pass in pnet_next
  int pnet_next;         //    at the start of every pass (k_render16p: the pass sequence depends on the task queue)
  __amdgpu_buffer_rsrc_t rsrc;
};

__device__ __forceinline__ void ring_init(Ring& rg, char* smem, const void* base, long long stride, int ppi, int wave,
                                          int lane) {
  rg.smem = smem;
  rg.base = (const char*)base;
  rg.stride = stride;
  rg.lane_off = wave * 4096 + lane * 16;
  rg.wave_lds = wave * 4096;
  rg.pslab = 0; rg.pslot = 0; rg.pphase = 0; rg.cslot = 0; rg.pnet_off = 0;
  rg.ppi = ppi;
  rg.pn0 = 1; rg.pn1 = 4;            // coarse pass, 3 fine passes, then (ppi = 7) 3 backward passes
  rg.dd = 0; rg.pnet_next = 0;
  rg.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
}

// SLABS: slabs per forward pass stream; SLABS_BWD: slabs of a backward pass (net 2) where that differs (bf16x3)
template <int NS = kRingSlots, int SLABS = kStreamSlabs, int SLABS_BWD = SLABS>
__device__ __forceinline__ void ring_issue(Ring& rg) {
  char* l = rg.smem + rg.pslot * kSlabBytes + rg.wave_lds;
#ifndef NSR_EXP_NODMA        // (NODMA: timing experiment only)
  // MUBUF LDS-DMA (buffer_load_dwordx4 ... lds): per-lane offset is a loop-invariant VGPR, everything that changes
  // (network, slab, chunk) sits in the scalar offset / immediate -> no per-instruction VALU address math; measured
  // +1 % on the layer GEMM over the flat global_load_lds form with 64-bit per-lane addresses.
  const int soff = rg.pnet_off + rg.pslab * kSlabBytes;
#define NSR_BUFDMA(C)                                                                                      \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rg.rsrc, (__attribute__((address_space(3))) void*)l, 16, rg.lane_off, \
                                           soff, (C) * 1024, 0) /* the immediate offsets BOTH the source and LDS */
  NSR_BUFDMA(0); NSR_BUFDMA(1); NSR_BUFDMA(2); NSR_BUFDMA(3);
#undef NSR_BUFDMA
#endif
  rg.pslot = (rg.pslot + 1 == NS) ? 0 : rg.pslot + 1;
  // branch-free advance (scalar selects only): a branch here would split the basic block and stop the scheduler
  // from interleaving the DMA issue with the MFMAs around it
  const int nslab = rg.pslab + 1;
  const bool wrap = nslab == ((SLABS_BWD != SLABS && rg.pphase >= rg.pn1) ? SLABS_BWD : SLABS);
  rg.pslab = wrap ? 0 : nslab;
  const int nphase = (rg.pphase + 1 == rg.ppi) ? 0 : rg.pphase + 1;
  rg.pphase = wrap ? nphase : rg.pphase;
  // arithmetic, not a pointer table: keeps Ring in SGPRs
  const int net = rg.pphase < rg.pn0 ? 0 : (rg.pphase < rg.pn1 ? 1 : 2);
  const int named = wrap ? rg.pnet_next : rg.pnet_off;     // data-driven form (dd is a compile-time 0 elsewhere)
  rg.pnet_off = rg.dd ? named : net * (int)rg.stride;
}

// The ring's scalar state is wave-uniform by construction.  When it is rewritten under control flow the compiler cannot
// prove uniform (k_render16p's ring restart), SIFixSGPRCopies moves the whole state to VGPRs and every LDS-DMA load gets
// a waterfall loop; re-asserting uniformity once per pass keeps it in SGPRs.
__device__ __forceinline__ void ring_assert_uniform(Ring& rg) {
  rg.pslab = __builtin_amdgcn_readfirstlane(rg.pslab);
  rg.pslot = __builtin_amdgcn_readfirstlane(rg.pslot);
  rg.pphase = __builtin_amdgcn_readfirstlane(rg.pphase);
  rg.cslot = __builtin_amdgcn_readfirstlane(rg.cslot);
  rg.pnet_off = __builtin_amdgcn_readfirstlane(rg.pnet_off);
  rg.pnet_next = __builtin_amdgcn_readfirstlane(rg.pnet_next);
}

// Fill the ring (NS slabs in flight), certify slab 0 and load the first step's fragments.
// A step = 4 chunks of 1 KiB = 4 float4 fragments per lane = 16 MFMAs; a slab = 4 steps.
#ifndef NSR_H2_SCHED
#define NSR_H2_SCHED 3     // step schedule of the f16x2 layer GEMMs (nsr_h2.inc: step_h2): 3 = one LDS-DMA piece per step (r05);
#endif                     // 0 = r03 / r04 (four pieces behind the slab change), kept for A/B builds (-DNSR_H2_SCHED=0)
constexpr int kH2RingLag = NSR_H2_SCHED == 3 ? 1 : 0;
// LAG = 1: the ring starts one slab short -- the consumer's steps then issue the pieces of slab n + NS - 1 DURING slab n,
// one piece per step (nsr_h2.inc, NSR_H2_SCHED 3), instead of the four pieces of slab n + NS at the end of slab n.
template <int NS = kRingSlots, int SLABS = kStreamSlabs, int SLABS_BWD = SLABS, int LAG = 0>
__device__ __forceinline__ void ring_start(Ring& rg, f32x4 (&A0)[4], int lane) {
#pragma unroll 1
  for (int s = 0; s < NS - LAG; ++s) ring_issue<NS, SLABS, SLABS_BWD>(rg);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (NS - LAG - 1)) : "memory");
  __builtin_amdgcn_s_barrier();
  const char* p = rg.smem + rg.cslot * kSlabBytes + lane * 16;
#pragma unroll
  for (int c = 0; c < 4; ++c) A0[c] = *(const f32x4*)(p + c * 1024);
}

// Steps 0..2 of a slab: the slab is already certified; fetch quarter u+1 for the next step.
template <int U>
__device__ __forceinline__ void ring_load_quarter(const Ring& rg, f32x4 (&A)[4], int lane) {
  const char* p = rg.smem + rg.cslot * kSlabBytes + U * kStepBytes + lane * 16;
#pragma unroll
  for (int c = 0; c < 4; ++c) A[c] = *(const f32x4*)(p + c * 1024);
}

// Last step of a slab: all of this wave's reads of the current slab are complete after lgkmcnt(0); my share of
// the next slab has landed after the counted vmcnt; the barrier makes both true for the whole workgroup, so the
// slot of the current slab can be refilled (slab n+NS) and the next slab's first quarter can be read.
template <int NS = kRingSlots>
__device__ __forceinline__ void ring_advance(Ring& rg, f32x4 (&A)[4], int lane) {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(4 * (NS - 2)) : "memory");
#ifndef NSR_EXP_NOBARRIER   // timing experiment only (results are wrong without the barrier)
  __builtin_amdgcn_s_barrier();
#endif
  rg.cslot = (rg.cslot + 1 == NS) ? 0 : rg.cslot + 1;
  const char* p = rg.smem + rg.cslot * kSlabBytes + lane * 16;
#pragma unroll
  for (int c = 0; c < 4; ++c) A[c] = *(const f32x4*)(p + c * 1024);
}

// ------------------------------------------------------------------------------------------------------
// MFMA segments.  A segment is a GEMM  acc[mo] += W_block(mo) * B  with NMO output blocks of 32 rows and NTQ
// k-quads (4 k-steps of 2 each); its weights are NMO*NTQ chunks in the stream, chunk n <-> (k-quad n / NMO,
// output block n % NMO).  `bop(t)` returns this lane's B operand (activation / gradient value) of k-step t;
// every index is a compile-time constant after unrolling.
// ------------------------------------------------------------------------------------------------------
#define NSR_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
// one name for both fp32 MFMA shapes: 32x32x2 (16 accumulator registers, 32 points per wave) and 16x16x4 (4 registers,
// 16 points per wave); exact fp32 products, fmaf-chain accumulate, 64 FLOP/clk/SIMD either way
__device__ __forceinline__ f32x16 mfma_op(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mfma_op(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// ZERO: the accumulators start from 0 -- the first MFMA of every output block takes the inline constant 0 as its C
// operand instead of a register block that 16 VALU writes per block would have to clear first.
template <int NMO, int KK0, int KK1, bool ZERO = false, int NACC, typename BOp, typename AccT>
__device__ __forceinline__ void consume(const f32x4 (&A)[4], int step, BOp bop, AccT (&acc)[NACC]) {
#pragma unroll
  for (int kk = KK0; kk < KK1; ++kk)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int n = 4 * step + c, tq = n / NMO, mo = n % NMO;
      if (ZERO && tq == 0 && kk == 0) acc[mo] = mfma_op(A[c][kk], bop(4 * tq + kk), AccT{0});
      else acc[mo] = mfma_op(A[c][kk], bop(4 * tq + kk), acc[mo]);
    }
}

// One step = [4 MFMAs][issue the next step's 4 fragment loads][12 MFMAs].  The loads are pinned there with
// sched_barrier: left alone, the scheduler sinks them to just above their first use (to save registers) and the
// waitcnt pass then waits lgkmcnt(0) right behind them, exposing the full LDS latency every 16 MFMAs (~10 % of the
// MFMA rate, measured).  Issued after the first 4 MFMAs they have 12 MFMAs (768 cycles) to land, and any
// lgkmcnt(0) the compiler places at the head of the next step finds nothing outstanding.
#define NSR_PIN() __builtin_amdgcn_sched_barrier(0)
// Instruction interleave of one step, given to the scheduler as a pattern (sched_group_barrier): an MFMA occupies
// the matrix pipe for 64 cycles and the wave can only run ~12 other instructions in its shadow, so the step's
// non-MFMA work (4 fragment loads for the next step; at a slab end also the DMA issue: 4 buffer_load..lds plus
// ~20 SALU) is spread one or two instructions per MFMA instead of sitting in one clump (a 25-instruction clump
// costs ~40-60 idle matrix-pipe cycles per slab: measured 149.4 -> 154.0 TFLOP/s on the isolated layer GEMM,
// 155 being the MFMA-only ceiling).
#define NSR_SGB(mask, n) __builtin_amdgcn_sched_group_barrier((mask), (n), 0)
#define NSR_MASK_VALU 0x002
#define NSR_MASK_SALU 0x004
#define NSR_MASK_MFMA 0x008
#define NSR_MASK_VMEM 0x010
#define NSR_MASK_DSRD 0x100
template <int N_MFMA>
__device__ __forceinline__ void step_pattern() {
#pragma unroll
  for (int i = 0; i < N_MFMA; ++i) {
    NSR_SGB(NSR_MASK_MFMA, 1);
    NSR_SGB(NSR_MASK_DSRD, 1);
    NSR_SGB(NSR_MASK_VALU, 2);
  }
}
template <int N_MFMA>
__device__ __forceinline__ void step_pattern_dma() {
#pragma unroll
  for (int i = 0; i < N_MFMA; ++i) {
    NSR_SGB(NSR_MASK_MFMA, 1);
    NSR_SGB(NSR_MASK_DSRD, 1);
    NSR_SGB(NSR_MASK_VMEM, 1);
    NSR_SGB(NSR_MASK_SALU, 3);
    NSR_SGB(NSR_MASK_VALU, 2);
  }
}

template <int NMO, int NTQ, int NS = kRingSlots, bool ZERO = false, int NACC, typename BOp, typename AccT>
__device__ __forceinline__ void seg(Ring& rg, f32x4 (&A0)[4], f32x4 (&A1)[4], BOp bop, AccT (&acc)[NACC],
                                    int lane) {
  static_assert((NMO * NTQ) % 16 == 0, "a segment is a whole number of slabs");
#pragma unroll
  for (int s = 0; s < NMO * NTQ / 4; s += 4) {
    NSR_PIN(); ring_load_quarter<1>(rg, A1, lane); consume<NMO, 0, 4, ZERO>(A0, s, bop, acc);     step_pattern<16>();
    NSR_PIN(); ring_load_quarter<2>(rg, A0, lane); consume<NMO, 0, 4, ZERO>(A1, s + 1, bop, acc); step_pattern<16>();
    NSR_PIN(); ring_load_quarter<3>(rg, A1, lane); consume<NMO, 0, 4, ZERO>(A0, s + 2, bop, acc); step_pattern<16>();
    NSR_PIN(); consume<NMO, 0, 1, ZERO>(A1, s + 3, bop, acc);
    NSR_PIN(); ring_advance<NS>(rg, A0, lane);      // counted wait + s_barrier, then the next slab's first loads
    ring_issue<NS>(rg); consume<NMO, 1, 4, ZERO>(A1, s + 3, bop, acc); step_pattern_dma<12>();
    NSR_PIN();
  }
}

// B-operand sources (references to register-resident arrays; all indices constant after unrolling)
template <int NG>
struct BRegs16 {   // previous layer's C fragment: k-step t <-> register (t>>4, t&15)
  const f32x16 (&v)[NG];
  __device__ __forceinline__ float operator()(int t) const { return v[t >> 4][t & 15]; }
};
template <int N>
struct BArr {      // plain per-lane array (encodings)
  const float (&v)[N];
  __device__ __forceinline__ float operator()(int t) const { return v[t]; }
};
struct BViews {    // cat([feature, input_views]) (RH:111): 128 k-steps of registers, then 16 of direction encoding
  const f32x16 (&v)[8];
  const float (&ed)[16];
  __device__ __forceinline__ float operator()(int t) const { return t < 128 ? v[(t & 127) >> 4][t & 15] : ed[t & 15]; }
};

// 4 * (lane half) as an opaque value: every aux-block access of a pass is then ONE per-lane base register plus an
// immediate offset.  (With the plain expression the compiler folds the half into ~70 distinct per-lane addresses,
// hoists them out of the main loop and spills them across the MFMA passes.)
__device__ __forceinline__ int aux_half(int lane) { return opaque_v((lane >> 5) * 4); }

// accumulator init = bias in C-fragment order (aux layout: [(mo*4+rq)*2+h] float4)
template <int NMO>
__device__ __forceinline__ void load_bias(const float* bias, int h4 /* 4 * lane half, see aux_half() */, f32x16 (&acc)[NMO]) {
#pragma unroll
  for (int mo = 0; mo < NMO; ++mo)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      f32x4 v = *(const f32x4*)(bias + (mo * 4 + rq) * 8 + h4);
      acc[mo][rq * 4 + 0] = v[0];
      acc[mo][rq * 4 + 1] = v[1];
      acc[mo][rq * 4 + 2] = v[2];
      acc[mo][rq * 4 + 3] = v[3];
    }
}

#include "nsr_b3.inc"
#include "nsr_h2.inc"

// arithmetic of the layer GEMMs of a network pass
constexpr int kMlpF32 = 0;      // fp32 MFMAs (seg)
constexpr int kMlpB3 = 1;       // bf16 MFMAs, operands split three ways (nsr_b3.inc)
constexpr int kMlpH2 = 2;       // fp16 MFMAs, operands split two ways + power-of-two range management (nsr_h2.inc)

// ------------------------------------------------------------------------------------------------------
// One network pass for this lane's point.  Lane (j = lane&31, h = lane>>5): both halves work on point j and
// hold complementary halves of every feature vector.  Returns raw = (r,g,b logits, sigma) in all lanes.
//   Embedder RH:18-48 (in-register), NeRF.forward RH:99-122.
// ------------------------------------------------------------------------------------------------------
// relu pattern of one layer output (lane-private C fragment) as a bit mask: bit (mo&1)*16 + r of word mo>>1
template <int NMO>
__device__ __forceinline__ uint4 relu_mask(const f32x16 (&acc)[NMO]) {
  // One v_alignbit per element shifts a sign bit into the word: element e = 16 (mo & 1) + r of word mo >> 1 ends
  // up at bit 31 - e, SET when the unit is OFF (x <= +0).  The sign comes from max(int(x), 0) - 1, which is -1
  // exactly for the units relu zeroes (negative floats, -0 and +0 have integer patterns <= 0) -- the same
  // predicate as torch's relu' (x > 0), without a compare + select + shift + or per element.
  unsigned w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int mo = 0; mo < NMO; ++mo)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int on = max(__float_as_int(acc[mo][r]), 0) - 1;
      w[mo >> 1] = __builtin_amdgcn_alignbit(w[mo >> 1], (unsigned)on, 31);
    }
  return make_uint4(w[0], w[1], w[2], w[3]);
}

// sin(a + k pi/2) for the encodings (RH:39-48; k = 0: sin, 1: cos, 2: -sin).  The library sincosf costs ~100
// VALU instructions per call (plus a Payne-Hanek path that cannot be kept out of the register budget), and a
// VALU instruction is never free here: it takes matrix-pipe issue time from the other wave on the SIMD
// (tools/probe_pairing.py).  So: one fp64 reduction -- t = a*(2/pi) has >= 24 spare bits for |a| < 2^24, and
// the split into quadrant n and fraction f is exact -- then the Cephes single-precision minimax polynomials on
// [-pi/4, pi/4] (1 ulp).  ~20 instructions.  Domain: |a| < 2^24; enc_domain() turns coordinates beyond it into
// NaN so that an out-of-range scene shows up as NaN pixels rather than as slightly wrong ones.
__device__ __forceinline__ float enc_trig(float a, int k) {
  const double kMagic = 6755399441055744.0;                 // 1.5 * 2^52: t + kMagic holds rint(t) in its low word
  const double t = (double)a * 0.63661977236758134308;      // a * 2/pi
  const double tm = t + kMagic;
  const int n = __double2loint(tm) + k;
  const double f = t - (tm - kMagic);                       // [-0.5, 0.5], exact
  const float r = (float)(f * 1.57079632679489661923);
  const float z = r * r;
  float ps = __builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
  ps = __builtin_fmaf(ps, z, -1.6666654611e-1f);
  ps = __builtin_fmaf(ps * z, r, r);
  float pc = __builtin_fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
  pc = __builtin_fmaf(pc, z, 4.166664568298827e-2f);
  pc = __builtin_fmaf(pc * z, z, __builtin_fmaf(-0.5f, z, 1.0f));
  const float v = (n & 1) ? pc : ps;
  return __uint_as_float(__float_as_uint(v) ^ ((unsigned)(n & 2) << 30));
}

// x if 2^(nfreq-1) |x| < 2^24 (the domain of enc_trig), NaN otherwise (also for inf / NaN inputs)
__device__ __forceinline__ float enc_domain(float x, int nfreq) {
  return __builtin_fabsf(x) < ldexpf(1.0f, 25 - nfreq) ? x : __builtin_nanf("");
}

// 0 if the point and direction are inside the encoder's domain, NaN otherwise.  Added to the network outputs so
// that a NaN / out-of-range input reaches the caller as NaN, as it does through torch's relu in the reference
// (the integer-max relu and v_max_f32 used here both let a NaN activation collapse to 0).
__device__ __forceinline__ float enc_poison(float px, float py, float pz, float vx, float vy, float vz) {
  const float a = (enc_domain(px, kMultires) * 0.0f + enc_domain(py, kMultires) * 0.0f) + enc_domain(pz, kMultires) * 0.0f;
  const float b = (enc_domain(vx, kMultiresViews) * 0.0f + enc_domain(vy, kMultiresViews) * 0.0f) +
                  enc_domain(vz, kMultiresViews) * 0.0f;
  return __builtin_fabsf(a + b);      // +0 or NaN
}

// MODE kMlpB3 / kMlpH2: the layer GEMMs run on bf16 / fp16 MFMAs with split operands (nsr_b3.inc, nsr_h2.inc) instead of
// fp32 MFMAs; the encodings, biases, activations and the two VALU heads are the same code (for kMlpH2 the packer has
// folded the layers' power-of-two scales into the biases and head weights of the aux block).
template <bool CAPTURE, int MODE = kMlpF32>
__device__ __forceinline__ void mlp_pass(Ring& rg, const float* aux, f32x4 (&A0)[4], f32x4 (&A1)[4], int lane,
                                         float px, float py, float pz, float vx, float vy, float vz,
                                         float (&raw)[4], uint4* mask_dst = nullptr /* uniform */, int mask_tid = 0,
                                         long long* tp = nullptr /* NSR_PHASE_TIMING: [4] encodings, GEMMs, between, heads */) {
#ifdef NSR_PHASE_TIMING
  long long tpl = clock64();
#define NSR_TP(i) do { if (tp) { const long long t_ = clock64(); tp[i] += t_ - tpl; tpl = t_; } } while (0)
#else
#define NSR_TP(i) do { } while (0)
#endif
  const int h = lane >> 5;
  const int h4 = aux_half(lane);
  const float poison = enc_poison(px, py, pz, vx, vy, vz);
  float e[32];   // position encoding, k-step t: h=0 -> sin(2^L p_ax), h=1 -> cos(2^L p_ax), t = 3L+ax
  float ed[16];  // direction encoding, same scheme with L < 4
  {
    const float p[3] = {px, py, pz};
    const float v[3] = {vx, vy, vz};
#pragma unroll
    for (int L = 0; L < kMultires; ++L)
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) e[3 * L + ax] = enc_trig(enc_domain(p[ax], kMultires) * (float)(1 << L), h);
#pragma unroll
    for (int L = 0; L < kMultiresViews; ++L)
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) ed[3 * L + ax] = enc_trig(enc_domain(v[ax], kMultiresViews) * (float)(1 << L), h);
    e[30] = h ? pz : px;
    e[31] = h ? 0.0f : py;
    ed[12] = h ? vz : vx;
    ed[13] = h ? 0.0f : vy;
    ed[14] = 0.0f;
    ed[15] = 0.0f;
  }

  f32x16 acc[8];
  f32x16 in[8];
  // layer 0
  auto enc_src = [&](int kb, int i) { return e[(8 * kb + i) & 31]; };
  auto in_src = [&](int kb, int i) { return in[(kb >> 1) & 7][8 * (kb & 1) + i]; };
  constexpr bool B3 = MODE == kMlpB3;
  constexpr bool H2M = MODE == kMlpH2;
  float amax = 0.0f;                       // kMlpH2: largest |scaled activation| this lane has split so far
  const float* h2s = aux + kAuxH2Scale;
  NSR_TP(0);
  load_bias<8>(aux + kAuxBias, h4, acc);
  NSR_TP(2);
  if constexpr (B3) gemm_b3<8, 2>(rg, A0, A1, enc_src, acc, lane);
  else if constexpr (H2M) gemm_h2<8, 4>(rg, A0, A1, enc_src, H2Scale{{h2s[0], 0.0f}, 4}, acc, lane, amax);
  else seg<8, 8>(rg, A0, A1, BArr<32>{e}, acc, lane);
  NSR_TP(1);
  if (CAPTURE) mask_dst[mask_tid] = relu_mask<8>(acc);
#pragma unroll
  for (int mo = 0; mo < 8; ++mo) in[mo] = relu16(acc[mo]);

  float alpha_part = 0.0f;
  // layers 1..7 (ReLU) and 8 = feature_linear (no activation)
#pragma unroll 1
  for (int L = 1; L <= 8; ++L) {
    load_bias<8>(aux + kAuxBias + L * 256, h4, acc);
    NSR_TP(2);
    if (L == 5) {                                                // skip: cat([input_pts, h]) -> input columns first (RH:105)
      if constexpr (B3) gemm_b3<8, 2>(rg, A0, A1, enc_src, acc, lane);
      else if constexpr (H2M) gemm_h2<8, 4>(rg, A0, A1, enc_src, H2Scale{{h2s[9], 0.0f}, 4}, acc, lane, amax);
      else seg<8, 8>(rg, A0, A1, BArr<32>{e}, acc, lane);
      NSR_TP(1);
    }
    if (L == 8) {
      // alpha_linear on h7 (RH:109): VALU dot product over this lane's 128 features, halves summed below
      const float* wa = aux + kAuxWAlpha;
#pragma unroll
      for (int tq = 0; tq < 32; ++tq) {
        f32x4 w = *(const f32x4*)(wa + tq * 8 + h4);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
          alpha_part = __builtin_fmaf(w[kk], in[(4 * tq + kk) >> 4][(4 * tq + kk) & 15], alpha_part);
      }
    }
    if constexpr (B3) gemm_b3<8, 8>(rg, A0, A1, in_src, acc, lane);
    else if constexpr (H2M) gemm_h2<8, 16>(rg, A0, A1, in_src, H2Scale{{h2s[L], 0.0f}, 16}, acc, lane, amax);
    else seg<8, 32>(rg, A0, A1, BRegs16<8>{in}, acc, lane);
    NSR_TP(1);
    if (CAPTURE && L < 8) mask_dst[L * 256 + mask_tid] = relu_mask<8>(acc);
    const int thr = (L == 8) ? (int)0x80000000 : 0;      // feature_linear has no activation
#pragma unroll
    for (int mo = 0; mo < 8; ++mo) in[mo] = clamp_bits16(acc[mo], thr);
  }

  // views_linears.0 (RH:111-115): cat([feature, input_views]) -> 128, ReLU
  f32x16 av[4];
  load_bias<4>(aux + kAuxBiasV, h4, av);
  NSR_TP(2);
  if constexpr (B3) {        // 16 blocks of features, 2 of direction encoding, 2 of padding (a group is 4 blocks)
    auto v_src = [&](int kb, int i) {
      return kb < 16 ? in[(kb >> 1) & 7][8 * (kb & 1) + i] : (kb < 18 ? ed[(8 * (kb - 16) + i) & 15] : 0.0f);
    };
    gemm_b3<4, 5>(rg, A0, A1, v_src, av, lane);
  } else if constexpr (H2M) {  // 16 k16 blocks of features, 2 of direction encoding
    auto v_src = [&](int kb, int i) { return kb < 16 ? in[(kb >> 1) & 7][8 * (kb & 1) + i] : ed[(8 * (kb - 16) + i) & 15]; };
    gemm_h2<4, 18>(rg, A0, A1, v_src, H2Scale{{h2s[10], h2s[11]}, 16}, av, lane, amax);
  } else {
    seg<4, 36>(rg, A0, A1, BViews{in, ed}, av, lane);
  }
  NSR_TP(1);
  if (CAPTURE) mask_dst[8 * 256 + mask_tid] = relu_mask<4>(av);

  // rgb_linear (RH:117) on relu(av): VALU
  float part[4] = {0.0f, 0.0f, 0.0f, alpha_part};
  if constexpr (H2M) {         // a scaled activation beyond the fp16 range: the point's outputs are NaN, not garbage
    const float ov = amax > kH2Max ? __builtin_nanf("") : 0.0f;
    part[0] = part[1] = part[2] = ov;
    part[3] = alpha_part + ov;
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float* wr = aux + kAuxWRgb + c * 128;
#pragma unroll
    for (int mo = 0; mo < 4; ++mo)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        f32x4 w = *(const f32x4*)(wr + (mo * 4 + rq) * 8 + h4);
#pragma unroll
        for (int ri = 0; ri < 4; ++ri)
          part[c] = __builtin_fmaf(w[ri], fmaxf(av[mo][rq * 4 + ri], 0.0f), part[c]);
      }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float other = __shfl_xor(part[c], 32);
    float lo_half = h ? other : part[c];
    float hi_half = h ? part[c] : other;
    raw[c] = ((lo_half + hi_half) + aux[(c < 3) ? (kAuxBRgb + c) : kAuxBAlpha]) + poison;
  }
  NSR_TP(3);
#undef NSR_TP
}

// ------------------------------------------------------------------------------------------------------
// per-item state in LDS (2 rays)
// ------------------------------------------------------------------------------------------------------
// NSS / NFS: row strides of the coarse / fine arrays = the largest N_samples / N_samples + N_importance the kernels using this
// layout are instantiated for.  ItemState (64, 192) is the YCB-V configuration's and every kernel's whose N_samples is <= 64;
// ItemStateBig (128, 256) serves N_samples = 128 (r05).
template <int NSS, int NFS>
struct ItemStateT {
  static constexpr int kNS = NSS, kNF = NFS;
  float tcoarse[NSS];       // torch.linspace(0,1,N_samples)   RN:439
  float ufine[128];         // torch.linspace(0,1,128)  RH:208
  float ray[2][16];         // o[0..2] d[3..5] viewdir[6..8] near far |d|
  float zc[2][NSS];         // coarse z               RN:441
  float rawc[2][NSS][4];    // coarse raw             RN:466
  float w0[2][NSS];         // coarse weights         RN:467
  float cdf[2][NSS];        // N_samples - 1 used     RH:203-204
  float zs[2][128];         // importance samples     RH:241
  float zf[2][NFS];         // sorted merged z        RN:477
  float rawf[2][NFS][4];    // fine raw               RN:483
  float alpha[2][NFS];      // compositing scratch
  float wf[2][NFS];         // fine weights           RN:485
  float tf[2][NFS];         // transmittance T_i (RN:376)
  float om[2][NFS];         // 1 - alpha + 1e-10 (RN:376) / pdf (RH:202): fp32 values, widened to fp64 inside the scans
  float bwd_scratch[2][NFS][2];   // backward compositing: A_i*w_i suffix sums and A_i*T_i
  float psum[2][NFS / 32][12];    // backward: per (pass, wave) partial sums of d/dpts, z*d/dpts, d/dviewdir
  float gnorm[2];           // backward: dL/d|rays_d| from dists*|d| (RN:361)
  float res[2][8];          // rgb(3) disp acc depth
};
typedef ItemStateT<64, 192> ItemState;
typedef ItemStateT<128, 256> ItemStateBig;
static_assert(sizeof(ItemState) == 23048, "the YCB-V layout is part of the shipped kernels' LDS budget");
template <int NS> struct ItemStateFor { typedef ItemState type; };
template <> struct ItemStateFor<128> { typedef ItemStateBig type; };

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// raw2outputs RN:343-387 for both rays of the item.  z: [2][S], raw: [2][S][4] (rgb overwritten by sigmoid),
// wout: [2][S] weights, tout: [2][S] transmittance.  Results in st.res.
// Three phases: (1) per-sample alpha / sigmoid / 1-alpha+1e-10, all threads; (2) the transmittance scan -- the only
// inherently serial part: torch-CPU cumprod is a sequential fp64 product with every prefix rounded to fp32
// (RN:376), one lane per ray, 8 factors per LDS round trip; (3) weights and the five weighted sums, one wave per
// ray (lane l owns samples l, l+64, l+128), xor-shuffle tree.
// STRIDE: samples per ray of the z / raw / wout / tout arrays (>= S; the item state is laid out for the largest S).
template <int S, int R = 2, int STRIDE = S, typename ST>
__device__ __forceinline__ void composite(ST& st, const float* z, float* raw, float* wout, float* tout, int tid,
                                          const float* noise = nullptr /* global [rays][S]: RN:365-374 */,
                                          long long row0 = 0, int valid = R) {
  static_assert(S % 8 == 0, "scan is unrolled by 8");
  for (int idx = tid; idx < R * S; idx += 256) {
    const int r = idx / S, i = idx - r * S;
    const float* zr = z + r * STRIDE;
    float dist = (i < S - 1) ? (zr[i + 1] - zr[i]) : 1e10f;   // RN:358-359
    dist = dist * st.ray[r][11];                               // RN:361
    float* q = raw + (r * STRIDE + i) * 4;
    if (noise) q[3] = q[3] + noise[(row0 + (r < valid ? r : 0)) * S + i];   // RN:374 (kept: the backward's relu' sees it too)
    const float sigma = fmaxf(q[3], 0.0f);
    const float a = 1.0f - expf(-sigma * dist);                // RN:356
    st.alpha[r][i] = a;
    st.om[r][i] = (1.0f - a) + 1e-10f;                         // RN:376 factor (widened to fp64 in the scan)
    q[0] = sigmoidf_(q[0]);                                    // RN:363
    q[1] = sigmoidf_(q[1]);
    q[2] = sigmoidf_(q[2]);
  }
  __syncthreads();
  if ((tid & 63) == 0 && tid < 64 * R) {
    const int r = tid >> 6;
    double T = 1.0;
#pragma unroll 1
    for (int i0 = 0; i0 < S; i0 += 8) {
      double f[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) f[k] = (double)st.om[r][i0 + k];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        tout[r * STRIDE + i0 + k] = (float)T;                  // exclusive product, rounded per prefix
        T = T * f[k];
      }
    }
  }
  __syncthreads();
  if (tid < 64 * R) {
    const int r = tid >> 6, l = tid & 63;
    const float* zr = z + r * STRIDE;
    const float* q = raw + r * STRIDE * 4;
    float cr = 0.f, cg = 0.f, cb = 0.f, depth = 0.f, acc = 0.f;
#pragma unroll
    for (int i = l; i < S; i += 64) {
      const float w = st.alpha[r][i] * tout[r * STRIDE + i];   // RN:376
      wout[r * STRIDE + i] = w;
      cr = cr + w * q[i * 4 + 0];                              // RN:378
      cg = cg + w * q[i * 4 + 1];
      cb = cb + w * q[i * 4 + 2];
      depth = depth + w * zr[i];                               // RN:380
      acc = acc + w;                                           // RN:382
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      cr += __shfl_xor(cr, m); cg += __shfl_xor(cg, m); cb += __shfl_xor(cb, m);
      depth += __shfl_xor(depth, m); acc += __shfl_xor(acc, m);
    }
    if (l == 0) {
      const float qd = depth / acc;
      float disp;
      if (qd != qd) disp = qd;                                 // 0/0 -> NaN propagates through torch.max (RN:381)
      else disp = 1.0f / fmaxf(1e-10f, qd);
      if (st.ray[r][12] != 0.0f) {                             // white_bkgd, RN:384-385
        const float bg = 1.0f - acc;
        cr = cr + bg; cg = cg + bg; cb = cb + bg;
      }
      st.res[r][0] = cr; st.res[r][1] = cg; st.res[r][2] = cb;
      st.res[r][3] = disp; st.res[r][4] = acc; st.res[r][5] = depth;
    }
  }
  __syncthreads();
}

// sample_pdf RH:199-243 (det=True) for both rays: weights w[r][0..61] (= coarse weights[1:-1]), bins = mid-points.
// Writes st.cdf, st.zs; optional inds.  `bins` is a callable: bins(r, k), k in 0..62.
// NI: importance samples per ray (128, or 64 / 32 for the kernels specialised to N_importance = 64 / 32); the thread
// mapping and the row stride of u_rays stay those of 128.
// sample_pdf_item for N_samples != 64 (NS - 2 weights, NS - 1 bins and cdf entries).  Same arithmetic: torch.sum in ATen's
// association order for n = NS - 2 contiguous floats (oracle/nerf_oracle.py: _torch_sum_lastdim -- 8 vector lanes, 4-way ILP:
// size_ilp = (n / 8) / 4 rounds into four partial vectors, the remaining whole vectors into the first, the partials folded
// 0 += 1, 2, 3, then the scalar tail and the eight lanes sequentially), the cdf as a sequential fp64 scan rounded per prefix.
template <int R, int NI, int NS, typename ST, typename BinsFn>
__device__ __forceinline__ void sample_pdf_item_ns(ST& st, const float* ufine, const float* w, int wstride, BinsFn bins,
                                                   int64_t* inds_out, int64_t inds_stride, int tid, int valid_rays,
                                                   const float* u_rays, long long row0, int write_mask) {
  constexpr int NW = NS - 2, NC = NS - 1;            // weights; bins = cdf entries
  constexpr int NVEC = NW / 8, SILP = NVEC / 4;
  for (int e = tid; e < R * NW; e += 256) {
    const int r = e / NW, i = e - r * NW;
    st.alpha[r][i] = (w + r * wstride)[i] + 1e-5f;               // RH:201
  }
  __syncthreads();
  if ((tid & 63) == 0 && tid < 64 * R) {
    const int r = tid >> 6;
    const float* x = st.alpha[r];
    float lanes[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float ps[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int i = 0; i < SILP; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) ps[k] = ps[k] + x[(i * 4 + k) * 8 + j];
#pragma unroll
      for (int v = SILP * 4; v < NVEC; ++v) ps[0] = ps[0] + x[v * 8 + j];
      lanes[j] = ((ps[0] + ps[1]) + ps[2]) + ps[3];
    }
    float total = 0.0f;
#pragma unroll
    for (int i = NVEC * 8; i < NW; ++i) total = total + x[i];
#pragma unroll
    for (int j = 0; j < 8; ++j) total = total + lanes[j];
    st.res[r][7] = total;
  }
  __syncthreads();
  for (int e = tid; e < R * NW; e += 256) {
    const int r = e / NW, i = e - r * NW;
    st.om[r][i] = st.alpha[r][i] / st.res[r][7];                // pdf (RH:202)
  }
  __syncthreads();
  if ((tid & 63) == 0 && tid < 64 * R) {
    const int r = tid >> 6;
    double run = 0.0;
    st.cdf[r][0] = 0.0f;
#pragma unroll 1
    for (int i = 0; i < NW; ++i) { run = run + (double)st.om[r][i]; st.cdf[r][i + 1] = (float)run; }      // RH:203-204
  }
  __syncthreads();
  if (tid < 128 * R && (tid & 127) < NI) {
    const int r = tid >> 7, k = tid & 127;
    const float u = u_rays ? u_rays[(row0 + (r < valid_rays ? r : 0)) * 128 + k] : ufine[k];
    const float* cdf = st.cdf[r];
    int lo = 0, hi = NC;                                         // searchsorted(cdf, u, right=True) among NC entries (RH:227)
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cdf[NSR_IDX(mid, NC)] <= u) lo = mid + 1; else hi = mid;
    }
    const int ind = lo;
    const int below = max(ind - 1, 0);
    const int above = min(ind, NC - 1);
    const float c0 = cdf[NSR_IDX(below, NC)], c1 = cdf[NSR_IDX(above, NC)];
    const float b0 = bins(r, NSR_IDX(below, NC)), b1 = bins(r, NSR_IDX(above, NC));
    float denom = c1 - c0;
    if (denom < 1e-5f) denom = 1.0f;                            // RH:238-239
    const float t = (u - c0) / denom;
    st.zs[r][k] = b0 + t * (b1 - b0);                           // RH:241
    if (inds_out && r < valid_rays && ((write_mask >> r) & 1)) inds_out[r * inds_stride + k] = (int64_t)ind;
  }
  __syncthreads();
}

template <int R = 2, int NI = 128, int NS = 64, typename ST, typename BinsFn>
__device__ __forceinline__ void sample_pdf_item(ST& st, const float* ufine /*[NI of 128], LDS or global*/,
                                                const float* w /*[R][stride]*/, int wstride, BinsFn bins,
                                                int64_t* inds_out /*[R][128] or null*/, int64_t inds_stride, int tid,
                                                int valid_rays, const float* u_rays = nullptr /* global [rays][128]: RH:211 */,
                                                long long row0 = 0, int write_mask = 3 /* rays whose inds are written */) {
  if constexpr (NS != 64) {      // N_samples 32 / 128 (r05): the same arithmetic over NS - 2 weights / NS - 1 bins, see below
    sample_pdf_item_ns<R, NI, NS>(st, ufine, w, wstride, bins, inds_out, inds_stride, tid, valid_rays, u_rays, row0, write_mask);
    return;
  }
  // (1) x = w + 1e-5 and the 8 vector-lane partial sums of ATen's cascade, 8 lanes per ray
  if (tid < 64 * R && (tid & 63) < 62) {
    const int r = tid >> 6, i = tid & 63;
    st.alpha[r][i] = (w + r * wstride)[i] + 1e-5f;              // RH:201 (alpha[] is free scratch here)
  }
  __syncthreads();
  if ((tid & 63) == 0 && tid < 64 * R) {
    const int r = tid >> 6;
    const float* x = st.alpha[r];
    // torch.sum over 62 contiguous floats: ATen's 8-lane x 4-ILP cascade (RH:202), exact association order
    float lanes[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float p0 = ((x[j] + x[32 + j]) + x[40 + j]) + x[48 + j];
      lanes[j] = ((p0 + x[8 + j]) + x[16 + j]) + x[24 + j];
    }
    float total = 0.0f;
#pragma unroll
    for (int i = 56; i < 62; ++i) total = total + x[i];
#pragma unroll
    for (int j = 0; j < 8; ++j) total = total + lanes[j];
    st.res[r][7] = total;
  }
  __syncthreads();
  if (tid < 64 * R && (tid & 63) < 62) {
    const int r = tid >> 6, i = tid & 63;
    st.om[r][i] = st.alpha[r][i] / st.res[r][7];               // pdf (RH:202), widened to fp64 in the scan
  }
  __syncthreads();
  if ((tid & 63) == 0 && tid < 64 * R) {
    const int r = tid >> 6;
    // cdf = [0, cumsum(pdf)]: sequential fp64 accumulator, each prefix rounded to fp32 (RH:203-204)
    double run = 0.0;
    st.cdf[r][0] = 0.0f;
#pragma unroll 1
    for (int i0 = 0; i0 < 56; i0 += 8) {
      double f[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) f[k] = (double)st.om[r][i0 + k];
#pragma unroll
      for (int k = 0; k < 8; ++k) { run = run + f[k]; st.cdf[r][i0 + k + 1] = (float)run; }
    }
#pragma unroll
    for (int i = 56; i < 62; ++i) { run = run + (double)st.om[r][i]; st.cdf[r][i + 1] = (float)run; }
  }
  __syncthreads();
  if (tid < 128 * R && (tid & 127) < NI) {
    const int r = tid >> 7, k = tid & 127;
    const float u = u_rays ? u_rays[(row0 + (r < valid_rays ? r : 0)) * 128 + k] : ufine[k];
    const float* cdf = st.cdf[r];
    // searchsorted(cdf, u, right=True): number of entries <= u among 63 (RH:227)
    int lo = 0, hi = 63;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cdf[NSR_IDX(mid, 63)] <= u) lo = mid + 1; else hi = mid;
    }
    const int ind = lo;
    const int below = max(ind - 1, 0);
    const int above = min(ind, 62);
    const float c0 = cdf[NSR_IDX(below, 63)], c1 = cdf[NSR_IDX(above, 63)];
    const float b0 = bins(r, NSR_IDX(below, 63)), b1 = bins(r, NSR_IDX(above, 63));
    float denom = c1 - c0;
    if (denom < 1e-5f) denom = 1.0f;                            // RH:238-239
    const float t = (u - c0) / denom;
    st.zs[r][k] = b0 + t * (b1 - b0);                           // RH:241
    if (inds_out && r < valid_rays && ((write_mask >> r) & 1)) inds_out[r * inds_stride + k] = (int64_t)ind;
  }
  __syncthreads();
}

// std(z_samples, unbiased=False) RN:495, fp64 two-pass; result valid in lane 0 of waves 0 / 1 (ray = wave).
template <int NI = 128, typename ST>
__device__ __forceinline__ float zstd_wave(const ST& st, int r, int lane) {
  const bool in0 = lane < NI, in1 = lane + 64 < NI;             // NI = 128: both, always
  double s = (in0 ? (double)st.zs[r][lane] : 0.0) + (in1 ? (double)st.zs[r][lane + 64] : 0.0);
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) s += shfl_xor_f64(s, m);
  const double mean = s * (1.0 / NI);
  const double d0 = in0 ? (double)st.zs[r][lane] - mean : 0.0, d1 = in1 ? (double)st.zs[r][lane + 64] - mean : 0.0;
  double v = d0 * d0 + d1 * d1;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += shfl_xor_f64(v, m);
  return (float)sqrt(v * (1.0 / NI));
}

// z_vals = sort(cat([z_coarse, z_samples])) RN:477, exact and stable: the rank of an element is the number of
// elements that sort before it (ties by position in the concatenation).  Both halves are almost always already
// non-decreasing (z_coarse by construction, z_samples because the inverse CDF is monotone -- but adjacent bins can
// produce a 1-ulp inversion), and then the rank is the element's own index plus one binary search in the other
// half: ~25 VALU instructions instead of ~400, which matters because this code runs next to another workgroup's
// MFMA stream ("Two waves per SIMD").  A workgroup-wide check picks the path; any inversion or NaN
// falls back to the full rank count, so the result never depends on the sortedness assumption.
template <int R = 2, int NI = 128, int NS = 64, typename ST>
__device__ __forceinline__ void merge_sort_item(ST& st, int tid) {
  constexpr int NF = NS + NI;                        // merged depths per ray (the arrays keep their row strides)
  constexpr bool kPow2 = (NI & (NI - 1)) == 0 && (NS & (NS - 1)) == 0;      // (NI = 96: plain binary searches, r05)
  int bad = 0;
  for (int e = tid; e < NF * R; e += 256) {
    const int r = e / NF, k = e - r * NF;
    if (k < NS - 1) bad |= !(st.zc[r][k] <= st.zc[r][k + 1]);
    else if (k >= NS && k < NF - 1) bad |= !(st.zs[r][k - NS] <= st.zs[r][k - NS + 1]);
  }
  // workgroup-wide OR through four LDS words (st.res is free here).  Not __syncthreads_or: its library implementation
  // rebuilds the flat thread id from threadIdx.y/z, which keeps two more VGPRs alive through the whole kernel.
  int* orw = (int*)&st.res[0][0];
  const bool wave_bad = __builtin_amdgcn_ballot_w64(bad != 0) != 0;
  if ((tid & 63) == 0) orw[tid >> 6] = wave_bad ? 1 : 0;
  __syncthreads();
  const int any_bad = (orw[0] | orw[1]) | (orw[2] | orw[3]);
  __syncthreads();
  if (!any_bad) {
    for (int e = tid; e < NF * R; e += 256) {
      const int r = e / NF, k = e - r * NF;
      int rank;
      float x;
      if (k < NS) {                                  // own index + #(z_samples < x)
        x = st.zc[r][k];
        int lb = 0;
        if constexpr (kPow2) {
#pragma unroll
          for (int sft = NI / 2; sft > 0; sft >>= 1) lb += (st.zs[r][lb + sft - 1] < x) ? sft : 0;
          lb += (st.zs[r][lb] < x) ? 1 : 0;
        } else {
          int hi = NI;
          while (lb < hi) { const int mid = (lb + hi) >> 1; if (st.zs[r][NSR_IDX(mid, NI)] < x) lb = mid + 1; else hi = mid; }
        }
        rank = k + lb;
      } else {                                       // own index + #(z_coarse <= x)
        x = st.zs[r][k - NS];
        int ub = 0;
        if constexpr (kPow2) {
#pragma unroll
          for (int sft = NS / 2; sft > 0; sft >>= 1) ub += (st.zc[r][ub + sft - 1] <= x) ? sft : 0;
          ub += (st.zc[r][ub] <= x) ? 1 : 0;
        } else {
          int hi = NS;
          while (ub < hi) { const int mid = (ub + hi) >> 1; if (st.zc[r][NSR_IDX(mid, NS)] <= x) ub = mid + 1; else hi = mid; }
        }
        rank = (k - NS) + ub;
      }
      st.zf[r][NSR_IDX(rank, NF)] = x;
    }
  } else {
    for (int e = tid; e < NF * R; e += 256) {
      const int r = e / NF, k = e - r * NF;
      const float x = (k < NS) ? st.zc[r][k] : st.zs[r][k - NS];
      int rank = 0;
      for (int j = 0; j < NS; ++j) {
        const float y = st.zc[r][j];
        rank += (y < x) || (y == x && j < k);
      }
      for (int j = 0; j < NI; ++j) {
        const float y = st.zs[r][j];
        rank += (y < x) || (y == x && (j + NS) < k);
      }
      st.zf[r][NSR_IDX(rank, NF)] = x;
    }
  }
  __syncthreads();
}

// RN:441 / RN:443: the coarse sample depth for table value t
__device__ __forceinline__ float coarse_z(float near_, float far_, float t, int lindisp) {
  if (lindisp) return 1.0f / (((1.0f / near_) * (1.0f - t)) + ((1.0f / far_) * t));
  return (near_ * (1.0f - t)) + (far_ * t);
}

// RN:447-459 (perturb > 0): one stratified sample per interval between the mid-points of the coarse depths of both rays of
// an item; t_rand [rays][64] are the caller's draws.  Called by the whole workgroup once st.zc is complete.
template <int NS = 64, typename ST>
__device__ __forceinline__ void perturb_coarse_z(ST& st, const float* t_rand, long long row0, int valid, int tid) {
  if constexpr (NS == 64) {
    const int r = (tid >> 6) & 1, i = tid & 63;
    float zn = 0.0f;
    if (tid < 128) {
      const float* zr = st.zc[r];
      const float lower = i > 0 ? 0.5f * (zr[i] + zr[i - 1]) : zr[0];
      const float upper = i < 63 ? 0.5f * (zr[i + 1] + zr[i]) : zr[63];
      zn = lower + (upper - lower) * t_rand[(row0 + (r < valid ? r : 0)) * 64 + i];
    }
    __syncthreads();
    if (tid < 128) st.zc[r][i] = zn;
    __syncthreads();
  } else {                                       // N_samples 32 / 128: thread tid owns sample tid % NS of ray tid / NS (2 NS <= 256)
    const int r = tid / NS, i = tid - r * NS;
    float zn = 0.0f;
    if (tid < 2 * NS) {
      const float* zr = st.zc[r];
      const float lower = i > 0 ? 0.5f * (zr[i] + zr[i - 1]) : zr[0];
      const float upper = i < NS - 1 ? 0.5f * (zr[i + 1] + zr[i]) : zr[NS - 1];
      zn = lower + (upper - lower) * t_rand[(row0 + (r < valid ? r : 0)) * NS + i];
    }
    __syncthreads();
    if (tid < 2 * NS) st.zc[r][i] = zn;
    __syncthreads();
  }
}

// get_rays RH:156-165 for pixel (row, col); cam = {c2w[12], fx, fy, cx, cy}
__device__ __forceinline__ void gen_ray(const float* __restrict__ c2w, float fx, float fy, float cx, float cy,
                                        int row, int col, float (&o)[3], float (&d)[3]) {
  const float dx = ((float)col - cx) / fx;
  const float dy = -(((float)row - cy) / fy);
  const float dz = -1.0f;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    d[a] = ((dx * c2w[a * 4 + 0]) + (dy * c2w[a * 4 + 1])) + (dz * c2w[a * 4 + 2]);
    o[a] = c2w[a * 4 + 3];
  }
}

struct RenderArgs {
  const float* nets;        // packed networks: coarse | fine | fine^T, `net_stride` BYTES apart
  long long net_stride;
  const float* aux[2];      // aux blocks (coarse, fine)
  const float* tcoarse;     // [64]
  const float* ufine;       // [128]
  const float* rays_o;      // [N,3]   (RAYS mode)
  const float* rays_d;      // [N,3]
  const float* c2w;         // [V,3,4] (CAMERA mode)
  float fx, fy, cx, cy;
  int H, W;
  long long n_rays;
  float near_, far_;
  int fine;                 // 0: coarse only
  int camera;               // 1: generate rays in-kernel
  int white_bkgd;           // RN:384-385: rgb_map += 1 - acc_map
  int lindisp;              // RN:443: coarse samples linear in inverse depth
  float *rgb, *disp, *acc, *rgb0, *disp0, *acc0, *z_std;
  float *dbg_w0, *dbg_zs, *dbg_zf, *dbg_raw0, *dbg_raw;
  long long* dbg_inds;
  float* zf_scratch;        // k_render16: [grid][chunk][192] sorted fine z values between the two phases of a chunk
  int chunk;                // k_render16: rays per workgroup per phase
  unsigned* sched_flags;    // k_render16p: [2][3 * super_rays] ready / taken generations of the z hand-off slots (zeroed per launch)
  unsigned* status;         // k_render16p: number of fine tasks that recomputed their coarse pass (hand-off not there in time)
  int super_lg;             // k_render16p: log2 of the rays per super-chunk
  int spin_max;             // k_render16p: looks at the ready flag before a fine task recomputes locally
  unsigned epoch;           // global phases: launch counter of the handle (1..4094), high bits of every generation tag;
                            // advanced ON THE DEVICE by k_set_args (epoch_counter), so that every replay of a captured
                            // launch gets a fresh one too
  unsigned* epoch_counter;  // device word behind `epoch` (null: the schedule is not in use)
  unsigned long long* work_counter;   // head of the work queue (chunks / items), zeroed by k_set_args; 64-bit: no wrap for any n_rays  // per-ray inputs of the options the reference's render() has beyond the deterministic test-time path; all nullable and
  // all read by the x32-structured kernels only (k_render / _b3 / _h2 and their VJPs) -- the launcher routes accordingly
  const float* viewdirs;    // [N,3] (RAYS mode): view directions given by the caller instead of rays_d / |rays_d| -- they are
                            // another camera's (c2w_staticcam, RN:91-96) or those of the rays before ndc_rays (RN:89-103)
  const float* t_rand;      // [N,64]  perturb > 0: stratified jitter in [0, 1) per coarse sample (RN:447-459)
  const float* u_rays;      // [N,128] sample_pdf with det=False: the uniforms (RH:211)
  const float* noise0;      // [N,64]  raw_noise_std * randn added to the coarse densities before the relu (RN:365-374)
  const float* noise1;      // [N,192] ... to the fine densities
  const float* near_rays;   // [N] per-ray bounds (RN:106-108: near / far may be arrays); both or neither
  const float* far_rays;
  // f16x2 range safety net (x32-structured kernels).  The f16x2 kernels REPORT: an item (2 rays) in which any network
  // output came out NaN -- a scaled activation / gradient beyond the fp16 range, nsr_h2.inc -- is appended to ovf_items
  // (ovf_stat[0] = items appended by this launch, zeroed by k_set_args; [1] points, [2] rays, [3] items beyond ovf_cap:
  // cumulative).  The launcher then runs the fp32 kernel of the same template over exactly that list (item_list /
  // item_count, read on the device: no host round trip), which overwrites the items' outputs.
  unsigned long long* ovf_items;
  unsigned* ovf_stat;
  unsigned ovf_cap;
  const unsigned long long* item_list;   // fallback launch: the items to render instead of 0 .. ceil(n_rays / 2) - 1
  const unsigned* item_count;            // ... how many (device word written by the launch before)
  unsigned item_cap;
};

__device__ __forceinline__ void load_aux(char* smem, const RenderArgs& a, int tid) {
  float* dst = (float*)(smem + kLdsAux);
  for (int i = tid; i < kAuxFloats; i += 256) {
    dst[i] = a.aux[0][i];
    dst[kAuxFloats + i] = a.aux[1][i];
  }
}

// ------------------------------------------------------------------------------------------------------
// The fused persistent render kernel.
// ------------------------------------------------------------------------------------------------------
// Argument block transport: the launcher writes the block to device memory with a 1-thread kernel
// (stream-ordered, no host staging buffer to keep alive), and the render kernel reads its fields with scalar
// loads at the point of use -- by-value kernel arguments were all preloaded into SGPRs and cost 70 more SGPR
// spills inside the MFMA passes.
#if NSR_UNIT_F32     // (launched by the API's unit only)
__global__ void k_set_args(const RenderArgs a, RenderArgs* dst) {
  *dst = a;
  *a.work_counter = 0ull;
  if (a.ovf_stat) a.ovf_stat[0] = 0u;
  if (a.epoch_counter) dst->epoch = *a.epoch_counter = *a.epoch_counter % 4094u + 1u;
}
#endif

// Work queue of the x32-structured kernels: the next item (2 rays) of this launch as  item | write mask << 62  (bit r of
// the mask: ray r of the item is this launch's to write), or -1 when there is none left.  A fallback launch (item_list)
// hands out the entries the f16x2 kernel reported: it renders the whole item and writes ONLY the reported rays, so a ray
// that stayed inside the fp16 range keeps the f16x2 kernel's bits whatever its neighbour did.  Thread 0 only.
constexpr long long kItemMask = (1ll << 62) - 1;
__device__ __forceinline__ long long queue_next_item(const RenderArgs& q) {
  const unsigned long long v = atomicAdd(q.work_counter, 1ull);
  if (q.item_list) {
    unsigned n = *q.item_count;
    n = n < q.item_cap ? n : q.item_cap;
    return v < (unsigned long long)n ? (long long)q.item_list[v] : -1ll;
  }
  return v < (unsigned long long)((q.n_rays + 1) >> 1) ? (long long)(v | (3ull << 62)) : -1ll;
}
__device__ __forceinline__ long long queue_items(const RenderArgs& q) {
  if (q.item_list) { const unsigned n = *q.item_count; return (long long)(n < q.item_cap ? n : q.item_cap); }
  return (q.n_rays + 1) >> 1;
}
// f16x2: a NaN network output marks the ray (LDS counters ovf[0], ovf[1] of the item's two rays: affected points)
__device__ __forceinline__ void range_mark(int* ovf, float chk, int lane) {
#ifndef NSR_EXP_NO_RANGE     // (timing experiment: the kernels without the safety net)
  if (chk != chk && lane < 32) atomicAdd(ovf, 1);
#endif
}
// ... and the item is reported once it is complete (thread 0).  Returns the mask of the rays that could NOT be listed (the
// list is full: only a launch captured into a graph before the list was grown to its size can get there, see
// ensure_range in nsr_api.hip); the caller poisons them -- a ray the fp32 kernel will not render again must not keep what
// the f16x2 kernel made of out-of-range activations (sigma = NaN composites as zero density: a finite, wrong pixel).
__device__ __forceinline__ unsigned range_report(const RenderArgs& a, int* ovf, long long item, int valid) {
#ifdef NSR_EXP_NO_RANGE
  return 0u;
#endif
  const int p0 = ovf[0], p1 = valid == 2 ? ovf[1] : 0;      // (an item's missing second ray repeats the first)
  if ((p0 | p1) != 0 && a.ovf_stat) {
    const unsigned long long mask = (p0 ? 1ull : 0ull) | (p1 ? 2ull : 0ull);
    const unsigned n = atomicAdd(a.ovf_stat, 1u);
    atomicAdd(a.ovf_stat + 1, (unsigned)(p0 + p1));
    if (n < a.ovf_cap) {
      a.ovf_items[n] = (unsigned long long)item | (mask << 62);
      atomicAdd(a.ovf_stat + 2, (unsigned)((p0 != 0) + (p1 != 0)));
    } else {
      atomicAdd(a.ovf_stat + 3, 1u);
      return (unsigned)mask;
    }
  }
  return 0u;
}
// NaN into every output of the rays of `mask` (thread 0, after the item's own output stores of the same wave)
__device__ __forceinline__ void range_poison(const RenderArgs& a, long long ray0, unsigned mask) {
  const float qn = __builtin_nanf("");
  for (int r = 0; r < 2; ++r) {
    if (!((mask >> r) & 1u)) continue;
    const long long rr = ray0 + r;
    float* const v3[2] = {a.rgb, a.rgb0};
    float* const v1[5] = {a.disp, a.acc, a.disp0, a.acc0, a.z_std};
    for (int k = 0; k < 2; ++k) if (v3[k]) { v3[k][rr * 3] = qn; v3[k][rr * 3 + 1] = qn; v3[k][rr * 3 + 2] = qn; }
    for (int k = 0; k < 5; ++k) if (v1[k]) v1[k][rr] = qn;
  }
}

#ifdef NSR_PHASE_TIMING      // diagnostic build: per-workgroup cycle totals of the item phases (thread 0), see tools
#define NSR_T(i) do { if (threadIdx.x == 0) { const long long t_ = clock64(); tacc[i] += t_ - tlast; tlast = t_; } } while (0)
#else
#define NSR_T(i) do { } while (0)
#endif

// NI: importance samples per ray.  128 = the YCB-V configuration; 64 / 32 = kernels specialised to N_importance = 64 / 32
// (RN:474: 64 + NI fine samples per ray, ceil(2 (64 + NI) / 128) = 2 fine passes per item instead of 3; a wave's 32 points
// belong to one ray because 64 + NI is a multiple of 32; with NI = 32 the last two waves of the second pass have no points:
// they run the pass on a repeated point -- the weight ring is consumed in lock-step -- and store nothing).
// NS: coarse samples per ray (N_samples, RN:439).  64 = the YCB-V configuration; 32 / 128 (r05): one coarse pass with two idle waves
// / two coarse passes per item, item state ItemStateBig for 128.
template <int MODE, int NI = 128, int NS = 64>
__device__ __forceinline__ void render32_body(const RenderArgs* __restrict__ ap, char* smem) {
  constexpr bool B3 = MODE == kMlpB3;
  constexpr int NF = NS + NI;                              // fine samples per ray
  constexpr int NP = (2 * NF + 127) / 128;                 // fine passes per item
  constexpr int NPC = (2 * NS + 127) / 128;                // coarse passes per item
  constexpr int LNS = NS == 32 ? 5 : (NS == 64 ? 6 : 7);   // N_samples is 32, 64 or 128
  typedef typename ItemStateFor<NS>::type ST;
  static_assert((NS == 32 || NS == 64 || NS == 128) && NF % 32 == 0 && NS <= ST::kNS && NF <= ST::kNF,
                "a wave's points belong to one ray; the item state holds kNS coarse / kNF fine samples per ray");
  const RenderArgs& a_setup = *ap;
#ifdef NSR_PHASE_TIMING
  long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tpass[4] = {0, 0, 0, 0};
  long long tlast = clock64();
#define NSR_TPASS tpass
#else
#define NSR_TPASS nullptr
#endif
  const int tid0 = threadIdx.x;
  const int lane = tid0 & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int j = lane & 31;
  ST& st = *(ST*)(smem + kLdsState);

  const long long n_rays = a_setup.n_rays;
  if ((long long)blockIdx.x >= queue_items(a_setup)) return;
  const int fine = a_setup.fine;
  int* ovf = (int*)&st.ray[1][14];                       // f16x2: [2] points with NaN network outputs, per ray of the current item

  Ring rg;
  ring_init(rg, smem, a_setup.nets, a_setup.net_stride, fine ? NPC + NP : NPC, wave, lane);
  rg.pn0 = NPC; rg.pn1 = NPC + NP;

  f32x4 A0[4], A1[4];
  if constexpr (B3) ring_start<kRingSlots, kStreamSlabsB3, kStreamSlabsB3Bwd>(rg, A0, lane);
  else if constexpr (MODE == kMlpH2) ring_start<kRingSlots, kStreamSlabs, kStreamSlabs, kH2RingLag>(rg, A0, lane);
  else ring_start(rg, A0, lane);   // weights start streaming while the aux blocks and tables are staged

  load_aux(smem, a_setup, tid0);
  if (tid0 < NS) st.tcoarse[tid0] = a_setup.tcoarse[tid0];
  if (tid0 < NI) st.ufine[tid0] = a_setup.ufine[tid0];
  __syncthreads();
  const float* aux_c = (const float*)(smem + kLdsAux);

  // items come from a global counter: nothing forces the workgroups to progress at the same rate
  long long* item_slot = (long long*)&st.ray[0][14];     // 8-byte slot in the unused tail of ray 0's block
  auto next_item = [&]() -> long long {
    if (opaque_v(tid0) == 0) *item_slot = queue_next_item(*opaque_s(ap));
    __syncthreads();
    const long long v = uniform64(*item_slot);
    __syncthreads();
    return v;
  };
  long long packed = next_item();
  int pass = 0;              // 0 .. NPC - 1 = coarse pass(es), NPC .. NPC + NP - 1 = fine passes of the current item
#pragma unroll 1
  while (packed != -1ll) {
    const long long item = packed & kItemMask;
    const long long ray0 = item * 2;
    const int valid = (ray0 + 1 < n_rays) ? 2 : 1;
    const int wmask = (int)((unsigned long long)packed >> 62) & (valid == 2 ? 3 : 1);     // rays this launch writes
    auto wr = [&](int r) { return ((wmask >> r) & 1) != 0; };
    if (pass == 0) {
      // ---- stage the two rays --------------------------------------------------------------------
      const RenderArgs& a = *opaque_s(ap);                // see opaque_v / opaque_s
      const int tid = opaque_v(tid0);
      const float near_ = a.near_, far_ = a.far_;
      if (tid < 2) {
        const long long rr = ray0 + (tid < valid ? tid : 0);
        float o[3], d[3];
        if (a.camera) {
          const long long hw = (long long)a.H * a.W;
          const long long v = rr / hw;
          const int pix = (int)(rr - v * hw);
          gen_ray(a.c2w + v * 12, a.fx, a.fy, a.cx, a.cy, pix / a.W, pix % a.W, o, d);
        } else {
#pragma unroll
          for (int c = 0; c < 3; ++c) { o[c] = a.rays_o[rr * 3 + c]; d[c] = a.rays_d[rr * 3 + c]; }
        }
        const float nrm = sqrtf(((d[0] * d[0]) + (d[1] * d[1])) + (d[2] * d[2]));   // torch.norm RN:97, RN:361
        const float* vd = a.camera ? nullptr : a.viewdirs;                           // given view directions: RN:91-103
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          st.ray[tid][c] = o[c]; st.ray[tid][3 + c] = d[c];
          st.ray[tid][6 + c] = vd ? vd[rr * 3 + c] : d[c] / nrm;
        }
        st.ray[tid][9] = a.near_rays ? a.near_rays[rr] : near_;       // per-ray bounds (RN:106-108) or the call's scalars
        st.ray[tid][10] = a.near_rays ? a.far_rays[rr] : far_;
        st.ray[tid][11] = nrm;
        st.ray[tid][12] = a.white_bkgd ? 1.0f : 0.0f;
      }
      if (MODE == kMlpH2 && (tid == 64 || tid == 65)) ovf[tid - 64] = 0;
      if (tid < 2 * NS) {
        const int r = tid >> LNS, i = tid & (NS - 1);
        const float t = st.tcoarse[i];
        const long long rb = ray0 + (r < valid ? r : 0);
        st.zc[r][i] = coarse_z(a.near_rays ? a.near_rays[rb] : near_, a.near_rays ? a.far_rays[rb] : far_, t, a.lindisp);
      }
      __syncthreads();
      if (a.t_rand) perturb_coarse_z<NS>(st, a.t_rand, ray0, valid, tid);
      NSR_T(0);
    }

    // ---- one network pass: 128 points -------------------------------------------------------------
    //   coarse p: point q = 128 p + 32w + j -> ray q/NS, sample q%NS   (RN:463-466; NS = 64: ray w>>1, sample 32 (w&1) + j)
    //   fine p:   point q = 128 (p - NPC) + 32w + j -> ray q/NF, sample q%NF (RN:478-483)
    {
      int r, i;
      const float* zsrc;
      float* dst;
      bool real = true;                                    // (NI = 32: the second pass has points for two waves only)
      if ((NPC == 1) ? (pass == 0) : (pass < NPC)) {
        if constexpr (NS == 64) {
          r = wave >> 1; i = 32 * (wave & 1) + j;
        } else {
          int q0 = 128 * pass + 32 * wave;
          if (2 * NS % 128 != 0 && q0 >= 2 * NS) { q0 = 2 * NS - 32; real = false; }      // (N_samples = 32: two waves have no points)
          r = q0 >> LNS; i = (q0 & (NS - 1)) + j;
        }
        zsrc = &st.zc[r][i]; dst = st.rawc[r][i];
      } else {
        int q0 = 128 * (pass - NPC) + 32 * wave;
        if (2 * NF % 128 != 0 && q0 >= 2 * NF) { q0 = 2 * NF - 32; real = false; }
        r = q0 / NF; i = q0 - r * NF + j;
        zsrc = &st.zf[r][i]; dst = st.rawf[r][i];
      }
      const float z = *zsrc;
      const float* ry = st.ray[r];
      float raw[4];
      mlp_pass<false, MODE>(rg, aux_c + (((NPC == 1) ? (pass == 0) : (pass < NPC)) ? 0 : kAuxFloats), A0, A1, lane, ry[0] + ry[3] * z, ry[1] + ry[4] * z,
               ry[2] + ry[5] * z, ry[6], ry[7], ry[8], raw, nullptr, 0, NSR_TPASS);
      if (lane < 32 && real) *(f32x4*)dst = f32x4{raw[0], raw[1], raw[2], raw[3]};
      if constexpr (MODE == kMlpH2) { if (real) range_mark(ovf + r, (raw[0] + raw[1]) + (raw[2] + raw[3]), lane); }
    }
    NSR_T(1);

    const RenderArgs& a = *opaque_s(ap);                  // nothing below may be hoisted above the network pass
    const int tid = opaque_v(tid0);
    if (NPC > 1 && pass < NPC - 1) {
      ++pass;                                              // (N_samples = 128: the item's second coarse pass)
    } else if ((NPC == 1) ? (pass == 0) : (pass < NPC)) {
      __syncthreads();
      if (a.dbg_raw0) {
        if constexpr (NS == ST::kNS) {
          for (int idx = tid; idx < 2 * NS * 4; idx += 256) if (wr(idx >> (LNS + 2))) a.dbg_raw0[ray0 * (NS * 4) + idx] = (&st.rawc[0][0][0])[idx];
        } else {
          for (int idx = tid; idx < 2 * NS * 4; idx += 256)
            if (wr(idx >> (LNS + 2))) a.dbg_raw0[ray0 * (NS * 4) + idx] = (&st.rawc[idx >> (LNS + 2)][0][0])[idx & (NS * 4 - 1)];
        }
        __syncthreads();
      }
      composite<NS, 2, ST::kNS>(st, &st.zc[0][0], &st.rawc[0][0][0], &st.w0[0][0], &st.tf[0][0], tid, a.noise0, ray0, valid);
      if (tid < 16 && wr(tid >> 3)) {
        const int r = tid >> 3, c = tid & 7;
        const long long rr = ray0 + r;
        const float v = st.res[r][c];
        float* rgb_dst = fine ? a.rgb0 : a.rgb;
        float* disp_dst = fine ? a.disp0 : a.disp;
        float* acc_dst = fine ? a.acc0 : a.acc;
        if (c < 3) { if (rgb_dst) rgb_dst[rr * 3 + c] = v; }
        else if (c == 3) { if (disp_dst) disp_dst[rr] = v; }
        else if (c == 4) { if (acc_dst) acc_dst[rr] = v; }
      }
      if (a.dbg_w0)
        for (int idx = tid; idx < 2 * NS; idx += 256)
          if (wr(idx >> LNS)) a.dbg_w0[ray0 * NS + idx] = NS == ST::kNS ? (&st.w0[0][0])[idx] : st.w0[idx >> LNS][idx & (NS - 1)];
      if (!fine) {
        if (MODE == kMlpH2 && tid == 0) { if (const unsigned m = range_report(a, ovf, item, valid)) range_poison(a, ray0, m); }
        __syncthreads();
        if constexpr (NPC > 1) pass = 0;                   // (N_samples = 128: the next item starts with its first coarse pass)
        packed = next_item(); continue;
      }
      NSR_T(2);

      // ---- hierarchical resampling ----------------------------------------------------------------
#ifdef NSR_PHASE_TIMING
      int64_t* inds = nullptr;                 // dbg_inds carries the cycle totals in this build
#else
      int64_t* inds = (int64_t*)a.dbg_inds;
#endif
      sample_pdf_item<2, NI, NS>(st, st.ufine, &st.w0[0][1], ST::kNS,
                             [&](int r, int k) { return 0.5f * (st.zc[r][k + 1] + st.zc[r][k]); },   // RN:473
                             inds ? inds + ray0 * NI : nullptr, NI, tid, valid, a.u_rays, ray0, wmask);
      NSR_T(3);
      if (wave < 2) {
        const float sd = zstd_wave<NI>(st, wave, lane);
        if (lane == 0 && wr(wave) && a.z_std) a.z_std[ray0 + wave] = sd;
      }
      if (a.dbg_zs)
        for (int idx = tid; idx < 2 * 128; idx += 256)      // st.zs rows are 128 apart, the tap's NI
          if (wr(idx >> 7) && (idx & 127) < NI) a.dbg_zs[(ray0 + (idx >> 7)) * NI + (idx & 127)] = (&st.zs[0][0])[idx];
      NSR_T(4);
      merge_sort_item<2, NI, NS>(st, tid);
      if (a.dbg_zf)
        for (int idx = tid; idx < 2 * ST::kNF; idx += 256)      // st.zf rows are kNF apart, the tap's NF
          if (wr(idx / ST::kNF) && idx % ST::kNF < NF) a.dbg_zf[(ray0 + idx / ST::kNF) * NF + idx % ST::kNF] = (&st.zf[0][0])[idx];
      NSR_T(5);
      pass = NPC;
    } else if (pass < NPC + NP - 1) {
      ++pass;
    } else {
      __syncthreads();
      if (a.dbg_raw) {
        for (int idx = tid; idx < 2 * ST::kNF * 4; idx += 256)      // st.rawf rows are kNF x 4 apart, the tap's NF x 4
          if (wr(idx / (ST::kNF * 4)) && idx % (ST::kNF * 4) < NF * 4)
            a.dbg_raw[(ray0 + idx / (ST::kNF * 4)) * (NF * 4) + idx % (ST::kNF * 4)] = (&st.rawf[0][0][0])[idx];
        __syncthreads();
      }
      composite<NF, 2, ST::kNF>(st, &st.zf[0][0], &st.rawf[0][0][0], &st.wf[0][0], &st.tf[0][0], tid, a.noise1, ray0, valid);
      if (tid < 16 && wr(tid >> 3)) {
        const int r = tid >> 3, c = tid & 7;
        const long long rr = ray0 + r;
        const float v = st.res[r][c];
        if (c < 3) { if (a.rgb) a.rgb[rr * 3 + c] = v; }
        else if (c == 3) { if (a.disp) a.disp[rr] = v; }
        else if (c == 4) { if (a.acc) a.acc[rr] = v; }
      }
      if (MODE == kMlpH2 && tid == 0) { if (const unsigned m = range_report(a, ovf, item, valid)) range_poison(a, ray0, m); }
      __syncthreads();
      NSR_T(6);
      pass = 0;
      packed = next_item();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the workgroup
#ifdef NSR_PHASE_TIMING
  if (tid0 == 0 && a_setup.dbg_raw == nullptr && a_setup.dbg_inds)      // diagnostic hijack: dbg_inds receives [grid][8] cycle totals
    for (int i = 0; i < 8; ++i) a_setup.dbg_inds[blockIdx.x * 8 + i] = tacc[i];
  if (tid0 == 0 && a_setup.dbg_raw == nullptr && a_setup.dbg_inds)      // ... and [grid][4] network-pass breakdowns behind them
    for (int i = 0; i < 4; ++i) a_setup.dbg_inds[(long long)gridDim.x * 8 + blockIdx.x * 4 + i] = tpass[i];
#endif
#undef NSR_TPASS
}

// The x32 forward kernels: one body, three arithmetics (fp32 MFMAs; bf16 MFMAs on three-way split operands, nsr_b3.inc; fp16
// MFMAs on two-way split operands with power-of-two range management, nsr_h2.inc) and the sample counts with kernels of their
// own -- N_importance = 64 / 32 / 96 at N_samples = 64 (two or three fine passes per item), N_samples = 32 with N_importance =
// 64, N_samples = 128 with N_importance = 128 (ItemStateBig): f16x2 handles, and the bf16x3 kernels their range safety net
// falls back to (fp32's exponent range -- no failure domain -- and fp32-grade error at 1.7x the fp32-MFMA kernels' speed).
#define NSR_RENDER_KERNEL_1(NAME, ...)                                                               \
  __global__ void __launch_bounds__(256, 1) NAME(const RenderArgs* __restrict__ ap) {                \
    extern __shared__ __attribute__((aligned(16))) char smem[];                                      \
    render32_body<__VA_ARGS__>(ap, smem);                                                            \
  }
#define NSR_RENDER_KERNEL_0(NAME, ...) __global__ void __launch_bounds__(256, 1) NAME(const RenderArgs* __restrict__ ap);
#define NSR_RENDER_KERNEL(UNIT, NAME, ...) NSR_CAT(NSR_RENDER_KERNEL_, UNIT)(NAME, __VA_ARGS__)
NSR_RENDER_KERNEL(NSR_UNIT_F32, k_render, kMlpF32)
NSR_RENDER_KERNEL(NSR_UNIT_B3, k_render_b3, kMlpB3)
NSR_RENDER_KERNEL(NSR_UNIT_H2, k_render_h2, kMlpH2)
NSR_RENDER_KERNEL(NSR_UNIT_H2, k_render_h2_n64, kMlpH2, 64)
NSR_RENDER_KERNEL(NSR_UNIT_H2, k_render_h2_n32, kMlpH2, 32)
NSR_RENDER_KERNEL(NSR_UNIT_B3, k_render_b3_n64, kMlpB3, 64)
NSR_RENDER_KERNEL(NSR_UNIT_B3, k_render_b3_n32, kMlpB3, 32)
NSR_RENDER_KERNEL(NSR_UNIT_H2, k_render_h2_n96, kMlpH2, 96, 64)
NSR_RENDER_KERNEL(NSR_UNIT_B3, k_render_b3_n96, kMlpB3, 96, 64)
NSR_RENDER_KERNEL(NSR_UNIT_H2, k_render_h2_c32_n64, kMlpH2, 64, 32)
NSR_RENDER_KERNEL(NSR_UNIT_B3, k_render_b3_c32_n64, kMlpB3, 64, 32)
NSR_RENDER_KERNEL(NSR_UNIT_H2, k_render_h2_c128_n128, kMlpH2, 128, 128)
NSR_RENDER_KERNEL(NSR_UNIT_B3, k_render_b3_c128_n128, kMlpB3, 128, 128)
#undef NSR_RENDER_KERNEL
#undef NSR_RENDER_KERNEL_0
#undef NSR_RENDER_KERNEL_1

// ------------------------------------------------------------------------------------------------------
// Backward (input-side VJP) of one network pass.  Same register-chained scheme with W^T as the A operand:
//   G_in^T[K_in x 32pts] = W^T[K_in x K_out] * (G_out^T (.) relu')        (weights are constants: no dW)
// The gradient fragment of layer l (C layout) is masked with the relu pattern captured in the forward pass
// and is, register for register, the B operand of layer l-1's transposed GEMM.
// Stream order: views^T 1 (encoding block) + 8 | feature^T 16 | L7^T 16 | L6^T 16 | L5^T 4 (2 encoding blocks) + 16 |
//               L4^T..L1^T 64 | L0^T 4 (2 blocks encoding)  = 145 slabs.  The encoding rows are their own small segments, run
//               BEFORE the 8-block GEMM of the same layer: 8 accumulator blocks instead of 10 (r02's 52 VGPR spills).
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x16 apply_mask(f32x16 x, unsigned word, int shift) {
  f32x16 r;
#pragma unroll
  for (int i = 0; i < 16; ++i)                        // bit 31 - (shift + i) set = unit off (relu_mask)
    r[i] = ((word >> (31 - (shift + i))) & 1u) ? 0.0f : x[i];
  return r;
}

// d(encoding)/d(x): this lane half holds G for sin features (h=0) or cos features (h=1) of x[ax]; identity
// terms sit in registers 3L and 3L+1.  Returns the half's contribution; the caller adds the two halves.
template <int NFREQ>
__device__ __forceinline__ void embed_bwd(const float (&x)[3], const float* G /* NFREQ*3+2 values */, int h,
                                          float (&out)[3]) {
  out[0] = h ? 0.0f : G[3 * NFREQ];
  out[1] = h ? 0.0f : G[3 * NFREQ + 1];
  out[2] = h ? G[3 * NFREQ] : 0.0f;
#pragma unroll
  for (int L = 0; L < NFREQ; ++L)
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {     // d sin = f cos, d cos = -f sin
      const float f = (float)(1 << L);
      out[ax] = __builtin_fmaf(f * G[3 * L + ax], enc_trig(enc_domain(x[ax], NFREQ) * f, 1 + h), out[ax]);
    }
}

#include "nsr_h2_bwd.inc"

// MODE kMlpB3: the transposed GEMMs on bf16 MFMAs (nsr_b3.inc): every 9- or 10-block segment becomes an 8-block GEMM for
// the hidden features plus a 4-block GEMM for the encoding rows (1 or 2 real blocks; the packer pads with zero weights).
// MODE kMlpH2: fp16 MFMAs, per-point normalised gradients (nsr_h2_bwd.inc).
template <int MODE = kMlpF32>
__device__ __forceinline__ void mlp_bwd_pass(Ring& rg, const float* aux, f32x4 (&A0)[4], f32x4 (&A1)[4], int lane,
                                             const uint4* mask_src /* uniform */, int mask_tid, float g0, float g1,
                                             float g2, float gs, const float* ry /* LDS: the ray block */,
                                             const float* zrow /* LDS: z of this wave's 32 points */,
                                             float (&dp)[3], float (&dv)[3]) {
  if constexpr (MODE == kMlpH2) {
    mlp_bwd_pass_h2(rg, aux, A0, A1, lane, mask_src, mask_tid, g0, g1, g2, gs, ry, zrow, dp, dv);
    return;
  }
  constexpr bool B3 = MODE == kMlpB3;
  const int h = lane >> 5;
  const int h4 = aux_half(lane);
  // rgb_linear^T (VALU) masked by the views layer's relu pattern
  f32x16 gv[4];
  {
    const uint4 mk = mask_src[8 * 256 + mask_tid];
    const unsigned mw[2] = {mk.x, mk.y};
#pragma unroll
    for (int mo = 0; mo < 4; ++mo)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const f32x4 w0 = *(const f32x4*)(aux + kAuxWRgb + 0 * 128 + (mo * 4 + rq) * 8 + h4);
        const f32x4 w1 = *(const f32x4*)(aux + kAuxWRgb + 1 * 128 + (mo * 4 + rq) * 8 + h4);
        const f32x4 w2 = *(const f32x4*)(aux + kAuxWRgb + 2 * 128 + (mo * 4 + rq) * 8 + h4);
#pragma unroll
        for (int ri = 0; ri < 4; ++ri) {
          const float v = __builtin_fmaf(w2[ri], g2, __builtin_fmaf(w1[ri], g1, w0[ri] * g0));
          const int r = rq * 4 + ri;
          gv[mo][r] = ((mw[mo >> 1] >> (31 - ((mo & 1) * 16 + r))) & 1u) ? 0.0f : v;     // set bit = unit off
        }
      }
  }
  // views^T: 256 feature rows (blocks 0-7) + 32 direction-encoding rows (block 8), K = 128
  f32x16 gin[8];
  f32x16 acc[8];              // gradient w.r.t. the 256 hidden features (the encoding rows have their own small GEMMs)
  auto gv_src = [&](int kb, int i) { return gv[(kb >> 1) & 3][8 * (kb & 1) + i]; };
  auto gin_src = [&](int kb, int i) { return gin[(kb >> 1) & 7][8 * (kb & 1) + i]; };
  // B3: the encoding rows are their own 4-block GEMM, run BEFORE the 8-block one of the same layer so that its
  // accumulators are dead again when the big one starts; embed_bwd is linear, so the position-encoding gradients of L5^T
  // and L0^T go through it separately and are added as d/dp
  auto point = [&](float (&p)[3]) {
    const float z = zrow[opaque_v(lane) & 31];
    p[0] = ry[0] + ry[3] * z; p[1] = ry[1] + ry[4] * z; p[2] = ry[2] + ry[5] * z;
  };
  float dp5[3] = {0.0f, 0.0f, 0.0f};
  {
    float Gd[16];
    if constexpr (B3) {
      {
        f32x16 ae[4];
        gemm_b3<4, 2, true, false>(rg, A0, A1, gv_src, ae, lane);
#pragma unroll
        for (int t = 0; t < 16; ++t) Gd[t] = ae[0][t];
      }
      gemm_b3<8, 4, true>(rg, A0, A1, gv_src, acc, lane);
#pragma unroll
      for (int mo = 0; mo < 8; ++mo) gin[mo] = acc[mo];          // feature_linear has no activation
    } else {
      {                                                          // the 32 direction-encoding rows: their own 1-block segment,
        f32x16 ae[1];                                            // run first (its accumulator is dead when the big one starts)
        seg<1, 16, kRingSlots, true>(rg, A0, A1, BRegs16<4>{gv}, ae, lane);
#pragma unroll
        for (int t = 0; t < 16; ++t) Gd[t] = ae[0][t];
      }
      seg<8, 16, kRingSlots, true>(rg, A0, A1, BRegs16<4>{gv}, acc, lane);
#pragma unroll
      for (int mo = 0; mo < 8; ++mo) gin[mo] = acc[mo];          // feature_linear has no activation
    }
    const float v[3] = {ry[6], ry[7], ry[8]};              // re-read where needed: not kept alive across the GEMMs
    float part[3];
    embed_bwd<kMultiresViews>(v, Gd, h, part);
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) dv[ax] = part[ax] + __shfl_xor(part[ax], 32);
  }

  // idx: 0 feature^T (+alpha head), 1 L7^T, 2 L6^T, 3 L5^T (10 blocks: first use of acc[8..9]), 4..7 L4^T..L1^T
#pragma unroll 1
  for (int idx = 0; idx < 8; ++idx) {
    const uint4 mk = mask_src[(7 - idx) * 256 + mask_tid];   // relu pattern of the layer this GEMM feeds back to
    if constexpr (B3) {
      if (idx == 3) {
        f32x16 ae[4];
        gemm_b3<4, 4, true, false>(rg, A0, A1, gin_src, ae, lane);
        float Ge[32];
#pragma unroll
        for (int t = 0; t < 32; ++t) Ge[t] = ae[t >> 4][t & 15];
        float p[3];
        point(p);
        embed_bwd<kMultires>(p, Ge, h, dp5);
      }
      gemm_b3<8, 8, true>(rg, A0, A1, gin_src, acc, lane);
    } else {
      if (idx == 3) {                                            // L5^T: its 64 encoding rows first (a 2-block segment)
        f32x16 ae[2];
        seg<2, 32, kRingSlots, true>(rg, A0, A1, BRegs16<8>{gin}, ae, lane);
        float Ge[32];
#pragma unroll
        for (int t = 0; t < 32; ++t) Ge[t] = ae[t >> 4][t & 15];
        float p[3];
        point(p);
        embed_bwd<kMultires>(p, Ge, h, dp5);
      }
      seg<8, 32, kRingSlots, true>(rg, A0, A1, BRegs16<8>{gin}, acc, lane);
    }
    if (idx == 0) {
      const float* wa = aux + kAuxWAlpha;                    // alpha_linear^T: rank-1 term w_alpha * dL/dsigma
#pragma unroll
      for (int tq = 0; tq < 32; ++tq) {
        const f32x4 w = *(const f32x4*)(wa + tq * 8 + h4);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
          acc[(4 * tq + kk) >> 4][(4 * tq + kk) & 15] =
              __builtin_fmaf(w[kk], gs, acc[(4 * tq + kk) >> 4][(4 * tq + kk) & 15]);
      }
    }
    const unsigned mw[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
    for (int mo = 0; mo < 8; ++mo) gin[mo] = apply_mask(acc[mo], mw[mo >> 1], (mo & 1) * 16);
  }
  // L0^T: the remaining 64 encoding rows
  {
    float Ge[32];
    if constexpr (B3) {
      f32x16 ae[4];
      gemm_b3<4, 4, true, false>(rg, A0, A1, gin_src, ae, lane);
#pragma unroll
      for (int t = 0; t < 32; ++t) Ge[t] = ae[t >> 4][t & 15];
    } else {
      f32x16 ae[2];
      seg<2, 32, kRingSlots, true>(rg, A0, A1, BRegs16<8>{gin}, ae, lane);
#pragma unroll
      for (int t = 0; t < 32; ++t) Ge[t] = ae[t >> 4][t & 15];
    }
    float p[3];
    point(p);
    float part[3];
    embed_bwd<kMultires>(p, Ge, h, part);
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
      const float both = part[ax] + dp5[ax];
      dp[ax] = both + __shfl_xor(both, 32);
    }
  }
}

// Backward of raw2outputs (RN:343-387) for both rays: dL/d raw (written over st.rawf) and dL/d|rays_d|.
// st.rawf holds sigmoid(rgb) and the raw sigma, st.alpha / st.wf / st.tf the forward alpha, weights, T.
//   dL/dw_i = g . c_i ;  dL/dalpha_i = A_i T_i - (sum_{k>i} A_k w_k) / (1 - alpha_i + 1e-10)
// Only the suffix sum is serial (one lane per ray, fp32, 8 terms per LDS round trip).
template <int S = 192, typename ST>
__device__ __forceinline__ void composite_bwd(ST& st, const float* grgb /* [2][3] in LDS */, int tid) {
  float* aw = &st.bwd_scratch[0][0][0];   // [2][S] A_i * w_i, then the exclusive suffix sums
  float* at = aw + 2 * S;                 // [2][S] A_i * T_i
  for (int idx = tid; idx < 2 * S; idx += 256) {
    const int r = idx / S, i = idx - r * S;
    const float* q = st.rawf[r][i];
    float a_i = (grgb[r * 3 + 0] * q[0] + grgb[r * 3 + 1] * q[1]) + grgb[r * 3 + 2] * q[2];
    if (st.ray[r][12] != 0.0f) a_i = a_i - ((grgb[r * 3 + 0] + grgb[r * 3 + 1]) + grgb[r * 3 + 2]);   // d(1 - acc)/dw
    aw[idx] = a_i * st.wf[r][i];
    at[idx] = a_i * st.tf[r][i];
  }
  __syncthreads();
  if ((tid & 63) == 0 && tid < 128) {
    const int r = tid >> 6;
    float suffix = 0.0f;
#pragma unroll 1
    for (int i0 = S - 8; i0 >= 0; i0 -= 8) {
      float f[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) f[k] = aw[r * S + i0 + k];
#pragma unroll
      for (int k = 7; k >= 0; --k) { aw[r * S + i0 + k] = suffix; suffix = suffix + f[k]; }
    }
  }
  __syncthreads();
  float dn = 0.0f;
  if (tid < 128) {
    const int r = tid >> 6, l = tid & 63;
    const float nrm = st.ray[r][11];
    const float g0 = grgb[r * 3 + 0], g1 = grgb[r * 3 + 1], g2 = grgb[r * 3 + 2];
#pragma unroll
    for (int i = l; i < S; i += 64) {
      float* q = st.rawf[r][i];
      const float a = st.alpha[r][i], w = st.wf[r][i];
      const float c0 = q[0], c1 = q[1], c2 = q[2], sigma = q[3];
      const float om = (1.0f - a) + 1e-10f;
      const float d_alpha = at[r * S + i] - aw[r * S + i] / om;          // T_k (k>i) carries the factor om_i
      const float dz = (i < S - 1) ? (st.zf[r][i + 1] - st.zf[r][i]) : 1e10f;
      const float e = 1.0f - a;                                           // exp(-relu(sigma) * delta)
      const float d_sigma = (sigma > 0.0f) ? d_alpha * (dz * nrm) * e : 0.0f;
      dn = dn + (d_alpha * fmaxf(sigma, 0.0f) * e) * dz;                  // d delta / d|d| = dz
      q[0] = w * g0 * c0 * (1.0f - c0);
      q[1] = w * g1 * c1 * (1.0f - c1);
      q[2] = w * g2 * c2 * (1.0f - c2);
      q[3] = d_sigma;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) dn += __shfl_xor(dn, m);
    if (l == 0) st.gnorm[r] = dn;
  }
  __syncthreads();
}

struct VjpArgs {
  RenderArgs r;             // forward arguments (fine must be 1)
  const float* grad_rgb;    // [N,3] cotangent
  float* grad_o;            // [N,3]
  float* grad_d;            // [N,3]
  uint4* mask_scratch;      // [gridDim][3 passes][9 layers][256 threads]
  const float* z_fine;      // optional [N,192]: sorted fine sample depths to use instead of the kernel's own resampling
                            // (z_samples is detached, RN:475: the depths are constants of the backward pass)
  float* grad_viewdirs;     // optional [N,3]: dL/d viewdirs when r.viewdirs is given (x32-structured kernels)
  // debug taps of the x32-structured kernels (include/nsr.h: NsrVjpDebugOut), all nullable
  uint4* dbg_masks;         // [ceil(N/2) items][3 fine passes][9 layers][256 threads]: the relu patterns the backward applied
  float* dbg_graw;          // [N,192,4] dL/d raw of the fine samples (output of the compositing backward)
  float* dbg_gpts;          // [N,192,6] per sample: dL/d pts, dL/d viewdirs (output of the network backward)
};

#if NSR_UNIT_F32
__global__ void k_set_vjp_args(const VjpArgs a, VjpArgs* dst) {
  *dst = a;
  *a.r.work_counter = 0ull;
  if (a.r.ovf_stat) a.r.ovf_stat[0] = 0u;
  if (a.r.epoch_counter) dst->r.epoch = *a.r.epoch_counter = *a.r.epoch_counter % 4094u + 1u;
}
#endif

// ------------------------------------------------------------------------------------------------------
// Fused forward + input-gradient kernel (render_path_grad, RN:168-178).  Per item (2 rays):
//   pass 0      coarse forward, compositing, resampling            (as k_render)
//   pass 1-3    fine forward, relu patterns captured to an L2-resident scratch (110 KB per workgroup)
//   --          compositing forward + backward -> dL/d raw per sample
//   pass 4-6    fine backward through the transposed network -> dL/d pts, dL/d viewdir per sample
//   --          per-ray reduction: dL/d rays_o, dL/d rays_d
// ------------------------------------------------------------------------------------------------------
template <int MODE, int NI = 128, int NS = 64>
__device__ __forceinline__ void render_vjp32_body(const VjpArgs* __restrict__ vp, char* smem) {
  constexpr bool B3 = MODE == kMlpB3;
  constexpr int NF = NS + NI;                              // fine samples per ray (see render32_body)
  constexpr int NP = (2 * NF + 127) / 128;                 // fine passes per item, forward and again backward
  constexpr int NPC = (2 * NS + 127) / 128;                // coarse passes per item
  constexpr int LNS = NS == 32 ? 5 : (NS == 64 ? 6 : 7);   // N_samples is 32, 64 or 128
  typedef typename ItemStateFor<NS>::type ST;
  constexpr int kMaskPasses = NP > 3 ? NP : 3;             // fine forward passes whose relu patterns the scratch holds per workgroup
  static_assert((NS == 32 || NS == 64 || NS == 128) && NF % 32 == 0 && NS <= ST::kNS && NF <= ST::kNF,
                "a wave's points belong to one ray; the item state holds kNS coarse / kNF fine samples per ray");
  const VjpArgs& va_setup = *vp;
  const RenderArgs& a_setup = va_setup.r;
  const int tid0 = threadIdx.x;
  const int lane = tid0 & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int j = lane & 31;
  ST& st = *(ST*)(smem + kLdsState);

  const long long n_rays = a_setup.n_rays;
  if ((long long)blockIdx.x >= queue_items(a_setup)) return;
  int* ovf = (int*)&st.ray[1][14];                       // f16x2: [2] points with NaN outputs / gradients, per ray of the item

  Ring rg;
  ring_init(rg, smem, a_setup.nets, a_setup.net_stride, NPC + 2 * NP, wave, lane);
  rg.pn0 = NPC; rg.pn1 = NPC + NP;

  f32x4 A0[4], A1[4];
  if constexpr (B3) ring_start<kRingSlots, kStreamSlabsB3, kStreamSlabsB3Bwd>(rg, A0, lane);
  else if constexpr (MODE == kMlpH2) ring_start<kRingSlots, kStreamSlabs, kStreamSlabsH2Bwd, kH2RingLag>(rg, A0, lane);
  else ring_start(rg, A0, lane);
  load_aux(smem, a_setup, tid0);
  if (tid0 < NS) st.tcoarse[tid0] = a_setup.tcoarse[tid0];
  if (tid0 < NI) st.ufine[tid0] = a_setup.ufine[tid0];
  __syncthreads();
  const float* aux_c = (const float*)(smem + kLdsAux);
  const float* aux_f = aux_c + kAuxFloats;
  // relu-pattern scratch of this workgroup: a UNIFORM base (scalar registers) + the thread index at each access, so
  // that no per-lane 64-bit address is kept alive across the passes
  uint4* my_masks = va_setup.mask_scratch + (size_t)blockIdx.x * (kMaskPasses * 9 * 256);
  float* grgb = &st.res[0][0];   // [2][3] cotangent staged here during the backward half (res is free then)

  long long* item_slot = (long long*)&st.ray[0][14];     // 8-byte slot in the unused tail of ray 0's block
  auto next_item = [&]() -> long long {
    if (opaque_v(tid0) == 0) *item_slot = queue_next_item(opaque_s(vp)->r);
    __syncthreads();
    const long long v = uniform64(*item_slot);
    __syncthreads();
    return v;
  };
  long long packed = next_item();
  int pass = 0;
#pragma unroll 1
  while (packed != -1ll) {
    const long long item = packed & kItemMask;
    const long long ray0 = item * 2;
    const int valid = (ray0 + 1 < n_rays) ? 2 : 1;
    const int wmask = (int)((unsigned long long)packed >> 62) & (valid == 2 ? 3 : 1);     // rays this launch writes
    auto wr = [&](int r) { return ((wmask >> r) & 1) != 0; };
    if (pass == 0) {
      const RenderArgs& a = opaque_s(vp)->r;              // see opaque_v / opaque_s
      const int tid = opaque_v(tid0);
      const float near_ = a.near_, far_ = a.far_;
      if (tid < 2) {
        const long long rr = ray0 + (tid < valid ? tid : 0);
        float o[3], d[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { o[c] = a.rays_o[rr * 3 + c]; d[c] = a.rays_d[rr * 3 + c]; }
        const float nrm = sqrtf(((d[0] * d[0]) + (d[1] * d[1])) + (d[2] * d[2]));
        const float* vd = a.viewdirs;                                                // given view directions: RN:91-103
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          st.ray[tid][c] = o[c]; st.ray[tid][3 + c] = d[c];
          st.ray[tid][6 + c] = vd ? vd[rr * 3 + c] : d[c] / nrm;
        }
        st.ray[tid][9] = a.near_rays ? a.near_rays[rr] : near_;       // per-ray bounds (RN:106-108) or the call's scalars
        st.ray[tid][10] = a.near_rays ? a.far_rays[rr] : far_;
        st.ray[tid][11] = nrm;
        st.ray[tid][12] = a.white_bkgd ? 1.0f : 0.0f;
      }
      if (MODE == kMlpH2 && (tid == 64 || tid == 65)) ovf[tid - 64] = 0;
      if (tid < 2 * NS) {
        const int r = tid >> LNS, i = tid & (NS - 1);
        const float t = st.tcoarse[i];
        const long long rb = ray0 + (r < valid ? r : 0);
        st.zc[r][i] = coarse_z(a.near_rays ? a.near_rays[rb] : near_, a.near_rays ? a.far_rays[rb] : far_, t, a.lindisp);
      }
      __syncthreads();
      if (a.t_rand) perturb_coarse_z<NS>(st, a.t_rand, ray0, valid, tid);
    }

    if ((NPC == 1) ? (pass <= NP) : (pass < NPC + NP)) {
      // ---- forward passes (coarse, then NP fine with relu capture) ----
      int r, i;
      const float* zsrc;
      float* dst;
      bool real = true;
      if ((NPC == 1) ? (pass == 0) : (pass < NPC)) {
        if constexpr (NS == 64) {
          r = wave >> 1; i = 32 * (wave & 1) + j;
        } else {
          int q0 = 128 * pass + 32 * wave;
          if (2 * NS % 128 != 0 && q0 >= 2 * NS) { q0 = 2 * NS - 32; real = false; }
          r = q0 >> LNS; i = (q0 & (NS - 1)) + j;
        }
        zsrc = &st.zc[r][i]; dst = st.rawc[r][i];
      } else {
        int q0 = 128 * (pass - NPC) + 32 * wave;
        if (2 * NF % 128 != 0 && q0 >= 2 * NF) { q0 = 2 * NF - 32; real = false; }
        r = q0 / NF; i = q0 - r * NF + j;
        zsrc = &st.zf[r][i]; dst = st.rawf[r][i];
      }
      const float z = *zsrc;
      const float* ry = st.ray[r];
      float raw[4];
      mlp_pass<true, MODE>(rg, ((NPC == 1) ? (pass == 0) : (pass < NPC)) ? aux_c : aux_f, A0, A1, lane, ry[0] + ry[3] * z, ry[1] + ry[4] * z,
                         ry[2] + ry[5] * z, ry[6], ry[7], ry[8], raw,
                         my_masks + (((NPC == 1) ? (pass == 0) : (pass < NPC)) ? 0 : (pass - NPC)) * (9 * 256), opaque_v(tid0));
      if (lane < 32 && real) *(f32x4*)dst = f32x4{raw[0], raw[1], raw[2], raw[3]};
      if constexpr (MODE == kMlpH2) { if (real) range_mark(ovf + r, (raw[0] + raw[1]) + (raw[2] + raw[3]), lane); }
    } else {
      // ---- backward passes: same point mapping as the fine forward pass p = pass - (NP + 1) ----
      int q0 = 128 * (pass - (NPC + NP)) + 32 * wave;
      bool real = true;
      if (2 * NF % 128 != 0 && q0 >= 2 * NF) { q0 = 2 * NF - 32; real = false; }
      const int r = q0 / NF, slot = (q0 - r * NF) >> 5, i = q0 - r * NF + j;
      const float z = st.zf[r][i];
      const float* ry = st.ray[r];
      const f32x4 g = *(const f32x4*)st.rawf[r][i];
      float dp[3], dv[3];
      mlp_bwd_pass<MODE>(rg, aux_f, A0, A1, lane, my_masks + (pass - (NPC + NP)) * (9 * 256), opaque_v(tid0), g[0], g[1], g[2], g[3],
                       ry, &st.zf[r][i - j], dp, dv);
      if constexpr (MODE == kMlpH2) { if (real) range_mark(ovf + r, dp[0] + dv[0], lane); }
      if (float* gp = opaque_s(vp)->dbg_gpts) {            // debug tap: the per-sample results of the network backward
        if (lane < 32 && wr(r) && real) {
          float* q = gp + ((ray0 + r) * NF + i) * 6;
          q[0] = dp[0]; q[1] = dp[1]; q[2] = dp[2]; q[3] = dv[0]; q[4] = dv[1]; q[5] = dv[2];
        }
      }
      // reduce the 32 points of this wave (all on ray r): sum dp, sum z*dp, sum dv
      float red[9] = {dp[0], dp[1], dp[2], z * dp[0], z * dp[1], z * dp[2], dv[0], dv[1], dv[2]};
#pragma unroll
      for (int m = 16; m >= 1; m >>= 1)
#pragma unroll
        for (int c = 0; c < 9; ++c) red[c] += __shfl_xor(red[c], m);
      if (lane == 0 && real)
#pragma unroll
        for (int c = 0; c < 9; ++c) st.psum[r][slot][c] = red[c];
    }

    const VjpArgs& va = *opaque_s(vp);                    // nothing below may be hoisted above the network passes
    const RenderArgs& a = va.r;
    const int tid = opaque_v(tid0);
    if (NPC > 1 && pass < NPC - 1) {
      ++pass;
    } else if ((NPC == 1) ? (pass == 0) : (pass < NPC)) {
      __syncthreads();
      composite<NS, 2, ST::kNS>(st, &st.zc[0][0], &st.rawc[0][0][0], &st.w0[0][0], &st.tf[0][0], tid, a.noise0, ray0, valid);
      int64_t* none = nullptr;
      sample_pdf_item<2, NI, NS>(st, st.ufine, &st.w0[0][1], ST::kNS,
                             [&](int r, int k) { return 0.5f * (st.zc[r][k + 1] + st.zc[r][k]); }, none, NI, tid, valid,
                             a.u_rays, ray0);
      merge_sort_item<2, NI, NS>(st, tid);
      if (va.z_fine) {                                       // caller-supplied depths replace the resampled ones
        for (int idx = tid; idx < 2 * NF; idx += 256) {
          const int r = idx / NF, k = idx - r * NF;
          st.zf[r][k] = va.z_fine[(ray0 + (r < valid ? r : 0)) * NF + k];
        }
        __syncthreads();
      }
      pass = NPC;
    } else if (pass < NPC + NP - 1) {
      ++pass;
    } else if (pass == NPC + NP - 1) {
      __syncthreads();
      composite<NF, 2, ST::kNF>(st, &st.zf[0][0], &st.rawf[0][0][0], &st.wf[0][0], &st.tf[0][0], tid, a.noise1, ray0, valid);
      if (tid < 16 && wr(tid >> 3)) {
        const int r = tid >> 3, c = tid & 7;
        const long long rr = ray0 + r;
        const float v = st.res[r][c];
        if (c < 3) { if (a.rgb) a.rgb[rr * 3 + c] = v; }
        else if (c == 3) { if (a.disp) a.disp[rr] = v; }
        else if (c == 4) { if (a.acc) a.acc[rr] = v; }
      }
      __syncthreads();
      if (tid < 6) {
        const int r = tid / 3;
        grgb[tid] = va.grad_rgb[(ray0 + (r < valid ? r : 0)) * 3 + (tid - r * 3)];
      }
      __syncthreads();
      composite_bwd<NF>(st, grgb, tid);
      if (va.dbg_graw)                                     // debug tap: dL/d raw as the network backward receives it
        for (int idx = tid; idx < 2 * ST::kNF * 4; idx += 256)
          if (wr(idx / (ST::kNF * 4)) && idx % (ST::kNF * 4) < NF * 4)
            va.dbg_graw[(ray0 + idx / (ST::kNF * 4)) * (NF * 4) + idx % (ST::kNF * 4)] = (&st.rawf[0][0][0])[idx];
      pass = NPC + NP;
    } else if (pass < NPC + 2 * NP - 1) {
      ++pass;
    } else {
      __syncthreads();
      if (tid < 2 && wr(tid)) {
        const int r = tid;
        float so[3] = {0, 0, 0}, sd[3] = {0, 0, 0}, sv[3] = {0, 0, 0};
        for (int s = 0; s < NF / 32; ++s)
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            so[c] += st.psum[r][s][c];
            sd[c] += st.psum[r][s][3 + c];
            sv[c] += st.psum[r][s][6 + c];
          }
        const float* ry = st.ray[r];
        const float nrm = ry[11];
        if (a.viewdirs) {          // view directions given by the caller: their gradient is an output of its own, and
                                   // rays_d is reached through the points and through dists * |d| only (RN:361)
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            va.grad_o[(ray0 + r) * 3 + c] = so[c];
            va.grad_d[(ray0 + r) * 3 + c] = sd[c] + st.gnorm[r] * (ry[3 + c] / nrm);
            if (va.grad_viewdirs) va.grad_viewdirs[(ray0 + r) * 3 + c] = sv[c];
          }
        } else {
          const float vdot = (sv[0] * ry[6] + sv[1] * ry[7]) + sv[2] * ry[8];
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const float v = ry[6 + c];
            va.grad_o[(ray0 + r) * 3 + c] = so[c];
            va.grad_d[(ray0 + r) * 3 + c] = sd[c] + (sv[c] - v * vdot) / nrm + st.gnorm[r] * v;   // RN:97, RN:361
          }
        }
      }
      if (va.dbg_masks && wmask == (valid == 2 ? 3 : 1))   // debug tap: the relu patterns of this item's three fine passes (the
        for (int k = 0; k < 9 * NP; ++k)                   // fallback launch rewrites them only when it owns the whole item)
          va.dbg_masks[((size_t)item * (9 * NP) + k) * 256 + tid] = my_masks[k * 256 + tid];
      if (MODE == kMlpH2 && tid == 0) {
        if (const unsigned m = range_report(a, ovf, item, valid)) {      // not listed: no fp32 re-render -> NaN, never a wrong number
          range_poison(a, ray0, m);
          const float qn = __builtin_nanf("");
          for (int r = 0; r < 2; ++r)
            if ((m >> r) & 1u)
              for (int c = 0; c < 3; ++c) {
                va.grad_o[(ray0 + r) * 3 + c] = qn; va.grad_d[(ray0 + r) * 3 + c] = qn;
                if (va.grad_viewdirs) va.grad_viewdirs[(ray0 + r) * 3 + c] = qn;
              }
        }
      }
      __syncthreads();
      pass = 0;
      packed = next_item();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// The x32 input-gradient kernels, same arithmetics and sample counts (two fine forward + two backward passes per item at
// N_importance = 64 / 32 instead of three + three; f16x2: the gradients are normalised per point, nsr_h2_bwd.inc).
#define NSR_VJP_KERNEL_1(NAME, ...)                                                                  \
  __global__ void __launch_bounds__(256, 1) NAME(const VjpArgs* __restrict__ vp) {                   \
    extern __shared__ __attribute__((aligned(16))) char smem[];                                      \
    render_vjp32_body<__VA_ARGS__>(vp, smem);                                                        \
  }
#define NSR_VJP_KERNEL_0(NAME, ...) __global__ void __launch_bounds__(256, 1) NAME(const VjpArgs* __restrict__ vp);
#define NSR_VJP_KERNEL(UNIT, NAME, ...) NSR_CAT(NSR_VJP_KERNEL_, UNIT)(NAME, __VA_ARGS__)
NSR_VJP_KERNEL(NSR_UNIT_F32, k_render_vjp, kMlpF32)
NSR_VJP_KERNEL(NSR_UNIT_B3, k_render_vjp_b3, kMlpB3)
NSR_VJP_KERNEL(NSR_UNIT_H2, k_render_vjp_h2, kMlpH2)
NSR_VJP_KERNEL(NSR_UNIT_H2, k_render_vjp_h2_n64, kMlpH2, 64)
NSR_VJP_KERNEL(NSR_UNIT_H2, k_render_vjp_h2_n32, kMlpH2, 32)
NSR_VJP_KERNEL(NSR_UNIT_B3, k_render_vjp_b3_n64, kMlpB3, 64)
NSR_VJP_KERNEL(NSR_UNIT_B3, k_render_vjp_b3_n32, kMlpB3, 32)
NSR_VJP_KERNEL(NSR_UNIT_H2, k_render_vjp_h2_n96, kMlpH2, 96, 64)
NSR_VJP_KERNEL(NSR_UNIT_B3, k_render_vjp_b3_n96, kMlpB3, 96, 64)
NSR_VJP_KERNEL(NSR_UNIT_H2, k_render_vjp_h2_c32_n64, kMlpH2, 64, 32)
NSR_VJP_KERNEL(NSR_UNIT_B3, k_render_vjp_b3_c32_n64, kMlpB3, 64, 32)
NSR_VJP_KERNEL(NSR_UNIT_H2, k_render_vjp_h2_c128_n128, kMlpH2, 128, 128)
NSR_VJP_KERNEL(NSR_UNIT_B3, k_render_vjp_b3_c128_n128, kMlpB3, 128, 128)
#undef NSR_VJP_KERNEL
#undef NSR_VJP_KERNEL_0
#undef NSR_VJP_KERNEL_1

#if NSR_UNIT_F32
// dL/d c2w[3][4] per patch of P consecutive pixels from dL/d rays (rays are linear in c2w, RH:160-164):
//   g[a][k] = sum_pix grad_d[pix][a] * dirs[pix][k]  (k < 3),  g[a][3] = sum_pix grad_o[pix][a]
__global__ void __launch_bounds__(256) k_pose_grad(const float* __restrict__ go, const float* __restrict__ gd,
                                                   float fx, float fy, float cx, float cy, int W, int n_pix,
                                                   int patch, float* out /* [n_patches][12] */) {
  __shared__ double red[256][12];
  const int p0 = blockIdx.x * patch;
  double acc[12];
#pragma unroll
  for (int c = 0; c < 12; ++c) acc[c] = 0.0;
  for (int pix = p0 + threadIdx.x; pix < min(p0 + patch, n_pix); pix += 256) {
    const int row = pix / W, col = pix - row * W;
    const float dirs[3] = {((float)col - cx) / fx, -(((float)row - cy) / fy), -1.0f};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
      for (int k = 0; k < 3; ++k) acc[a * 4 + k] += (double)gd[pix * 3 + a] * (double)dirs[k];
      acc[a * 4 + 3] += (double)go[pix * 3 + a];
    }
  }
#pragma unroll
  for (int c = 0; c < 12; ++c) red[threadIdx.x][c] = acc[c];
  __syncthreads();
  for (int s = 128; s >= 1; s >>= 1) {
    if ((int)threadIdx.x < s)
#pragma unroll
      for (int c = 0; c < 12; ++c) red[threadIdx.x][c] += red[threadIdx.x + s][c];
    __syncthreads();
  }
  if (threadIdx.x < 12) out[blockIdx.x * 12 + threadIdx.x] = (float)red[0][threadIdx.x];
}
#endif  // NSR_UNIT_F32


// ------------------------------------------------------------------------------------------------------
// x16 variant of the forward kernel: v_mfma_f32_16x16x4_f32, 16 points per wave, 64 per workgroup, registers
// <= 256 per lane so that TWO workgroups share a CU (two waves per SIMD).  Each workgroup still runs its four
// waves in lock-step on its own 3-slab weight ring, but the two workgroups of a CU drift against each other, so
// one's VALU-only stretches (encoding, layer epilogues, heads, compositing, resampling) run in the shadow of the
// other's MFMAs.  Work item = ONE ray: a coarse pass (64 points) + 3 fine passes.  Weight layout: pack_network16
// (no K permutation: register (block mo, r) of lane group g holds feature 16*mo + 4*g + r, which is what k-step
// 4*mo + r reads in group g; biases and heads in natural order).
// ------------------------------------------------------------------------------------------------------
constexpr int kAux16Floats = 3080;                                     // 3076 used, 16-byte multiple
constexpr int kLds16Aux = kRing16 * kSlabBytes;                        // 49152
constexpr int kLds16State = kLds16Aux + 2 * kAux16Floats * 4;          // 73792

struct ItemState16 {            // one ray; 7520 bytes
  float ray[1][16];
  float zc[1][64];
  float w0[1][64];
  float cdf[1][64];
  float zs[1][128];
  float zf[1][192];
  float rawf[1][192][4];        // the coarse pass uses the first 64 entries
  float alpha[1][192];
  float om[1][192];
  float tf[1][192];
  float res[1][8];
};
static_assert(kLds16State + sizeof(ItemState16) <= 81920, "two workgroups must fit in the 160 KiB LDS of a CU");
#if NSR_UNIT_X16     // (the state structs above and below size the launches: every unit sees them; the code is the x16 unit's)

template <int NG>
struct BRegs4 {    // previous layer's C fragment (16x16x4): k-step t <-> register (t>>2, t&3)
  const f32x4 (&v)[NG];
  __device__ __forceinline__ float operator()(int t) const { return v[t >> 2][t & 3]; }
};
struct BViews4 {   // cat([feature, input_views]) (RH:111): 64 k-steps of registers, then 8 of direction encoding
  const f32x4 (&v)[16];
  const float (&ed)[8];
  __device__ __forceinline__ float operator()(int t) const { return t < 64 ? v[(t & 63) >> 2][t & 3] : ed[t & 7]; }
};

__device__ __forceinline__ f32x4 clamp_bits4(f32x4 x, int thr) {
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  i32x4 b = __builtin_bit_cast(i32x4, x);
  i32x4 t = thr;
  b = __builtin_elementwise_max(b, t);
  return __builtin_bit_cast(f32x4, b);
}

// One network pass for this lane's point: lane (j = lane&15, g = lane>>4), the four lanes of a point hold
// complementary quarters of every feature vector.  Embedder RH:18-48, NeRF.forward RH:99-122.
// relu pattern of one x16 layer output (lane-private C fragment): element e = 4 (mo & 7) + r of word mo >> 3 ends up at
// bit 31 - e, SET when the unit is ON (x > +0, torch's relu').  0 - max(int(x), 0) is negative exactly then; one
// v_alignbit per element shifts that sign into the word.
template <int NMO>
__device__ __forceinline__ uint2 relu_mask4(const f32x4 (&acc)[NMO]) {
  unsigned w[2] = {0u, 0u};
#pragma unroll
  for (int mo = 0; mo < NMO; ++mo)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int on = 0 - max(__float_as_int(acc[mo][r]), 0);
      w[mo >> 3] = __builtin_amdgcn_alignbit(w[mo >> 3], (unsigned)on, 31);
    }
  return make_uint2(w[0], w[1]);
}

// x where the unit was on, +0 where it was off: one v_bfe_i32 (bit -> 0 / -1) and one v_and per element
__device__ __forceinline__ f32x4 apply_mask4(f32x4 x, unsigned word, int shift) {
  f32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    r[i] = __int_as_float(__float_as_int(x[i]) & __builtin_amdgcn_sbfe((int)word, 31 - (shift + i), 1));
  return r;
}

// LOCAL_G: derive everything that depends on the lane group (the two encoding constants, the aux base) from an opaque
// copy of it inside the pass instead of letting it be hoisted out of the kernel's main loop -- for the kernel whose other
// passes need every register (k_render_vjp16).
template <bool CAPTURE = false, bool LOCAL_G = false>
__device__ __forceinline__ void mlp_pass16(Ring& rg, const float* aux, f32x4 (&A0)[4], f32x4 (&A1)[4], int lane,
                                           float px, float py, float pz, float vx, float vy, float vz,
                                           float (&raw)[4], uint2* mask_dst = nullptr /* uniform */, unsigned mask_tid = 0u) {
  const int g = LOCAL_G ? opaque_v(lane >> 4) : (lane >> 4);
  const float poison = enc_poison(px, py, pz, vx, vy, vz);
  float e[16];   // 60 sin/cos columns dealt 15 per lane group (reference order), then the identity column g
  float ed[8];   // directions: group g holds frequency 2^g (sin xyz, cos xyz), then the identity column g
  {
    const float p[3] = {enc_domain(px, kMultires), enc_domain(py, kMultires), enc_domain(pz, kMultires)};
    const float v[3] = {enc_domain(vx, kMultiresViews), enc_domain(vy, kMultiresViews), enc_domain(vz, kMultiresViews)};
    // position columns: lane group g holds sin (g even) or cos (g odd) of the five octaves 5 (g >> 1) + f, f = t / 3,
    // of axis t % 3 -- so only TWO per-lane constants exist (the group's base scale and its sin/cos selector) and the
    // rest of the column assignment is compile-time (pack.py: eps16)
    const float base = (g & 2) ? 32.0f : 1.0f;
    const int sc = g & 1;
#pragma unroll
    for (int t = 0; t < 15; ++t) e[t] = enc_trig(p[t % 3] * ((float)(1 << (t / 3)) * base), sc);
#pragma unroll
    for (int t = 0; t < 6; ++t) ed[t] = enc_trig(ldexpf(v[t % 3], g), t >= 3);
    e[15] = g == 0 ? px : (g == 1 ? py : (g == 2 ? pz : 0.0f));
    ed[6] = g == 0 ? vx : (g == 1 ? vy : (g == 2 ? vz : 0.0f));
    ed[7] = 0.0f;
  }

  f32x4 acc[16], in[16];
  const float* bias_g = aux + kAuxBias + 4 * g;
  auto load_bias16 = [&](const float* b) {
#pragma unroll
    for (int mo = 0; mo < 16; ++mo) acc[mo] = *(const f32x4*)(b + 16 * mo);
  };
  load_bias16(bias_g);
  seg<16, 4, kRing16>(rg, A0, A1, BArr<16>{e}, acc, lane);
  if (CAPTURE) mask_dst[mask_tid] = relu_mask4<16>(acc);     // uniform base + 32-bit lane offset (saddr form)
#pragma unroll
  for (int mo = 0; mo < 16; ++mo) in[mo] = clamp_bits4(acc[mo], 0);

  float alpha_part = 0.0f;
#pragma unroll 1
  for (int L = 1; L <= 8; ++L) {
    load_bias16(bias_g + L * 256);
    if (L == 5) seg<16, 4, kRing16>(rg, A0, A1, BArr<16>{e}, acc, lane);   // skip columns first (RH:105)
    if (L == 8) {                                                           // alpha_linear on h7 (RH:109)
      const float* wa = aux + kAuxWAlpha + 4 * g;
#pragma unroll
      for (int mo = 0; mo < 16; ++mo) {
        const f32x4 w = *(const f32x4*)(wa + 16 * mo);
#pragma unroll
        for (int r = 0; r < 4; ++r) alpha_part = __builtin_fmaf(w[r], in[mo][r], alpha_part);
      }
    }
    seg<16, 16, kRing16>(rg, A0, A1, BRegs4<16>{in}, acc, lane);
    if (CAPTURE && L < 8) mask_dst[(unsigned)L * 256u + mask_tid] = relu_mask4<16>(acc);
    const int thr = (L == 8) ? (int)0x80000000 : 0;                         // feature_linear has no activation
#pragma unroll
    for (int mo = 0; mo < 16; ++mo) in[mo] = clamp_bits4(acc[mo], thr);
  }

  f32x4 av[8];                                                              // views_linears.0 (RH:111-115)
#pragma unroll
  for (int mo = 0; mo < 8; ++mo) av[mo] = *(const f32x4*)(aux + kAuxBiasV + 4 * g + 16 * mo);
  seg<8, 18, kRing16>(rg, A0, A1, BViews4{in, ed}, av, lane);
  if (CAPTURE) mask_dst[8u * 256u + mask_tid] = relu_mask4<8>(av);

  float part[4] = {0.0f, 0.0f, 0.0f, alpha_part};                           // rgb_linear (RH:117)
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int mo = 0; mo < 8; ++mo) {
      const f32x4 w = *(const f32x4*)(aux + kAuxWRgb + c * 128 + 4 * g + 16 * mo);
#pragma unroll
      for (int r = 0; r < 4; ++r) part[c] = __builtin_fmaf(w[r], fmaxf(av[mo][r], 0.0f), part[c]);
    }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float x = part[c];
    x = x + __shfl_xor(x, 16);
    x = x + __shfl_xor(x, 32);
    raw[c] = (x + aux[(c < 3) ? (kAuxBRgb + c) : kAuxBAlpha]) + poison;
  }
}

// Schedule: the rays are cut into chunks of `chunk` consecutive rays; workgroups take chunks from a global counter
// (the two workgroups of a CU share its matrix pipe unevenly, so a static split would leave the slower one
// finishing alone) and process a chunk in two phases: (A) the coarse pass + compositing + resampling + sort of every
// ray of the chunk (the 192 sorted z values go to a small global scratch, 768 B per ray), then (B) the three fine
// passes + compositing of the same rays.  Every chunk has the same shape (a short last chunk is padded with
// repeats of the last ray whose results are dropped), so the LDS-DMA producer can run ahead across chunk
// boundaries without knowing which chunk comes next; and workgroups that start together stream ONE network at a
// time: 2.3 MiB against the 4 MiB L2 of an XCD, instead of both networks (4.6 MiB) thrashing it.
__global__ void __launch_bounds__(256, 2) k_render16(const RenderArgs* __restrict__ ap) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const RenderArgs& a_setup = *ap;
#ifdef NSR_PHASE_TIMING
  long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tlast = clock64();
#endif
  const int tid0 = threadIdx.x;
  const int lane = tid0 & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int j = lane & 15;
  ItemState16& st = *(ItemState16*)(smem + kLds16State);

  const long long n_items = a_setup.n_rays;
  const int fine = a_setup.fine;
  const int K = a_setup.chunk;
  const long long n_chunks = (n_items + K - 1) / K;
  float* zscr = a_setup.zf_scratch + (size_t)blockIdx.x * K * 192;
  long long* chunk_slot = (long long*)&st.res[0][6];       // 8-byte LDS slot that broadcasts the chunk id

  Ring rg;
  ring_init(rg, smem, a_setup.nets, a_setup.net_stride, 1, wave, lane);
  rg.pn0 = fine ? K : 0x7fffffff;                          // K coarse passes, then 3K fine passes, repeating
  rg.pn1 = fine ? 4 * K : 0x7fffffff;
  rg.ppi = fine ? 4 * K : 1;
  f32x4 A0[4], A1[4];
  ring_start<kRing16>(rg, A0, lane);
  {
    float* dst = (float*)(smem + kLds16Aux);
    for (int i = tid0; i < kAux16Floats; i += 256) {
      dst[i] = a_setup.aux[0][i];
      dst[kAux16Floats + i] = a_setup.aux[1][i];
    }
  }
  __syncthreads();
  const float* aux_c = (const float*)(smem + kLds16Aux);

  long long c0;                                            // first ray of the current chunk
  int jr = 0;                                              // ray within the chunk
  int pass = 0;                                            // 0 = coarse pass (phase A), 1..3 = fine passes (phase B)
  auto next_chunk = [&]() -> bool {
    if (opaque_v(tid0) == 0) *chunk_slot = (long long)atomicAdd(opaque_s(ap)->work_counter, 1ull);
    __syncthreads();
    const long long c = uniform64(*chunk_slot);
    __syncthreads();
    c0 = c * K;
    return c < n_chunks;
  };
  bool more = next_chunk();
#pragma unroll 1
  while (more) {
    // a short last chunk is padded with repeats of the last ray (recomputed and rewritten with identical values)
    const long long rr = (c0 + jr) < n_items ? (c0 + jr) : n_items - 1;
    if (pass == 0 || (pass == 1 && K > 1)) {
      const RenderArgs& a = *opaque_s(ap);
      const int tid = opaque_v(tid0);               // a new ray enters the workgroup state (with one ray
                                                           // per chunk phase B simply continues in LDS)
      const float near_ = a.near_, far_ = a.far_;
      if (tid == 0) {
        float o[3], d[3];
        if (a.camera) {
          const long long hw = (long long)a.H * a.W;
          const long long v = rr / hw;
          const int pix = (int)(rr - v * hw);
          gen_ray(a.c2w + v * 12, a.fx, a.fy, a.cx, a.cy, pix / a.W, pix % a.W, o, d);
        } else {
#pragma unroll
          for (int c = 0; c < 3; ++c) { o[c] = a.rays_o[rr * 3 + c]; d[c] = a.rays_d[rr * 3 + c]; }
        }
        const float nrm = sqrtf(((d[0] * d[0]) + (d[1] * d[1])) + (d[2] * d[2]));   // torch.norm RN:97, RN:361
#pragma unroll
        for (int c = 0; c < 3; ++c) { st.ray[0][c] = o[c]; st.ray[0][3 + c] = d[c]; st.ray[0][6 + c] = d[c] / nrm; }
        st.ray[0][9] = near_; st.ray[0][10] = far_; st.ray[0][11] = nrm;
        st.ray[0][12] = a.white_bkgd ? 1.0f : 0.0f;
      }
      if (pass == 0) {
        if (tid < 64) {
          const float t = a.tcoarse[tid];
          st.zc[0][tid] = coarse_z(near_, far_, t, a.lindisp);
        }
      } else if (tid < 192) {                              // phase B: the sorted z values phase A left for this ray
        st.zf[0][tid] = __uint_as_float(__hip_atomic_load((const unsigned*)zscr + jr * 192 + tid, __ATOMIC_RELAXED,
                                                          __HIP_MEMORY_SCOPE_AGENT));
      }
      __syncthreads();
      NSR_T(0);
    }

    // one network pass: 64 points; coarse: sample 16w + j; fine p: sample 64(p-1) + 16w + j
    {
      const int i = (pass == 0 ? 0 : 64 * (pass - 1)) + 16 * wave + j;
      const float z = (pass == 0) ? st.zc[0][i] : st.zf[0][i];
      const float* ry = st.ray[0];
      float raw[4];
      mlp_pass16(rg, aux_c + (pass == 0 ? 0 : kAux16Floats), A0, A1, lane, ry[0] + ry[3] * z, ry[1] + ry[4] * z,
                 ry[2] + ry[5] * z, ry[6], ry[7], ry[8], raw);
      if (lane < 16) *(f32x4*)st.rawf[0][i] = f32x4{raw[0], raw[1], raw[2], raw[3]};
    }
    NSR_T(1);

    const RenderArgs& a = *opaque_s(ap);                  // see opaque_v / opaque_s: nothing below may be hoisted
    const int tid = opaque_v(tid0);                       // above the network pass
    if (pass == 0) {
      __syncthreads();
      if (a.dbg_raw0) {
        for (int idx = tid; idx < 256; idx += 256) a.dbg_raw0[rr * 256 + idx] = (&st.rawf[0][0][0])[idx];
        __syncthreads();
      }
      composite<64, 1>(st, &st.zc[0][0], &st.rawf[0][0][0], &st.w0[0][0], &st.tf[0][0], tid);
      if (tid < 8) {
        const int c = tid;
        const float v = st.res[0][c];
        float* rgb_dst = fine ? a.rgb0 : a.rgb;
        float* disp_dst = fine ? a.disp0 : a.disp;
        float* acc_dst = fine ? a.acc0 : a.acc;
        if (c < 3) { if (rgb_dst) rgb_dst[rr * 3 + c] = v; }
        else if (c == 3) { if (disp_dst) disp_dst[rr] = v; }
        else if (c == 4) { if (acc_dst) acc_dst[rr] = v; }
      }
      if (a.dbg_w0 && tid < 64) a.dbg_w0[rr * 64 + tid] = st.w0[0][tid];
      if (fine) {
        NSR_T(2);
#ifdef NSR_PHASE_TIMING
        int64_t* inds = nullptr;                 // dbg_inds carries the cycle totals in this build
#else
        int64_t* inds = (int64_t*)a.dbg_inds;
#endif
        sample_pdf_item<1>(st, a.ufine, &st.w0[0][1], 64,
                           [&](int r, int k) { return 0.5f * (st.zc[r][k + 1] + st.zc[r][k]); },   // RN:473
                           inds ? inds + rr * 128 : nullptr, 128, tid, 1);
        if (wave == 0) {
          const float sd = zstd_wave(st, 0, lane);
          if (lane == 0 && a.z_std) a.z_std[rr] = sd;
        }
        if (a.dbg_zs && tid < 128) a.dbg_zs[rr * 128 + tid] = st.zs[0][tid];
        NSR_T(3);
        merge_sort_item<1>(st, tid);
        if (tid < 192) {
          const float zv = st.zf[0][tid];
          if (K > 1)
            __hip_atomic_store((unsigned*)zscr + jr * 192 + tid, __float_as_uint(zv), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
          if (a.dbg_zf) a.dbg_zf[rr * 192 + tid] = zv;
        }
        NSR_T(5);
      }
      __syncthreads();
      ++jr;
      if (jr == K) {                                       // phase A of the chunk is done
        jr = 0;
        if (fine) pass = 1;
        else more = next_chunk();
      }
    } else if (pass < 3) {
      ++pass;
    } else {
      __syncthreads();
      if (a.dbg_raw) {
        for (int idx = tid; idx < 768; idx += 256) a.dbg_raw[rr * 768 + idx] = (&st.rawf[0][0][0])[idx];
        __syncthreads();
      }
      composite<192, 1>(st, &st.zf[0][0], &st.rawf[0][0][0], &st.alpha[0][0], &st.tf[0][0], tid);
      if (tid < 8) {
        const int c = tid;
        const float v = st.res[0][c];
        if (c < 3) { if (a.rgb) a.rgb[rr * 3 + c] = v; }
        else if (c == 3) { if (a.disp) a.disp[rr] = v; }
        else if (c == 4) { if (a.acc) a.acc[rr] = v; }
      }
      __syncthreads();
      NSR_T(6);
      pass = 1;
      ++jr;
      if (jr == K) {                                       // phase B of the chunk is done: next chunk
        jr = 0; pass = 0;
        more = next_chunk();
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the workgroup
#ifdef NSR_PHASE_TIMING
  if (tid0 == 0 && a_setup.dbg_raw == nullptr && a_setup.dbg_inds)
    for (int i = 0; i < 8; ++i) a_setup.dbg_inds[blockIdx.x * 8 + i] = tacc[i];
#endif
}

// ------------------------------------------------------------------------------------------------------
// k_render16p: k_render16 with the "global phases" schedule (NSR_FLAG_SCHED_PHASES).  Same arithmetic, same results;
// what changes is WHEN each workgroup streams which network.  With the default queue every workgroup alternates
// coarse net / fine net per ray, so the 64 workgroups of an XCD keep both weight images (4.6 MiB) cycling through its
// 4 MiB L2 and 3-5 % of the weight traffic misses to the fabric.  Here the rays are cut into super-chunks of S = 2^lg
// rays and ONE global queue hands out, per super-chunk, first its S coarse tasks (1 pass each) and then its S fine tasks
// (3 passes each): all workgroups work on the same network most of the time whatever their individual speed, and any
// workgroup may run the fine task of a ray whose coarse task another workgroup ran.
//   * hand-off: the coarse task leaves the ray's 192 sorted z values in slot (ray mod 3S) of a global ring.  The
//     hand-off is an OPTIMISATION, never a dependency: nothing in this kernel blocks on another workgroup.
//       - publisher (coarse task of super-chunk k): if the slot's previous content (super-chunk k-3) has been consumed
//         (taken[slot] == k-2, one non-blocking look) and it can CLAIM the slot (compare-and-swap of ready[slot] from
//         an older generation to "busy": exclusive even against a publisher that was descheduled mid-write), it writes
//         the values with agent-scope (sc1, write-through) stores, drains them (vmcnt(0)) and publishes ready[slot] =
//         k+1; otherwise it skips;
//       - consumer (fine task): one lane polls ready[slot] == tag at most `spin_max` times (default 64, about the time
//         of one network pass), all lanes read their 8-byte granule {value, tag} with one agent-scope load and compare
//         the tag (every value proves its own generation: a stale line or a later publisher cannot pass), and
//         taken[slot] = tag is raised.  If the values never showed up or any tag is wrong, the workgroup RECOMPUTES the
//         coarse pass of that ray itself
//         (ring reset to the coarse network, one extra pass, identical arithmetic -> identical bits) and carries on.
//     So the result never depends on timing, residency or what else runs on the GPU (other streams, other processes:
//     a descheduled publisher costs the consumer one extra pass); RenderArgs::status counts the recomputed rays.
//   * the LDS-DMA producer runs two slabs ahead of the consumer, across pass boundaries, so the network of the pass
//     AFTER the current one must be known when a pass starts: each workgroup holds its current AND its next task
//     (the queue is pulled one task ahead) and names the following network in rg.pnet_next (Ring::dd).
// ------------------------------------------------------------------------------------------------------
constexpr unsigned kSlotBusy = 0xffffffffu;

// generation tag of super-chunk k in launch `epoch` (the host counts launches per handle, 1..4094): unique across the
// launches that can still have values in a cache, monotonic within a launch, never equal to kSlotBusy
__device__ __forceinline__ unsigned handoff_tag(unsigned epoch, long long k) { return (epoch << 20) | (unsigned)(k + 1); }

// every thread's `ok` AND-ed over the workgroup, through four LDS words
__device__ __forceinline__ bool wg_all(bool ok, int* lds4, int tid) {
  const bool wave_ok = __builtin_amdgcn_ballot_w64(!ok) == 0;
  if ((tid & 63) == 0) lds4[tid >> 6] = wave_ok ? 1 : 0;
  __syncthreads();
  const bool all = (lds4[0] & lds4[1] & lds4[2] & lds4[3]) != 0;
  __syncthreads();
  return all;
}

// Hand-off slot: 192 granules of 8 bytes {value bits, generation tag}, each written by ONE agent-scope 8-byte store and
// read by one 8-byte load, so every value carries its own proof of freshness (MI355X_MICROARCH.md: data-tagged granules):
// a granule from another generation -- a stale cache line, a publisher of a later super-chunk -- fails the tag compare.
__device__ __forceinline__ void handoff_store(float* scratch, long long slot, int tid, float v, unsigned tag) {
  const unsigned long long g = ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v);
  __hip_atomic_store((unsigned long long*)scratch + slot * 192 + tid, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool handoff_load(const float* scratch, long long slot, int tid, unsigned tag, float& v) {
  const unsigned long long g =
      __hip_atomic_load((const unsigned long long*)scratch + slot * 192 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  v = __uint_as_float((unsigned)g);
  return (unsigned)(g >> 32) == tag;
}

__global__ void __launch_bounds__(256, 2) k_render16p(const RenderArgs* __restrict__ ap) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const RenderArgs& a_setup = *ap;
#ifdef NSR_PHASE_TIMING
  long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tlast = clock64();
#endif
  const int tid0 = threadIdx.x;
  const int lane = tid0 & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int j = lane & 15;
  ItemState16& st = *(ItemState16*)(smem + kLds16State);

  const long long n_rays = a_setup.n_rays;
  const long long total = 2 * n_rays;                      // tasks: one coarse + one fine per ray
  const int lg = a_setup.super_lg;
  const long long S = 1ll << lg;
  Ring rg;                                                 // (before any control flow: the descriptor stays in SGPRs)
  ring_init(rg, smem, a_setup.nets, a_setup.net_stride, 1, wave, lane);
  long long* task_slot = (long long*)&st.res[0][6];        // 8-byte LDS slot that broadcasts a task id / a flag
  auto pull = [&]() -> long long {
    if (opaque_v(tid0) == 0) *task_slot = (long long)atomicAdd(opaque_s(ap)->work_counter, 1ull);
    __syncthreads();
    const long long tv = uniform64(*task_slot);
    __syncthreads();
    return tv;
  };
  auto vote = [&](bool mine) -> bool {                     // thread 0's verdict for the whole workgroup
    if (opaque_v(tid0) == 0) *task_slot = mine ? 1 : 0;
    __syncthreads();
    const bool v = uniform64(*task_slot) != 0;
    __syncthreads();
    return v;
  };
  // task t -> (super-chunk k, fine?, ray): super-chunk k owns ids [2kS, 2kS + 2 S_k), coarse tasks first
  auto decode = [&](long long t, long long& k, bool& is_fine, long long& ray) {
    k = t >> (lg + 1);
    const long long off = t & (2 * S - 1);
    const long long left = n_rays - (k << lg);
    const long long Sk = left < S ? left : S;
    is_fine = off >= Sk;
    ray = (k << lg) + off - (is_fine ? Sk : 0);
  };
  long long cur = pull();
  if (cur >= total) return;
  long long nxt = pull();
  long long k, rr;
  bool cur_fine;
  decode(cur, k, cur_fine, rr);

  rg.dd = 1;
  rg.ppi = 0x7fffffff;
  const int fine_off = (int)a_setup.net_stride;
  rg.pnet_off = cur_fine ? fine_off : 0;
  rg.pnet_next = rg.pnet_off;
  f32x4 A0[4], A1[4];
  ring_start<kRing16>(rg, A0, lane);
  {
    float* dst = (float*)(smem + kLds16Aux);
    for (int i = tid0; i < kAux16Floats; i += 256) {
      dst[i] = a_setup.aux[0][i];
      dst[kAux16Floats + i] = a_setup.aux[1][i];
    }
  }
  __syncthreads();
  const float* aux_c = (const float*)(smem + kLds16Aux);

  int pass = cur_fine ? 1 : 0;                             // 0 = a coarse pass, 1..3 = the fine task's passes
  int local = 0;     // 0: normal; 1: this fine task is recomputing its own coarse pass; 2: ... and resumes its fine passes
#pragma unroll 1
  while (true) {
    const long long slots = 3 * S;
    const long long slot = NSR_IDX64(rr % slots, slots);
    if (pass <= 1 && local == 0) {                         // a task starts
      const RenderArgs& a = *opaque_s(ap);
      const int tid = opaque_v(tid0);
      unsigned* ready = a.sched_flags;
      unsigned* taken = a.sched_flags + slots;
      const unsigned gen = handoff_tag(a.epoch, k);
      bool got = false;
      if (tid == 0) {
        if (pass == 1) {                                   // is the coarse task's result there?  (bounded look)
#pragma unroll 1
          for (int it = 0; it < a.spin_max && !got; ++it) {
            got = __hip_atomic_load(ready + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen;
            if (!got) __builtin_amdgcn_s_sleep(16);
          }
        }
        const float near_ = a.near_, far_ = a.far_;
        float o[3], d[3];
        if (a.camera) {
          const long long hw = (long long)a.H * a.W;
          const long long v = rr / hw;
          const int pix = (int)(rr - v * hw);
          gen_ray(a.c2w + v * 12, a.fx, a.fy, a.cx, a.cy, pix / a.W, pix % a.W, o, d);
        } else {
#pragma unroll
          for (int c = 0; c < 3; ++c) { o[c] = a.rays_o[rr * 3 + c]; d[c] = a.rays_d[rr * 3 + c]; }
        }
        const float nrm = sqrtf(((d[0] * d[0]) + (d[1] * d[1])) + (d[2] * d[2]));   // torch.norm RN:97, RN:361
#pragma unroll
        for (int c = 0; c < 3; ++c) { st.ray[0][c] = o[c]; st.ray[0][3 + c] = d[c]; st.ray[0][6 + c] = d[c] / nrm; }
        st.ray[0][9] = near_; st.ray[0][10] = far_; st.ray[0][11] = nrm;
        st.ray[0][12] = a.white_bkgd ? 1.0f : 0.0f;
      }
      if (pass == 1) {
        got = vote(got);
        if (got) {
          bool fresh = true;                               // every granule proves its own generation
          if (tid < 192) {
            float zv;
            fresh = handoff_load(a.zf_scratch, slot, tid, gen, zv);
            st.zf[0][tid] = zv;
          }
          got = wg_all(fresh, (int*)&st.res[0][0], tid);
        }
        if (tid == 0) {
          __hip_atomic_store(taken + slot, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // consumed or given up
          if (!got) atomicAdd(a.status, 1u);
        }
        if (!got) {
          // recompute the coarse pass of this ray here: restart the weight stream on the coarse network
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
          rg.pslab = 0; rg.pslot = 0; rg.cslot = 0;
          rg.pnet_off = 0;
          ring_start<kRing16>(rg, A0, lane);
          local = 1;
          pass = 0;
        }
      }
      if (pass == 0) {
        if (tid < 64) st.zc[0][tid] = coarse_z(a.near_, a.far_, a.tcoarse[tid], a.lindisp);
      }
      __syncthreads();
    }
    NSR_T(0);
    {
      // the network of the pass AFTER this one: the fine net inside a fine task, else the first pass of the next task
      bool nf = true;
      if ((pass == 0 && local == 0) || pass == 3) {
        long long k2, r2;
        nf = cur_fine;
        if (nxt < total) decode(nxt, k2, nf, r2);
      }
      rg.pnet_next = nf ? fine_off : 0;
      ring_assert_uniform(rg);
    }

    // one network pass: 64 points; coarse: sample 16w + j; fine p: sample 64(p-1) + 16w + j
    {
      const int i = (pass == 0 ? 0 : 64 * (pass - 1)) + 16 * wave + j;
      const float z = (pass == 0) ? st.zc[0][i] : st.zf[0][i];
      const float* ry = st.ray[0];
      float raw[4];
      mlp_pass16<false, false>(rg, aux_c + (pass == 0 ? 0 : kAux16Floats), A0, A1, lane, ry[0] + ry[3] * z,
                              ry[1] + ry[4] * z, ry[2] + ry[5] * z, ry[6], ry[7], ry[8], raw);
      if (lane < 16) *(f32x4*)st.rawf[0][i] = f32x4{raw[0], raw[1], raw[2], raw[3]};
    }

    NSR_T(1);
    const RenderArgs& a = *opaque_s(ap);                  // nothing below may be hoisted above the network pass
    const int tid = opaque_v(tid0);
    bool task_done = false;
    if (pass == 0) {
      __syncthreads();
      if (a.dbg_raw0) {
        if (tid < 256) a.dbg_raw0[rr * 256 + tid] = (&st.rawf[0][0][0])[tid];
        __syncthreads();
      }
      composite<64, 1>(st, &st.zc[0][0], &st.rawf[0][0][0], &st.w0[0][0], &st.tf[0][0], tid);
      if (tid < 8) {                                       // (a recomputing fine task rewrites identical values)
        const int c = tid;
        const float v = st.res[0][c];
        if (c < 3) { if (a.rgb0) a.rgb0[rr * 3 + c] = v; }
        else if (c == 3) { if (a.disp0) a.disp0[rr] = v; }
        else if (c == 4) { if (a.acc0) a.acc0[rr] = v; }
      }
      if (a.dbg_w0 && tid < 64) a.dbg_w0[rr * 64 + tid] = st.w0[0][tid];
#ifdef NSR_PHASE_TIMING
      int64_t* inds = nullptr;                 // dbg_inds carries the cycle totals in this build
#else
      int64_t* inds = (int64_t*)a.dbg_inds;
#endif
      sample_pdf_item<1>(st, a.ufine, &st.w0[0][1], 64,
                         [&](int r, int kk) { return 0.5f * (st.zc[r][kk + 1] + st.zc[r][kk]); },   // RN:473
                         inds ? inds + rr * 128 : nullptr, 128, tid, 1);
      if (wave == 0) {
        const float sd = zstd_wave(st, 0, lane);
        if (lane == 0 && a.z_std) a.z_std[rr] = sd;
      }
      if (a.dbg_zs && tid < 128) a.dbg_zs[rr * 128 + tid] = st.zs[0][tid];
      merge_sort_item<1>(st, tid);
      if (a.dbg_zf && tid < 192) a.dbg_zf[rr * 192 + tid] = st.zf[0][tid];
      NSR_T(2);
      if (local == 1) {                                    // the fine passes follow right here, on the values in LDS
        local = 2;                                         // (pass 1 must not look like the start of a task)
        pass = 1;
      } else {
        // publish for the fine task -- if the slot's previous content has been consumed; never wait for it
        unsigned* ready = a.sched_flags;
        unsigned* taken = a.sched_flags + slots;
        bool can = false;
        if (tid == 0) {
          can = k < 3 || __hip_atomic_load(taken + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == handoff_tag(a.epoch, k - 3);
          if (can) {
            // claim the slot: it must hold an OLDER generation and nobody may be writing it (a publisher that was
            // descheduled in the middle of its stores still owns it); the compare-and-swap makes the claim exclusive
            const unsigned r = __hip_atomic_load(ready + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            can = r != kSlotBusy && r < handoff_tag(a.epoch, k) && atomicCAS(ready + slot, r, kSlotBusy) == r;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // "busy" is out before any value is
          }
        }
        can = vote(can);
        if (can) {
          if (tid < 192) handoff_store(a.zf_scratch, slot, tid, st.zf[0][tid], handoff_tag(a.epoch, k));
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my stores have left (write-through) before the flag is raised
          __syncthreads();
          if (tid == 0)
            __hip_atomic_store(ready + slot, handoff_tag(a.epoch, k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        NSR_T(3);
        task_done = true;
      }
    } else if (pass < 3) {
      ++pass;
      local = 0;
    } else {
      __syncthreads();
      if (a.dbg_raw) {
        for (int idx = tid; idx < 768; idx += 256) a.dbg_raw[rr * 768 + idx] = (&st.rawf[0][0][0])[idx];
        __syncthreads();
      }
      composite<192, 1>(st, &st.zf[0][0], &st.rawf[0][0][0], &st.alpha[0][0], &st.tf[0][0], tid);
      if (tid < 8) {
        const int c = tid;
        const float v = st.res[0][c];
        if (c < 3) { if (a.rgb) a.rgb[rr * 3 + c] = v; }
        else if (c == 3) { if (a.disp) a.disp[rr] = v; }
        else if (c == 4) { if (a.acc) a.acc[rr] = v; }
      }
      __syncthreads();
      NSR_T(4);
      task_done = true;
    }
    if (task_done) {
      cur = nxt;
      if (cur >= total) break;
      nxt = pull();
      decode(cur, k, cur_fine, rr);
      pass = cur_fine ? 1 : 0;
      NSR_T(5);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the workgroup
#ifdef NSR_PHASE_TIMING
  if (tid0 == 0 && a_setup.dbg_raw == nullptr && a_setup.dbg_inds)
    for (int i = 0; i < 8; ++i) a_setup.dbg_inds[blockIdx.x * 8 + i] = tacc[i];
#endif
}

// ------------------------------------------------------------------------------------------------------
// x16 variant of the forward + input-gradient kernel (k_render_vjp16): the scheme of k_render16 (16 points per wave,
// <= 256 registers, 80 KB of LDS, TWO workgroups per CU) applied to the seven passes of an item.  A work item is ONE
// ray: coarse forward (64 points), three fine forward passes with the relu patterns captured, compositing forward +
// backward, three backward passes through the transposed fine network (pack_network_backward16), per-ray reduction.
// ------------------------------------------------------------------------------------------------------
// d(encoding)/d(x) for the x16 encoding registers: lane group g holds G[t] = dL/d e[t] of ITS registers
// (t < PER: see mlp_pass16 for the column each register holds; t = PER: identity column g).
// Returns this group's contribution; the caller adds the four groups.
template <int NFREQ>
__device__ __forceinline__ void embed_bwd16(const float (&x)[3], const float* G, int g, float (&out)[3]) {
  constexpr int PER = 6 * NFREQ / 4;
  const float xd[3] = {enc_domain(x[0], NFREQ), enc_domain(x[1], NFREQ), enc_domain(x[2], NFREQ)};
  out[0] = g == 0 ? G[PER] : 0.0f;
  out[1] = g == 1 ? G[PER] : 0.0f;
  out[2] = g == 2 ? G[PER] : 0.0f;
#pragma unroll
  for (int t = 0; t < PER; ++t) {
    if (NFREQ == 4) {                      // PER = 6: group g holds frequency 2^g, t = 3 sc + ax
      const int sc = t / 3, ax = t % 3;
      const float f = ldexpf(1.0f, g);
      out[ax] = __builtin_fmaf(f * G[t], enc_trig(xd[ax] * f, 1 + sc), out[ax]);   // d sin = f cos, d cos = -f sin
    } else {                               // PER = 15: octave 5 (g >> 1) + t / 3, sin (g even) / cos (g odd), axis t % 3
      const int ax = t % 3;
      const float f = (float)(1 << (t / 3)) * ((g & 2) ? 32.0f : 1.0f);
      out[ax] = __builtin_fmaf(f * G[t], enc_trig(xd[ax] * f, 1 + (g & 1)), out[ax]);
    }
  }
}

// Backward (input-side VJP) of one x16 network pass, stream order as mlp_bwd_pass:
//   views^T 9 slabs (16 feature blocks + 2 direction-encoding blocks, K = 128) | feature^T 16 | L7^T 16 | L6^T 16 |
//   L5^T 20 (16 blocks h4 + 4 blocks encoding) | L4^T..L1^T 64 | L0^T 4 (4 encoding blocks) = 145 slabs.
// The gradient fragment of layer l (C layout: register (mo, r) of group g = feature 16 mo + 4 g + r) is masked with the
// relu pattern captured in the forward pass and is, register for register, the B operand of layer l-1's transposed GEMM.
__device__ __forceinline__ void mlp_bwd_pass16(Ring& rg, const float* aux, f32x4 (&A0)[4], f32x4 (&A1)[4], int lane,
                                               const uint2* mask_src /* uniform */, unsigned mask_tid, float g0, float g1,
                                               float g2, float gs, const float* ry /* LDS: the ray block */,
                                               const float* zrow /* LDS: z of this wave's 16 points */, float (&dp)[3],
                                               float (&dv)[3]) {
  const int g = lane >> 4;
  const int g4 = opaque_v(4 * g);                            // one per-lane base register for every aux access
  f32x4 gin[16];
  f32x4 acc[20];   // 0-15: gradient w.r.t. the 256 hidden features; 16-19: gradient w.r.t. the 64 encoding registers
  {
    // rgb_linear^T (VALU) masked by the views layer's relu pattern -> dL/d(views pre-activation), 128 features
    f32x4 gv[8];
    const unsigned mk = mask_src[8u * 256u + mask_tid].x;
#pragma unroll
    for (int mo = 0; mo < 8; ++mo) {
      const f32x4 w0 = *(const f32x4*)(aux + kAuxWRgb + 0 * 128 + 16 * mo + g4);
      const f32x4 w1 = *(const f32x4*)(aux + kAuxWRgb + 1 * 128 + 16 * mo + g4);
      const f32x4 w2 = *(const f32x4*)(aux + kAuxWRgb + 2 * 128 + 16 * mo + g4);
      f32x4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = __builtin_fmaf(w2[r], g2, __builtin_fmaf(w1[r], g1, w0[r] * g0));
      gv[mo] = apply_mask4(v, mk, 4 * mo);
    }
    // views^T: 256 feature rows (blocks 0-15) + 32 direction-encoding rows (blocks 16-17), K = 128
    seg<18, 8, kRing16, true>(rg, A0, A1, BRegs4<8>{gv}, acc, lane);
  }
  {
    float Gd[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) Gd[t] = acc[16 + (t >> 2)][t & 3];
    const float v[3] = {ry[6], ry[7], ry[8]};              // re-read where needed: not kept alive across the GEMMs
    float part[3];
    embed_bwd16<kMultiresViews>(v, Gd, opaque_v(g), part);   // opaque: keep the per-group constants out of the main loop's live set
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
      float x = part[ax];
      x = x + __shfl_xor(x, 16);
      dv[ax] = x + __shfl_xor(x, 32);
    }
  }
#pragma unroll
  for (int mo = 0; mo < 16; ++mo) gin[mo] = acc[mo];        // feature_linear has no activation
  // idx: 0 feature^T (+alpha head), 1 L7^T, 2 L6^T, 3 L5^T (20 blocks: first use of acc[16..19]), 4..7 L4^T..L1^T
#pragma unroll 1
  for (int idx = 0; idx < 8; ++idx) {
    const uint2 mk = mask_src[(unsigned)(7 - idx) * 256u + mask_tid];   // relu pattern of the layer this GEMM feeds back to
    if (idx == 3) seg<20, 16, kRing16, true>(rg, A0, A1, BRegs4<16>{gin}, acc, lane);
    else seg<16, 16, kRing16, true>(rg, A0, A1, BRegs4<16>{gin}, acc, lane);
    if (idx == 0) {                                          // alpha_linear^T: rank-1 term w_alpha * dL/dsigma
#pragma unroll
      for (int mo = 0; mo < 16; ++mo) {
        const f32x4 w = *(const f32x4*)(aux + kAuxWAlpha + 16 * mo + g4);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[mo][r] = __builtin_fmaf(w[r], gs, acc[mo][r]);
      }
    }
#pragma unroll
    for (int mo = 0; mo < 16; ++mo) gin[mo] = apply_mask4(acc[mo], mo < 8 ? mk.x : mk.y, 4 * (mo & 7));
  }
  // L0^T: the remaining contribution to the 64 encoding rows (accumulates onto L5^T's)
  seg<4, 16, kRing16>(rg, A0, A1, BRegs4<16>{gin}, *(f32x4(*)[4]) & acc[16], lane);
  {
    float Ge[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) Ge[t] = acc[16 + (t >> 2)][t & 3];
    const float z = zrow[opaque_v(lane) & 15];
    const float p[3] = {ry[0] + ry[3] * z, ry[1] + ry[4] * z, ry[2] + ry[5] * z};
    float part[3];
    embed_bwd16<kMultires>(p, Ge, opaque_v(g), part);
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
      float x = part[ax];
      x = x + __shfl_xor(x, 16);
      dp[ax] = x + __shfl_xor(x, 32);
    }
  }
}

#endif  // NSR_UNIT_X16
struct ItemStateV16 {           // one ray; the coarse-phase arrays are dead once the samples are sorted and are reused
  float ray[1][16];
  union {
    struct { float zc[1][64]; float w0[1][64]; float cdf[1][64]; float zs[1][128]; };     // pass 0
    struct { float at[192]; float psum[12][9]; float gnorm[1]; float grgb[3]; };           // backward half
  };
  float zf[1][192];
  float rawf[1][192][4];        // coarse raw (first 64), fine raw -> sigmoid(rgb), sigma -> dL/d raw
  float alpha[1][192];
  union { float om[1][192]; float wf[1][192]; };   // 1 - alpha + 1e-10 during the scan, then the fine weights
  union { float tf[1][192]; float aw[192]; };      // T_i, then A_i w_i and its exclusive suffix sums
  float res[1][8];
};
static_assert(kLds16State + sizeof(ItemStateV16) <= 81920, "two workgroups must fit in the 160 KiB LDS of a CU");
#if NSR_UNIT_X16

// Backward of raw2outputs (RN:343-387) for the ray of an x16 item, as composite_bwd: dL/d raw written over st.rawf,
// dL/d|rays_d| in st.gnorm.  st.rawf holds sigmoid(rgb) and the raw sigma, st.alpha / st.wf / st.tf the forward
// alpha, weights, T; A_i T_i goes to st.at, A_i w_i (then its suffix sums) replaces T_i in place.
__device__ __forceinline__ void composite_bwd_ray(ItemStateV16& st, int tid) {
  constexpr int S = 192;
  const float g0 = st.grgb[0], g1 = st.grgb[1], g2 = st.grgb[2];
  if (tid < S) {
    const float* q = st.rawf[0][tid];
    float a_i = (g0 * q[0] + g1 * q[1]) + g2 * q[2];
    if (st.ray[0][12] != 0.0f) a_i = a_i - ((g0 + g1) + g2);       // white_bkgd: d(1 - acc)/dw
    const float T = st.tf[0][tid];
    st.at[tid] = a_i * T;
    st.aw[tid] = a_i * st.wf[0][tid];                              // same thread, same slot as T: read above
  }
  __syncthreads();
  if (tid == 0) {
    float suffix = 0.0f;
#pragma unroll 1
    for (int i0 = S - 8; i0 >= 0; i0 -= 8) {
      float f[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) f[k] = st.aw[i0 + k];
#pragma unroll
      for (int k = 7; k >= 0; --k) { st.aw[i0 + k] = suffix; suffix = suffix + f[k]; }
    }
  }
  __syncthreads();
  if (tid < 64) {
    const float nrm = st.ray[0][11];
    float dn = 0.0f;
#pragma unroll
    for (int i = tid; i < S; i += 64) {
      float* q = st.rawf[0][i];
      const float a = st.alpha[0][i], w = st.wf[0][i];
      const float c0 = q[0], c1 = q[1], c2 = q[2], sigma = q[3];
      const float om = (1.0f - a) + 1e-10f;
      const float d_alpha = st.at[i] - st.aw[i] / om;                    // T_k (k>i) carries the factor om_i
      const float dz = (i < S - 1) ? (st.zf[0][i + 1] - st.zf[0][i]) : 1e10f;
      const float e = 1.0f - a;                                           // exp(-relu(sigma) * delta)
      const float d_sigma = (sigma > 0.0f) ? d_alpha * (dz * nrm) * e : 0.0f;
      dn = dn + (d_alpha * fmaxf(sigma, 0.0f) * e) * dz;                  // d delta / d|d| = dz
      q[0] = w * g0 * c0 * (1.0f - c0);
      q[1] = w * g1 * c1 * (1.0f - c1);
      q[2] = w * g2 * c2 * (1.0f - c2);
      q[3] = d_sigma;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) dn += __shfl_xor(dn, m);
    if (tid == 0) st.gnorm[0] = dn;
  }
  __syncthreads();
}

// PHASES = false: per-ray queue, an item is the seven passes of one ray (k_render_vjp16).
// PHASES = true : the global-phases schedule of k_render16p (k_render_vjp16p): per super-chunk first the coarse tasks
//                 (one pass), then the "fine + backward" tasks (six passes) of its rays, any workgroup; the sorted depths
//                 travel through the same non-blocking hand-off ring (a value that is not there in time is recomputed).
template <bool PHASES>
__device__ __forceinline__ void render_vjp16_body(const VjpArgs* __restrict__ vp, char* smem) {
  const VjpArgs& va_setup = *vp;
  const RenderArgs& a_setup = va_setup.r;
  const int tid0 = threadIdx.x;
  const int lane = tid0 & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int j = lane & 15;
  ItemStateV16& st = *(ItemStateV16*)(smem + kLds16State);
  const long long n_rays = a_setup.n_rays;                   // per-ray queue: a work item is one ray
  const long long total = PHASES ? 2 * n_rays : n_rays;      // tasks in the queue
  const int lg = PHASES ? a_setup.super_lg : 0;
  const long long S = 1ll << lg;

  Ring rg;                                                   // (before any control flow: the descriptor stays in SGPRs)
  ring_init(rg, smem, a_setup.nets, a_setup.net_stride, 7, wave, lane);   // coarse | 3 x fine | 3 x fine^T per item
  long long* item_slot = (long long*)&st.res[0][6];
  auto pull = [&]() -> long long {
    if (opaque_v(tid0) == 0) *item_slot = (long long)atomicAdd(opaque_s(vp)->r.work_counter, 1ull);
    __syncthreads();
    const long long v = uniform64(*item_slot);
    __syncthreads();
    return v;
  };
  auto vote = [&](bool mine) -> bool {                       // thread 0's verdict for the whole workgroup
    if (opaque_v(tid0) == 0) *item_slot = mine ? 1 : 0;
    __syncthreads();
    const bool v = uniform64(*item_slot) != 0;
    __syncthreads();
    return v;
  };
  // PHASES: task t -> (super-chunk k, fine+backward?, ray): super-chunk k owns ids [2kS, 2kS + 2 S_k), coarse tasks first
  auto decode = [&](long long t, long long& k, bool& is_fine, long long& ray) {
    if (!PHASES) { k = 0; is_fine = false; ray = t; return; }
    k = t >> (lg + 1);
    const long long off = t & (2 * S - 1);
    const long long left = n_rays - (k << lg);
    const long long Sk = left < S ? left : S;
    is_fine = off >= Sk;
    ray = (k << lg) + off - (is_fine ? Sk : 0);
  };
  long long cur = pull();
  if (cur >= total) return;
  long long nxt = PHASES ? pull() : 0;
  long long k, rr;
  bool cur_fine;
  decode(cur, k, cur_fine, rr);
  const int stride = (int)a_setup.net_stride;
  if (PHASES) {
    rg.dd = 1;
    rg.ppi = 0x7fffffff;
    rg.pnet_off = cur_fine ? stride : 0;
    rg.pnet_next = rg.pnet_off;
  }
  f32x4 A0[4], A1[4];
  ring_start<kRing16>(rg, A0, lane);
  {
    float* dst = (float*)(smem + kLds16Aux);
    for (int i = tid0; i < kAux16Floats; i += 256) {
      dst[i] = a_setup.aux[0][i];
      dst[kAux16Floats + i] = a_setup.aux[1][i];
    }
  }
  __syncthreads();
  const float* aux_c = (const float*)(smem + kLds16Aux);
  const float* aux_f = aux_c + kAux16Floats;
  // relu-pattern scratch of this workgroup: uniform base (re-read from the argument block where it is used, so that
  // it lives in scalar registers only around the pass) + thread index at each access
  auto my_masks = [&]() { return (uint2*)opaque_s(vp)->mask_scratch + (size_t)blockIdx.x * (3 * 9 * 256); };
  int pass = cur_fine ? 1 : 0;
  int local = 0;     // PHASES: 1 = this fine task is recomputing its own coarse pass, 2 = ... and resumes at pass 1
#pragma unroll 1
  while (true) {
    const long long slots = 3 * S;
    const long long slot = PHASES ? NSR_IDX64(rr % slots, slots) : 0;
    if (pass == 0 || (PHASES && pass == 1 && local == 0)) {  // a ray enters the workgroup state
      const RenderArgs& a = opaque_s(vp)->r;               // see opaque_v / opaque_s
      const int tid = opaque_v(tid0);
      const float near_ = a.near_, far_ = a.far_;
      unsigned* ready = a.sched_flags;
      unsigned* taken = a.sched_flags + slots;
      const unsigned gen = handoff_tag(a.epoch, k);
      bool got = false;
      if (tid == 0) {
        if (PHASES && pass == 1) {                         // is the coarse task's result there?  (bounded look)
#pragma unroll 1
          for (int it = 0; it < a.spin_max && !got; ++it) {
            got = __hip_atomic_load(ready + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen;
            if (!got) __builtin_amdgcn_s_sleep(16);
          }
        }
        float o[3], d[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { o[c] = a.rays_o[rr * 3 + c]; d[c] = a.rays_d[rr * 3 + c]; }
        const float nrm = sqrtf(((d[0] * d[0]) + (d[1] * d[1])) + (d[2] * d[2]));   // torch.norm RN:97, RN:361
#pragma unroll
        for (int c = 0; c < 3; ++c) { st.ray[0][c] = o[c]; st.ray[0][3 + c] = d[c]; st.ray[0][6 + c] = d[c] / nrm; }
        st.ray[0][9] = near_; st.ray[0][10] = far_; st.ray[0][11] = nrm;
        st.ray[0][12] = a.white_bkgd ? 1.0f : 0.0f;
      }
      if (PHASES && pass == 1) {
        got = vote(got);
        if (got) {
          bool fresh = true;                               // every granule proves its own generation
          if (tid < 192) {
            float zv;
            fresh = handoff_load(a.zf_scratch, slot, tid, gen, zv);
            st.zf[0][tid] = zv;
          }
          got = wg_all(fresh, (int*)&st.res[0][0], tid);
        }
        if (tid == 0) {
          __hip_atomic_store(taken + slot, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // consumed or given up
          if (!got) atomicAdd(a.status, 1u);
        }
        if (!got) {                                        // recompute the coarse pass here: restart the ring on net 0
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
          rg.pslab = 0; rg.pslot = 0; rg.cslot = 0;
          rg.pnet_off = 0;
          ring_start<kRing16>(rg, A0, lane);
          local = 1;
          pass = 0;
        }
      }
      if (pass == 0) {
        if (tid < 64) st.zc[0][tid] = coarse_z(near_, far_, a.tcoarse[tid], a.lindisp);
      }
      __syncthreads();
    }
    if (PHASES) {
      // the network of the pass AFTER this one (0 coarse, 1 fine, 2 fine^T)
      int nn;
      if (pass == 0 && local != 0) nn = 1;
      else if (pass == 0 || pass == 6) {
        long long k2, r2;
        bool nf = cur_fine;
        if (nxt < total) decode(nxt, k2, nf, r2);
        nn = nf ? 1 : 0;
      } else nn = pass < 3 ? 1 : 2;
      rg.pnet_next = nn * stride;
      ring_assert_uniform(rg);
    }

    if (pass <= 3) {
      // forward passes: 64 points; coarse: sample 16w + j; fine p: sample 64(p-1) + 16w + j
      const int i = (pass == 0 ? 0 : 64 * (pass - 1)) + 16 * wave + j;
      const float z = (pass == 0) ? st.zc[0][i] : st.zf[0][i];
      const float* ry = st.ray[0];
      float raw[4];
      if (pass == 0)
        mlp_pass16<false, true>(rg, aux_c, A0, A1, lane, ry[0] + ry[3] * z, ry[1] + ry[4] * z, ry[2] + ry[5] * z, ry[6], ry[7],
                          ry[8], raw);
      else
        mlp_pass16<true, true>(rg, aux_f, A0, A1, lane, ry[0] + ry[3] * z, ry[1] + ry[4] * z, ry[2] + ry[5] * z, ry[6], ry[7],
                         ry[8], raw, my_masks() + (pass - 1) * (9 * 256), (unsigned)opaque_v(tid0));
      const int lo = opaque_v(lane);
      if (lo < 16) *(f32x4*)st.rawf[0][(pass == 0 ? 0 : 64 * (pass - 1)) + 16 * wave + lo] = f32x4{raw[0], raw[1], raw[2], raw[3]};
    } else {
      // backward passes: same point mapping as the fine forward pass p = pass - 4
      const int i = 64 * (pass - 4) + 16 * wave + j;
      const f32x4 g = *(const f32x4*)st.rawf[0][i];
      float dp[3], dv[3];
      mlp_bwd_pass16(rg, aux_f, A0, A1, lane, my_masks() + (pass - 4) * (9 * 256), (unsigned)opaque_v(tid0), g[0], g[1], g[2], g[3],
                     st.ray[0], &st.zf[0][64 * (pass - 4) + 16 * wave], dp, dv);
      // reduce the 16 points of this wave: sum dp, sum z*dp, sum dv (every lane group holds the same totals); z is
      // re-read rather than kept alive across the pass
      const float zz = st.zf[0][64 * (pass - 4) + 16 * wave + (opaque_v(lane) & 15)];
      float red[9] = {dp[0], dp[1], dp[2], zz * dp[0], zz * dp[1], zz * dp[2], dv[0], dv[1], dv[2]};
#pragma unroll
      for (int m = 8; m >= 1; m >>= 1)
#pragma unroll
        for (int c = 0; c < 9; ++c) red[c] += __shfl_xor(red[c], m);
      if (lane == 0)
#pragma unroll
        for (int c = 0; c < 9; ++c) st.psum[(pass - 4) * 4 + wave][c] = red[c];
    }

    const VjpArgs& va = *opaque_s(vp);                      // nothing below may be hoisted above the network passes
    const RenderArgs& a = va.r;
    const int tid = opaque_v(tid0);
    bool task_done = false;
    if (pass == 0) {
      __syncthreads();
      composite<64, 1>(st, &st.zc[0][0], &st.rawf[0][0][0], &st.w0[0][0], &st.tf[0][0], tid);
      int64_t* none = nullptr;
      sample_pdf_item<1>(st, a.ufine, &st.w0[0][1], 64,
                         [&](int r, int kk) { return 0.5f * (st.zc[r][kk + 1] + st.zc[r][kk]); }, none, 128, tid, 1);
      merge_sort_item<1>(st, tid);
      if (!PHASES) {
        if (va.z_fine) {                                     // caller-supplied depths replace the resampled ones
          if (tid < 192) st.zf[0][tid] = va.z_fine[rr * 192 + tid];
          __syncthreads();
        }
        pass = 1;
      } else if (local == 1) {                               // the fine passes follow right here, on the values in LDS
        local = 2;
        pass = 1;
      } else {
        // publish for the fine+backward task -- if the slot's previous content has been consumed; never wait for it
        unsigned* ready = a.sched_flags;
        unsigned* taken = a.sched_flags + slots;
        bool can = false;
        if (tid == 0) {
          can = k < 3 || __hip_atomic_load(taken + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == handoff_tag(a.epoch, k - 3);
          if (can) {
            // claim the slot: it must hold an OLDER generation and nobody may be writing it (a publisher that was
            // descheduled in the middle of its stores still owns it); the compare-and-swap makes the claim exclusive
            const unsigned r = __hip_atomic_load(ready + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            can = r != kSlotBusy && r < handoff_tag(a.epoch, k) && atomicCAS(ready + slot, r, kSlotBusy) == r;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // "busy" is out before any value is
          }
        }
        can = vote(can);
        if (can) {
          if (tid < 192) handoff_store(a.zf_scratch, slot, tid, st.zf[0][tid], handoff_tag(a.epoch, k));
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
          if (tid == 0)
            __hip_atomic_store(ready + slot, handoff_tag(a.epoch, k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        task_done = true;
      }
    } else if (pass < 3) {
      ++pass;
      local = 0;
    } else if (pass == 3) {
      __syncthreads();
      composite<192, 1>(st, &st.zf[0][0], &st.rawf[0][0][0], &st.wf[0][0], &st.tf[0][0], tid);
      if (tid < 8) {
        const int c = tid;
        const float v = st.res[0][c];
        if (c < 3) { if (a.rgb) a.rgb[rr * 3 + c] = v; }
        else if (c == 3) { if (a.disp) a.disp[rr] = v; }
        else if (c == 4) { if (a.acc) a.acc[rr] = v; }
      }
      if (tid < 3) st.grgb[tid] = va.grad_rgb[rr * 3 + tid];   // the coarse-phase arrays are dead: backward half
      __syncthreads();
      composite_bwd_ray(st, tid);
      pass = 4;
    } else if (pass < 6) {
      ++pass;
    } else {
      __syncthreads();
      if (tid == 0) {
        float so[3] = {0, 0, 0}, sd[3] = {0, 0, 0}, sv[3] = {0, 0, 0};
        for (int s = 0; s < 12; ++s)
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            so[c] += st.psum[s][c];
            sd[c] += st.psum[s][3 + c];
            sv[c] += st.psum[s][6 + c];
          }
        const float* ry = st.ray[0];
        const float nrm = ry[11];
        const float vdot = (sv[0] * ry[6] + sv[1] * ry[7]) + sv[2] * ry[8];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float v = ry[6 + c];
          va.grad_o[rr * 3 + c] = so[c];
          va.grad_d[rr * 3 + c] = sd[c] + (sv[c] - v * vdot) / nrm + st.gnorm[0] * v;   // RN:97, RN:361
        }
      }
      __syncthreads();
      task_done = true;
    }
    if (task_done) {
      if (PHASES) {
        cur = nxt;
        if (cur >= total) break;
        nxt = pull();
      } else {
        cur = pull();
        if (cur >= total) break;
      }
      decode(cur, k, cur_fine, rr);
      pass = cur_fine ? 1 : 0;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the workgroup
}

__global__ void __launch_bounds__(256, 2) k_render_vjp16(const VjpArgs* __restrict__ vp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  render_vjp16_body<false>(vp, smem);
}

__global__ void __launch_bounds__(256, 2) k_render_vjp16p(const VjpArgs* __restrict__ vp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  render_vjp16_body<true>(vp, smem);
}

#else    // another unit: the x16 kernels are nsr_fused_x16.hip's
__global__ void __launch_bounds__(256, 2) k_render16(const RenderArgs* __restrict__ ap);
__global__ void __launch_bounds__(256, 2) k_render16p(const RenderArgs* __restrict__ ap);
__global__ void __launch_bounds__(256, 2) k_render_vjp16(const VjpArgs* __restrict__ vp);
__global__ void __launch_bounds__(256, 2) k_render_vjp16p(const VjpArgs* __restrict__ vp);
#endif  // NSR_UNIT_X16

#if NSR_UNIT_F32     // the stage kernels live in the API's unit
// ------------------------------------------------------------------------------------------------------
// run_network (RN:26-40) as a stage kernel: 128 points per workgroup pass.
// ------------------------------------------------------------------------------------------------------
struct NetArgs {
  const float* stream;
  const float* aux;
  const float* pts;     // [P,3]
  const float* dirs;    // [P,3]
  float* raw;           // [P,4]
  long long n_pts;
};

__global__ void __launch_bounds__(256, 1) k_run_network(NetArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long long n_tiles = (a.n_pts + 127) >> 7;
  if ((long long)blockIdx.x >= n_tiles) return;

  Ring rg;
  ring_init(rg, smem, a.stream, 0, 1, wave, lane);
  f32x4 A0[4], A1[4];
  ring_start(rg, A0, lane);
  float* auxl = (float*)(smem + kLdsAux);
  for (int i = tid; i < kAuxFloats; i += 256) auxl[i] = a.aux[i];
  __syncthreads();

  for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    long long p = tile * 128 + wave * 32 + (lane & 31);
    const bool ok = p < a.n_pts;
    if (!ok) p = a.n_pts - 1;
    float raw[4];
    mlp_pass<false>(rg, auxl, A0, A1, lane, a.pts[p * 3 + 0], a.pts[p * 3 + 1], a.pts[p * 3 + 2], a.dirs[p * 3 + 0],
             a.dirs[p * 3 + 1], a.dirs[p * 3 + 2], raw);
    if (ok && lane < 32) *(f32x4*)(a.raw + p * 4) = f32x4{raw[0], raw[1], raw[2], raw[3]};
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ------------------------------------------------------------------------------------------------------
// small stage kernels (same device functions as the fused kernel)
// ------------------------------------------------------------------------------------------------------
__global__ void k_get_rays(const float* __restrict__ c2w, float fx, float fy, float cx, float cy, int H, int W,
                           float* rays_o, float* rays_d) {
  const int pix = blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= H * W) return;
  float o[3], d[3];
  gen_ray(c2w + 12 * blockIdx.y, fx, fy, cx, cy, pix / W, pix % W, o, d);       // blockIdx.y: the view (nsr_get_rays_views)
  const long long at = ((long long)blockIdx.y * H * W + pix) * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) { rays_o[at + c] = o[c]; rays_d[at + c] = d[c]; }
}

// ndc_rays (RH:168-186) in torch's fp32 op order: the python scalars -1/(W/(2 focal)), -1/(H/(2 focal)), 2 near, -2 near
// are formed in double by the caller and reach the tensors as fp32 (cw, ch, two_near, m2near).
__global__ void k_ndc_rays(const float* __restrict__ ro, const float* __restrict__ rd, long long n, float near_, float cw,
                           float ch, float two_near, float m2near, float* __restrict__ oo, float* __restrict__ od) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float ox = ro[i * 3], oy = ro[i * 3 + 1], oz = ro[i * 3 + 2];
  const float dx = rd[i * 3], dy = rd[i * 3 + 1], dz = rd[i * 3 + 2];
  const float t = -(near_ + oz) / dz;                                  // RH:170
  const float sx = ox + t * dx, sy = oy + t * dy, sz = oz + t * dz;    // RH:171
  oo[i * 3 + 0] = (cw * sx) / sz;                                      // RH:174-176
  oo[i * 3 + 1] = (ch * sy) / sz;
  oo[i * 3 + 2] = 1.0f + two_near / sz;
  od[i * 3 + 0] = cw * (dx / dz - sx / sz);                            // RH:178-180
  od[i * 3 + 1] = ch * (dy / dz - sy / sz);
  od[i * 3 + 2] = m2near / sz;
}

// input-side VJP of the above: (dL/d o', dL/d d') -> (dL/d rays_o, dL/d rays_d), what autograd gives the reference's
// render(rays=..., ndc=True) under torch.autograd.grad (RN:101-103 sits between the rays and the renderer)
__global__ void k_ndc_rays_vjp(const float* __restrict__ ro, const float* __restrict__ rd, long long n, float near_,
                               float cw, float ch, float two_near, const float* __restrict__ g_o,
                               const float* __restrict__ g_d, float* __restrict__ go, float* __restrict__ gd) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float ox = ro[i * 3], oy = ro[i * 3 + 1], oz = ro[i * 3 + 2];
  const float dx = rd[i * 3], dy = rd[i * 3 + 1], dz = rd[i * 3 + 2];
  const float t = -(near_ + oz) / dz;
  const float sx = ox + t * dx, sy = oy + t * dy, sz = oz + t * dz;
  const float a0 = g_o[i * 3] - g_d[i * 3], a1 = g_o[i * 3 + 1] - g_d[i * 3 + 1];     // o0 and d0 hold -+ cw sx / sz
  const float gsx = a0 * cw / sz, gsy = a1 * ch / sz;
  const float gsz = ((-a0 * cw * sx - a1 * ch * sy) - two_near * g_o[i * 3 + 2] + two_near * g_d[i * 3 + 2]) / (sz * sz);
  float gdx = g_d[i * 3] * cw / dz, gdy = g_d[i * 3 + 1] * ch / dz;
  float gdz = -(g_d[i * 3] * cw * dx + g_d[i * 3 + 1] * ch * dy) / (dz * dz);
  const float gt = (gsx * dx + gsy * dy) + gsz * dz;                                   // s = o + t d, t = -(near + oz) / dz
  go[i * 3 + 0] = gsx;
  go[i * 3 + 1] = gsy;
  go[i * 3 + 2] = gsz - gt / dz;
  gd[i * 3 + 0] = gdx + gsx * t;
  gd[i * 3 + 1] = gdy + gsy * t;
  gd[i * 3 + 2] = (gdz + gsz * t) + gt * (near_ + oz) / (dz * dz);
}

// Embedder.embed (RH:39-48): [n,3] -> [n, 3 + 6 L] = [x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)]
__global__ void k_embed(const float* __restrict__ x, long long n, int L, float* __restrict__ out) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // one thread per (point, frequency)
  if (idx >= n * (L + 1)) return;
  const long long pt = idx / (L + 1);
  const int f = (int)(idx - pt * (L + 1));       // 0 = identity block, f >= 1: frequency 2^(f-1)
  float* o = out + pt * (3 + 6 * L);
#pragma unroll
  for (int ax = 0; ax < 3; ++ax) {
    const float v = x[pt * 3 + ax];
    if (f == 0) { o[ax] = v; continue; }
    const float a = ldexpf(enc_domain(v, L), f - 1);
    o[3 + 6 * (f - 1) + ax] = enc_trig(a, 0);
    o[3 + 6 * (f - 1) + 3 + ax] = enc_trig(a, 1);
  }
}

struct R2OArgs {
  const float *raw, *z, *rays_d;
  long long n_rays;
  int white_bkgd;
  float *rgb, *disp, *acc, *weights, *depth;
};

template <int S>
__global__ void __launch_bounds__(256) k_raw2outputs(R2OArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  ItemState& st = *(ItemState*)smem;
  float* zbuf = &st.zf[0][0];       // [2][S] (S <= 192)
  float* rawbuf = &st.rawf[0][0][0];
  float* wbuf = &st.wf[0][0];
  const int tid = threadIdx.x;
  const long long n_items = (a.n_rays + 1) >> 1;
  for (long long item = blockIdx.x; item < n_items; item += gridDim.x) {
    const long long ray0 = item * 2;
    const int valid = (ray0 + 1 < a.n_rays) ? 2 : 1;
    for (int idx = tid; idx < 2 * S; idx += 256) {
      const int r = idx / S, i = idx - r * S;
      const long long rr = ray0 + (r < valid ? r : 0);
      zbuf[r * S + i] = a.z[rr * S + i];
#pragma unroll
      for (int c = 0; c < 4; ++c) rawbuf[(r * S + i) * 4 + c] = a.raw[(rr * S + i) * 4 + c];
    }
    if (tid < 2) {
      const long long rr = ray0 + (tid < valid ? tid : 0);
      const float dx = a.rays_d[rr * 3], dy = a.rays_d[rr * 3 + 1], dz = a.rays_d[rr * 3 + 2];
      st.ray[tid][11] = sqrtf(((dx * dx) + (dy * dy)) + (dz * dz));
      st.ray[tid][12] = a.white_bkgd ? 1.0f : 0.0f;
    }
    __syncthreads();
    composite<S>(st, zbuf, rawbuf, wbuf, &st.tf[0][0], tid);
    for (int idx = tid; idx < valid * S; idx += 256) a.weights[ray0 * S + idx] = wbuf[idx];
    if (tid < valid * 8) {
      const int r = tid >> 3, c = tid & 7;
      const float v = st.res[r][c];
      if (c < 3) a.rgb[(ray0 + r) * 3 + c] = v;
      else if (c == 3) a.disp[ray0 + r] = v;
      else if (c == 4) a.acc[ray0 + r] = v;
      else if (c == 5) a.depth[ray0 + r] = v;
    }
    __syncthreads();
  }
}

struct PdfArgs {
  const float *bins, *weights, *ufine;
  long long n_rays;
  float* samples;
  long long* inds;
};

__global__ void __launch_bounds__(256) k_sample_pdf(PdfArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  ItemState& st = *(ItemState*)smem;
  const int tid = threadIdx.x;
  float* binbuf = &st.zc[0][0];     // [2][64] (63 used)
  float* wbuf = &st.w0[0][0];       // [2][64] (62 used)
  if (tid < 128) st.ufine[tid] = a.ufine[tid];
  const long long n_items = (a.n_rays + 1) >> 1;
  for (long long item = blockIdx.x; item < n_items; item += gridDim.x) {
    const long long ray0 = item * 2;
    const int valid = (ray0 + 1 < a.n_rays) ? 2 : 1;
    if (tid < 128) {
      const int r = tid >> 6, i = tid & 63;
      const long long rr = ray0 + (r < valid ? r : 0);
      binbuf[r * 64 + i] = (i < 63) ? a.bins[rr * 63 + i] : 0.0f;
      wbuf[r * 64 + i] = (i < 62) ? a.weights[rr * 62 + i] : 0.0f;
    }
    __syncthreads();
    sample_pdf_item(st, st.ufine, wbuf, 64, [&](int r, int k) { return binbuf[r * 64 + k]; },
                    a.inds ? (int64_t*)a.inds + ray0 * 128 : nullptr, 128, tid, valid);
    for (int idx = tid; idx < valid * 128; idx += 256) a.samples[ray0 * 128 + idx] = (&st.zs[0][0])[idx];
    __syncthreads();
  }
}

// z_vals, _ = torch.sort(torch.cat([z_vals, z_samples], -1), -1) (RN:477) as a stage kernel
__global__ void __launch_bounds__(256) k_sort_merge(const float* __restrict__ zc, const float* __restrict__ zs,
                                                    long long n_rays, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  ItemState& st = *(ItemState*)smem;
  const int tid = threadIdx.x;
  const long long n_items = (n_rays + 1) >> 1;
  for (long long item = blockIdx.x; item < n_items; item += gridDim.x) {
    const long long ray0 = item * 2;
    const int valid = (ray0 + 1 < n_rays) ? 2 : 1;
    for (int idx = tid; idx < 2 * 192; idx += 256) {
      const int r = idx / 192, k = idx - r * 192;
      const long long rr = ray0 + (r < valid ? r : 0);
      if (k < 64) st.zc[r][k] = zc[rr * 64 + k];
      else st.zs[r][k - 64] = zs[rr * 128 + (k - 64)];
    }
    __syncthreads();
    merge_sort_item(st, tid);
    for (int idx = tid; idx < valid * 192; idx += 256) out[ray0 * 192 + idx] = (&st.zf[0][0])[idx];
    __syncthreads();
  }
}

// MFMA fragment layout self-test: D = A(32x2) * B(2x32) with asymmetric operands, plus a chained second product
// that uses D's registers as B operands the way the network layers do.
__global__ void k_selftest(float* out /*[64][16] + [64][16]*/) {
  const int lane = threadIdx.x;
  const int i = lane & 31, k = lane >> 5;
  // A[i][k] = 1 + i + 100k ; B[k][j] = 3 + 7j - 1000k (j = lane&31)
  const float a = 1.0f + (float)i + 100.0f * (float)k;
  const float b = 3.0f + 7.0f * (float)i - 1000.0f * (float)k;
  f32x16 acc = {0};
  acc = NSR_MFMA(a, b, acc);
#pragma unroll
  for (int r = 0; r < 16; ++r) out[lane * 16 + r] = acc[r];
}

#endif  // NSR_UNIT_F32

}  // namespace nsr
