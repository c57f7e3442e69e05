// nsr_device.h -- constants shared by the kernels, the host API and (via nsr_pack_layout) the Python packer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nsr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kMultires = 10;        // NM:1268
constexpr int kMultiresViews = 4;    // NM:1270
constexpr int kRingSlots = 6;        // 16 KiB slabs resident in LDS
constexpr int kSlabBytes = 16384;
constexpr int kStepBytes = 4096;   // a step = 4 chunks of 1 KiB = 16 MFMAs
constexpr int kSlabFloats = 4096;
// slabs per network pass: L0 4 | L1-4 64 | L5 skip 4 + 16 | L6-7 32 | feature 16 | views 9
constexpr int kStreamSlabs = 145;

// aux block (floats): biases in C-fragment order, alpha/rgb heads
constexpr int kAuxBias = 0;          // 9 x 256 (pts_linears.0-7, feature_linear)
constexpr int kAuxBiasV = 2304;      // 128      (views_linears.0)
constexpr int kAuxWAlpha = 2432;     // 256      (alpha_linear.weight, B-operand order)
constexpr int kAuxWRgb = 2688;       // 3 x 128  (rgb_linear.weight, C-fragment order of the views layer)
constexpr int kAuxBAlpha = 3072;     // 1
constexpr int kAuxBRgb = 3073;       // 3
constexpr int kAuxH2Scale = 3080;    // f16x2 layout only (nsr_h2.inc): the powers of two activations are multiplied with before
                                     // the fp16 split -- [0] encoding @ L0, [1..8] hidden input of L1..L7 / feature_linear,
                                     // [9] encoding @ L5, [10] feature @ views layer, [11] direction encoding @ views layer
constexpr int kAuxH2Bwd = 3096;      // f16x2, transposed network (nsr_h2_bwd.inc): 14 power-of-two multipliers of the backward chain
constexpr int kAuxFloats = 3328;     // padded to 13 KiB

// LDS map of the fused kernel (bytes)
constexpr int kLdsRing = 0;
constexpr int kLdsAux = kRingSlots * kSlabBytes;              //  98304
constexpr int kLdsState = kLdsAux + 2 * kAuxFloats * 4;       // 124928

typedef int i32x16 __attribute__((ext_vector_type(16)));

// relu as ONE integer max per element: negative floats are negative ints, so max_i32(bits, 0) clamps them to +0
// and leaves non-negative values (and their NaNs) untouched; with thr = INT_MIN it is the identity (used for the
// activation-free feature layer without a branch).  fmaxf would cost a canonicalising v_max plus the max itself.
__device__ __forceinline__ f32x16 clamp_bits16(f32x16 x, int thr) {
  i32x16 b = __builtin_bit_cast(i32x16, x);
  i32x16 t = thr;
  b = __builtin_elementwise_max(b, t);
  return __builtin_bit_cast(f32x16, b);
}
__device__ __forceinline__ f32x16 relu16(f32x16 x) { return clamp_bits16(x, 0); }

// Bounds-checked indexing for the data-dependent LDS / scratch indices (searchsorted results, merge ranks, hand-off
// slots).  Release build: the index itself.  `make debug` (-DNSR_DEBUG_BOUNDS, libnsr_debug.so): an out-of-range index
// is clamped -- no wild LDS write, the kernel finishes -- and its source line is recorded in a device word that
// nsr_debug_bounds_status() reads back.  (A trap would take the whole HSA queue down; this keeps the evidence.)
#ifdef NSR_DEBUG_BOUNDS
static __device__ unsigned g_bounds_violation = 0u;      // first (highest) offending source line, 0 = clean; one copy per
                                                         // translation unit (nsr_unit_bounds.inc collects them)
__device__ __forceinline__ long long checked_index(long long i, long long n, unsigned line) {
  if (i < 0 || i >= n) {
    atomicMax(&g_bounds_violation, line);
    return i < 0 ? 0 : n - 1;
  }
  return i;
}
#define NSR_IDX(i, n) ((int)nsr::checked_index((long long)(i), (long long)(n), __LINE__))
#define NSR_IDX64(i, n) (nsr::checked_index((long long)(i), (long long)(n), __LINE__))
#else
#define NSR_IDX(i, n) (i)
#define NSR_IDX64(i, n) (i)
#endif

__device__ __forceinline__ double shfl_xor_f64(double v, int mask) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl_xor(lo, mask);
  hi = __shfl_xor(hi, mask);
  return __hiloint2double(hi, lo);
}

}  // namespace nsr
