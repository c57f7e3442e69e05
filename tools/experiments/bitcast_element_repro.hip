// hipcc 7.2 (gfx950, -O3): __builtin_bit_cast(float, v[e]) of ONE element of an ext_vector_type(4) of unsigned takes element 0 for
// every e of an unrolled loop -- all four MFMAs below read v16 / v18 (hipcc --offload-arch=gfx950 -O3 -c -save-temps=obj, then grep
// v_mfma in the .s).  Casting the whole vector first (__builtin_bit_cast(f32x4, v)[e]) compiles to v16..v19 / v20..v23 as it should.
// Found on the first run of kw_gemm_f32 (csrc/nsr_wide_b3.inc); the stage-wise parity tests caught it.
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const u32x4* a, const u32x4* b, f32x16* out) {
  u32x4 fa = a[threadIdx.x], fb = b[threadIdx.x];
  f32x16 acc = out[threadIdx.x];
#pragma unroll
  for (int e = 0; e < 4; ++e)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, fa[e & 3]), __builtin_bit_cast(float, fb[e & 3]), acc, 0, 0, 0);
  out[threadIdx.x] = acc;
}
