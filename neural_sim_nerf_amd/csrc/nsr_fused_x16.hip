// nsr_fused_x16.hip -- translation unit of the x16 fp32 fused kernels (v_mfma_f32_16x16x4_f32, two workgroups per CU): k_render16,
// k_render16p, k_render_vjp16, k_render_vjp16p (nsr_kernels.hip: "Translation units").  Linked into libnsr.so.
#define NSR_UNIT_X16 1
#include "nsr_kernels.hip"
#include "nsr_unit_bounds.inc"
NSR_UNIT_BOUNDS(x16)
