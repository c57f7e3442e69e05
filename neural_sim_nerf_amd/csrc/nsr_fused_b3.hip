// nsr_fused_b3.hip -- translation unit of the bf16x3 fused kernels: k_render_b3 / k_render_vjp_b3 and their sample-count variants
// (nsr_kernels.hip: "Translation units"; nsr_b3.inc).  Linked into libnsr.so.
#define NSR_UNIT_B3 1
#include "nsr_kernels.hip"
#include "nsr_unit_bounds.inc"
NSR_UNIT_BOUNDS(b3)
