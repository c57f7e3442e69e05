// nsr_fused_h2.hip -- translation unit of the f16x2 fused kernels (the default arithmetic): k_render_h2 / k_render_vjp_h2 and their
// sample-count variants (nsr_kernels.hip: "Translation units"; nsr_h2.inc, nsr_h2_bwd.inc).  Linked into libnsr.so.
#define NSR_UNIT_H2 1
#include "nsr_kernels.hip"
#include "nsr_unit_bounds.inc"
NSR_UNIT_BOUNDS(h2)
