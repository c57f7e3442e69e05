/*
 * nsr_probe.h -- C ABI of libnsr_probe.so: diagnostic micro-kernels, NOT part of the product library libnsr.so
 * (built by `make -C neural_sim_nerf_amd/csrc probe`).  Nothing in the reference corresponds to this; it exists to
 * attribute MFMA-rate losses of the render kernels.
 */
#ifndef NSR_PROBE_H_
#define NSR_PROBE_H_
#ifdef __cplusplus
extern "C" {
#endif

const char* nsr_probe_last_error(void);

/* The 256x256 layer GEMM of the render kernels in isolation on every CU of `device`, `iters` layer-equivalents per
 * wave; returns ms.  Synchronises; allocates its own operands.
 * mode 0: MFMAs only, 1: + LDS fragment reads, 2: + LDS-DMA ring and barriers = the production x32 segment,
 * 3: the x16 segment with two workgroups per CU (ms = kernel time).  Modes 4..8: the x16 segment in the first
 * workgroup of every CU while the second one runs nothing / a dense fp32 VALU chain / sin-cos / an LDS pointer
 * chase / an fp64 chain (ms = mean duration of the GEMM workgroups; NSR_PROBE_VERBOSE=1 prints the partner's loop
 * rate; partner_prio = 1 raises its priority).  Mode 10: the x32 layer followed by its relu + re-bias epilogue;
 * mode 9: the two-tiles-per-wave 16x16x4 scheme with the epilogue of one tile interleaved into the other tile's MFMAs.
 * Modes 11..27: the bf16x3 layer and its ablations (tools/probe_bf16x3.py).  Modes 28 / 30 / 29 (r05): the f16x2 layer as the
 * render kernels run it (gemm_h2<8, 16>: ring, fragment reads, LDS-DMA, split stages) with / without the layer epilogue, and its
 * MFMAs alone; NSR_PROBE_VERBOSE=1 prints shader cycles per MFMA and the clock of workgroup 0's loop (tools/probe_h2.py). */
int nsr_probe(int device, int mode, int iters, int partner_prio, float* ms);

#ifdef __cplusplus
}
#endif
#endif /* NSR_PROBE_H_ */
