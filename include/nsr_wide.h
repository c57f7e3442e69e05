/*
 * nsr_wide.h -- C ABI of the LAYERED renderer inside libnsr.so: the same path as include/nsr.h (render_rays RN:390-501 and its
 * input-side VJP, RN:168-178) for the configurations the fused kernels are not built for -- a NeRF (RH:70-122) of ANY depth,
 * width and skip list, any N_samples / N_importance (RN:439, RN:474), with or without view directions.
 *
 * (RN = optimization/utils/run_nerf_noscale.py, RH = optimization/utils/run_nerf_helpers.py of the reference.)
 *
 * Design: one MFMA GEMM kernel per network layer over ALL points of a chunk of rays -- on fp16 MFMAs with two-piece
 * operands, on bf16 MFMAs with three-piece operands or on fp32 MFMAs (NsrwConfig.flags), fp32 in HBM and fp32-grade results in
 * every case -- activations resident in HBM between the layers (288 GB: a chunk is thousands of rays), the per-ray stages (depths,
 * compositing RN:343-387, resampling RH:199-243, sort RN:477) as small kernels of their own between them, and for the gradient the
 * same GEMM kernel on the transposed weights with the relu masks read back from the stored activations.  Nothing here runs on the host CPU
 * and nothing falls back to another library: a network the fused kernels cannot hold costs layer-by-layer HBM traffic, not
 * correctness.
 *
 * Conventions are those of include/nsr.h: int status (0 = OK, message via nsrw_last_error), caller-owned DEVICE buffers passed
 * as raw pointers, stream-ordered, no hidden synchronisation in the launch calls, one handle per (model, stream); a handle is
 * not thread-safe, distinct handles are.  The scratch memory is the CALLER's too: nsrw_workspace_bytes says how much one
 * chunk of rays needs, the launch calls split the rays into as many chunks as the workspace they are handed allows.
 */
#ifndef NSR_WIDE_H_
#define NSR_WIDE_H_
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct NsrwHandle_* nsrw_handle;

#define NSRW_MAX_SKIPS 16
#define NSRW_MAX_DEPTH 64
#define NSRW_MAX_WIDTH 4096
#define NSRW_MAX_SAMPLES 512         /* N_samples, N_importance (torch.sum's first cascade level: oracle/_torch_sum_lastdim) */

enum { NSRW_FLAG_WHITE_BKGD = 1,     /* RN:384-385 */
       NSRW_FLAG_LINDISP = 2,        /* RN:443 */
       NSRW_FLAG_MLP_BF16X3 = 4,     /* r06: the layer GEMMs on bf16 MFMAs, every fp32 operand split exactly into three bf16 pieces, six
                                      * piece products per product, fp32 accumulate (csrc/nsr_wide_b3.inc): fp32-grade results, fp32's
                                      * exponent range, 2.67x the matrix-pipe rate of the fp32 MFMAs.  Without a flag: fp32 MFMAs. */
       NSRW_FLAG_MLP_F16X2 = 8 };    /* r06: the layer GEMMs on fp16 MFMAs, every fp32 operand as two fp16 pieces, three piece products
                                      * (the arithmetic of the fused default kernel, csrc/nsr_h2.inc: |error| <= 2^-22 per product;
                                      * weights pre-scaled per matrix, gradients normalised per point).  A network pass -- forward or
                                      * backward -- in which a value reaches fp16's largest number is run again on bf16x3 inside the
                                      * same call (nsrw_range_status counts them), so no result depends on the range. */

typedef struct NsrwConfig {
  int32_t device;
  int32_t n_samples;                 /* N_samples >= 2 (RN:439) */
  int32_t n_importance;              /* N_importance >= 0 (RN:474); 0 = coarse only */
  int32_t flags;
  int32_t reserved[4];
} NsrwConfig;

/* One NeRF module (RH:70-97).  input_ch = 3 + 6 multires, input_ch_views = 3 + 6 multires_views (get_embedder RH:51-66;
 * multires = 0 is i_embed = -1, the identity).  skips: layer indices i whose OUTPUT is concatenated behind the input
 * encoding (RH:105-106), each < D - 1.  use_viewdirs = 0: outputs = output_linear(h) with output_ch rows (RH:119-120), of
 * which render_rays reads the first four. */
typedef struct NsrwNet {
  int32_t D, W;
  int32_t multires, multires_views;
  int32_t use_viewdirs;
  int32_t output_ch;
  int32_t n_skips;
  int32_t skips[NSRW_MAX_SKIPS];
} NsrwNet;

/* Per-ray inputs of the options beyond the deterministic test-time path, all nullable DEVICE pointers (cf. NsrRayExtras):
 * viewdirs [N,3] given view directions (c2w_staticcam RN:91-96, ndc RN:101-103); near / far [N] (RN:106-108);
 * t_rand [N, N_samples] (RN:451), u [N, N_importance] (RH:211), noise0 [N, N_samples] / noise1 [N, N_samples + N_importance]
 * (RN:365-374, already multiplied by raw_noise_std); z_fine [N, N_samples + N_importance]: sorted sample depths the fine pass
 * is to use INSTEAD of its own resampling (z_samples is detached, RN:475: the depths are constants of the gradient -- this is how
 * the parity tests differentiate at the reference's own depths). */
typedef struct NsrwExtras {
  const float* d_viewdirs;
  const float* d_near;
  const float* d_far;
  const float* d_t_rand;
  const float* d_u;
  const float* d_noise0;
  const float* d_noise1;
  const float* d_z_fine;
} NsrwExtras;

/* Outputs, all nullable DEVICE pointers: the returns of render_rays (RN:488-495) -- rgb [N,3], disp [N], acc [N] of the last
 * pass, rgb0 / disp0 / acc0 of the coarse pass and z_std [N] when N_importance > 0 -- plus the taps the parity tests read:
 * raw [N, S, C] of the last pass (retraw; C = 4, or output_ch without view directions), z_vals [N, S] of the last pass,
 * weights0 [N, N_samples] of the coarse pass, z_samples [N, N_importance], inds int64 [N, N_importance], raw0 [N, N_samples, C]
 * of the coarse pass when there is a fine one. */
typedef struct NsrwOut {
  float* d_rgb;
  float* d_disp;
  float* d_acc;
  float* d_rgb0;
  float* d_disp0;
  float* d_acc0;
  float* d_z_std;
  float* d_raw;
  float* d_z_vals;
  float* d_weights0;
  float* d_z_samples;
  int64_t* d_inds;
  float* d_raw0;
} NsrwOut;

const char* nsrw_last_error(void);

int nsrw_create(const NsrwConfig* cfg, nsrw_handle* out);
int nsrw_destroy(nsrw_handle h);

/* Network net_id (0 = network_fn, 1 = network_fine) from HOST memory: the parameters in the module's own order and layout
 * (torch [out, in] row-major), weight then bias per layer -- pts_linears.0 .. D-1, then with view directions feature_linear,
 * alpha_linear, views_linears.0, rgb_linear, without them output_linear.  SETUP call: allocates, copies, synchronises. */
int nsrw_upload_network(nsrw_handle h, int net_id, const NsrwNet* net, const float* weights, size_t n_floats);
/* Number of floats nsrw_upload_network expects for `net` (0 if the description is invalid; nsrw_last_error says why). */
size_t nsrw_network_floats(const NsrwNet* net);

/* torch.linspace(0, 1, N_samples) and torch.linspace(0, 1, N_importance) as the HOST's torch computes them (RN:439, RH:208:
 * ATen's values are not np.linspace's); n_fine may be 0.  SETUP call. */
int nsrw_upload_tables(nsrw_handle h, const float* t_coarse, int n_coarse, const float* u_fine, int n_fine);

/* Bytes of workspace ONE chunk of `rays` rays needs (with_grad: for nsrw_render_rays_vjp, which keeps every activation of the
 * fine network).  The launch calls accept any workspace that holds a chunk of at least 64 rays. */
int nsrw_workspace_bytes(nsrw_handle h, int64_t rays, int with_grad, size_t* bytes);

/* render_rays (RN:390-501) over n_rays rays.  near / far: scalars, overridden per ray by ex->d_near / d_far. */
int nsrw_render_rays(nsrw_handle h, const float* d_rays_o, const float* d_rays_d, int64_t n_rays, float near_, float far_,
                     const NsrwExtras* ex, const NsrwOut* out, void* d_workspace, size_t workspace_bytes, void* stream);

/* The same forward plus d(sum(rgb * grad_rgb)) / d(rays_o, rays_d) (RN:177; weights frozen, z_samples detached RN:475, so
 * the gradient flows through the LAST pass only): d_grad_o, d_grad_d [N,3]; d_grad_viewdirs [N,3] iff ex->d_viewdirs is given
 * (d_grad_d then holds no view-direction term).  out (nullable) receives the forward's results. */
int nsrw_render_rays_vjp(nsrw_handle h, const float* d_rays_o, const float* d_rays_d, int64_t n_rays, float near_, float far_,
                         const NsrwExtras* ex, const float* d_grad_rgb, const NsrwOut* out, float* d_grad_o, float* d_grad_d,
                         float* d_grad_viewdirs, void* d_workspace, size_t workspace_bytes, void* stream);

/* run_network (RN:26-40): d_pts [P,3], d_viewdirs [P,3] (unit length; ignored without view directions) -> d_raw [P, C],
 * C = 4 or output_ch.  Workspace: nsrw_workspace_bytes(h, ceil(P / max(N_samples + N_importance, 1)), 0) suffices. */
int nsrw_run_network(nsrw_handle h, int net_id, const float* d_pts, const float* d_viewdirs, int64_t n_pts, float* d_raw,
                     void* d_workspace, size_t workspace_bytes, void* stream);

/* Stage entries that need no handle (stream-ordered launch calls on `device`; they restore the caller's current device).
 * nsrw_sample_pdf = sample_pdf (RH:199-243) for ANY bin and sample count: d_bins [N, n_bins], d_weights [N, n_bins - 1], the
 * uniforms d_u -- [n_samples] shared by all rows (det=True: the HOST's torch.linspace(0, 1, n_samples), RH:208) or, u_per_row != 0,
 * [N, n_samples] (det=False: the caller's torch.rand, RH:211; the library has no generator) -> d_samples [N, n_samples], d_inds
 * (nullable) int64 [N, n_samples] = searchsorted(cdf, u, right=True) (RH:227).  d_scratch: 2 N (n_bins + 1) floats.  Bit-exact
 * against the reference: torch.sum's association order, fp64-accumulate cdf.
 * nsrw_embed_vjp = the VJP of Embedder.embed (RH:39-48): d_x [P,3], d_grad_out [P, 3 + 6 multires] -> d_grad_x [P,3]. */
int nsrw_sample_pdf(int device, const float* d_bins, const float* d_weights, int64_t n_rows, int n_bins, const float* d_u,
                    int u_per_row, int n_samples, float* d_samples, int64_t* d_inds, float* d_scratch, void* stream);
int nsrw_embed_vjp(int device, const float* d_x, const float* d_grad_out, int64_t n_points, int multires, float* d_grad_x, void* stream);

/* NSRW_FLAG_MLP_F16X2 handles: network passes (forward and backward) launched so far and how many of them were re-run on bf16x3
 * because an activation or a gradient left fp16's range (both 0 for the other arithmetics).  Synchronises the device. */
int nsrw_range_status(nsrw_handle h, unsigned long long* passes, unsigned long long* passes_rerun);

/* Device time of the last launch call (ms; synchronises on its closing event) and the number of chunks it ran. */
int nsrw_last_ms(nsrw_handle h, float* ms, int* chunks);

/* libnsr_debug.so (-DNSR_DEBUG_BOUNDS) checks every global / LDS index of this unit's kernels against its extent; a violation is
 * recorded, not trapped.  built_with_checks: 1 in that build, 0 in the product library; first_bad_line: 0 = clean, else the source
 * line of the (highest) failed check -- nsr_wide.hip's line, or 100000 + the line of nsr_wide_b3.inc.  Synchronises the device. */
int nsrw_debug_bounds_status(int* built_with_checks, unsigned* first_bad_line);

#ifdef __cplusplus
}
#endif
#endif /* NSR_WIDE_H_ */
