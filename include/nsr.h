/*
 * nsr.h -- C ABI of libnsr.so, the MI355X-native NeRF volumetric renderer.
 *
 * This is the drop-in boundary below the reference's Python render API.  The reference
 * (gyhandy/Neural-Sim-NeRF) has no native layer: its hot path is a chain of PyTorch ops in
 *   RN = optimization/utils/run_nerf_noscale.py, RH = optimization/utils/run_nerf_helpers.py.
 * Each entry point below replaces the reference functions cited next to it.  The binding a maintainer
 * adds on the reference side is a ctypes stub; `neural_sim_nerf_amd/_lib.py` is that stub.
 *
 * Conventions
 *   - plain C types only; every `const float*` / `float*` / `int64_t*` named d_* is a DEVICE pointer
 *     owned by the caller (e.g. a PyTorch-ROCm tensor's data_ptr); the library never frees them;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is stream-ordered.
 *     SETUP calls (nsr_create, nsr_destroy, nsr_upload_*, nsr_reserve_bbox, nsr_reserve_range, nsr_selftest,
 *     nsr_last_kernel_ms) may allocate and synchronise.  LAUNCH calls (everything else) only enqueue kernels on `stream`:
 *     no synchronisation, no environment reads, the calling thread's current HIP device is left as it was found, and no
 *     allocation -- with ONE exception outside stream capture: an f16x2 handle's render / input-gradient launch that is
 *     larger than every launch before it first grows the range safety net's list (nsr_reserve_range).  They can be
 *     captured into a hipGraph and replayed (tests/test_gpu_parity.py);
 *   - return value: 0 = OK, non-zero = error, message via nsr_last_error() (thread-local);
 *   - one handle per (model, stream): a handle owns one argument block, one work-queue head and its scratch
 *     buffers, so launches are ordered by the stream they are issued on; a launch on a DIFFERENT stream while the
 *     handle's previous launch is still running is refused with an error (never a silent race).  A handle is not
 *     thread-safe, distinct handles are.  The refusal covers EAGER launches: a launch that is being captured records
 *     no completion event, so a graph replay still running on stream A is invisible to a later eager launch (or a
 *     second replay) of the same handle on stream B -- order replays of one handle yourself (one stream, or events);
 *   - the arithmetic is fp32 everywhere outside the layer GEMMs, fp64 inside the two sequential scans where
 *     torch-CPU accumulates in fp64 (RN:376 cumprod, RH:203 cumsum).  The layer GEMMs run, per NsrConfig.flags, on
 *     fp32 MFMAs (v_mfma_f32_16x16x4_f32 / 32x32x2: exact fp32 products), on bf16 MFMAs with every fp32 operand split
 *     exactly into three bf16 pieces (NSR_FLAG_MLP_BF16X3) or on fp16 MFMAs with two fp16 pieces and power-of-two
 *     range management (NSR_FLAG_MLP_F16X2): all three accumulate in fp32 and deliver fp32-grade results (error
 *     against an fp64 evaluation within that of an fp32 GEMM chain: tests/test_host_logic.py, tests/test_gpu_parity.py).
 */
#ifndef NSR_H_
#define NSR_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NSR_ABI_VERSION 5       /* r06: + nsr_get_rays_views (nothing else changed since 4) */

/* Fixed architecture of the path (configs/nerf_param_ycbv_general.txt:12-13; NM:1232-1272). */
#define NSR_N_SAMPLES     64   /* N_samples    (coarse, RN:439)          */
#define NSR_N_IMPORTANCE  128  /* N_importance (fine,   RN:474)          */
#define NSR_NET_WIDTH     256  /* netwidth                               */
#define NSR_NET_DEPTH     8    /* netdepth, skip after layer 4 (RN:269)  */
#define NSR_MULTIRES      10   /* -> 63 position channels (RH:51-66)     */
#define NSR_MULTIRES_VIEWS 4   /* -> 27 direction channels               */

/* Size in floats of one packed network as produced by the host packer (pack.py / nsr_pack_layout). */
#define NSR_SLAB_FLOATS   4096                 /* one 16 KiB LDS ring slab                          */
#define NSR_STREAM_SLABS  145                  /* slabs per network pass                            */
#define NSR_AUX_FLOATS    3328                 /* biases + alpha/rgb heads, 13 KiB                  */
#define NSR_PACKED_FLOATS (NSR_STREAM_SLABS * NSR_SLAB_FLOATS + NSR_AUX_FLOATS)
/* the bf16x3 layout (NSR_FLAG_MLP_BF16X3): every weight as three bf16 pieces, 1.5x the stream */
#define NSR_STREAM_SLABS_B3  219
#define NSR_PACKED_B3_FLOATS (NSR_STREAM_SLABS_B3 * NSR_SLAB_FLOATS + NSR_AUX_FLOATS)
#define NSR_STREAM_SLABS_B3_BWD 234            /* slabs of the transposed fine network in that layout */
#define NSR_STREAM_SLABS_H2_BWD 146            /* ... and in the f16x2 layout (NSR_FLAG_MLP_F16X2)     */

typedef struct nsr_handle_s* nsr_handle;

typedef struct NsrConfig {
  int32_t abi_version;     /* must be NSR_ABI_VERSION                                              */
  int32_t device;          /* HIP device ordinal                                                   */
  int32_t n_samples;       /* N_samples (RN:439): 64; NSR_FLAG_MLP_F16X2 handles also take 32 (with n_importance 64 or 0) and 128
                              (with n_importance 128 or 0) -- r05: kernels specialised to those counts (one coarse network pass
                              with two idle waves / two coarse passes per item)                                              */
  int32_t n_importance;    /* N_importance (RN:474): 128, or 0 for a coarse-only render (BASELINE config 1); NSR_FLAG_MLP_F16X2
                              handles with n_samples = 64 also take 96, 64 and 32: kernels specialised to 64 + n fine samples
                              per ray -- two fine network passes per item instead of three for 64 and 32.  For a handle of
                              (n_samples, n_importance) = (S, n) every [.,64] / [.,128] / [.,192] array below is [.,S] / [.,n] /
                              [.,S + n] (d_u keeps its row stride of 128, the first n entries of a row are read; u_fine of
                              nsr_upload_tables: its first n entries)                                                         */
  int32_t max_workgroups;  /* 0 = fill the chip (one workgroup per CU for x32, two for x16)        */
  int32_t variant;         /* render AND input-gradient kernels: 0 = library default (= 16); 16 = 16 points/wave, two
                              workgroups per CU (needs nsr_upload_weights16 / _bwd16); 32 = 32 points/wave, one
                              workgroup per CU (nsr_upload_weights / _bwd)                                   */
  int32_t flags;           /* NSR_FLAG_* render options (0 = the YCB-V configuration)                */
  int32_t chunk;           /* x16 kernel, per-ray queue: rays per chunk of the work queue; 0 = default (1).  Larger chunks
                              trade load balance for L2 locality of the weight streams ("Chunk queue").
                              With NSR_FLAG_SCHED_PHASES: chunk - 1 = how many times a fine task looks for the handed-
                              over depths before it recomputes them (0 = default 64 looks; 1 = never look: test hook)  */
} NsrConfig;

#define NSR_FLAG_WHITE_BKGD 1   /* white_bkgd (RN:384-385): rgb_map += 1 - acc_map, coarse and fine; also in the VJP */
#define NSR_FLAG_LINDISP    2   /* lindisp (RN:443): coarse samples linear in inverse depth                          */
#define NSR_FLAG_SCHED_PHASES 4 /* x16 forward kernel: "global phases" work schedule (k_render16p) -- bit-identical
                                   results; every workgroup streams the same network most of the time, which cuts the
                                   L2-miss (fabric) traffic of the weight streams; see DESIGN.md 4 for speed vs traffic.
                                   The cross-workgroup hand-off it uses is non-blocking: a value that is not there in
                                   time is recomputed locally (nsr_schedule_stats counts those rays)                   */

#define NSR_FLAG_MLP_BF16X3 8    /* forward render kernel k_render_b3: the layer GEMMs run on bf16 MFMAs with every fp32 operand
                                   split exactly into three bf16 pieces and the six significant piece products
                                   accumulated in fp32 -- fp32-grade results (same error against fp64 as an fp32 GEMM)
                                   at ~1.9x the fp32-MFMA rate.  One workgroup per CU, 32 points per wave, per-item
                                   queue; needs nsr_upload_weights_b3.  nsr_render_rays_vjp runs the same scheme
                                   (k_render_vjp_b3, forward and transposed GEMMs) and needs nsr_upload_weights_bwd_b3;
                                   the stage kernels follow `variant` as before                                         */

#define NSR_FLAG_MLP_F16X2 16    /* forward render kernel k_render_h2: the layer GEMMs run on fp16 MFMAs with every fp32 operand
                                   split into two fp16 pieces (hi = fp16(x), lo = fp16(x - hi): 22 significand bits) and
                                   the three significant piece products accumulated in fp32; weights and biases carry exact
                                   power-of-two scales chosen by the packer so that no piece leaves the fp16 range -- fp32-
                                   grade results (error against fp64 within that of an fp32 GEMM chain) at half the MFMA
                                   work of bf16x3.  Domain: a hidden activation whose scaled magnitude reaches 65504 makes
                                   that point's outputs NaN inside the kernel (never a wrong number) -- see RANGE SAFETY NET below for what the caller gets.  One workgroup per CU, 32 points
                                   per wave; needs nsr_upload_weights_h2.  Mutually exclusive with NSR_FLAG_MLP_BF16X3.
                                   nsr_render_rays_vjp runs the same scheme (k_render_vjp_h2: forward and transposed GEMMs
                                   on fp16 MFMAs, the gradients of every point normalised by a power of two on entry) once
                                   nsr_upload_weights_bwd_h2 has been called; before that, the fp32 kernels of `variant`.
                                   RANGE SAFETY NET (ABI 3): a NaN never reaches the caller because of the fp16 range.  The
                                   f16x2 kernels append every work item (2 consecutive rays) in which a ray had a NaN
                                   network output or gradient to a device-side list, with the mask of those rays, and the
                                   same launch call enqueues the fp32 kernel of the same template (k_render / k_render_vjp,
                                   RH:99-118 has no range limit) over exactly that list -- list and length stay on the
                                   device, no host round trip; with an empty list every workgroup of the second launch
                                   returns at once (~10 us).  The second launch writes ONLY the reported rays: they then
                                   hold the fp32 kernel's results (a NaN the fp32 arithmetic itself produces -- NaN inputs,
                                   the encoder's domain -- stays a NaN), every other ray keeps the f16x2 kernel's bits,
                                   so a ray's result never depends on its neighbours.  nsr_range_status counts what happened.  The input-
                                   gradient launch needs nsr_upload_weights_bwd (fp32 transposed stream) for its fallback;
                                   without it every output of the affected rays is NaN and they are counted as dropped.    */

/* Optional per-ray debug taps of the fused kernel (all device pointers, any may be NULL).  128 / 192 = N_importance /
 * 64 + N_importance of the handle. */
typedef struct NsrDebugOut {
  float*   d_weights0;   /* [N,64]   coarse weights             (RN:467)   */
  float*   d_z_samples;  /* [N,128]  importance samples         (RH:241)   */
  int64_t* d_inds;       /* [N,128]  searchsorted indices       (RH:227)   */
  float*   d_z_fine;     /* [N,192]  sorted merged z            (RN:477)   */
  float*   d_raw0;       /* [N,64,4] coarse network output      (RN:466)   */
  float*   d_raw;        /* [N,192,4] fine network output       (RN:483)   */
} NsrDebugOut;

typedef struct NsrRenderOut {
  float* d_rgb;    /* [N,3]  rgb_map  (fine, or coarse when n_importance==0) */
  float* d_disp;   /* [N]    disp_map */
  float* d_acc;    /* [N]    acc_map  */
  float* d_rgb0;   /* [N,3]  coarse rgb_map  (NULL allowed)                  */
  float* d_disp0;  /* [N]                                                     */
  float* d_acc0;   /* [N]                                                     */
  float* d_z_std;  /* [N]    std of the importance samples (RN:495)           */
} NsrRenderOut;

const char* nsr_last_error(void);
int nsr_abi_version(void);

/* create_nerf (RN:258-340): allocates device-side weight storage + tables for one model. */
int nsr_create(const NsrConfig* cfg, nsr_handle* out);
int nsr_destroy(nsr_handle h);

/* Weight upload: `packed` is a HOST buffer of NSR_PACKED_FLOATS floats in the kernel layout
 * (pack.py).  net_id 0 = network_fn (coarse), 1 = network_fine.  Replaces the .to(device) of RN:269-278. */
int nsr_upload_weights(nsr_handle h, int net_id, const float* packed, size_t n_floats);

/* The same networks in the layout of the x16 forward kernel (pack.py: pack_network16). */
int nsr_upload_weights16(nsr_handle h, int net_id, const float* packed, size_t n_floats);

/* The same networks in the bf16x3 layout (pack.py: pack_network_b3; NSR_PACKED_B3_FLOATS floats: the stream holds
 * packed bf16 pairs, the aux block is the fp32 one of nsr_upload_weights).  Handles created with NSR_FLAG_MLP_BF16X3. */
int nsr_upload_weights_b3(nsr_handle h, int net_id, const float* packed, size_t n_floats);

/* The same networks in the f16x2 layout (pack.py: pack_network_h2; NSR_PACKED_FLOATS floats: the stream holds packed
 * fp16 pairs -- 2 pieces x 2 bytes = the fp32 stream's size -- the aux block holds the scaled biases / heads and the
 * activation scales).  Handles created with NSR_FLAG_MLP_F16X2. */
int nsr_upload_weights_h2(nsr_handle h, int net_id, const float* packed, size_t n_floats);

/* Transposed stream of the FINE network in the f16x2 layout (pack.py: pack_network_backward_h2; NSR_STREAM_SLABS_H2_BWD *
 * NSR_SLAB_FLOATS floats; its multipliers live in the aux block nsr_upload_weights_h2 received for net_id 1). */
int nsr_upload_weights_bwd_h2(nsr_handle h, const float* stream, size_t n_floats);

/* Transposed stream of the FINE network in the bf16x3 layout (pack.py: pack_network_backward_b3;
 * NSR_STREAM_SLABS_B3_BWD * NSR_SLAB_FLOATS floats), for nsr_render_rays_vjp on an NSR_FLAG_MLP_BF16X3 handle. */
int nsr_upload_weights_bwd_b3(nsr_handle h, const float* stream, size_t n_floats);

/* Transposed stream of the FINE network for the input-gradient kernel (pack.py: pack_network_backward);
 * host buffer of NSR_STREAM_SLABS*NSR_SLAB_FLOATS floats.  Needed only by nsr_render_rays_vjp. */
int nsr_upload_weights_bwd(nsr_handle h, const float* stream, size_t n_floats);

/* The same transposed stream in the x16 layout (pack.py: pack_network_backward16), for the x16 forward+input-gradient
 * kernel k_render_vjp16 that variant 0 / 16 handles use (two workgroups per CU, one ray per work item). */
int nsr_upload_weights_bwd16(nsr_handle h, const float* stream, size_t n_floats);

/* The two linspace tables the reference builds on the host and moves to the device
 * (RN:439 t_vals[N_samples], RH:208 u[128]); host buffers; n_coarse = the handle's n_samples, n_fine = 128. */
int nsr_upload_tables(nsr_handle h, const float* t_coarse, int n_coarse, const float* u_fine, int n_fine);

/* render(rays=...) -> batchify_rays -> render_rays (RN:58-123, RN:43-55, RN:390-501), use_viewdirs=True, on the
 * deterministic test-time path (ndc=False, perturb=0, raw_noise_std=0; white_bkgd / lindisp per the handle's flags).
 * d_rays_o, d_rays_d: [N,3].  viewdirs = rays_d/|rays_d| are computed in-kernel (RN:97). */
int nsr_render_rays(nsr_handle h, const float* d_rays_o, const float* d_rays_d, int64_t n_rays,
                    float near_, float far_, const NsrRenderOut* out, const NsrDebugOut* dbg, void* stream);

/* The per-ray inputs of the options render() / render_rays() have beyond that path.  Every pointer is nullable (NULL =
 * the option is off) and addresses caller-owned DEVICE memory.  The random draws are the CALLER's (the reference takes
 * them from torch's global generator inside render_rays, once per 'chunk' of rays and in the order t_rand, noise0, u,
 * noise1): the library adds no generator of its own, so a caller that hands over the draws the reference made gets the
 * reference's render (tests/golden/g14_stochastic.npz), and one that draws [N, .] arrays in the same order reproduces the
 * reference's stream whenever N <= chunk.
 *   d_viewdirs [N,3]  view directions to use instead of rays_d/|rays_d|: render(c2w_staticcam=...) takes the rays of the
 *                     static camera and the view directions of c2w (RN:91-96); render(ndc=True) takes them from the
 *                     rays BEFORE ndc_rays (RN:89-103, nsr_ndc_rays below)
 *   d_t_rand  [N,64]  perturb > 0: stratified jitter in [0,1), z = lower + (upper - lower) * t_rand (RN:447-459)
 *   d_u       [N,128] sample_pdf with det=False (perturb > 0, RN:474): the uniforms of RH:211; the importance samples then
 *                     arrive unsorted and the merged depths are fully sorted (RN:477)
 *   d_noise0  [N,64]  raw_noise_std > 0: raw_noise_std * randn added to the coarse densities before the relu (RN:365-374)
 *   d_noise1  [N,192] ... to the fine densities (the raw returned through NsrDebugOut stays the network's output)
 *   d_near, d_far [N] per-ray bounds: render()'s near / far may be arrays (RN:106-108); both or neither, they replace
 *                     the call's scalars
 * Served by the x32-structured kernels: an fp32 handle of any `variant` runs k_render / k_render_vjp for such a call
 * (the VJP then needs nsr_upload_weights_bwd), bf16x3 / f16x2 handles run their usual kernels. */
typedef struct NsrRayExtras {
  const float* d_viewdirs;
  const float* d_t_rand;
  const float* d_u;
  const float* d_noise0;
  const float* d_noise1;
  const float* d_near;
  const float* d_far;
} NsrRayExtras;

/* nsr_render_rays with the extras above (ex may be NULL: identical to nsr_render_rays). */
int nsr_render_rays_ex(nsr_handle h, const float* d_rays_o, const float* d_rays_d, int64_t n_rays, float near_,
                       float far_, const NsrRayExtras* ex, const NsrRenderOut* out, const NsrDebugOut* dbg, void* stream);

/* render(c2w=...) for a batch of views (RN:84-86 get_rays + the above; render_path's loop RN:229-235 is
 * folded into one launch).  d_c2w: [n_views,3,4] row-major; K9: HOST 3x3 intrinsics (row-major, as the
 * python floats of LL:177); ray r of view v is pixel (row = r / W, col = r % W); outputs are
 * [n_views*H*W, ...].  Rays are generated in-kernel: no HBM traffic for ray origins/directions. */
int nsr_render_views(nsr_handle h, const float* d_c2w, int n_views, int H, int W, const double* K9,
                     float near_, float far_, const NsrRenderOut* out, const NsrDebugOut* dbg, void* stream);

/* Forward + input-side VJP in one launch: what render_path_grad obtains per 512-ray patch with
 * torch.autograd.grad(rgb_p, batch_rays, grad_outputs=patch_grad_E) (RN:168-178).  Network weights are
 * constants and z_samples is detached (RN:475), so the gradient reaches the rays only through the fine pass.
 * d_grad_rgb [N,3] cotangent of rgb_map -> d_grad_o, d_grad_d [N,3].  `out` (optional, may be NULL; only
 * d_rgb/d_disp/d_acc are written) receives the forward render of the same launch.
 * d_z_fine (optional, may be NULL) [N,192]: sorted fine sample depths (RN:477) to use INSTEAD of the kernel's own
 * resampling -- they are constants of the backward pass (RN:475), so a caller that already holds them (or a parity
 * test holding the reference's) gets the gradient at exactly those depths. */
int nsr_render_rays_vjp(nsr_handle h, const float* d_rays_o, const float* d_rays_d, int64_t n_rays, float near_,
                        float far_, const float* d_grad_rgb, float* d_grad_o, float* d_grad_d,
                        const float* d_z_fine, const NsrRenderOut* out, void* stream);

/* Debug taps of the input-gradient launch (x32-structured kernels: f16x2, bf16x3 and fp32 `variant` 32 handles; an fp32
 * handle of another variant runs k_render_vjp for a call with taps, like for the extras).  All pointers nullable.
 *   d_relu_masks [ceil(N/2)][3][9][256][4] uint32 (3 = fine passes per item = ceil(2 (n_samples + n_importance) / 128): 2 on
 *       (64, 64), (64, 32) and (32, 64) handles, 4 on a (128, 128) handle; there the
 *       point mapping is q = 128 p + 32 w + j -> sample q % (64 + n) of ray 2t + q / (64 + n)): the relu patterns the backward pass applied, as captured by the
 *       forward passes of the same launch -- item t = rays 2t, 2t+1; fine pass p of the item covers the 128 points
 *       q = 128 p + 32 w + j (wave w = thread / 64, j = thread % 32) = sample q % 192 of ray 2t + q / 192; layer 0..7 =
 *       pts_linears, 8 = views_linears.0; thread (w, lane) holds, for its point, the units
 *       32 mo + (r & 3) + 8 (r >> 2) + 4 (lane >> 5) in bit 31 - (16 (mo & 1) + r) of word mo >> 1, SET = unit OFF
 *       (pre-activation <= 0)
 *   d_grad_raw  [N,192,4]  dL/d raw of the fine samples (rgb logits, sigma): what the compositing backward hands to the
 *       network backward
 *   d_grad_pts  [N,192,6]  per fine sample: dL/d pts (3) and dL/d viewdirs (3), the results of the network backward
 * oracle/vjp_census.py replays the oracle's fp64 backprop with these to attribute a ray's gradient error to relu units
 * whose pre-activation sits at the discontinuity. */
typedef struct NsrVjpDebugOut {
  uint32_t* d_relu_masks;
  float*    d_grad_raw;
  float*    d_grad_pts;
} NsrVjpDebugOut;

/* nsr_render_rays_vjp with the extras: the forward half of the launch repeats the render with the same draws (or takes
 * d_z_fine), the backward half sees the noisy fine densities (the relu' of RN:374).  With ex->d_viewdirs the view
 * directions are an input of their own: d_grad_viewdirs [N,3] (nullable) receives dL/d viewdirs and d_grad_d holds only
 * the paths through the points and through dists * |rays_d| (RN:361); without it d_grad_viewdirs must be NULL. */
int nsr_render_rays_vjp_ex(nsr_handle h, const float* d_rays_o, const float* d_rays_d, int64_t n_rays, float near_,
                           float far_, const NsrRayExtras* ex, const float* d_grad_rgb, float* d_grad_o, float* d_grad_d,
                           float* d_grad_viewdirs, const float* d_z_fine, const NsrRenderOut* out, void* stream);

/* ... and with the debug taps (dbg may be NULL: identical to nsr_render_rays_vjp_ex). */
int nsr_render_rays_vjp_dbg(nsr_handle h, const float* d_rays_o, const float* d_rays_d, int64_t n_rays, float near_,
                            float far_, const NsrRayExtras* ex, const float* d_grad_rgb, float* d_grad_o, float* d_grad_d,
                            float* d_grad_viewdirs, const float* d_z_fine, const NsrRenderOut* out,
                            const NsrVjpDebugOut* dbg, void* stream);

/* ndc_rays (RH:168-186), the projection render(ndc=True) applies to the rays of forward-facing scenes before rendering
 * them with near=0, far=1 (RN:101-103): [N,3] x 2 -> [N,3] x 2 in torch's fp32 op order (bit-exact against the reference,
 * tests/golden/g14_stochastic.npz).  focal = K[0][0]; near_ = 1.0 at the reference's call site.  _vjp: (dL/d o', dL/d d')
 * -> (dL/d rays_o, dL/d rays_d), the link autograd adds between the rays and nsr_render_rays_vjp_ex. */
int nsr_ndc_rays(nsr_handle h, const float* d_rays_o, const float* d_rays_d, int64_t n_rays, int H, int W, double focal,
                 double near_, float* d_o_out, float* d_d_out, void* stream);
int nsr_ndc_rays_vjp(nsr_handle h, const float* d_rays_o, const float* d_rays_d, int64_t n_rays, int H, int W, double focal,
                     double near_, const float* d_grad_o_ndc, const float* d_grad_d_ndc, float* d_grad_o, float* d_grad_d,
                     void* stream);

/* Chain rule through get_rays (RH:160-164, linear in c2w): per patch of `patch` consecutive pixels (row-major,
 * the order of RN:150-157), d_out[p] = dL/d c2w[3][4] given dL/d rays.  n_patches = ceil(H*W / patch). */
int nsr_pose_grad(nsr_handle h, const float* d_grad_o, const float* d_grad_d, int H, int W, const double* K9,
                  int patch, float* d_out, void* stream);

/* Stage entry points (same device code as the fused kernel; used by the parity tests and usable alone). */

/* get_rays (RH:156-165): d_rays_o, d_rays_d [H*W,3] for one c2w [3,4]. */
int nsr_get_rays(nsr_handle h, const float* d_c2w, int H, int W, const double* K9,
                 float* d_rays_o, float* d_rays_d, void* stream);
/* ... for n_views cameras d_c2w [n_views,3,4] in ONE launch: d_rays_o, d_rays_d [n_views*H*W,3], view-major (r06: what the
 * layered renderer's render(c2w=...) / render_path feed to nsrw_render_rays; the fused kernels generate their rays themselves). */
int nsr_get_rays_views(nsr_handle h, const float* d_c2w, int n_views, int H, int W, const double* K9,
                       float* d_rays_o, float* d_rays_d, void* stream);

/* Embedder.embed (RH:18-48, get_embedder RH:51-66 with i_embed = 0): d_x [n,3] -> d_out [n, 3 + 6*multires] in the
 * reference's channel order [x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)].  This is the
 * encoding the fused kernels evaluate in registers; coordinates with 2^(L-1)|x| >= 2^24 (outside any NeRF scene)
 * encode to NaN rather than to an inaccurate value. */
int nsr_embed(nsr_handle h, const float* d_x, int64_t n, int multires, float* d_out, void* stream);

/* run_network (RN:26-40) = Embedder (RH:18-48) + NeRF MLP (RH:99-122): d_pts [P,3], d_viewdirs [P,3]
 * (already unit length) -> d_raw [P,4]. */
int nsr_run_network(nsr_handle h, int net_id, const float* d_pts, const float* d_viewdirs, int64_t n_pts,
                    float* d_raw, void* stream);

/* raw2outputs (RN:343-387), n_samples in {64,192}: d_raw [N,S,4], d_z [N,S], d_rays_d [N,3]. */
int nsr_raw2outputs(nsr_handle h, const float* d_raw, const float* d_z, const float* d_rays_d, int64_t n_rays,
                    int n_samples, float* d_rgb, float* d_disp, float* d_acc, float* d_weights, float* d_depth,
                    void* stream);

/* sample_pdf (RH:199-243), det=True: d_bins [N,63], d_weights [N,62] -> d_samples [N,128], d_inds [N,128]. */
int nsr_sample_pdf(nsr_handle h, const float* d_bins, const float* d_weights, int64_t n_rays,
                   float* d_samples, int64_t* d_inds, void* stream);

/* z_vals, _ = torch.sort(torch.cat([z_vals, z_samples], -1), -1) (RN:477): d_z_coarse [N,64], d_z_samples [N,128]
 * -> d_z_sorted [N,192].  Exact for any input (a merge when both halves are already ordered, a full rank count
 * otherwise). */
int nsr_sort_merge(nsr_handle h, const float* d_z_coarse, const float* d_z_samples, int64_t n_rays,
                   float* d_z_sorted, void* stream);

/* Image hand-off to the detector's loader without the PNG round trip.
 * nsr_to8b: to8b (RH:14) = (255 * clip(x, 0, 1)).astype(uint8) over n floats (truncation; NaN -> 0).
 * nsr_find_bbox: what get_annotation / find_bbox (NM:786-797) derive from the PNG read back with cv2, for
 *   d_rgb8 [n_images,H,W,3] uint8 RGB: gray = cv2.cvtColor(BGR image, COLOR_RGB2GRAY) (the reference's channel
 *   swap kept), mask = gray > 1, 8-connected components with statistics, the largest-area row dropped, then the row
 *   with the largest w*h (NM:691-692).  d_bbox [n_images,4] int32 = x, y, w, h; d_count [n_images] = number of rows
 *   left after the drop (0 = the reference would raise on this image; the box is then 0,0,0,0); d_mask (nullable)
 *   [n_images,H,W] uint8 0/255.  H*W <= 2^20.  Its scratch (24 B per pixel for 16 images in flight) is allocated by
 *   the SETUP call nsr_reserve_bbox(h, H, W), which must have been made for an image at least this large. */
int nsr_to8b(nsr_handle h, const float* d_x, int64_t n, uint8_t* d_out, void* stream);
int nsr_find_bbox(nsr_handle h, const uint8_t* d_rgb8, int n_images, int H, int W, int32_t* d_bbox,
                  int32_t* d_count, uint8_t* d_mask, void* stream);
int nsr_reserve_bbox(nsr_handle h, int H, int W);

/* Content fingerprint of n_tensors device buffers (d_ptrs[i]: n_words[i] 32-bit words; both tables in device memory):
 * an order-independent 64-bit hash of (tensor, index, bits) written to *d_out.  LAUNCH call (a memset node + one
 * kernel).  The drop-in API keys its packed-weight cache on it, so that in-place parameter writes which bump no autograd
 * version are seen without concatenating and reducing 0.6 M parameters per render call. */
int nsr_fingerprint(nsr_handle h, const void* const* d_ptrs, const int64_t* d_n_words, int n_tensors, uint64_t* d_out,
                    void* stream);

/* psi -> camera pose on the device (LL = optimization/utils/load_LINEMOD_noscale.py, GU = utils/gumble.py).
 * The random draws stay where the reference makes them (numpy on the host, recorded in sample_log, LL:273-297); these
 * are the deterministic maps (probabilities, recorded noise) -> poses, written straight into device memory.
 * nsr_sample_pose = sample_pose LL:202-247 (GU:57-63, pose_spherical LL:62-71) in torch's fp32 arithmetic:
 *   d_prob [n_cat] fp32; d_gumbel [K,n_cat], d_uniform [K], d_theta [K] fp64 (the python floats of sample_log);
 *   outputs (each nullable): d_poses44 [K,4,4], d_c2w34 [K,3,4] (= what nsr_render_views reads), d_jac [K,12,n_cat] =
 *   d c2w[:3,:4] / d prob (the chain RN:179-181 needs).  Bin centres are 45 j + 22.5 degrees (LL:217); n_cat <= 16.
 * nsr_sample_pose_nograd = sample_pose_nograd LL:250-301 (GU:46-47, GU:64-70, pose_spherical_nograd LL:89-94) in numpy's
 *   fp64 arithmetic: d_logits [n_cat] fp64 = np.log(probs) exactly as the caller's numpy computed it (NM:88 makes the
 *   probabilities float16, so the log is a float16 value). */
int nsr_sample_pose(nsr_handle h, const float* d_prob, const double* d_gumbel, const double* d_uniform,
                    const double* d_theta, int K, int n_cat, double gumbel_T, double radius, float* d_poses44,
                    float* d_c2w34, float* d_jac, void* stream);
int nsr_sample_pose_nograd(nsr_handle h, const double* d_logits, const double* d_gumbel, const double* d_uniform,
                           const double* d_theta, int K, int n_cat, double gumbel_T, double radius, float* d_poses44,
                           float* d_c2w34, void* stream);

/* Device self-test of the MFMA fragment-layout assumptions the packer relies on. Returns 0 if they hold. */
int nsr_selftest(nsr_handle h, void* stream);

/* Global-phases schedule: number of rays (cumulative over the handle's launches) whose fine task recomputed the coarse
 * pass because the handed-over depths were not there in time -- 0 in normal operation, > 0 when the GPU is shared with
 * other work; never an error.  0 for handles without NSR_FLAG_SCHED_PHASES.  Synchronises the device. */
int nsr_schedule_stats(nsr_handle h, unsigned* recomputed_rays);

/* f16x2 range safety net (NSR_FLAG_MLP_F16X2): what the handle's launches reported so far.  *last_items = items (2 rays)
 * the LAST launch handed to its fp32 fallback; cumulative over the handle's life: *points = network evaluations whose
 * outputs / gradients were NaN, *rays = rays rendered again by the fp32 kernel, *dropped_items = items that could not be:
 * EVERY OUTPUT OF THEIR REPORTED RAYS IS NaN (never a finite number computed from out-of-range activations).  An item is
 * dropped only (a) by a launch CAPTURED into a hipGraph that is larger than the list was when the capture began (a capture
 * cannot allocate: call nsr_reserve_range first) or (b) by an input-gradient launch without nsr_upload_weights_bwd (no fp32
 * transposed stream to fall back to).  (A launch that cannot get memory for its list FAILS: r06.)  Any pointer may be NULL.
 * All zero for other handles.  Synchronises the device. */
int nsr_range_status(nsr_handle h, unsigned* last_items, unsigned* points, unsigned* rays, unsigned* dropped_items);

/* SETUP call: make the safety net's list large enough for launches of up to n_rays rays (8 bytes per 2 rays).  EAGER launch
 * calls do this themselves -- nsr_render_rays* / nsr_render_views / nsr_render_rays_vjp* grow the list to their own size
 * before they enqueue anything (one hipMalloc per new largest size, no synchronisation; the outgrown list stays allocated
 * until nsr_destroy because a launch in flight or a graph captured earlier still addresses it) -- so this call is needed
 * only before CAPTURING a launch larger than anything the handle has launched so far (at most 2^33 - 2 rays).  Fails while
 * the handle's stream is being captured.  No-op for handles without NSR_FLAG_MLP_F16X2. */
int nsr_reserve_range(nsr_handle h, int64_t n_rays);

/* Debug build (`make -C neural_sim_nerf_amd/csrc debug` -> libnsr_debug.so, -DNSR_DEBUG_BOUNDS): every data-dependent
 * LDS / scratch index of the kernels (searchsorted results, merge ranks, hand-off slots) is range-checked; a violation
 * is clamped and its source line recorded.  *built_with_checks = 0 for the release library (then *first_bad_line = 0).
 * Synchronises the device. */
int nsr_debug_bounds_status(nsr_handle h, int* built_with_checks, unsigned* first_bad_line);

/* Timing helper for bench.py: HIP-event time in ms of the last EAGER nsr_render_* launch on this handle
 * (events recorded on the launch stream; this call synchronises on the stop event).  Launches made while the
 * stream is being captured into a graph are not timed. */
int nsr_last_kernel_ms(nsr_handle h, float* ms);

#ifdef __cplusplus
}
#endif
#endif /* NSR_H_ */
