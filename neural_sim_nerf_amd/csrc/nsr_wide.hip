// nsr_wide.hip -- the LAYERED renderer (include/nsr_wide.h): render_rays (RN:390-501) and its input-side VJP (RN:168-178) for
// NeRFs (RH:70-122) of any depth / width / skip list and any sample counts, i.e. everything the fused kernels of
// nsr_kernels.hip are not specialised to.  A translation unit of its own (its kernels share nothing with the fused ones but the
// arithmetic they restate), linked into libnsr.so.
//
// Shape of the computation.  The rays of a call are cut into chunks of R rays; per chunk
//   depths -> [encode -> D+3 GEMMs -> composite] (coarse) -> sample_pdf -> sort -> [encode -> GEMMs -> composite] (fine)
// with every activation matrix [points, width] resident in the caller's workspace (HBM), and for the gradient
//   composite_bwd -> the same GEMM kernel on the transposed weights, relu masks read from the stored activations
//   -> encode_bwd + the per-ray reductions.
// The GEMM (kw_gemm) is fp32 in, fp32 MFMA (v_mfma_f32_32x32x2_f32), fp32 out: the reference's arithmetic, no operand
// splitting, no range to manage.  Every K and N extent is padded to a multiple of 32 with zero weights, so no kernel carries a
// K tail; M (points) is arbitrary.
//
// (RN = optimization/utils/run_nerf_noscale.py, RH = optimization/utils/run_nerf_helpers.py of the reference.)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/nsr_wide.h"

namespace nsrw {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// `make debug` (-DNSR_DEBUG_BOUNDS, libnsr_debug.so): every global / LDS index the kernels of this unit compute is checked against
// the extent it must stay inside; a violation records (file tag + source line) in a device word -- no trap (that would take the
// HSA queue down), the evidence is read back by nsrw_debug_bounds_status().  Release build: nothing.
#ifdef NSR_DEBUG_BOUNDS
__device__ unsigned g_wide_violation = 0u;
#define NSRW_CHECK(cond)                                                                             \
  do {                                                                                               \
    if (!(cond)) atomicMax(&nsrw::g_wide_violation, (unsigned)(NSRW_FILE_TAG + __LINE__));           \
  } while (0)
#else
#define NSRW_CHECK(cond) do { } while (0)
#endif
#define NSRW_FILE_TAG 0            /* nsr_wide.hip: the line itself; nsr_wide_b3.inc: 100000 + line */

// ------------------------------------------------------------------------------------------------------------------------
// GEMM: C[m][n] = epi( sum_k A[m][k] * Wt[n][k] + bias[n] ), A = [A1 | A2] (two row-major segments: the skip concatenation
// cat[input_pts, h] RH:105-106 and cat[feature, input_views] RH:113 without materialising them), both K extents multiples of 32.
// ONE kernel body for the three arithmetics (nsr_wide_b3.inc: gemm_split_body -> kw_gemm_h2 / kw_gemm_b3 / kw_gemm_f32): weights as an
// LDS image moved by LDS-DMA, activations staged through registers, 256 x 256 tiles in 512-thread workgroups, persistent tiles with
// the next tile's first stages in flight, XCD-aware tile order (the N-tiles of one M-block go to ONE XCD back to back: an A tile
// is read from HBM once and then from that XCD's L2).  MFMA operand roles: A-operand = activations, B-operand = weights, so a
// lane's 16 accumulators share ONE output column: bias and relu mask cost one load per 16 values and every store instruction
// writes two full 128-byte rows.  (r05's separate fp32 kernel kw_gemm -- 128 x 128 tiles, XOR-swizzled register-staged LDS tiles --
// was retired in r06: the shared body on fp32 MFMAs is 9-12 % faster, profiles/r06/extra/layered_fp32_ab.txt.)
// ------------------------------------------------------------------------------------------------------------------------
enum { kRelu = 1, kAccum = 2, kMaskEpi = 4 };

#include "nsr_wide_b3.inc"
#undef NSRW_FILE_TAG
#define NSRW_FILE_TAG 0

// ------------------------------------------------------------------------------------------------------------------------
// per-ray and per-point stages
// ------------------------------------------------------------------------------------------------------------------------
struct RayArgs {                  // one chunk of R rays, first ray = row0 of the call's arrays
  const float* rays_o; const float* rays_d;      // [N,3] (already offset to the chunk)
  const float* viewdirs_in;                      // nullable [N,3] (offset)
  const float* near_in; const float* far_in;     // nullable [N] (offset)
  float near_, far_;
  int R, S0, NI, flags;
  const float* t_tab; const float* u_tab;        // linspace tables
  const float* t_rand; const float* u_rays; const float* noise0; const float* noise1;   // nullable, offset to the chunk
  float* vd; float* nrm; float* z0;              // chunk scratch: [R,3], [R], [R,S0]
};

__device__ __forceinline__ float coarse_z(float near_, float far_, float t, int lindisp) {      // RN:441-443
  if (lindisp) return 1.0f / (((1.0f / near_) * (1.0f - t)) + ((1.0f / far_) * t));
  return (near_ * (1.0f - t)) + (far_ * t);
}

// viewdirs = d / |d| (RN:97-98; torch.norm = sqrt of the sequential fp32 sum of squares), |d| for RN:361, the coarse depths
// (RN:439-443) and their stratified perturbation (RN:447-459).  One thread per ray.
__global__ void __launch_bounds__(256) kw_ray_setup(const RayArgs a) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= a.R) return;
  const float dx = a.rays_d[r * 3 + 0], dy = a.rays_d[r * 3 + 1], dz = a.rays_d[r * 3 + 2];
  const float nrm = sqrtf(((dx * dx) + (dy * dy)) + (dz * dz));
  a.nrm[r] = nrm;
  if (a.viewdirs_in) {
    a.vd[r * 3 + 0] = a.viewdirs_in[r * 3 + 0]; a.vd[r * 3 + 1] = a.viewdirs_in[r * 3 + 1]; a.vd[r * 3 + 2] = a.viewdirs_in[r * 3 + 2];
  } else {
    a.vd[r * 3 + 0] = dx / nrm; a.vd[r * 3 + 1] = dy / nrm; a.vd[r * 3 + 2] = dz / nrm;
  }
  const float nr = a.near_in ? a.near_in[r] : a.near_, fr = a.far_in ? a.far_in[r] : a.far_;
  const int lindisp = a.flags & NSRW_FLAG_LINDISP;
  float* z = a.z0 + (long long)r * a.S0;
  if (!a.t_rand) {
    for (int i = 0; i < a.S0; ++i) z[i] = coarse_z(nr, fr, a.t_tab[i], lindisp);
    return;
  }
  const float* tr = a.t_rand + (long long)r * a.S0;
  float prev = coarse_z(nr, fr, a.t_tab[0], lindisp), cur = prev;     // z[i-1], z[i]
  for (int i = 0; i < a.S0; ++i) {
    const float next = i + 1 < a.S0 ? coarse_z(nr, fr, a.t_tab[i + 1], lindisp) : cur;
    const float lower = i > 0 ? 0.5f * (cur + prev) : cur;            // RN:449-450: mids = .5 (z[1:] + z[:-1])
    const float upper = i + 1 < a.S0 ? 0.5f * (next + cur) : cur;
    z[i] = lower + (upper - lower) * tr[i];                           // RN:459
    prev = cur; cur = next;
  }
}

// gamma(x) (RH:18-48): sin / cos of the fp32 argument 2^l x (the scaling is exact) by sincos_enc: <= 1 ulp from the true value,
// i.e. inside the reference's own libm error.  sin / cos of inf / NaN are NaN, as in torch.
struct EmbedArgs {
  const float* rays_o; const float* rays_d;   // rays form: point (r, s) = o + d z[r][s]  (RN:463, RN:478)
  const float* z; const float* vd;            // [R,S], [R,3]
  const float* pts; const float* dirs;        // points form (run_network): [P,3] each; used when rays_o == nullptr
  long long P; int S;
  int L, Lv;                                  // frequencies; Lv < 0: no direction encoding
  float* E; int ldE;                          // [P, Ci]
  float* ED; int ldED;                        // [P, Cv]
};

// sin and cos of the fp32 argument a for the encodings.  |a| < 2^24 (every scene: 2^14 |x| < 2^24 means |x| < 1024): ONE fp64
// range reduction -- t = a * 2 / pi has >= 24 spare bits, the split into quadrant and fraction is exact -- and the Cephes
// single-precision minimax polynomials on [-pi / 4, pi / 4] (1 ulp), as the fused kernels' enc_trig (nsr_kernels.hip); ~30 VALU
// for the pair where the fp64 library sincos costs several hundred (r06: kw_embed was 2-9 % of a layered render).  Beyond that,
// and for inf / NaN: the fp64 library routine, rounded once, as before.
__device__ __forceinline__ void sincos_enc(float a, float& s, float& c) {
  if (!(fabsf(a) < 16777216.0f)) {
    double sd, cd;
    sincos((double)a, &sd, &cd);
    s = (float)sd; c = (float)cd;
    return;
  }
  const double kMagic = 6755399441055744.0;                 // 1.5 * 2^52: t + kMagic holds rint(t) in its low word
  const double t = (double)a * 0.63661977236758134308;      // a * 2 / pi
  const double tm = t + kMagic;
  const int n = __double2loint(tm);
  const double f = t - (tm - kMagic);                       // [-0.5, 0.5], exact
  const float r = (float)(f * 1.57079632679489661923);
  const float z = r * r;
  float ps = __builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
  ps = __builtin_fmaf(ps, z, -1.6666654611e-1f);
  ps = __builtin_fmaf(ps * z, r, r);
  float pc = __builtin_fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
  pc = __builtin_fmaf(pc, z, 4.166664568298827e-2f);
  pc = __builtin_fmaf(pc * z, z, __builtin_fmaf(-0.5f, z, 1.0f));
  const float vs = (n & 1) ? pc : ps, vc = (n & 1) ? ps : pc;              // sin(a) = sin(r + n pi/2), cos(a) = sin(r + (n + 1) pi/2)
  s = __uint_as_float(__float_as_uint(vs) ^ ((unsigned)(n & 2) << 30));
  c = __uint_as_float(__float_as_uint(vc) ^ ((unsigned)((n + 1) & 2) << 30));
}

// 16 threads per point, 16 points per workgroup: thread q computes x (q = 0) or band q - 1 (1 <= q <= L) of the position row and the
// same of the direction row into LDS; the workgroup then writes its 16 consecutive rows of E (and of ED) -- one contiguous piece of
// memory -- 16 bytes per thread, padding columns included (r06: the rows used to be written 24 strided bytes per thread, twelve
// dword stores each; the kernel was 4-15 % of a layered render).
constexpr int kEmbedLdMax = 96;          // pad32(3 + 6 * 15)
__global__ void __launch_bounds__(256) kw_embed(const EmbedArgs a) {
  __shared__ __attribute__((aligned(16))) float sE[16 * kEmbedLdMax];
  __shared__ __attribute__((aligned(16))) float sD[16 * kEmbedLdMax];
  const int tid = threadIdx.x;
  const long long p0 = (long long)blockIdx.x * 16;
  const int lp = tid >> 4, q = tid & 15;
  const long long p = p0 + lp;
  const int nE = 16 * a.ldE / 4, nD = a.Lv >= 0 ? 16 * a.ldED / 4 : 0;          // float4s of the workgroup's rows
  for (int t = tid; t < nE; t += 256) reinterpret_cast<f32x4*>(sE)[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  for (int t = tid; t < nD; t += 256) reinterpret_cast<f32x4*>(sD)[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  __syncthreads();
  if (p < a.P) {
    float x[3], v[3] = {0.0f, 0.0f, 0.0f};
    if (a.rays_o) {
      const long long r = p / a.S;
      const float zz = a.z[p];
#pragma unroll
      for (int c = 0; c < 3; ++c) x[c] = a.rays_o[r * 3 + c] + a.rays_d[r * 3 + c] * zz;
      if (a.Lv >= 0)
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = a.vd[r * 3 + c];
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c) x[c] = a.pts[p * 3 + c];
      if (a.Lv >= 0)
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = a.dirs[p * 3 + c];
    }
    float* e = sE + lp * a.ldE;
    NSRW_CHECK(a.ldE <= kEmbedLdMax && a.ldED <= kEmbedLdMax && 3 + 6 * a.L <= a.ldE);
    if (q == 0) {
      e[0] = x[0]; e[1] = x[1]; e[2] = x[2];
    } else if (q <= a.L) {
      const float f = ldexpf(1.0f, q - 1);
#pragma unroll
      for (int c = 0; c < 3; ++c) sincos_enc(x[c] * f, e[3 + 6 * (q - 1) + c], e[6 + 6 * (q - 1) + c]);
    }
    if (a.Lv >= 0) {
      float* ed = sD + lp * a.ldED;
      if (q == 0) {
        ed[0] = v[0]; ed[1] = v[1]; ed[2] = v[2];
      } else if (q <= a.Lv) {
        const float f = ldexpf(1.0f, q - 1);
#pragma unroll
        for (int c = 0; c < 3; ++c) sincos_enc(v[c] * f, ed[3 + 6 * (q - 1) + c], ed[6 + 6 * (q - 1) + c]);
      }
    }
  }
  __syncthreads();
  const long long rows = a.P - p0 < 16 ? a.P - p0 : 16;
  const int mE = (int)(rows * a.ldE / 4), mD = a.Lv >= 0 ? (int)(rows * a.ldED / 4) : 0;
  f32x4* gE = reinterpret_cast<f32x4*>(a.E + p0 * a.ldE);
  for (int t = tid; t < mE; t += 256) gE[t] = reinterpret_cast<const f32x4*>(sE)[t];
  if (mD) {
    f32x4* gD = reinterpret_cast<f32x4*>(a.ED + p0 * a.ldED);
    for (int t = tid; t < mD; t += 256) gD[t] = reinterpret_cast<const f32x4*>(sD)[t];
  }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float relu_nan(float x) { return (x < 0.0f) ? 0.0f : x; }      // F.relu: a NaN density stays NaN (RN:356)

// raw2outputs (RN:343-387), one thread per ray, the reference's operations in its order: dists (last = 1e10) * |d|, sigmoid,
// alpha = 1 - exp(-relu(sigma + noise) dists), the exclusive transmittance product as torch-CPU's cumprod computes it -- a
// sequential fp64 product, every prefix rounded to fp32 -- weights, the five weighted sums (sequential fp32), disparity with
// its NaN case, white background.
struct CompositeArgs {
  int R, S, flags;
  const float* rgb; int ld_rgb;        // raw rgb logits of point p at rgb[p * ld_rgb + 0..2]
  const float* sigma; int ld_sigma;    // raw density at sigma[p * ld_sigma]
  const float* z; const float* nrm;    // [R,S], [R]
  const float* noise;                  // nullable [R,S]
  float* weights;                      // [R,S]
  float* rgb_map; float* disp; float* acc;       // nullable outputs [R,3], [R], [R]
  float* raw_out; int raw_ch;          // nullable [R,S,raw_ch]: the raw the reference returns with retraw (RN:490-491)
  const float* raw_src; int ld_raw;    // raw_ch channels per point (raw_out given, raw_ch != 4: the output_linear rows)
};

__global__ void __launch_bounds__(256) kw_composite(const CompositeArgs a) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= a.R) return;
  const float* z = a.z + (long long)r * a.S;
  const float nrm = a.nrm[r];
  double T = 1.0;
  float cr = 0.f, cg = 0.f, cb = 0.f, depth = 0.f, acc = 0.f;
  for (int i = 0; i < a.S; ++i) {
    const long long p = (long long)r * a.S + i;
    float dist = (i < a.S - 1) ? (z[i + 1] - z[i]) : 1e10f;               // RN:358-359
    dist = dist * nrm;                                                     // RN:361
    float sg = a.sigma[p * a.ld_sigma];
    if (a.noise) sg = sg + a.noise[p];                                     // RN:374
    const float al = 1.0f - expf(-relu_nan(sg) * dist);                    // RN:356
    const float w = al * (float)T;                                         // RN:376
    T = T * (double)((1.0f - al) + 1e-10f);
    a.weights[p] = w;
    const float* q = a.rgb + p * a.ld_rgb;
    cr = cr + w * sigmoidf_(q[0]);                                         // RN:363, RN:378
    cg = cg + w * sigmoidf_(q[1]);
    cb = cb + w * sigmoidf_(q[2]);
    depth = depth + w * z[i];                                              // RN:380
    acc = acc + w;                                                         // RN:382
    if (a.raw_out) {
      float* o = a.raw_out + p * a.raw_ch;
      if (a.raw_src) { for (int c = 0; c < a.raw_ch; ++c) o[c] = a.raw_src[p * a.ld_raw + c]; }
      else { o[0] = q[0]; o[1] = q[1]; o[2] = q[2]; o[3] = a.sigma[p * a.ld_sigma]; }
    }
  }
  if (a.rgb_map) {
    const float qd = depth / acc;
    const float disp = (qd != qd) ? qd : 1.0f / fmaxf(1e-10f, qd);         // RN:381: 0/0 -> NaN propagates through torch.max
    if (a.flags & NSRW_FLAG_WHITE_BKGD) {                                   // RN:384-385
      const float bg = 1.0f - acc;
      cr = cr + bg; cg = cg + bg; cb = cb + bg;
    }
    a.rgb_map[r * 3 + 0] = cr; a.rgb_map[r * 3 + 1] = cg; a.rgb_map[r * 3 + 2] = cb;
    a.disp[r] = disp; a.acc[r] = acc;
  }
}

// sample_pdf (RH:199-243) for bins = mid-points of the coarse depths (RN:473) and the coarse weights[1:-1], one thread per ray:
// weights + 1e-5, torch.sum in ATen's association order for a contiguous fp32 row (oracle/nerf_oracle.py:_torch_sum_lastdim
// -- 8 vector lanes x 4 partial vectors, the left-over whole vectors into the first, partials folded 0 += 1, 2, 3, then the
// scalar tail and the 8 lanes in sequence), pdf, the cdf as a sequential fp64 sum rounded per prefix (torch.cumsum on the CPU),
// searchsorted(right=True), the gathers, denom < 1e-5 -> 1, the interpolation.  Also z_std = std(z_samples, unbiased=False)
// (RN:495), fp64 two-pass.
struct PdfArgs {
  int R, S0, NI;
  const float* z0; const float* w0;      // [R,S0] coarse depths and weights
  const float* u_tab; const float* u_rays;      // [NI] table, or nullable [R,NI] draws
  float* pdf; float* cdf;                // scratch [R,S0] each
  float* zs;                             // [R,NI]
  int64_t* inds;                         // nullable [R,NI]
  float* z_std;                          // nullable [R]
  const float* bins;                     // nullable: the stage entry nsrw_sample_pdf -- bins [R,S0-1] GIVEN (z0 unused), w0 = weights [R,S0-2]
};

__global__ void __launch_bounds__(256) kw_sample_pdf(const PdfArgs a) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= a.R) return;
  const int NW = a.S0 - 2, NC = a.S0 - 1;
  const float* z = a.z0 + (long long)r * a.S0;
  const float* w = a.bins ? a.w0 + (long long)r * NW : a.w0 + (long long)r * a.S0 + 1;
  const float* bn = a.bins ? a.bins + (long long)r * NC : nullptr;
  float* x = a.pdf + (long long)r * a.S0;
  float* cdf = a.cdf + (long long)r * a.S0;
  for (int i = 0; i < NW; ++i) x[i] = w[i] + 1e-5f;                        // RH:201
  const int nvec = NW / 8, silp = nvec / 4;
  float ps[4][8];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) ps[k][jj] = 0.0f;
  for (int i = 0; i < silp; ++i)
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) ps[k][jj] = ps[k][jj] + x[(i * 4 + k) * 8 + jj];
  for (int v = silp * 4; v < nvec; ++v)
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) ps[0][jj] = ps[0][jj] + x[v * 8 + jj];
  float total = 0.0f;
  for (int i = nvec * 8; i < NW; ++i) total = total + x[i];
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) total = total + (((ps[0][jj] + ps[1][jj]) + ps[2][jj]) + ps[3][jj]);
  double run = 0.0;
  cdf[0] = 0.0f;
  for (int i = 0; i < NW; ++i) {
    const float pdf = x[i] / total;                                        // RH:202
    run = run + (double)pdf;                                               // RH:203
    cdf[i + 1] = (float)run;
  }
  float* zs = a.zs + (long long)r * a.NI;
  double sum = 0.0;
  for (int k = 0; k < a.NI; ++k) {
    const float u = a.u_rays ? a.u_rays[(long long)r * a.NI + k] : a.u_tab[k];
    int lo = 0, hi = NC;                                                   // RH:227: first index with cdf > u
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
    }
    const int below = max(lo - 1, 0), above = min(lo, NC - 1);             // RH:228-229
    NSRW_CHECK(lo >= 0 && lo <= NC && below >= 0 && above + 1 < a.S0 && below + 1 < a.S0 && NC < a.S0);
    const float c0 = cdf[below], c1 = cdf[above];
    const float b0 = bn ? bn[below] : 0.5f * (z[below + 1] + z[below]);     // RN:473
    const float b1 = bn ? bn[above] : 0.5f * (z[above + 1] + z[above]);
    float denom = c1 - c0;
    if (denom < 1e-5f) denom = 1.0f;                                       // RH:238-239
    const float t = (u - c0) / denom;
    const float s = b0 + t * (b1 - b0);                                    // RH:241
    zs[k] = s;
    sum += (double)s;
    if (a.inds) a.inds[(long long)r * a.NI + k] = (int64_t)lo;
  }
  if (a.z_std) {
    const double mean = sum / a.NI;
    double var = 0.0;
    for (int k = 0; k < a.NI; ++k) { const double d = (double)zs[k] - mean; var += d * d; }
    a.z_std[r] = (float)sqrt(var / a.NI);
  }
}

// z_vals, _ = torch.sort(torch.cat([z_vals, z_samples], -1), -1) (RN:477): one wave per ray, exact and stable for ANY input --
// the rank of an element is the number of elements that sort before it (NaN after every number, ties by position).
__global__ void __launch_bounds__(64) kw_sort(const float* __restrict__ z0, const float* __restrict__ zs, int S0, int NI,
                                              float* __restrict__ zf) {
  extern __shared__ float sz[];
  const long long r = blockIdx.x;
  const int S = S0 + NI, lane = threadIdx.x;
  for (int i = lane; i < S; i += 64) sz[i] = i < S0 ? z0[r * S0 + i] : zs[r * NI + (i - S0)];
  __syncthreads();
  for (int i = lane; i < S; i += 64) {
    const float x = sz[i];
    const bool xn = x != x;
    int rank = 0;
    for (int k = 0; k < S; ++k) {
      const float y = sz[k];
      const bool yn = y != y;
      const bool before = xn ? (!yn || k < i) : (!yn && (y < x || (y == x && k < i)));
      rank += before ? 1 : 0;
    }
    NSRW_CHECK(rank >= 0 && rank < S);
    zf[r * S + rank] = x;
  }
}

// Backward of the compositing (the last pass's raw2outputs) for the cotangent of rgb_map, one thread per ray, fp32 forward
// quantities recomputed exactly as kw_composite computed them, derivative arithmetic in fp64:
//   w_i = alpha_i T_i, T_k (k > i) carries the factor om_i = 1 - alpha_i + 1e-10 -> dL/dalpha_i = A_i T_i - (sum_{k>i} A_k w_k) / om_i
//   with A_i = g . sigmoid(rgb_i) [- sum g with a white background]; alpha = 1 - exp(-relu(sigma) dz |d|).
// Writes dL/draw [P,32] (rgb logits, sigma, zero padding: the K extent of the first backward GEMM) and dL/d|d| per ray.
struct CompositeBwdArgs {
  int R, S, flags;
  const float* rgb; int ld_rgb; const float* sigma; int ld_sigma;
  const float* z; const float* nrm; const float* noise;
  const float* grad_rgb;              // [R,3]
  float* draw;                        // [P,32]
  float* gnorm;                       // [R]
  float* scr_a; float* scr_t;         // scratch [R,S] each: alpha_i, T_i
  float* gscale;                      // nullable [P]: f16x2 handles -- the row leaves NORMALISED, gscale[p] undoes it (kw_embed_bwd)
};

// f16x2 gradient GEMMs (csrc/nsr_h2_bwd.inc has the reasoning): the backward chain is LINEAR in a point's dL/d raw, whose size is
// whatever the cotangent and the compositing weights make it (1e-12 for an occluded sample, 1e+3 for a summed loss) -- no
// upload-time scale can fit it into fp16.  So every point's four inputs are multiplied by s = 2^(7 - floor(log2 max|.|)) (exact),
// the chain runs on gradients of magnitude ~2^7 (2^9 of head-room; a point that outgrows it raises the range flag and the chain is
// re-run on bf16x3), and kw_embed_bwd multiplies what comes out by 1 / s.  0 / denormal-small / inf / NaN rows stay as they are.
__device__ __forceinline__ float grad_norm_scale(float g0, float g1, float g2, float g3, float& inv) {
  const float m = fmaxf(fmaxf(fabsf(g0), fabsf(g1)), fmaxf(fabsf(g2), fabsf(g3)));
  const unsigned ef = (__float_as_uint(m) >> 23) & 0xffu;
  const bool ok = ef >= 8u && ef <= 253u;
  inv = ok ? __uint_as_float((ef - 7u) << 23) : 1.0f;                  // 2^(ef - 127 - 7)
  return ok ? __uint_as_float((254u - ef + 7u) << 23) : 1.0f;          // 2^(127 - ef + 7)
}

__global__ void __launch_bounds__(256) kw_composite_bwd(const CompositeBwdArgs a) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= a.R) return;
  const float* z = a.z + (long long)r * a.S;
  const float nrm = a.nrm[r];
  float* al_s = a.scr_a + (long long)r * a.S;
  float* t_s = a.scr_t + (long long)r * a.S;
  double T = 1.0;
  for (int i = 0; i < a.S; ++i) {
    const long long p = (long long)r * a.S + i;
    float dist = (i < a.S - 1) ? (z[i + 1] - z[i]) : 1e10f;
    dist = dist * nrm;
    float sg = a.sigma[p * a.ld_sigma];
    if (a.noise) sg = sg + a.noise[p];
    const float al = 1.0f - expf(-relu_nan(sg) * dist);
    al_s[i] = al;
    t_s[i] = (float)T;
    T = T * (double)((1.0f - al) + 1e-10f);
  }
  const double g0 = a.grad_rgb[r * 3 + 0], g1 = a.grad_rgb[r * 3 + 1], g2 = a.grad_rgb[r * 3 + 2];
  const double gsum = (a.flags & NSRW_FLAG_WHITE_BKGD) ? (g0 + g1 + g2) : 0.0;
  double suffix = 0.0, dn = 0.0;
  for (int i = a.S - 1; i >= 0; --i) {
    const long long p = (long long)r * a.S + i;
    const float* q = a.rgb + p * a.ld_rgb;
    const double c0 = sigmoidf_(q[0]), c1 = sigmoidf_(q[1]), c2 = sigmoidf_(q[2]);
    const double al = al_s[i], Ti = t_s[i];
    const double w = (double)(al_s[i] * t_s[i]);
    const double A = g0 * c0 + g1 * c1 + g2 * c2 - gsum;
    const double om = (double)((1.0f - al_s[i]) + 1e-10f);
    const double d_alpha = A * Ti - suffix / om;
    suffix += A * w;
    float sg = a.sigma[p * a.ld_sigma];
    if (a.noise) sg = sg + a.noise[p];
    const double dz = (i < a.S - 1) ? (double)(z[i + 1] - z[i]) : 1e10;
    const double e = 1.0 - al;
    float* o = a.draw + p * 32;
    float o0 = (float)(w * g0 * c0 * (1.0 - c0));
    float o1 = (float)(w * g1 * c1 * (1.0 - c1));
    float o2 = (float)(w * g2 * c2 * (1.0 - c2));
    float o3 = sg > 0.0f ? (float)(d_alpha * (dz * (double)nrm) * e) : 0.0f;
    if (a.gscale) {
      float inv;
      const float sc = grad_norm_scale(o0, o1, o2, o3, inv);
      o0 *= sc; o1 *= sc; o2 *= sc; o3 *= sc;
      a.gscale[p] = inv;
    }
    o[0] = o0; o[1] = o1; o[2] = o2; o[3] = o3;
    dn += d_alpha * (double)relu_nan(sg) * e * dz;                       // d(dz |d|) / d|d| = dz
  }
  a.gnorm[r] = (float)dn;
}

__device__ __forceinline__ double shfl_xor_f64(double v, int mask) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl_xor(lo, mask);
  hi = __shfl_xor(hi, mask);
  return __hiloint2double(hi, lo);
}

// Backward of the encodings and of pts = o + d z (RN:478), viewdirs = d / |d| (RN:97), dists |d| (RN:361), one wave per ray:
// lane l owns samples l, l + 64, ...; fp64 throughout (the per-ray sums run over hundreds of samples of either sign).
struct EmbedBwdArgs {
  int R, S, L, Lv;
  const float* rays_o; const float* rays_d; const float* z; const float* vd; const float* nrm; const float* gnorm;
  const float* GE; int ldE; const float* GED; int ldED;       // dL/d encodings [P, Ci], [P, Cv] (GED nullable)
  int given_viewdirs;
  float* grad_o; float* grad_d; float* grad_v;                // [R,3]; grad_v only with given view directions
  const float* gscale;                                        // nullable [P]: 1 / s of the point's normalised chain (kw_composite_bwd)
};

__device__ __forceinline__ void embed_bwd(const float* G, const double (&x)[3], int L, double (&out)[3]) {
#pragma unroll
  for (int c = 0; c < 3; ++c) out[c] = (double)G[c];
  for (int l = 0; l < L; ++l) {
    const double f = (double)ldexpf(1.0f, l);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      double s, co;
      sincos((double)((float)x[c] * (float)f), &s, &co);     // the forward's fp32 argument
      out[c] += f * ((double)G[3 + 6 * l + c] * co - (double)G[6 + 6 * l + c] * s);
    }
  }
}

__global__ void __launch_bounds__(64) kw_embed_bwd(const EmbedBwdArgs a) {
  const long long r = blockIdx.x;
  const int lane = threadIdx.x;
  double o[3], d[3], v[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) { o[c] = a.rays_o[r * 3 + c]; d[c] = a.rays_d[r * 3 + c]; v[c] = a.vd[r * 3 + c]; }
  double go[3] = {0, 0, 0}, gd[3] = {0, 0, 0}, gv[3] = {0, 0, 0};
  for (int i = lane; i < a.S; i += 64) {
    const long long p = r * a.S + i;
    const float zz = a.z[p];
    double x[3], gp[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) x[c] = (double)(a.rays_o[r * 3 + c] + a.rays_d[r * 3 + c] * zz);
    embed_bwd(a.GE + p * a.ldE, x, a.L, gp);
    const double un = a.gscale ? (double)a.gscale[p] : 1.0;             // (a power of two: exact)
#pragma unroll
    for (int c = 0; c < 3; ++c) { go[c] += gp[c] * un; gd[c] += gp[c] * un * (double)zz; }
    if (a.GED) {
      double g2[3];
      embed_bwd(a.GED + p * a.ldED, v, a.Lv, g2);
#pragma unroll
      for (int c = 0; c < 3; ++c) gv[c] += g2[c] * un;
    }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      go[c] += shfl_xor_f64(go[c], m); gd[c] += shfl_xor_f64(gd[c], m); gv[c] += shfl_xor_f64(gv[c], m);
    }
  if (lane != 0) return;
  const double nrm = a.nrm[r], gn = a.gnorm[r];
  double u[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) u[c] = d[c] / nrm;                       // d|d| / dd = d / |d|
  if (a.given_viewdirs) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      a.grad_o[r * 3 + c] = (float)go[c];
      a.grad_d[r * 3 + c] = (float)(gd[c] + gn * u[c]);
      if (a.grad_v) a.grad_v[r * 3 + c] = (float)gv[c];
    }
  } else {
    const double dot = gv[0] * u[0] + gv[1] * u[1] + gv[2] * u[2];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      a.grad_o[r * 3 + c] = (float)go[c];
      a.grad_d[r * 3 + c] = (float)(gd[c] + gn * u[c] + (gv[c] - u[c] * dot) / nrm);      // d(d / |d|)
    }
  }
}

__global__ void __launch_bounds__(256) kw_copy_rows(const float* __restrict__ src, int ld_src, float* __restrict__ dst, int ld_dst,
                                                    long long rows, int cols) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long p = gid / cols;
  const int c = (int)(gid - p * cols);
  if (p < rows) dst[p * ld_dst + c] = src[p * ld_src + c];
}

// VJP of Embedder.embed (RH:39-48) alone: gx[c] = g[c] + sum_l 2^l (g_sin[l][c] cos(2^l x_c) - g_cos[l][c] sin(2^l x_c)), the
// trigonometric values at the forward's own fp32 arguments, the sum in fp64 rounded once.  One thread per point.
__global__ void __launch_bounds__(256) kw_embed_vjp(const float* __restrict__ x, const float* __restrict__ g, long long P, int L,
                                                    float* __restrict__ gx) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const int C = 3 + 6 * L;
  const float* gp = g + p * C;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float xc = x[p * 3 + c];
    double acc = (double)gp[c];
    for (int l = 0; l < L; ++l) {
      const float f = ldexpf(1.0f, l);
      float sn, cs;
      sincos_enc(xc * f, sn, cs);
      acc += (double)f * ((double)gp[3 + 6 * l + c] * (double)cs - (double)gp[6 + 6 * l + c] * (double)sn);
    }
    gx[p * 3 + c] = (float)acc;
  }
}

}  // namespace nsrw

// ------------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------------
namespace {

using namespace nsrw;

thread_local std::string g_err;

int fail(const std::string& m) {
  g_err = m;
  return 1;
}

#define NSRW_HIP(expr)                                                                      \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess) return fail(std::string(#expr) + ": " + hipGetErrorString(e_));   \
  } while (0)

struct DeviceGuard {
  int prev = -1, dev;
  hipError_t err = hipSuccess;
  explicit DeviceGuard(int d) : dev(d) {
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != dev) err = hipSetDevice(dev);
  }
  ~DeviceGuard() { if (prev >= 0 && prev != dev) (void)hipSetDevice(prev); }
};
#define NSRW_DEVICE(h) DeviceGuard guard_((h)->cfg.device); NSRW_HIP(guard_.err)

inline int pad32(int x) { return (x + 31) / 32 * 32; }

struct Mat {                 // one packed matrix [Np][K1p + K2p] and its (nullable) bias, as offsets into Net::dW
  size_t w = 0, b = (size_t)-1;
  int Np = 0, K1p = 0, K2p = 0;
  size_t wb = 0;             // bf16x3 handles: byte offset of the split image [col-block][k16 block][piece][64][8 bf16] in Net::dWb
  int ncb = 0;               // ... and its col-blocks (Np / 32 rounded up to an even count; the padding holds zero weights)
  size_t wh = 0;             // f16x2 handles: byte offset of the two-piece fp16 image (same layout, 2 KiB per block) in Net::dWh
  float cscale = 1.0f;       // ... and 2^-sw, which undoes the image's weight scale in the epilogue
  size_t wf = 0;             // fp32 handles: byte offset of the fp32 image [col-block][k16 block][2][64 lanes][4 floats] in Net::dWf
};

struct Net {
  NsrwNet d{};
  bool loaded = false;
  int in_ch = 0, in_v = 0, Ci = 0, Cv = 0, Wp = 0, W2 = 0, W2p = 0, ldfa = 0, raw_ch = 4;
  int ldraw = 32;                     // row length of the raw outputs: 4 = [r g b sigma] dense (view directions), else 32
  std::vector<char> skip_in;          // skip_in[i]: layer i takes cat[input_pts, h]  (i - 1 in skips)
  float* dW = nullptr;
  char* dWb = nullptr;                // bf16x3 handles: the split images of every matrix (nsr_wide_b3.inc)
  size_t wb_bytes = 0;
  char* dWh = nullptr;                // f16x2 handles: the scaled two-piece fp16 images of every matrix (kw_gemm_h2)
  size_t wh_bytes = 0;
  char* dWf = nullptr;                // fp32 handles: the fp32 images in the same LDS-image layout (kw_gemm_f32)
  size_t wf_bytes = 0;
  std::vector<Mat> fwd;               // pts_linears
  Mat fa, al, hv, rgb, out;           // feature_linear, alpha_linear, views_linears.0, rgb_linear / output_linear
  std::vector<Mat> bwd_h, bwd_e;      // per pts layer: G W_i[:, hidden part] (i >= 1), G W_i[:, encoding part] (layer 0, skip layers)
  Mat b_rgb, b_feat, b_ed, b_head;
};

std::string check_net(const NsrwNet& n) {
  char buf[160];
  if (n.D < 1 || n.D > NSRW_MAX_DEPTH) { snprintf(buf, sizeof buf, "netdepth %d (1..%d)", n.D, NSRW_MAX_DEPTH); return buf; }
  if (n.W < 2 || n.W > NSRW_MAX_WIDTH) { snprintf(buf, sizeof buf, "netwidth %d (2..%d)", n.W, NSRW_MAX_WIDTH); return buf; }
  if (n.multires < 0 || n.multires > 15 || n.multires_views < 0 || n.multires_views > 15) return "multires / multires_views (0..15)";
  if (n.n_skips < 0 || n.n_skips > NSRW_MAX_SKIPS) return "more skips than NSRW_MAX_SKIPS";
  for (int i = 0; i < n.n_skips; ++i)
    if (n.skips[i] < 0 || n.skips[i] >= n.D - 1) {
      snprintf(buf, sizeof buf, "skip after layer %d of %d (the reference itself fails on a skip behind the last layer, RH:109)", n.skips[i], n.D);
      return buf;
    }
  if (!n.use_viewdirs && (n.output_ch < 4 || n.output_ch > 32)) return "output_ch (4..32) of a network without view directions";
  return "";
}

bool is_skip(const NsrwNet& n, int i) {
  for (int k = 0; k < n.n_skips; ++k) if (n.skips[k] == i) return true;
  return false;
}

size_t net_floats(const NsrwNet& n) {
  const size_t in_ch = 3 + 6 * n.multires, in_v = 3 + 6 * n.multires_views, W = n.W;
  size_t t = W * in_ch + W;
  for (int i = 1; i < n.D; ++i) t += W * (W + (is_skip(n, i - 1) ? in_ch : 0)) + W;
  if (n.use_viewdirs) t += W * W + W + W + 1 + (W / 2) * (W + in_v) + W / 2 + 3 * (W / 2) + 3;
  else t += (size_t)n.output_ch * W + n.output_ch;
  return t;
}

// How the GEMM kernels are launched: fixed per handle by nsrw_create (a setup call; launch calls read no environment).
struct GemmCfg {
  int cus = 256;       // compute units of the handle's device
  bool b3 = false;     // NSRW_FLAG_MLP_BF16X3: every GEMM on bf16 MFMAs with three-way split operands (kw_gemm_b3)
  int b3_wm = 4;       // ... 256-column tiles with 256 rows / 512 threads (4) or 128 rows / 256 threads (2: NSRW_B3_WM = 2)
  // NSRW_FLAG_MLP_F16X2 (b3 is set as well: it is the re-run arithmetic and the backward's).  A forward network pass runs on
  // kw_gemm_h2 (two fp16 pieces, three products); an activation that leaves fp16's range sets d_range[0], and the same pass
  // follows on kw_gemm_b3 in launches that return at once unless the flag is set (net_forward).
  bool h2 = false;
  unsigned* d_range = nullptr;          // device: [0] flag of the pass in flight, [1] passes run, [2] passes re-run on bf16x3
  bool use_h2 = false;                  // (per call, net_forward) this chain runs on kw_gemm_h2
  const unsigned* run_if = nullptr;     // (per call, net_forward) this chain's launches are conditional on *run_if
};

struct Handle {
  NsrwConfig cfg{};
  GemmCfg gemm_cfg;
  Net net[2];
  float* d_tab = nullptr;           // [n_samples] t, [n_importance] u
  bool tables = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  unsigned* d_range = nullptr;      // GemmCfg::d_range
  bool timed = false;
  int chunks = 0;
};

// appends a packed [Np][Kp] matrix to `img`; get(n, k) is consulted for n < rows, k < cols only
template <typename F>
size_t pack(std::vector<float>& img, int Np, int Kp, F get) {
  const size_t off = img.size();
  img.resize(off + (size_t)Np * Kp, 0.0f);
  for (int n = 0; n < Np; ++n)
    for (int k = 0; k < Kp; ++k) img[off + (size_t)n * Kp + k] = get(n, k);
  return off;
}

size_t pack_bias(std::vector<float>& img, int Np, const float* b, int n_real, int at = 0) {
  const size_t off = img.size();
  img.resize(off + Np, 0.0f);
  for (int n = 0; n < n_real; ++n) img[off + at + n] = b[n];
  return off;
}

// fp32 -> nearest-even bf16 (bit pattern); the three round-to-nearest pieces of a weight sum to it exactly
inline uint16_t bf16_rn(float x) {
  uint32_t u;
  memcpy(&u, &x, 4);
  if ((u & 0x7f800000u) == 0x7f800000u) return (uint16_t)((u >> 16) | ((u & 0xffffu) ? 0x40u : 0u));      // inf / NaN
  return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
inline float bf16_f(uint16_t h) {
  const uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

// appends the bf16x3 image of the packed fp32 matrix src [Np][Kp] (Kp a multiple of 16): [col-block][k16 block][piece][lane][8],
// lane l <-> (column 32 cb + l % 32, k = 16 kb + 8 (l / 32) .. + 7): what one LDS-DMA instruction of kw_gemm_b3 moves is one
// B fragment of v_mfma_f32_32x32x16_bf16.  Returns the byte offset; *ncb = col-blocks, rounded up to an even count.
size_t pack_b3(std::vector<uint16_t>& imgb, const float* src, int Np, int Kp, int* ncb) {
  const int cbs = ((Np + 31) / 32 + 1) / 2 * 2, KB = Kp / 16;
  const size_t off = imgb.size();
  imgb.resize(off + (size_t)cbs * KB * 3 * 512, 0);
  for (int cb = 0; cb < cbs; ++cb)
    for (int kb = 0; kb < KB; ++kb) {
      uint16_t* blk = imgb.data() + off + ((size_t)cb * KB + kb) * 3 * 512;
      for (int l = 0; l < 64; ++l) {
        const int n = 32 * cb + (l & 31);
        if (n >= Np) continue;
        for (int e = 0; e < 8; ++e) {
          const float w = src[(size_t)n * Kp + 16 * kb + 8 * (l >> 5) + e];
          const uint16_t p0 = bf16_rn(w);
          const float r1 = w - bf16_f(p0);
          const uint16_t p1 = bf16_rn(r1);
          const uint16_t p2 = bf16_rn(r1 - bf16_f(p1));
          blk[0 * 512 + l * 8 + e] = p0; blk[1 * 512 + l * 8 + e] = p1; blk[2 * 512 + l * 8 + e] = p2;
        }
      }
    }
  *ncb = cbs;
  return off * sizeof(uint16_t);
}

// f16x2: the image of src scaled by 2^sw -- sw the power of two that puts the largest |entry| into [2^14, 2^15), 0 for a zero
// matrix -- as two round-to-nearest fp16 pieces in pack_b3's layout (2 KiB per (col-block, k16 block)).  *cscale = 2^-sw.
size_t pack_h2(std::vector<uint16_t>& imgh, const float* src, int Np, int Kp, float* cscale) {
  const int cbs = ((Np + 31) / 32 + 1) / 2 * 2, KB = Kp / 16;
  float amax = 0.0f;
  for (size_t i = 0; i < (size_t)Np * Kp; ++i) if (std::isfinite(src[i])) amax = std::max(amax, std::fabs(src[i]));
  int sw = 0;
  if (amax > 0.0f) { int e; (void)std::frexp(amax, &e); sw = 15 - e; }        // amax = f 2^e, f in [0.5, 1): amax 2^sw in [2^14, 2^15)
  sw = std::max(-100, std::min(100, sw));
  const float up = std::ldexp(1.0f, sw);
  *cscale = std::ldexp(1.0f, -sw);
  const size_t off = imgh.size();
  imgh.resize(off + (size_t)cbs * KB * 2 * 512, 0);
  auto bits = [](_Float16 v) { uint16_t u; memcpy(&u, &v, 2); return u; };
  for (int cb = 0; cb < cbs; ++cb)
    for (int kb = 0; kb < KB; ++kb) {
      uint16_t* blk = imgh.data() + off + ((size_t)cb * KB + kb) * 2 * 512;
      for (int l = 0; l < 64; ++l) {
        const int n = 32 * cb + (l & 31);
        if (n >= Np) continue;
        for (int e = 0; e < 8; ++e) {
          const float w = src[(size_t)n * Kp + 16 * kb + 8 * (l >> 5) + e] * up;       // exact (a non-finite weight stays non-finite)
          const _Float16 p0 = (_Float16)w;
          const _Float16 p1 = (_Float16)(w - (float)p0);
          blk[0 * 512 + l * 8 + e] = bits(p0); blk[1 * 512 + l * 8 + e] = bits(p1);
        }
      }
    }
  return off * sizeof(uint16_t);
}

// fp32 handles: the matrix as it is, in the LDS-image layout of kw_gemm_f32 -- [col-block][k16 block][chunk 2][64 lanes][4 floats],
// lane l <-> (column 32 cb + l % 32, k = 16 kb + 8 (l / 32) + 4 chunk .. + 3)
size_t pack_f32img(std::vector<float>& imgf, const float* src, int Np, int Kp) {
  const int cbs = ((Np + 31) / 32 + 1) / 2 * 2, KB = Kp / 16;
  const size_t off = imgf.size();
  imgf.resize(off + (size_t)cbs * KB * 512, 0.0f);
  for (int cb = 0; cb < cbs; ++cb)
    for (int kb = 0; kb < KB; ++kb) {
      float* blk = imgf.data() + off + ((size_t)cb * KB + kb) * 512;
      for (int l = 0; l < 64; ++l) {
        const int n = 32 * cb + (l & 31);
        if (n >= Np) continue;
        for (int e = 0; e < 8; ++e) blk[(e >> 2) * 256 + l * 4 + (e & 3)] = src[(size_t)n * Kp + 16 * kb + 8 * (l >> 5) + e];
      }
    }
  return off * sizeof(float);
}

// Chunk workspace: the same carve runs with base = nullptr to size it.
struct Carve {
  char* base; size_t off = 0;
  explicit Carve(void* b) : base(static_cast<char*>(b)) {}
  float* f(size_t n) {
    float* p = base ? reinterpret_cast<float*>(base + off) : nullptr;
    off += (n * sizeof(float) + 255) / 256 * 256;
    return p;
  }
};

struct Chunk {
  float *vd, *nrm, *z0, *w0, *zs, *zf, *wf, *pdf, *cdf, *gnorm, *scr_a, *scr_t;
  float *E, *ED, *FA, *HV, *RAW;
  std::vector<float*> H;
  float *DRAW, *GV, *G0, *G1, *GEP, *GED, *gscale;
};

// floats per point of the forward buffers of `n`
void carve_chunk(const Handle& h, long long R, bool grad, Carve& c, Chunk& k) {
  const int S0 = h.cfg.n_samples, NI = h.cfg.n_importance, S1 = NI > 0 ? S0 + NI : 0;
  const long long Smax = std::max(S0, S1), P = R * Smax;
  const Net& n0 = h.net[0];
  const Net& n1 = (NI > 0 && h.net[1].loaded) ? h.net[1] : h.net[0];
  auto mx = [&](int Net::*f) { return (size_t)std::max(n0.*f, n1.*f); };
  k.vd = c.f(R * 3); k.nrm = c.f(R); k.gnorm = c.f(R);
  k.z0 = c.f(R * S0); k.w0 = c.f(R * S0); k.pdf = c.f(R * S0); k.cdf = c.f(R * S0);
  k.zs = c.f(R * std::max(NI, 1)); k.zf = c.f(R * std::max(S1, 1)); k.wf = c.f(R * std::max(S1, 1));
  k.E = c.f(P * mx(&Net::Ci)); k.ED = c.f(P * mx(&Net::Cv));
  k.FA = c.f(P * mx(&Net::ldfa)); k.HV = c.f(P * mx(&Net::W2p)); k.RAW = c.f(P * 32);
  const Net& last = NI > 0 ? n1 : n0;
  const int nH = grad ? std::max(2, last.d.D) : 2;
  k.H.resize(nH);
  for (int i = 0; i < nH; ++i) k.H[i] = c.f(P * mx(&Net::Wp));
  if (grad) {
    const long long Pg = R * (S1 > 0 ? S1 : S0);
    k.scr_a = c.f(Pg); k.scr_t = c.f(Pg); k.gscale = c.f(Pg);
    k.DRAW = c.f(Pg * 32); k.GV = c.f(Pg * last.W2p); k.G0 = c.f(Pg * last.Wp); k.G1 = c.f(Pg * last.Wp);
    k.GEP = c.f(Pg * last.Ci); k.GED = c.f(Pg * last.Cv);
  } else {
    k.scr_a = k.scr_t = k.DRAW = k.GV = k.G0 = k.G1 = k.GEP = k.GED = k.gscale = nullptr;
  }
}

template <int NJ, int WM>
void launch_gemm_b3(hipStream_t st, unsigned grid, const GemmB3Args& g, int epi) {
  switch (epi) {
    case kRelu: hipLaunchKernelGGL((kw_gemm_b3<NJ, kRelu, WM>), dim3(grid), dim3(128 * WM), 0, st, g); break;
    case kMaskEpi: hipLaunchKernelGGL((kw_gemm_b3<NJ, kMaskEpi, WM>), dim3(grid), dim3(128 * WM), 0, st, g); break;
    case kAccum: hipLaunchKernelGGL((kw_gemm_b3<NJ, kAccum, WM>), dim3(grid), dim3(128 * WM), 0, st, g); break;
    default: hipLaunchKernelGGL((kw_gemm_b3<NJ, 0, WM>), dim3(grid), dim3(128 * WM), 0, st, g); break;
  }
}

template <int NJ, int WM>
void launch_gemm_h2(hipStream_t st, unsigned grid, const GemmB3Args& g, int epi) {
  switch (epi) {
    case kRelu: hipLaunchKernelGGL((kw_gemm_h2<NJ, kRelu, WM>), dim3(grid), dim3(128 * WM), 0, st, g); break;
    case kMaskEpi: hipLaunchKernelGGL((kw_gemm_h2<NJ, kMaskEpi, WM>), dim3(grid), dim3(128 * WM), 0, st, g); break;
    case kAccum: hipLaunchKernelGGL((kw_gemm_h2<NJ, kAccum, WM>), dim3(grid), dim3(128 * WM), 0, st, g); break;
    default: hipLaunchKernelGGL((kw_gemm_h2<NJ, 0, WM>), dim3(grid), dim3(128 * WM), 0, st, g); break;
  }
}

template <int NJ, int WM>
void launch_gemm_f32(hipStream_t st, unsigned grid, const GemmB3Args& g, int epi) {
  switch (epi) {
    case kRelu: hipLaunchKernelGGL((kw_gemm_f32<NJ, kRelu, WM>), dim3(grid), dim3(128 * WM), 0, st, g); break;
    case kMaskEpi: hipLaunchKernelGGL((kw_gemm_f32<NJ, kMaskEpi, WM>), dim3(grid), dim3(128 * WM), 0, st, g); break;
    case kAccum: hipLaunchKernelGGL((kw_gemm_f32<NJ, kAccum, WM>), dim3(grid), dim3(128 * WM), 0, st, g); break;
    default: hipLaunchKernelGGL((kw_gemm_f32<NJ, 0, WM>), dim3(grid), dim3(128 * WM), 0, st, g); break;
  }
}

// bf16x3 handles: the N extent is cut into tiles of 256 columns, then one of 128 and one of 64 for what is left (the image is
// padded to an even number of 32-column blocks with zero weights; columns >= N are never stored)
int gemm_b3(const GemmCfg& cfg, hipStream_t st, const Net& net, const Mat& m, const float* bias, const float* A1, int lda1,
            const float* A2, int lda2, float* C, int ldc, int N, long long M, int epi, const float* mask, int ldm) {
  GemmB3Args g{};
  g.A1 = A1; g.A2 = m.K2p ? A2 : nullptr; g.lda1 = lda1; g.lda2 = lda2; g.K1 = m.K1p; g.K2 = m.K2p;
  g.M = M; g.ldc = ldc; g.ldm = ldm;
  const bool h2 = cfg.use_h2, f32 = !cfg.b3;
  const int kfrag = (h2 || f32) ? 2048 : 3072;                      // bytes of one (col-block, k16 block) of the image
  g.cscale = h2 ? m.cscale : 1.0f; g.range_flag = cfg.d_range; g.run_if = cfg.run_if;
  const int KB = (m.K1p + m.K2p) / 16;
  int cb = 0;
  auto part = [&](int nj, int tiles) {
    // 256-row tiles (one 512-thread workgroup per CU) for the 256-column tiles; the narrow remainders keep 128 rows
    const bool tall = nj >= 2 && cfg.b3_wm == 4;
    const int tm = tall ? 256 : 128;
    const long long mblocks = (M + tm - 1) / tm, mgroups = (mblocks + 7) / 8;
    GemmB3Args t = g;
    const size_t boff = (size_t)cb * KB * kfrag;
    t.Wb = (h2 ? net.dWh + m.wh : f32 ? net.dWf + m.wf : net.dWb + m.wb) + boff;
    t.wb_bytes = (unsigned)((size_t)m.ncb * KB * kfrag - boff);
    t.bias = bias ? bias + cb * 32 : nullptr;
    t.C = C + cb * 32; t.N = N - cb * 32;
    t.mask = mask ? mask + cb * 32 : nullptr;
    t.n_tiles = tiles;
    if (t.N > 0) {
      const long long all = mgroups * 8 * tiles;
      const int per_cu = tall ? (nj >= 3 ? 1 : 2) : nj == 4 ? 2 : nj == 2 ? 3 : 4;
      const unsigned grid = (unsigned)std::max<long long>(8, std::min<long long>(all, (long long)cfg.cus * per_cu / 8 * 8));
      if (f32) {
        if (tall && nj == 4) launch_gemm_f32<4, 4>(st, grid, t, epi);
        else if (tall && nj == 3) launch_gemm_f32<3, 4>(st, grid, t, epi);
        else if (tall) launch_gemm_f32<2, 4>(st, grid, t, epi);
        else if (nj == 4) launch_gemm_f32<4, 2>(st, grid, t, epi);
        else if (nj == 2) launch_gemm_f32<2, 2>(st, grid, t, epi);
        else launch_gemm_f32<1, 2>(st, grid, t, epi);
      } else if (h2) {
        if (tall && nj == 4) launch_gemm_h2<4, 4>(st, grid, t, epi);
        else if (tall && nj == 3) launch_gemm_h2<3, 4>(st, grid, t, epi);
        else if (tall) launch_gemm_h2<2, 4>(st, grid, t, epi);
        else if (nj == 4) launch_gemm_h2<4, 2>(st, grid, t, epi);
        else if (nj == 2) launch_gemm_h2<2, 2>(st, grid, t, epi);
        else launch_gemm_h2<1, 2>(st, grid, t, epi);
      } else if (tall && nj == 4) launch_gemm_b3<4, 4>(st, grid, t, epi);
      else if (tall && nj == 3) launch_gemm_b3<3, 4>(st, grid, t, epi);
      else if (tall) launch_gemm_b3<2, 4>(st, grid, t, epi);
      else if (nj == 4) launch_gemm_b3<4, 2>(st, grid, t, epi);
      else if (nj == 2) launch_gemm_b3<2, 2>(st, grid, t, epi);
      else launch_gemm_b3<1, 2>(st, grid, t, epi);
    }
    cb += tiles * nj * 2;
  };
  // N in units of 64 columns: tiles of 256 where they divide it; else tiles of 192 where THEY do (r06: 384 = 2 x 192 and 192 itself are ONE
  // launch -- as 256 + 128 resp. 128 + 64 the second launch read the whole A matrix from HBM again); else 256s and one remainder tile
  const int u = m.ncb / 2, three = cfg.b3_wm == 4;
  if (u % 4 != 0 && u % 3 == 0 && three) {
    part(3, u / 3);
  } else {
    if (u / 4) part(4, u / 4);
    const int r = u % 4;
    if (r == 3 && three) part(3, 1);
    else {
      if (r >= 2) part(2, 1);
      if (r & 1) part(1, 1);
    }
  }
  return 0;
}

int gemm(const GemmCfg& cfg, hipStream_t st, const Net& net, const Mat& m, const float* A1, int lda1, const float* A2, int lda2,
         float* C, int ldc, int N, long long M, int flags, const float* mask = nullptr, int ldm = 0, bool with_bias = true) {
  if (M <= 0) return 0;
  const int epi = mask ? kMaskEpi : (flags & kAccum) ? kAccum : (flags & kRelu) ? kRelu : 0;     // (never combined: net_*)
  return gemm_b3(cfg, st, net, m, (with_bias && m.b != (size_t)-1) ? net.dW + m.b : nullptr, A1, lda1, A2, lda2, C, ldc, N, M, epi, mask, ldm);
}

// RH:99-122 over P points whose encodings are in k.E / k.ED: leaves the rgb logits in k.RAW (ld 32; all output_linear rows
// without view directions) and the density at sigma / ld_sigma.  keep: every pts layer's activation stays (k.H[i]).
int net_forward_chain(const GemmCfg& cfg, hipStream_t st, const Net& n, const Chunk& k, long long P, bool keep, const float** sigma,
                      int* ld_sigma) {
  const float* h = nullptr;
  for (int i = 0; i < n.d.D; ++i) {
    float* out = k.H[keep ? i : (i & 1)];
    int rc;
    if (i == 0) rc = gemm(cfg, st, n, n.fwd[0], k.E, n.Ci, nullptr, 0, out, n.Wp, n.Wp, P, kRelu);
    else if (n.skip_in[i]) rc = gemm(cfg, st, n, n.fwd[i], k.E, n.Ci, h, n.Wp, out, n.Wp, n.Wp, P, kRelu);
    else rc = gemm(cfg, st, n, n.fwd[i], h, n.Wp, nullptr, 0, out, n.Wp, n.Wp, P, kRelu);
    if (rc) return rc;
    h = out;
  }
  if (!n.d.use_viewdirs) {
    if (gemm(cfg, st, n, n.out, h, n.Wp, nullptr, 0, k.RAW, 32, n.d.output_ch, P, 0)) return 1;
    *sigma = k.RAW + 3; *ld_sigma = 32;
    return 0;
  }
  // the raw outputs of a point are FOUR dense floats [r g b sigma] (r06: rows of 32 with 3 + 1 useful values made every per-ray
  // kernel fetch two 128-byte lines per point): alpha_linear writes column 3, rgb_linear columns 0..2
  if (gemm(cfg, st, n, n.fa, h, n.Wp, nullptr, 0, k.FA, n.ldfa, n.ldfa, P, 0)) return 1;
  if (gemm(cfg, st, n, n.al, h, n.Wp, nullptr, 0, k.RAW + 3, n.ldraw, 1, P, 0)) return 1;
  if (gemm(cfg, st, n, n.hv, k.FA, n.ldfa, k.ED, n.Cv, k.HV, n.W2p, n.W2p, P, kRelu)) return 1;
  if (gemm(cfg, st, n, n.rgb, k.HV, n.W2p, nullptr, 0, k.RAW, n.ldraw, 3, P, 0)) return 1;
  *sigma = k.RAW + 3; *ld_sigma = n.ldraw;
  return 0;
}

// the bookkeeping of one f16x2 network pass: counts it, counts its re-run, clears the flag for the next pass
__global__ void kw_range_tally(unsigned* r) {
  if (threadIdx.x == 0) {
    r[1] += 1u;
    if (r[0]) { r[2] += 1u; r[0] = 0u; }
  }
}

int net_forward(const GemmCfg& cfg, hipStream_t st, const Net& n, const Chunk& k, long long P, bool keep, const float** sigma,
                int* ld_sigma) {
  if (!cfg.h2 || P <= 0) return net_forward_chain(cfg, st, n, k, P, keep, sigma, ld_sigma);
  // f16x2: the pass on fp16 MFMAs; then the same pass on bf16x3 in launches that are empty unless an activation left fp16's
  // range (its inputs -- the encodings -- are untouched, every buffer it writes is written again); then the tally
  GemmCfg c = cfg;
  c.use_h2 = true;
  if (net_forward_chain(c, st, n, k, P, keep, sigma, ld_sigma)) return 1;
  c.use_h2 = false; c.run_if = cfg.d_range;
  if (net_forward_chain(c, st, n, k, P, keep, sigma, ld_sigma)) return 1;
  hipLaunchKernelGGL(kw_range_tally, dim3(1), dim3(64), 0, st, cfg.d_range);
  return 0;
}

// Input-side backward of net_forward (activations kept): k.DRAW [P,32] = dL/d(rgb logits, sigma) -> k.GEP [P,Ci], k.GED [P,Cv].
int net_backward_chain(const GemmCfg& cfg, hipStream_t st, const Net& n, const Chunk& k, long long P) {
  float* g = k.G0;
  float* g2 = k.G1;
  const int D = n.d.D;
  if (n.d.use_viewdirs) {
    if (gemm(cfg, st, n, n.b_rgb, k.DRAW, 32, nullptr, 0, k.GV, n.W2p, n.W2p, P, 0, k.HV, n.W2p)) return 1;   // (g W_rgb) relu'(views)
    if (gemm(cfg, st, n, n.b_feat, k.GV, n.W2p, nullptr, 0, g2, n.Wp, n.Wp, P, 0)) return 1;                  // dL/d feature
    if (gemm(cfg, st, n, n.b_ed, k.GV, n.W2p, nullptr, 0, k.GED, n.Cv, n.Cv, P, 0)) return 1;                 // dL/d direction encoding
    if (gemm(cfg, st, n, n.b_head, g2, n.Wp, k.DRAW, 32, g, n.Wp, n.Wp, P, 0, k.H[D - 1], n.Wp)) return 1;    // feature, alpha -> h_{D-1}
  } else {
    if (gemm(cfg, st, n, n.b_head, k.DRAW, 32, nullptr, 0, g, n.Wp, n.Wp, P, 0, k.H[D - 1], n.Wp)) return 1;
  }
  bool first_e = true;
  for (int i = D - 1; i >= 0; --i) {          // g = dL/d pre-activation of layer i
    if (i == 0 || n.skip_in[i]) {
      if (gemm(cfg, st, n, n.bwd_e[i], g, n.Wp, nullptr, 0, k.GEP, n.Ci, n.Ci, P, first_e ? 0 : kAccum)) return 1;
      first_e = false;
    }
    if (i > 0) {
      if (gemm(cfg, st, n, n.bwd_h[i], g, n.Wp, nullptr, 0, g2, n.Wp, n.Wp, P, 0, k.H[i - 1], n.Wp)) return 1;
      std::swap(g, g2);
    }
  }
  return 0;
}

// f16x2: the chain on fp16 MFMAs over the per-point NORMALISED gradients (kw_composite_bwd), then -- in launches that are empty
// unless a gradient left fp16's range -- once more on bf16x3 from the same inputs (k.DRAW and the stored activations are read
// only; the first write of k.GEP does not accumulate), then the tally
int net_backward(const GemmCfg& cfg, hipStream_t st, const Net& n, const Chunk& k, long long P) {
  if (!cfg.h2 || P <= 0) return net_backward_chain(cfg, st, n, k, P);
  GemmCfg c = cfg;
  c.use_h2 = true;
  if (net_backward_chain(c, st, n, k, P)) return 1;
  c.use_h2 = false; c.run_if = cfg.d_range;
  if (net_backward_chain(c, st, n, k, P)) return 1;
  hipLaunchKernelGGL(kw_range_tally, dim3(1), dim3(64), 0, st, cfg.d_range);
  return 0;
}

// the timing events are skipped while the stream is being captured into a hipGraph (an event recorded under capture cannot be
// waited on by nsrw_last_ms); everything else a launch call does -- kernels, memset / memcpy nodes -- is capturable
bool capturing(hipStream_t st) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  return hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
}

int check_common(Handle* h, const float* ro, const float* rd, long long n, const NsrwExtras* ex) {
  if (!h) return fail("nsrw: null handle");
  if (n < 0) return fail("nsrw: n_rays < 0");
  if (n > 0 && (!ro || !rd)) return fail("nsrw: null rays");
  if (!h->net[0].loaded) return fail("nsrw: network 0 not uploaded");
  if (!h->tables) return fail("nsrw: nsrw_upload_tables not called");
  if (ex && (ex->d_u || ex->d_z_fine) && h->cfg.n_importance == 0) return fail("nsrw: u draws / fine depths without N_importance");
  return 0;
}

long long chunk_rays(Handle* h, size_t bytes, bool grad, long long n) {
  // largest R (multiple of 64, <= n rounded up) whose chunk fits `bytes`; 0 if not even 64 rays fit
  auto need = [&](long long R) { Carve c(nullptr); Chunk k; carve_chunk(*h, R, grad, c, k); return c.off; };
  if (need(64) > bytes) return 0;
  long long lo = 64, hi = std::max<long long>(64, (n + 63) / 64 * 64);
  if (need(hi) <= bytes) return hi;
  while (hi - lo > 64) {
    const long long mid = (lo + hi) / 2 / 64 * 64;
    if (need(mid) <= bytes) lo = mid; else hi = mid;
  }
  return lo;
}

int render_impl(Handle* h, const float* ro, const float* rd, long long n, float near_, float far_, const NsrwExtras* ex,
                const float* grad_rgb, const NsrwOut* out, float* grad_o, float* grad_d, float* grad_v, void* ws, size_t ws_bytes,
                hipStream_t st) {
  const bool grad = grad_rgb != nullptr;
  const int S0 = h->cfg.n_samples, NI = h->cfg.n_importance, S1 = NI > 0 ? S0 + NI : 0, SL = NI > 0 ? S1 : S0;
  const Net& n0 = h->net[0];
  const Net& n1 = (NI > 0 && h->net[1].loaded) ? h->net[1] : h->net[0];
  const Net& last = NI > 0 ? n1 : n0;
  NsrwExtras e{};
  if (ex) e = *ex;
  NsrwOut o{};
  if (out) o = *out;
  if (n == 0) return 0;
  if (!ws) return fail("nsrw: null workspace");
  if ((reinterpret_cast<uintptr_t>(ws) & 255) != 0) return fail("nsrw: workspace must be 256-byte aligned");
  const long long R = chunk_rays(h, ws_bytes, grad, n);
  if (R == 0) return fail("nsrw: workspace too small for a chunk of 64 rays (nsrw_workspace_bytes)");
  if ((size_t)S1 * sizeof(float) > 64 * 1024) return fail("nsrw: too many samples per ray for the sort kernel");
  h->chunks = 0;
  const bool cap = capturing(st);
  if (!cap) NSRW_HIP(hipEventRecord(h->ev0, st));
  for (long long r0 = 0; r0 < n; r0 += R) {
    const int Rc = (int)std::min<long long>(R, n - r0);
    Carve c(ws);
    Chunk k;
    carve_chunk(*h, R, grad, c, k);
    if (c.off > ws_bytes) return fail("nsrw: internal error: the chunk's carve exceeds the workspace");      // (chunk_rays sized it)
    const unsigned rb = (Rc + 255) / 256;
    RayArgs ra{};
    ra.rays_o = ro + r0 * 3; ra.rays_d = rd + r0 * 3;
    ra.viewdirs_in = e.d_viewdirs ? e.d_viewdirs + r0 * 3 : nullptr;
    ra.near_in = e.d_near ? e.d_near + r0 : nullptr; ra.far_in = e.d_far ? e.d_far + r0 : nullptr;
    ra.near_ = near_; ra.far_ = far_; ra.R = Rc; ra.S0 = S0; ra.NI = NI; ra.flags = h->cfg.flags;
    ra.t_tab = h->d_tab; ra.u_tab = h->d_tab + S0;
    ra.t_rand = e.d_t_rand ? e.d_t_rand + r0 * S0 : nullptr;
    ra.vd = k.vd; ra.nrm = k.nrm; ra.z0 = k.z0;
    hipLaunchKernelGGL(kw_ray_setup, dim3(rb), dim3(256), 0, st, ra);

    auto encode = [&](const Net& nn, const float* z, int S) {
      EmbedArgs ea{};
      ea.rays_o = ra.rays_o; ea.rays_d = ra.rays_d; ea.z = z; ea.vd = k.vd; ea.P = (long long)Rc * S; ea.S = S;
      ea.L = nn.d.multires; ea.Lv = nn.d.use_viewdirs ? nn.d.multires_views : -1;
      ea.E = k.E; ea.ldE = nn.Ci; ea.ED = k.ED; ea.ldED = nn.Cv;
      const long long thr = ea.P * 16;
      hipLaunchKernelGGL(kw_embed, dim3((unsigned)((thr + 255) / 256)), dim3(256), 0, st, ea);
    };
    // ---- coarse pass (RN:463-467) ----
    const float* sigma; int ld_sigma;
    encode(n0, k.z0, S0);
    const bool grad_coarse = grad && NI == 0;
    if (net_forward(h->gemm_cfg, st, n0, k, (long long)Rc * S0, grad_coarse, &sigma, &ld_sigma)) return 1;
    CompositeArgs ca{};
    ca.R = Rc; ca.S = S0; ca.flags = h->cfg.flags; ca.rgb = k.RAW; ca.ld_rgb = n0.ldraw; ca.sigma = sigma; ca.ld_sigma = ld_sigma;
    ca.z = k.z0; ca.nrm = k.nrm; ca.noise = e.d_noise0 ? e.d_noise0 + r0 * S0 : nullptr; ca.weights = k.w0;
    float* rgb0 = NI > 0 ? o.d_rgb0 : o.d_rgb;
    float* disp0 = NI > 0 ? o.d_disp0 : o.d_disp;
    float* acc0 = NI > 0 ? o.d_acc0 : o.d_acc;
    ca.rgb_map = rgb0 ? rgb0 + r0 * 3 : nullptr; ca.disp = disp0 ? disp0 + r0 : nullptr; ca.acc = acc0 ? acc0 + r0 : nullptr;
    float* raw_c = NI == 0 ? o.d_raw : o.d_raw0;
    if (raw_c) {
      ca.raw_out = raw_c + r0 * S0 * n0.raw_ch; ca.raw_ch = n0.raw_ch;
      if (!n0.d.use_viewdirs) { ca.raw_src = k.RAW; ca.ld_raw = 32; }
    }
    if (ca.rgb_map && !(ca.disp && ca.acc)) return fail("nsrw: rgb / disp / acc outputs of a pass must be given together");
    hipLaunchKernelGGL(kw_composite, dim3(rb), dim3(256), 0, st, ca);
    if (o.d_weights0) NSRW_HIP(hipMemcpyAsync(o.d_weights0 + r0 * S0, k.w0, (size_t)Rc * S0 * 4, hipMemcpyDeviceToDevice, st));
    const float* z_last = k.z0;
    const float* noise_last = ca.noise;
    if (NI > 0) {
      // ---- resampling (RN:473-477) ----
      PdfArgs pa{};
      pa.R = Rc; pa.S0 = S0; pa.NI = NI; pa.z0 = k.z0; pa.w0 = k.w0; pa.u_tab = h->d_tab + S0;
      pa.u_rays = e.d_u ? e.d_u + r0 * NI : nullptr; pa.pdf = k.pdf; pa.cdf = k.cdf; pa.zs = k.zs;
      pa.inds = o.d_inds ? o.d_inds + r0 * NI : nullptr; pa.z_std = o.d_z_std ? o.d_z_std + r0 : nullptr;
      hipLaunchKernelGGL(kw_sample_pdf, dim3(rb), dim3(256), 0, st, pa);
      if (o.d_z_samples) NSRW_HIP(hipMemcpyAsync(o.d_z_samples + r0 * NI, k.zs, (size_t)Rc * NI * 4, hipMemcpyDeviceToDevice, st));
      hipLaunchKernelGGL(kw_sort, dim3((unsigned)Rc), dim3(64), (size_t)S1 * sizeof(float), st, k.z0, k.zs, S0, NI, k.zf);
      if (e.d_z_fine)          // given depths (constants of the gradient, RN:475) replace the resampled ones for the fine pass
        NSRW_HIP(hipMemcpyAsync(k.zf, e.d_z_fine + r0 * S1, (size_t)Rc * S1 * 4, hipMemcpyDeviceToDevice, st));
      // ---- fine pass (RN:478-485) ----
      encode(n1, k.zf, S1);
      if (net_forward(h->gemm_cfg, st, n1, k, (long long)Rc * S1, grad, &sigma, &ld_sigma)) return 1;
      CompositeArgs cf = ca;
      cf.S = S1; cf.ld_rgb = n1.ldraw; cf.sigma = sigma; cf.ld_sigma = ld_sigma; cf.z = k.zf; cf.noise = e.d_noise1 ? e.d_noise1 + r0 * S1 : nullptr;
      cf.weights = k.wf;
      cf.rgb_map = o.d_rgb ? o.d_rgb + r0 * 3 : nullptr; cf.disp = o.d_disp ? o.d_disp + r0 : nullptr;
      cf.acc = o.d_acc ? o.d_acc + r0 : nullptr;
      cf.raw_out = nullptr; cf.raw_src = nullptr;
      if (o.d_raw) {
        cf.raw_out = o.d_raw + r0 * S1 * last.raw_ch; cf.raw_ch = last.raw_ch;
        if (!last.d.use_viewdirs) { cf.raw_src = k.RAW; cf.ld_raw = 32; }
      }
      if (cf.rgb_map && !(cf.disp && cf.acc)) return fail("nsrw: rgb / disp / acc outputs of a pass must be given together");
      hipLaunchKernelGGL(kw_composite, dim3(rb), dim3(256), 0, st, cf);
      z_last = k.zf;
      noise_last = cf.noise;
    }
    if (o.d_z_vals) NSRW_HIP(hipMemcpyAsync(o.d_z_vals + r0 * SL, z_last, (size_t)Rc * SL * 4, hipMemcpyDeviceToDevice, st));
    if (grad) {
      const long long P = (long long)Rc * SL;
      NSRW_HIP(hipMemsetAsync(k.DRAW, 0, (size_t)P * 32 * 4, st));
      CompositeBwdArgs cb{};
      cb.R = Rc; cb.S = SL; cb.flags = h->cfg.flags; cb.rgb = k.RAW; cb.ld_rgb = last.ldraw; cb.sigma = sigma; cb.ld_sigma = ld_sigma;
      cb.z = z_last; cb.nrm = k.nrm; cb.noise = noise_last; cb.grad_rgb = grad_rgb + r0 * 3; cb.draw = k.DRAW; cb.gnorm = k.gnorm;
      cb.scr_a = k.scr_a; cb.scr_t = k.scr_t; cb.gscale = h->gemm_cfg.h2 ? k.gscale : nullptr;
      hipLaunchKernelGGL(kw_composite_bwd, dim3(rb), dim3(256), 0, st, cb);
      if (net_backward(h->gemm_cfg, st, last, k, P)) return 1;
      EmbedBwdArgs eb{};
      eb.R = Rc; eb.S = SL; eb.L = last.d.multires; eb.Lv = last.d.multires_views;
      eb.rays_o = ra.rays_o; eb.rays_d = ra.rays_d; eb.z = z_last; eb.vd = k.vd; eb.nrm = k.nrm; eb.gnorm = k.gnorm;
      eb.GE = k.GEP; eb.ldE = last.Ci; eb.GED = last.d.use_viewdirs ? k.GED : nullptr; eb.ldED = last.Cv;
      eb.given_viewdirs = e.d_viewdirs != nullptr; eb.gscale = cb.gscale;
      eb.grad_o = grad_o + r0 * 3; eb.grad_d = grad_d + r0 * 3; eb.grad_v = grad_v ? grad_v + r0 * 3 : nullptr;
      hipLaunchKernelGGL(kw_embed_bwd, dim3((unsigned)Rc), dim3(64), 0, st, eb);
    }
    ++h->chunks;
  }
  if (!cap) NSRW_HIP(hipEventRecord(h->ev1, st));
  h->timed = !cap;
  NSRW_HIP(hipGetLastError());
  return 0;
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------------------------------
extern "C" {

const char* nsrw_last_error(void) { return g_err.c_str(); }

size_t nsrw_network_floats(const NsrwNet* net) {
  if (!net) { fail("nsrw_network_floats: null description"); return 0; }
  const std::string why = check_net(*net);
  if (!why.empty()) { fail("nsrw: " + why); return 0; }
  return net_floats(*net);
}

int nsrw_create(const NsrwConfig* cfg, nsrw_handle* out) {
  if (!cfg || !out) return fail("nsrw_create: null argument");
  if (cfg->n_samples < 3 || cfg->n_samples > NSRW_MAX_SAMPLES) return fail("nsrw_create: N_samples must be 3.." + std::to_string(NSRW_MAX_SAMPLES));
  if (cfg->n_importance < 0 || cfg->n_importance > NSRW_MAX_SAMPLES) return fail("nsrw_create: N_importance must be 0.." + std::to_string(NSRW_MAX_SAMPLES));
  int count = 0;
  NSRW_HIP(hipGetDeviceCount(&count));
  if (cfg->device < 0 || cfg->device >= count) return fail("nsrw_create: no such device");
  GemmCfg gc;
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess && prop.multiProcessorCount > 0) gc.cus = prop.multiProcessorCount;
  }
  gc.h2 = (cfg->flags & NSRW_FLAG_MLP_F16X2) != 0;
  gc.b3 = gc.h2 || (cfg->flags & NSRW_FLAG_MLP_BF16X3) != 0;
  if (const char* wm = getenv("NSRW_B3_WM")) gc.b3_wm = atoi(wm) == 2 ? 2 : 4;
  Handle* h = new Handle();
  h->cfg = *cfg;
  h->gemm_cfg = gc;
  DeviceGuard guard(cfg->device);
  if (guard.err != hipSuccess) { delete h; return fail("nsrw_create: hipSetDevice failed"); }
  if (hipMalloc(&h->d_tab, (size_t)(cfg->n_samples + std::max(cfg->n_importance, 1)) * 4) != hipSuccess ||
      hipMalloc(&h->d_range, 4 * sizeof(unsigned)) != hipSuccess || hipMemset(h->d_range, 0, 4 * sizeof(unsigned)) != hipSuccess ||
      hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess) {
    if (h->d_tab) (void)hipFree(h->d_tab);
    if (h->d_range) (void)hipFree(h->d_range);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    delete h;
    return fail("nsrw_create: out of device memory");
  }
  h->gemm_cfg.d_range = h->d_range;
  *out = reinterpret_cast<nsrw_handle>(h);
  return 0;
}

int nsrw_range_status(nsrw_handle hh, unsigned long long* passes, unsigned long long* passes_rerun) {
  Handle* h = reinterpret_cast<Handle*>(hh);
  if (!h) return fail("nsrw_range_status: null handle");
  NSRW_DEVICE(h);
  unsigned r[4] = {0, 0, 0, 0};
  NSRW_HIP(hipDeviceSynchronize());
  NSRW_HIP(hipMemcpy(r, h->d_range, sizeof r, hipMemcpyDeviceToHost));
  if (passes) *passes = r[1];
  if (passes_rerun) *passes_rerun = r[2];
  return 0;
}

int nsrw_destroy(nsrw_handle hh) {
  Handle* h = reinterpret_cast<Handle*>(hh);
  if (!h) return 0;
  DeviceGuard guard(h->cfg.device);
  for (Net& n : h->net) {
    if (n.dW) (void)hipFree(n.dW);
    if (n.dWb) (void)hipFree(n.dWb);
    if (n.dWh) (void)hipFree(n.dWh);
    if (n.dWf) (void)hipFree(n.dWf);
  }
  if (h->d_tab) (void)hipFree(h->d_tab);
  if (h->d_range) (void)hipFree(h->d_range);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  delete h;
  return 0;
}

int nsrw_upload_tables(nsrw_handle hh, const float* t_coarse, int n_coarse, const float* u_fine, int n_fine) {
  Handle* h = reinterpret_cast<Handle*>(hh);
  if (!h || !t_coarse) return fail("nsrw_upload_tables: null argument");
  if (n_coarse != h->cfg.n_samples || n_fine != h->cfg.n_importance || (n_fine > 0 && !u_fine))
    return fail("nsrw_upload_tables: table sizes must be the handle's N_samples / N_importance");
  NSRW_DEVICE(h);
  NSRW_HIP(hipMemcpy(h->d_tab, t_coarse, (size_t)n_coarse * 4, hipMemcpyHostToDevice));
  if (n_fine > 0) NSRW_HIP(hipMemcpy(h->d_tab + n_coarse, u_fine, (size_t)n_fine * 4, hipMemcpyHostToDevice));
  h->tables = true;
  return 0;
}

int nsrw_upload_network(nsrw_handle hh, int net_id, const NsrwNet* desc, const float* wts, size_t n_floats) {
  Handle* h = reinterpret_cast<Handle*>(hh);
  if (!h || !desc || !wts) return fail("nsrw_upload_network: null argument");
  if (net_id < 0 || net_id > 1) return fail("nsrw_upload_network: net_id must be 0 or 1");
  const std::string why = check_net(*desc);
  if (!why.empty()) return fail("nsrw_upload_network: " + why);
  if (n_floats != net_floats(*desc)) return fail("nsrw_upload_network: expected " + std::to_string(net_floats(*desc)) + " floats, got " + std::to_string(n_floats));
  Net n;
  n.d = *desc;
  const int D = desc->D, W = desc->W;
  n.in_ch = 3 + 6 * desc->multires; n.in_v = 3 + 6 * desc->multires_views;
  n.Ci = pad32(n.in_ch); n.Cv = desc->use_viewdirs ? pad32(n.in_v) : 32; n.Wp = pad32(W); n.W2 = W / 2; n.W2p = pad32(std::max(n.W2, 1));
  n.ldfa = n.Wp;
  n.ldraw = desc->use_viewdirs ? 4 : 32;
  n.raw_ch = desc->use_viewdirs ? 4 : desc->output_ch;
  n.skip_in.assign(D, 0);
  for (int i = 1; i < D; ++i) n.skip_in[i] = is_skip(*desc, i - 1);
  // views of the parameter block
  std::vector<const float*> Wl(D), Bl(D);
  std::vector<int> Kin(D);
  const float* p = wts;
  for (int i = 0; i < D; ++i) {
    Kin[i] = i == 0 ? n.in_ch : (n.skip_in[i] ? n.in_ch + W : W);
    Wl[i] = p; p += (size_t)W * Kin[i];
    Bl[i] = p; p += W;
  }
  const float *Wf = nullptr, *Bf = nullptr, *Wa = nullptr, *Ba = nullptr, *Wv = nullptr, *Bv = nullptr, *Wr = nullptr, *Br = nullptr,
              *Wo = nullptr, *Bo = nullptr;
  const int Kv = W + n.in_v;
  if (desc->use_viewdirs) {
    Wf = p; p += (size_t)W * W; Bf = p; p += W;
    Wa = p; p += W; Ba = p; p += 1;
    Wv = p; p += (size_t)n.W2 * Kv; Bv = p; p += n.W2;
    Wr = p; p += (size_t)3 * n.W2; Br = p; p += 3;
  } else {
    Wo = p; p += (size_t)desc->output_ch * W; Bo = p; p += desc->output_ch;
  }
  std::vector<float> img;
  n.fwd.resize(D); n.bwd_h.resize(D); n.bwd_e.resize(D);
  const int in_ch = n.in_ch, Ci = n.Ci, Wp = n.Wp;
  for (int i = 0; i < D; ++i) {
    Mat& m = n.fwd[i];
    const float* w = Wl[i];
    const int K = Kin[i];
    m.Np = Wp;
    if (i == 0) {
      m.K1p = Ci;
      m.w = pack(img, Wp, Ci, [&](int nn, int kk) { return (nn < W && kk < in_ch) ? w[(size_t)nn * K + kk] : 0.0f; });
    } else if (n.skip_in[i]) {                      // cat[input_pts, h]: encoding columns first (RH:106)
      m.K1p = Ci; m.K2p = Wp;
      m.w = pack(img, Wp, Ci + Wp, [&](int nn, int kk) {
        if (nn >= W) return 0.0f;
        if (kk < Ci) return kk < in_ch ? w[(size_t)nn * K + kk] : 0.0f;
        const int c = kk - Ci;
        return c < W ? w[(size_t)nn * K + in_ch + c] : 0.0f;
      });
    } else {
      m.K1p = Wp;
      m.w = pack(img, Wp, Wp, [&](int nn, int kk) { return (nn < W && kk < W) ? w[(size_t)nn * K + kk] : 0.0f; });
    }
    m.b = pack_bias(img, Wp, Bl[i], W);
    // transposed pieces for the backward: out[n'] = sum_k g[k] W_i[k][col(n')]
    const int hoff = (i > 0 && n.skip_in[i]) ? in_ch : 0;
    if (i > 0) {
      Mat& bh = n.bwd_h[i];
      bh.Np = Wp; bh.K1p = Wp;
      bh.w = pack(img, Wp, Wp, [&](int nn, int kk) { return (nn < W && kk < W) ? w[(size_t)kk * K + hoff + nn] : 0.0f; });
    }
    if (i == 0 || n.skip_in[i]) {
      Mat& be = n.bwd_e[i];
      be.Np = Ci; be.K1p = Wp;
      be.w = pack(img, Ci, Wp, [&](int nn, int kk) { return (nn < in_ch && kk < W) ? w[(size_t)kk * K + nn] : 0.0f; });
    }
  }
  if (desc->use_viewdirs) {
    const int W2 = n.W2, W2p = n.W2p, Cv = n.Cv, in_v = n.in_v;
    // feature_linear, and alpha_linear as a matrix of its own (row 0; its one output column goes to the raw block, net_forward)
    n.fa.Np = Wp; n.fa.K1p = Wp;
    n.fa.w = pack(img, Wp, Wp, [&](int nn, int kk) { return (nn < W && kk < W) ? Wf[(size_t)nn * W + kk] : 0.0f; });
    n.fa.b = pack_bias(img, Wp, Bf, W);
    n.al.Np = 32; n.al.K1p = Wp;
    n.al.w = pack(img, 32, Wp, [&](int nn, int kk) { return (nn == 0 && kk < W) ? Wa[kk] : 0.0f; });
    n.al.b = pack_bias(img, 32, Ba, 1);
    // views_linears.0 on cat[feature, input_views] (RH:113): A1 = FA's feature columns, A2 = direction encoding
    n.hv.Np = W2p; n.hv.K1p = Wp; n.hv.K2p = Cv;
    n.hv.w = pack(img, W2p, Wp + Cv, [&](int nn, int kk) {
      if (nn >= W2) return 0.0f;
      if (kk < Wp) return kk < W ? Wv[(size_t)nn * Kv + kk] : 0.0f;
      const int c = kk - Wp;
      return c < in_v ? Wv[(size_t)nn * Kv + W + c] : 0.0f;
    });
    n.hv.b = pack_bias(img, W2p, Bv, W2);
    n.rgb.Np = 32; n.rgb.K1p = W2p;
    n.rgb.w = pack(img, 32, W2p, [&](int nn, int kk) { return (nn < 3 && kk < W2) ? Wr[(size_t)nn * W2 + kk] : 0.0f; });
    n.rgb.b = pack_bias(img, 32, Br, 3);
    // backward heads
    n.b_rgb.Np = W2p; n.b_rgb.K1p = 32;
    n.b_rgb.w = pack(img, W2p, 32, [&](int nn, int kk) { return (nn < W2 && kk < 3) ? Wr[(size_t)kk * W2 + nn] : 0.0f; });
    n.b_feat.Np = Wp; n.b_feat.K1p = W2p;
    n.b_feat.w = pack(img, Wp, W2p, [&](int nn, int kk) { return (nn < W && kk < W2) ? Wv[(size_t)kk * Kv + nn] : 0.0f; });
    n.b_ed.Np = Cv; n.b_ed.K1p = W2p;
    n.b_ed.w = pack(img, Cv, W2p, [&](int nn, int kk) { return (nn < in_v && kk < W2) ? Wv[(size_t)kk * Kv + W + nn] : 0.0f; });
    n.b_head.Np = Wp; n.b_head.K1p = Wp; n.b_head.K2p = 32;       // [dL/dfeature | dL/draw]: feature_linear^T, alpha_linear^T at column 3
    n.b_head.w = pack(img, Wp, Wp + 32, [&](int nn, int kk) {
      if (nn >= W) return 0.0f;
      if (kk < Wp) return kk < W ? Wf[(size_t)kk * W + nn] : 0.0f;
      return kk - Wp == 3 ? Wa[nn] : 0.0f;
    });
  } else {
    const int oc = desc->output_ch;
    n.out.Np = 32; n.out.K1p = Wp;
    n.out.w = pack(img, 32, Wp, [&](int nn, int kk) { return (nn < oc && kk < W) ? Wo[(size_t)nn * W + kk] : 0.0f; });
    n.out.b = pack_bias(img, 32, Bo, oc);
    n.b_head.Np = Wp; n.b_head.K1p = 32;                            // render_rays reads rows 0..3 of output_linear (RN:363-374)
    n.b_head.w = pack(img, Wp, 32, [&](int nn, int kk) { return (nn < W && kk < 4) ? Wo[(size_t)kk * W + nn] : 0.0f; });
  }
  // bf16x3 handles: every matrix once more as three bf16 pieces in the layout kw_gemm_b3's LDS-DMA reads (6 bytes per weight)
  std::vector<uint16_t> imgb;
  if (h->gemm_cfg.b3) {
    std::vector<Mat*> all;
    for (Mat& m : n.fwd) all.push_back(&m);
    for (Mat& m : n.bwd_h) all.push_back(&m);
    for (Mat& m : n.bwd_e) all.push_back(&m);
    for (Mat* m : {&n.fa, &n.al, &n.hv, &n.rgb, &n.out, &n.b_rgb, &n.b_feat, &n.b_ed, &n.b_head}) all.push_back(m);
    for (Mat* m : all)
      if (m->Np > 0) m->wb = pack_b3(imgb, img.data() + m->w, m->Np, m->K1p + m->K2p, &m->ncb);
    if (imgb.size() * sizeof(uint16_t) >= 0x7fffffffull) return fail("nsrw_upload_network: network too large for the bf16x3 image");
  }
  // f16x2 handles: ... and as scaled two-piece fp16 images (4 bytes per weight) for kw_gemm_h2
  std::vector<uint16_t> imgh;
  if (h->gemm_cfg.h2) {
    std::vector<Mat*> all;
    for (Mat& m : n.fwd) all.push_back(&m);
    for (Mat& m : n.bwd_h) all.push_back(&m);
    for (Mat& m : n.bwd_e) all.push_back(&m);
    for (Mat* m : {&n.fa, &n.al, &n.hv, &n.rgb, &n.out, &n.b_rgb, &n.b_feat, &n.b_ed, &n.b_head}) all.push_back(m);
    for (Mat* m : all)
      if (m->Np > 0) m->wh = pack_h2(imgh, img.data() + m->w, m->Np, m->K1p + m->K2p, &m->cscale);
    if (imgh.size() * sizeof(uint16_t) >= 0x7fffffffull) return fail("nsrw_upload_network: network too large for the f16x2 image");
  }
  // fp32 handles: the matrices in the LDS-image layout of kw_gemm_f32 (4 bytes per weight; dW keeps the biases)
  std::vector<float> imgf;
  if (!h->gemm_cfg.b3) {
    std::vector<Mat*> all;
    for (Mat& m : n.fwd) all.push_back(&m);
    for (Mat& m : n.bwd_h) all.push_back(&m);
    for (Mat& m : n.bwd_e) all.push_back(&m);
    for (Mat* m : {&n.fa, &n.al, &n.hv, &n.rgb, &n.out, &n.b_rgb, &n.b_feat, &n.b_ed, &n.b_head}) all.push_back(m);
    for (Mat* m : all)
      if (m->Np > 0) {
        m->wf = pack_f32img(imgf, img.data() + m->w, m->Np, m->K1p + m->K2p);
        m->ncb = ((m->Np + 31) / 32 + 1) / 2 * 2;
      }
    if (imgf.size() * sizeof(float) >= 0x7fffffffull) return fail("nsrw_upload_network: network too large for the fp32 image");
  }
  NSRW_DEVICE(h);
  // (every failure path below frees what this call allocated: ADVICE r05)
  hipError_t e = hipMalloc(&n.dW, img.size() * sizeof(float));
  if (e == hipSuccess) e = hipMemcpy(n.dW, img.data(), img.size() * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess && !imgb.empty()) {
    n.wb_bytes = imgb.size() * sizeof(uint16_t);
    e = hipMalloc(&n.dWb, n.wb_bytes);
    if (e == hipSuccess) e = hipMemcpy(n.dWb, imgb.data(), n.wb_bytes, hipMemcpyHostToDevice);
  }
  if (e == hipSuccess && !imgh.empty()) {
    n.wh_bytes = imgh.size() * sizeof(uint16_t);
    e = hipMalloc(&n.dWh, n.wh_bytes);
    if (e == hipSuccess) e = hipMemcpy(n.dWh, imgh.data(), n.wh_bytes, hipMemcpyHostToDevice);
  }
  if (e == hipSuccess && !imgf.empty()) {
    n.wf_bytes = imgf.size() * sizeof(float);
    e = hipMalloc(&n.dWf, n.wf_bytes);
    if (e == hipSuccess) e = hipMemcpy(n.dWf, imgf.data(), n.wf_bytes, hipMemcpyHostToDevice);
  }
  Net& slot = h->net[net_id];
  if (e == hipSuccess && (slot.dW || slot.dWb)) e = hipDeviceSynchronize();      // a launch may still read the images being replaced
  if (e != hipSuccess) {
    if (n.dW) (void)hipFree(n.dW);
    if (n.dWb) (void)hipFree(n.dWb);
    if (n.dWh) (void)hipFree(n.dWh);
    if (n.dWf) (void)hipFree(n.dWf);
    return fail(std::string("nsrw_upload_network: ") + hipGetErrorString(e));
  }
  if (slot.dW) (void)hipFree(slot.dW);
  if (slot.dWb) (void)hipFree(slot.dWb);
  if (slot.dWh) (void)hipFree(slot.dWh);
  if (slot.dWf) (void)hipFree(slot.dWf);
  slot = n;
  slot.loaded = true;
  return 0;
}

int nsrw_workspace_bytes(nsrw_handle hh, int64_t rays, int with_grad, size_t* bytes) {
  Handle* h = reinterpret_cast<Handle*>(hh);
  if (!h || !bytes) return fail("nsrw_workspace_bytes: null argument");
  if (!h->net[0].loaded) return fail("nsrw_workspace_bytes: network 0 not uploaded");
  if (rays < 1) rays = 1;
  const long long R = (rays + 63) / 64 * 64;
  Carve c(nullptr);
  Chunk k;
  carve_chunk(*h, R, with_grad != 0, c, k);
  *bytes = c.off;
  return 0;
}

int nsrw_render_rays(nsrw_handle hh, const float* d_rays_o, const float* d_rays_d, int64_t n_rays, float near_, float far_,
                     const NsrwExtras* ex, const NsrwOut* out, void* ws, size_t ws_bytes, void* stream) {
  Handle* h = reinterpret_cast<Handle*>(hh);
  if (check_common(h, d_rays_o, d_rays_d, n_rays, ex)) return 1;
  NSRW_DEVICE(h);
  return render_impl(h, d_rays_o, d_rays_d, n_rays, near_, far_, ex, nullptr, out, nullptr, nullptr, nullptr, ws, ws_bytes,
                     static_cast<hipStream_t>(stream));
}

int nsrw_render_rays_vjp(nsrw_handle hh, const float* d_rays_o, const float* d_rays_d, int64_t n_rays, float near_, float far_,
                         const NsrwExtras* ex, const float* d_grad_rgb, const NsrwOut* out, float* d_grad_o, float* d_grad_d,
                         float* d_grad_viewdirs, void* ws, size_t ws_bytes, void* stream) {
  Handle* h = reinterpret_cast<Handle*>(hh);
  if (check_common(h, d_rays_o, d_rays_d, n_rays, ex)) return 1;
  if (n_rays > 0 && (!d_grad_rgb || !d_grad_o || !d_grad_d)) return fail("nsrw_render_rays_vjp: null gradient buffer");
  if (d_grad_viewdirs && !(ex && ex->d_viewdirs)) return fail("nsrw_render_rays_vjp: grad_viewdirs without given view directions");
  NSRW_DEVICE(h);
  return render_impl(h, d_rays_o, d_rays_d, n_rays, near_, far_, ex, d_grad_rgb, out, d_grad_o, d_grad_d, d_grad_viewdirs, ws,
                     ws_bytes, static_cast<hipStream_t>(stream));
}

int nsrw_run_network(nsrw_handle hh, int net_id, const float* d_pts, const float* d_viewdirs, int64_t n_pts, float* d_raw,
                     void* ws, size_t ws_bytes, void* stream) {
  Handle* h = reinterpret_cast<Handle*>(hh);
  if (!h) return fail("nsrw_run_network: null handle");
  if (net_id < 0 || net_id > 1 || !h->net[net_id].loaded) return fail("nsrw_run_network: network not uploaded");
  if (n_pts < 0 || (n_pts > 0 && (!d_pts || !d_raw))) return fail("nsrw_run_network: null argument");
  const Net& n = h->net[net_id];
  if (n.d.use_viewdirs && n_pts > 0 && !d_viewdirs) return fail("nsrw_run_network: this network takes view directions");
  if (n_pts == 0) return 0;
  if (!ws || (reinterpret_cast<uintptr_t>(ws) & 255) != 0) return fail("nsrw_run_network: workspace must be 256-byte aligned");
  NSRW_DEVICE(h);
  hipStream_t st = static_cast<hipStream_t>(stream);
  // points per pass: what the workspace holds of [E | ED | FA | HV | RAW | H0 | H1]
  const size_t per_pt = (size_t)(n.Ci + n.Cv + n.ldfa + n.W2p + 32 + 2 * n.Wp) * 4 + 64;
  long long PB = (long long)((ws_bytes > 4096 ? ws_bytes - 4096 : 0) / per_pt) / 128 * 128;
  if (PB < 128) return fail("nsrw_run_network: workspace too small");
  PB = std::min<long long>(PB, (n_pts + 127) / 128 * 128);
  Carve c(ws);
  Chunk k;
  k.E = c.f(PB * n.Ci); k.ED = c.f(PB * n.Cv); k.FA = c.f(PB * n.ldfa); k.HV = c.f(PB * n.W2p); k.RAW = c.f(PB * 32);
  k.H = {c.f(PB * n.Wp), c.f(PB * n.Wp)};
  if (c.off > ws_bytes) return fail("nsrw_run_network: workspace too small");
  const bool cap = capturing(st);
  if (!cap) NSRW_HIP(hipEventRecord(h->ev0, st));
  h->chunks = 0;
  for (long long p0 = 0; p0 < n_pts; p0 += PB) {
    const long long P = std::min<long long>(PB, n_pts - p0);
    EmbedArgs ea{};
    ea.pts = d_pts + p0 * 3; ea.dirs = d_viewdirs ? d_viewdirs + p0 * 3 : nullptr; ea.P = P; ea.S = 1;
    ea.L = n.d.multires; ea.Lv = n.d.use_viewdirs ? n.d.multires_views : -1;
    ea.E = k.E; ea.ldE = n.Ci; ea.ED = k.ED; ea.ldED = n.Cv;
    hipLaunchKernelGGL(kw_embed, dim3((unsigned)((P * 16 + 255) / 256)), dim3(256), 0, st, ea);
    const float* sigma; int ld_sigma;
    if (net_forward(h->gemm_cfg, st, n, k, P, false, &sigma, &ld_sigma)) return 1;
    const int C = n.raw_ch;
    if (n.d.use_viewdirs) {          // the raw block IS [P, 4] = [r g b sigma]
      NSRW_HIP(hipMemcpyAsync(d_raw + p0 * 4, k.RAW, (size_t)P * 4 * sizeof(float), hipMemcpyDeviceToDevice, st));
    } else {
      hipLaunchKernelGGL(kw_copy_rows, dim3((unsigned)((P * C + 255) / 256)), dim3(256), 0, st, k.RAW, 32, d_raw + p0 * C, C, P, C);
    }
    ++h->chunks;
  }
  if (!cap) NSRW_HIP(hipEventRecord(h->ev1, st));
  h->timed = !cap;
  NSRW_HIP(hipGetLastError());
  return 0;
}

int nsrw_debug_bounds_status(int* built_with_checks, unsigned* first_bad_line) {
  if (!built_with_checks || !first_bad_line) return fail("nsrw_debug_bounds_status: null argument");
  *first_bad_line = 0;
#ifdef NSR_DEBUG_BOUNDS
  *built_with_checks = 1;
  NSRW_HIP(hipDeviceSynchronize());
  NSRW_HIP(hipMemcpyFromSymbol(first_bad_line, HIP_SYMBOL(nsrw::g_wide_violation), sizeof(unsigned)));
#else
  *built_with_checks = 0;
#endif
  return 0;
}

int nsrw_sample_pdf(int device, const float* d_bins, const float* d_weights, int64_t n_rows, int n_bins, const float* d_u,
                    int u_per_row, int n_samples, float* d_samples, int64_t* d_inds, float* d_scratch, void* stream) {
  if (!d_bins || !d_weights || !d_u || !d_samples || !d_scratch) return fail("nsrw_sample_pdf: null argument");
  if (n_rows < 0 || n_rows > 0x7fffffffll) return fail("nsrw_sample_pdf: 0 .. 2^31 - 1 rows");
  if (n_bins < 2 || n_bins > NSRW_MAX_SAMPLES || n_samples < 1 || n_samples > NSRW_MAX_SAMPLES)
    return fail("nsrw_sample_pdf: 2.." + std::to_string(NSRW_MAX_SAMPLES) + " bins, 1.." + std::to_string(NSRW_MAX_SAMPLES) + " samples");
  if (n_rows == 0) return 0;
  DeviceGuard guard_(device);
  NSRW_HIP(guard_.err);
  PdfArgs pa{};
  pa.R = (int)n_rows; pa.S0 = n_bins + 1; pa.NI = n_samples; pa.z0 = d_bins; pa.w0 = d_weights; pa.bins = d_bins;
  pa.u_tab = u_per_row ? nullptr : d_u; pa.u_rays = u_per_row ? d_u : nullptr;
  pa.pdf = d_scratch; pa.cdf = d_scratch + (size_t)n_rows * (n_bins + 1);
  pa.zs = d_samples; pa.inds = d_inds; pa.z_std = nullptr;
  hipLaunchKernelGGL(kw_sample_pdf, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), pa);
  NSRW_HIP(hipGetLastError());
  return 0;
}

int nsrw_embed_vjp(int device, const float* d_x, const float* d_grad_out, int64_t n_points, int multires, float* d_grad_x, void* stream) {
  if (!d_x || !d_grad_out || !d_grad_x) return fail("nsrw_embed_vjp: null argument");
  if (n_points < 0 || multires < 0 || multires > 15) return fail("nsrw_embed_vjp: n_points >= 0, multires 0..15");
  if (n_points == 0) return 0;
  DeviceGuard guard_(device);
  NSRW_HIP(guard_.err);
  hipLaunchKernelGGL(kw_embed_vjp, dim3((unsigned)((n_points + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), d_x, d_grad_out,
                     (long long)n_points, multires, d_grad_x);
  NSRW_HIP(hipGetLastError());
  return 0;
}

int nsrw_last_ms(nsrw_handle hh, float* ms, int* chunks) {
  Handle* h = reinterpret_cast<Handle*>(hh);
  if (!h || !ms) return fail("nsrw_last_ms: null argument");
  if (!h->timed) return fail("nsrw_last_ms: no timed launch yet (launches captured into a hipGraph are not timed)");
  NSRW_DEVICE(h);
  NSRW_HIP(hipEventSynchronize(h->ev1));
  NSRW_HIP(hipEventElapsedTime(ms, h->ev0, h->ev1));
  if (chunks) *chunks = h->chunks;
  return 0;
}

}  // extern "C"
