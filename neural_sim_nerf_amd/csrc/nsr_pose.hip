// nsr_pose.hip -- psi -> camera pose on the device: the pose-distribution sampler of the bilevel
// loop, writing [K,4,4] / [K,3,4] poses straight into the buffer nsr_render_views reads, plus the Jacobian
// d c2w[:3,:4] / d categorical_prob that render_path_grad's chain rule needs (RN:179-181).
//   LL = optimization/utils/load_LINEMOD_noscale.py, GU = optimization/utils/gumble.py
//   sample_pose LL:202-247 (+ differentiable_sample GU:57-63, pose_spherical LL:62-71): torch fp32 op by op
//   sample_pose_nograd LL:250-301 (+ GU:46-47, GU:64-70, pose_spherical_nograd LL:89-94): numpy fp64, then fp32 matrices
// The random draws stay where the reference makes them (numpy's Mersenne Twister on the host, recorded in sample_log,
// LL:273-297); this kernel is the deterministic map (probabilities, recorded noise) -> poses.  One thread per pose.
//
// Matrix products: c2w = FLIP @ (R_theta @ (R_phi @ T(radius))) (LL:64-70).  Every entry of every product has exactly
// one non-zero term (the factors are rotations about coordinate axes times a z translation), so each is ONE rounded
// fp32 product whatever the GEMM's summation order or FMA use -- the closed form below is bit-equal to torch's `@`:
//   c2w = [[-ct,  st*s,  st*c,  st*(c*r)],
//          [ st,  ct*s,  ct*c,  ct*(c*r)],
//          [  0,     c,    -s,    -(s*r)],
//          [  0,     0,     0,         1]]      c, s = cos/sin(phi), ct, st = cos/sin(theta), r = radius.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nsr {

constexpr int kMaxCat = 16;

struct PoseArgs {
  const float* prob;        // [n_cat] categorical probabilities (fp32; the fp64 path reads logits64 instead)
  const double* logits64;   // [n_cat] np.log(probs) as the reference computed it (fp16 probs -> fp16 log, widened)
  const double* gumbel;     // [K,n_cat] recorded Gumbel noise (fp64, sample_log['gumbel_noises'])
  const double* uniform;    // [K]
  const double* theta;      // [K] degrees
  int K, n_cat;
  double gumbel_T, radius;
  float* poses44;           // [K,4,4] or null
  float* c2w34;             // [K,3,4] or null
  float* jac;               // [K,12,n_cat] d c2w[:3,:4] / d prob, or null (fp32 path only)
};

__device__ __forceinline__ void write_pose(const PoseArgs& a, int k, float c, float s, float ct, float st, float r) {
  const float cr = c * r, sr = s * r;
  const float m[12] = {-ct, st * s, st * c, st * cr,
                       st, ct * s, ct * c, ct * cr,
                       0.0f, c, -s, -sr};
  if (a.c2w34)
    for (int i = 0; i < 12; ++i) a.c2w34[k * 12 + i] = m[i];
  if (a.poses44) {
    for (int i = 0; i < 12; ++i) a.poses44[k * 16 + i] = m[i];
    a.poses44[k * 16 + 12] = 0.0f; a.poses44[k * 16 + 13] = 0.0f; a.poses44[k * 16 + 14] = 0.0f; a.poses44[k * 16 + 15] = 1.0f;
  }
}

// bin centres: [0, 45, ..., 315] + 22.5 (LL:217 / LL:266), n_cat = 8 in the reference
__device__ __forceinline__ float degree_of(int j) { return 45.0f * (float)j + 22.5f; }

// LL:202-247 in torch's fp32 arithmetic, op by op (no contraction: this file is built with -ffp-contract=off)
__global__ void k_sample_pose(PoseArgs a) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= a.K) return;
  const int n = a.n_cat;
  const float T = (float)a.gumbel_T;
  float x[kMaxCat], w[kMaxCat];
  float mx = -INFINITY;
  for (int j = 0; j < n; ++j) {
    x[j] = (logf(a.prob[j]) + (float)a.gumbel[k * n + j]) / T;     // GU:60: (logits + torch.Tensor(noise)) / temperature
    mx = fmaxf(mx, x[j]);
  }
  float sum = 0.0f;
  for (int j = 0; j < n; ++j) { w[j] = expf(x[j] - mx); sum = sum + w[j]; }     // torch.softmax: max-shifted
  float sample = 0.0f;
  for (int j = 0; j < n; ++j) { w[j] = w[j] / sum; sample = sample + w[j] * degree_of(j); }   // GU:62
  const float phi_deg = ((sample - 22.5f) + (float)(45.0 * a.uniform[k])) - 180.0f;         // LL:232, LL:245
  const float ph = (phi_deg / 180.0f) * 3.14159274101257324f;                                // LL:67 (fp32 pi)
  const float th = ((float)a.theta[k] / 180.0f) * 3.14159274101257324f;                      // LL:69, theta = torch.Tensor([t])
  const float c = cosf(ph), s = sinf(ph), ct = cosf(th), st = sinf(th);
  const float r = (float)a.radius;
  write_pose(a, k, c, s, ct, st, r);
  if (a.jac) {
    // d c2w / d phi_rad (closed form above), d phi_rad / d phi_deg = pi/180, d phi_deg / d p_j = w_j (deg_j - sample) / (T p_j)
    const float dm[12] = {0.0f, st * c, -(st * s), -(st * s) * r,
                          0.0f, ct * c, -(ct * s), -(ct * s) * r,
                          0.0f, -s, -c, -(c * r)};
    for (int j = 0; j < n; ++j) {
      const float dphi = (w[j] * (degree_of(j) - sample)) / (T * a.prob[j]) * (3.14159274101257324f / 180.0f);
      for (int i = 0; i < 12; ++i) a.jac[(k * 12 + i) * n + j] = dm[i] * dphi;
    }
  }
}

// LL:250-301 in numpy's fp64 arithmetic; the rotation entries are rounded to fp32 (torch.Tensor(...).float(), LL:74-86)
// before the fp32 matrix products.
__global__ void k_sample_pose_nograd(PoseArgs a) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= a.K) return;
  const int n = a.n_cat;
  double e[kMaxCat];
  auto pairwise = [&](const double* v) {             // np.sum of n <= 16 contiguous doubles (pairwise_sum's 8-way unrolled head)
    if (n < 8) { double r = 0.0; for (int j = 0; j < n; ++j) r += v[j]; return r; }
    double r8[8];
    for (int j = 0; j < 8; ++j) r8[j] = v[j];
    int i = 8;
    for (; i < n - (n % 8); i += 8)
      for (int j = 0; j < 8; ++j) r8[j] += v[i + j];
    double res = ((r8[0] + r8[1]) + (r8[2] + r8[3])) + ((r8[4] + r8[5]) + (r8[6] + r8[7]));
    for (; i < n; ++i) res += v[i];
    return res;
  };
  for (int j = 0; j < n; ++j) e[j] = exp((a.logits64[j] + a.gumbel[k * n + j]) / a.gumbel_T);   // GU:46-47, GU:67
  const double tot = pairwise(e);
  for (int j = 0; j < n; ++j) e[j] = (e[j] / tot) * (double)degree_of(j);
  const double sample = pairwise(e);                                                             // GU:69
  const double phi_deg = ((sample - 22.5) + 45.0 * a.uniform[k]) - 180.0;                         // LL:284, LL:293
  const double ph = phi_deg / 180.0 * 3.141592653589793, th = a.theta[k] / 180.0 * 3.141592653589793;
  write_pose(a, k, (float)cos(ph), (float)sin(ph), (float)cos(th), (float)sin(th), (float)a.radius);
}

}  // namespace nsr
