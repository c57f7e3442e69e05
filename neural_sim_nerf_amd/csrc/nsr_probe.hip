// nsr_probe.hip -- libnsr_probe.so: diagnostic micro-kernels that isolate the layer GEMM of the render kernels
// (same ring / segment device code, included from nsr_kernels.hip) to attribute MFMA-rate losses.
// NOT part of the product library: built by `make probe`, declared in include/nsr_probe.h, used by tools/probe_*.py.
// Self-contained: allocates its own weight-stream stand-in, synchronises freely.
#include "nsr_kernels.hip"

namespace nsr {
#include "nsr_probe_kernels.inc"

__global__ void k_probe_fill(float* p, long long n) {     // non-trivial operands: zero-filled inputs clock higher (DVFS)
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[i] = ((float)(x & 0xffff) - 32768.0f) * (1.0f / 262144.0f);
  }
}
__global__ void k_probe_fill_bf16(unsigned* p, long long n) {     // two bf16 weight pieces per word, |w| < 1/8
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    const float a = ((float)(x & 0xffff) - 32768.0f) * (1.0f / 262144.0f);
    const float b = ((float)(x >> 16) - 32768.0f) * (1.0f / 262144.0f);
    p[i] = (__float_as_uint(a) >> 16) | (__float_as_uint(b) & 0xffff0000u);
  }
}
}  // namespace nsr

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/nsr_probe.h"

namespace {
thread_local std::string g_perr;
int pfail(const std::string& m) { g_perr = m; return 1; }
#define NSRP_HIP(expr)                                                                        \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess) return pfail(std::string(#expr) + ": " + hipGetErrorString(e_));    \
  } while (0)
}  // namespace

extern "C" {

const char* nsr_probe_last_error(void) { return g_perr.c_str(); }

int nsr_probe(int device, int mode, int iters, int partner_prio, float* ms) {
  if (!ms) return pfail("nsr_probe: null argument");
  if (mode < 0 || mode > 30 || iters <= 0) return pfail("nsr_probe: mode in 0..30, iters > 0");
  NSRP_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  NSRP_HIP(hipGetDeviceProperties(&prop, device));
  const int n_cu = prop.multiProcessorCount;
  hipStream_t s = nullptr;
  const long long n_stream = (long long)nsr::kStreamSlabsB3 * nsr::kSlabFloats + nsr::kAuxFloats;   // the longer of the two streams
  float *wstream = nullptr, *out = nullptr;
  int* done = nullptr;
  hipEvent_t ev0, ev1;
  NSRP_HIP(hipMalloc(&wstream, sizeof(float) * n_stream));
  NSRP_HIP(hipMalloc(&out, sizeof(float) * 512 * n_cu));
  NSRP_HIP(hipMalloc(&done, sizeof(int)));
  NSRP_HIP(hipMemset(done, 0, sizeof(int)));
  NSRP_HIP(hipEventCreate(&ev0));
  NSRP_HIP(hipEventCreate(&ev1));
  if (mode >= 11) hipLaunchKernelGGL(nsr::k_probe_fill_bf16, dim3(1024), dim3(256), 0, s, (unsigned*)wstream, n_stream);
  else hipLaunchKernelGGL(nsr::k_probe_fill, dim3(1024), dim3(256), 0, s, wstream, n_stream);
  const size_t lds = nsr::kRingSlots * nsr::kSlabBytes, l16 = nsr::kRing16 * nsr::kSlabBytes, lepi = lds + 1024;
#define NSRP_LDS(K, L) NSRP_HIP(hipFuncSetAttribute((const void*)(K), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(L)))
  NSRP_LDS(nsr::k_probe<0>, lds); NSRP_LDS(nsr::k_probe<1>, lds); NSRP_LDS(nsr::k_probe<2>, lds);
  NSRP_LDS(nsr::k_probe16, l16);
  NSRP_LDS(nsr::k_probe_epi<9>, lepi); NSRP_LDS(nsr::k_probe_epi<10>, lepi);
  NSRP_LDS(nsr::k_probe16_pair<0>, l16); NSRP_LDS(nsr::k_probe16_pair<1>, l16); NSRP_LDS(nsr::k_probe16_pair<2>, l16);
  NSRP_LDS(nsr::k_probe16_pair<3>, l16); NSRP_LDS(nsr::k_probe16_pair<4>, l16);
  NSRP_LDS(nsr::k_probe_b3<11>, lepi); NSRP_LDS(nsr::k_probe_b3<12>, lepi); NSRP_LDS(nsr::k_probe_b3<13>, lepi);
  NSRP_LDS(nsr::k_probe_b3<14>, lepi); NSRP_LDS(nsr::k_probe_b3<15>, lepi); NSRP_LDS(nsr::k_probe_b3<16>, lepi); NSRP_LDS(nsr::k_probe_b3<17>, lepi); NSRP_LDS(nsr::k_probe_b3<18>, lepi);
  NSRP_LDS(nsr::k_probe_b3<19>, lepi); NSRP_LDS(nsr::k_probe_b3<20>, lepi); NSRP_LDS(nsr::k_probe_b3<21>, lepi); NSRP_LDS(nsr::k_probe_b3<22>, lepi); NSRP_LDS(nsr::k_probe_b3<23>, lepi); NSRP_LDS(nsr::k_probe_b3<24>, lepi); NSRP_LDS(nsr::k_probe_b3<25>, lepi); NSRP_LDS(nsr::k_probe_b3<26>, lepi); NSRP_LDS(nsr::k_probe_b3<27>, lepi);
  NSRP_LDS(nsr::k_probe_h2<28>, lepi); NSRP_LDS(nsr::k_probe_h2<29>, lepi); NSRP_LDS(nsr::k_probe_h2<30>, lepi);
  NSRP_HIP(hipDeviceSynchronize());
  NSRP_HIP(hipEventRecord(ev0, s));
  const dim3 b(256), g1(n_cu), g2(2 * n_cu);
  switch (mode) {
    case 0: hipLaunchKernelGGL(nsr::k_probe<0>, g1, b, lds, s, wstream, out, iters); break;
    case 1: hipLaunchKernelGGL(nsr::k_probe<1>, g1, b, lds, s, wstream, out, iters); break;
    case 2: hipLaunchKernelGGL(nsr::k_probe<2>, g1, b, lds, s, wstream, out, iters); break;
    case 3: hipLaunchKernelGGL(nsr::k_probe16, g2, b, l16, s, wstream, out, iters); break;
    case 4: hipLaunchKernelGGL(nsr::k_probe16_pair<0>, g2, b, l16, s, wstream, out, iters, done, n_cu, partner_prio); break;
    case 5: hipLaunchKernelGGL(nsr::k_probe16_pair<1>, g2, b, l16, s, wstream, out, iters, done, n_cu, partner_prio); break;
    case 6: hipLaunchKernelGGL(nsr::k_probe16_pair<2>, g2, b, l16, s, wstream, out, iters, done, n_cu, partner_prio); break;
    case 7: hipLaunchKernelGGL(nsr::k_probe16_pair<3>, g2, b, l16, s, wstream, out, iters, done, n_cu, partner_prio); break;
    case 8: hipLaunchKernelGGL(nsr::k_probe16_pair<4>, g2, b, l16, s, wstream, out, iters, done, n_cu, partner_prio); break;
    case 9: hipLaunchKernelGGL(nsr::k_probe_epi<9>, g1, b, lepi, s, wstream, out, iters); break;
    case 10: hipLaunchKernelGGL(nsr::k_probe_epi<10>, g1, b, lepi, s, wstream, out, iters); break;
    case 11: hipLaunchKernelGGL(nsr::k_probe_b3<11>, g1, b, lepi, s, wstream, out, iters); break;
    case 12: hipLaunchKernelGGL(nsr::k_probe_b3<12>, g1, b, lepi, s, wstream, out, iters); break;
    case 13: hipLaunchKernelGGL(nsr::k_probe_b3<13>, g1, b, lepi, s, wstream, out, iters); break;
    case 14: hipLaunchKernelGGL(nsr::k_probe_b3<14>, g1, b, lepi, s, wstream, out, iters); break;
    case 15: hipLaunchKernelGGL(nsr::k_probe_b3<15>, g1, b, lepi, s, wstream, out, iters); break;
    case 16: hipLaunchKernelGGL(nsr::k_probe_b3<16>, g1, b, lepi, s, wstream, out, iters); break;
    case 17: hipLaunchKernelGGL(nsr::k_probe_b3<17>, g1, b, lepi, s, wstream, out, iters); break;
    case 18: hipLaunchKernelGGL(nsr::k_probe_b3<18>, g1, b, lepi, s, wstream, out, iters); break;
    case 19: hipLaunchKernelGGL(nsr::k_probe_b3<19>, g1, b, lepi, s, wstream, out, iters); break;
    case 20: hipLaunchKernelGGL(nsr::k_probe_b3<20>, g1, b, lepi, s, wstream, out, iters); break;
    case 21: hipLaunchKernelGGL(nsr::k_probe_b3<21>, g1, b, lepi, s, wstream, out, iters); break;
    case 22: hipLaunchKernelGGL(nsr::k_probe_b3<22>, g1, b, lepi, s, wstream, out, iters); break;
    case 23: hipLaunchKernelGGL(nsr::k_probe_b3<23>, g1, b, lepi, s, wstream, out, iters); break;
    case 24: hipLaunchKernelGGL(nsr::k_probe_b3<24>, g1, b, lepi, s, wstream, out, iters); break;
    case 25: hipLaunchKernelGGL(nsr::k_probe_b3<25>, g1, b, lepi, s, wstream, out, iters); break;
    case 26: hipLaunchKernelGGL(nsr::k_probe_b3<26>, g1, b, lepi, s, wstream, out, iters); break;
    case 27: hipLaunchKernelGGL(nsr::k_probe_b3<27>, g1, b, lepi, s, wstream, out, iters); break;
    case 28: hipLaunchKernelGGL(nsr::k_probe_h2<28>, g1, b, lepi, s, wstream, out, iters); break;
    case 29: hipLaunchKernelGGL(nsr::k_probe_h2<29>, g1, b, lepi, s, wstream, out, iters); break;
    case 30: hipLaunchKernelGGL(nsr::k_probe_h2<30>, g1, b, lepi, s, wstream, out, iters); break;
  }
  NSRP_HIP(hipGetLastError());
  NSRP_HIP(hipEventRecord(ev1, s));
  NSRP_HIP(hipEventSynchronize(ev1));
  NSRP_HIP(hipEventElapsedTime(ms, ev0, ev1));
  int rc = 0;
  if (mode >= 11 && getenv("NSR_PROBE_VERBOSE")) {     // shader cycles and 100 MHz ticks of workgroup 0's loop
    float cw[2];
    NSRP_HIP(hipMemcpy(cw, out + (size_t)n_cu * 256, sizeof(cw), hipMemcpyDeviceToHost));
    fprintf(stderr, "nsr_probe mode %d: %.0f shader cycles in %.0f ticks of 10 ns -> %.3f GHz, %.2f cycles per MFMA\n", mode,
            cw[0], cw[1], cw[0] / cw[1] / 10.0, cw[0] / ((double)iters * (mode >= 28 ? 384.0 : 768.0)));
  }
  if (mode >= 4 && mode <= 8) {        // mean duration of the GEMM workgroups (100 MHz ticks -> ms), not the whole kernel
    std::vector<float> host(2 * n_cu);
    NSRP_HIP(hipMemcpy(host.data(), out, sizeof(float) * host.size(), hipMemcpyDeviceToHost));
    double sum = 0.0, ps = 0.0;
    int n = 0, pn = 0;
    for (float v : host) {
      if (v > 0.0f) { sum += v; ++n; }
      if (v < 0.0f) { ps -= v; ++pn; }
    }
    *ms = n ? (float)(sum / n * 1e-5) : 0.0f;
    if (pn && getenv("NSR_PROBE_VERBOSE"))
      fprintf(stderr, "nsr_probe mode %d: partner workgroups ran %.4f loop iterations per 10 ns tick\n", mode, ps / pn);
    if (n != n_cu) rc = pfail("nsr_probe: the dispatcher did not place one first workgroup per CU");
  }
  hipFree(wstream); hipFree(out); hipFree(done);
  hipEventDestroy(ev0); hipEventDestroy(ev1);
  return rc;
}

}  // extern "C"
