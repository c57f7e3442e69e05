// probe_f16_mfma.hip -- what v_mfma_f32_32x32x16_f16 does with fp16 subnormal inputs and how exact its 16-term sums are.
// Diagnostic for the two-piece fp16 split (csrc/nsr_h2.inc): build with
//   hipcc --offload-arch=gfx950 -O2 tools/probe_f16_mfma.hip -o tools/build/probe_f16_mfma
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ void k(const _Float16* A /*[32][16]*/, const _Float16* B /*[16][32]*/, float* C /*[32][32]*/) {
  const int l = threadIdx.x, i = l & 31, h = l >> 5;
  h8 a, b;
  for (int s = 0; s < 8; ++s) { a[s] = A[i * 16 + 8 * h + s]; b[s] = B[(8 * h + s) * 32 + i]; }
  f16v c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = c[r];
}

int main() {
  std::vector<_Float16> A(512), B(512);
  std::vector<float> C(1024);
  _Float16 *dA, *dB; float* dC;
  hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dC, 4096);
  auto run = [&]() {
    hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
  };
  // 1. subnormal A (2^-24 * (i+1)) times B = 2^(j/2): one non-zero term per output
  for (auto& v : A) v = (_Float16)0.0f;
  for (auto& v : B) v = (_Float16)0.0f;
  for (int i = 0; i < 32; ++i) A[i * 16 + 3] = (_Float16)ldexpf((float)(i + 1), -24);
  for (int j = 0; j < 32; ++j) B[3 * 32 + j] = (_Float16)ldexpf(1.0f, j / 2 - 2);
  run();
  int bad = 0, zero = 0;
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
    const double want = (double)(float)A[i * 16 + 3] * (double)(float)B[3 * 32 + j];
    if (C[i * 32 + j] == 0.0f && want != 0.0) ++zero;
    if ((double)C[i * 32 + j] != want) ++bad;
  }
  printf("subnormal A x normal B: %d / 1024 wrong, %d flushed to zero  -> A denormals %s\n", bad, zero, zero ? "FLUSHED" : "honoured");
  // 2. subnormal B
  for (auto& v : A) v = (_Float16)0.0f;
  for (auto& v : B) v = (_Float16)0.0f;
  for (int i = 0; i < 32; ++i) A[i * 16 + 9] = (_Float16)ldexpf(1.0f, i / 2 - 2);
  for (int j = 0; j < 32; ++j) B[9 * 32 + j] = (_Float16)ldexpf((float)(j + 1), -24);
  run();
  bad = zero = 0;
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
    const double want = (double)(float)A[i * 16 + 9] * (double)(float)B[9 * 32 + j];
    if (C[i * 32 + j] == 0.0f && want != 0.0) ++zero;
    if ((double)C[i * 32 + j] != want) ++bad;
  }
  printf("normal A x subnormal B: %d / 1024 wrong, %d flushed to zero  -> B denormals %s\n", bad, zero, zero ? "FLUSHED" : "honoured");
  // 3. tiny products (fp32-subnormal results): 2^-14 * 2^-24 ... informational
  // 4. accuracy of 16-term sums on random normal data, wide dynamic range, vs exact
  srand(1);
  double worst = 0, worst_rel_sum = 0;
  for (int rep = 0; rep < 50; ++rep) {
    for (auto& v : A) v = (_Float16)(((rand() % 2001) - 1000) / 1000.0f * ldexpf(1.0f, rand() % 12 - 6));
    for (auto& v : B) v = (_Float16)(((rand() % 2001) - 1000) / 1000.0f * ldexpf(1.0f, rand() % 12 - 6));
    run();
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
      double want = 0, mag = 0;
      for (int kk = 0; kk < 16; ++kk) { const double p = (double)(float)A[i * 16 + kk] * (double)(float)B[kk * 32 + j]; want += p; mag += fabs(p); }
      const double e = fabs((double)C[i * 32 + j] - want);
      if (e / mag > worst) worst = e / mag;
      if (want != 0 && e / fabs(want) > worst_rel_sum) worst_rel_sum = e / fabs(want);
    }
  }
  printf("16-term sums: worst |err| / sum|terms| = %.3e (2^-24 = 5.96e-8), worst |err| / |sum| = %.3e\n", worst, worst_rel_sum);
  return 0;
}
