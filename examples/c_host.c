/* c_host.c -- a torch-free, Python-free host of libnsr.so: plain C, the HIP runtime C API and include/nsr.h.
 *
 * Shows that the drop-in boundary really is a C ABI (plain pointers and sizes): a C program creates a handle, uploads
 * the packed networks, renders one view into hipMalloc'd buffers and writes rgb_map to a file.  The GPU test
 * tests/test_gpu_parity.py::test_c_host_matches_python_engine compiles it with gcc, runs it and compares the image
 * with the one the Python engine renders from the same inputs, bit for bit.
 *
 *   gcc -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/c_host.c \
 *       -Lneural_sim_nerf_amd/csrc -lnsr -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/neural_sim_nerf_amd/csrc -o c_host
 *   ./c_host inputs.bin out.bin H W [fp32|f16x2]
 *
 * inputs.bin (float32, little endian): packed32 coarse | packed32 fine | packed16 coarse | packed16 fine | packed f16x2
 * coarse | packed f16x2 fine (each NSR_PACKED_FLOATS, from pack.py) | t_coarse[64] | u_fine[128] | c2w[12] | K[9] | near | far
 * The 5th argument picks the forward kernel: fp32 (k_render16p, the default here) or f16x2 (NSR_FLAG_MLP_F16X2, k_render_h2:
 * the Python engine's default).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime_api.h>
#include "nsr.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_NSR(x) do { if ((x) != 0) { fprintf(stderr, "%s: %s\n", #x, nsr_last_error()); return 3; } } while (0)

int main(int argc, char** argv) {
  if (argc != 5 && argc != 6) { fprintf(stderr, "usage: %s inputs.bin out.bin H W [fp32|f16x2]\n", argv[0]); return 1; }
  const int H = atoi(argv[3]), W = atoi(argv[4]);
  const int h2 = argc == 6 && strcmp(argv[5], "f16x2") == 0;
  const size_t n_in = 6 * (size_t)NSR_PACKED_FLOATS + 64 + 128 + 12 + 9 + 2;
  float* in = (float*)malloc(n_in * sizeof(float));
  FILE* f = fopen(argv[1], "rb");
  if (!f || fread(in, sizeof(float), n_in, f) != n_in) { fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
  fclose(f);
  const float* p32c = in;
  const float* p32f = p32c + NSR_PACKED_FLOATS;
  const float* p16c = p32f + NSR_PACKED_FLOATS;
  const float* p16f = p16c + NSR_PACKED_FLOATS;
  const float* ph2c = p16f + NSR_PACKED_FLOATS;
  const float* ph2f = ph2c + NSR_PACKED_FLOATS;
  const float* t64 = ph2f + NSR_PACKED_FLOATS;
  const float* u128 = t64 + 64;
  const float* c2w = u128 + 128;
  const float* Kf = c2w + 12;
  const float near_ = Kf[9], far_ = Kf[10];
  double K9[9];
  for (int i = 0; i < 9; ++i) K9[i] = (double)Kf[i];

  NsrConfig cfg = {NSR_ABI_VERSION, 0, NSR_N_SAMPLES, NSR_N_IMPORTANCE, 0, 0,
                   NSR_FLAG_SCHED_PHASES | (h2 ? NSR_FLAG_MLP_F16X2 : 0), 0};
  nsr_handle h = NULL;
  CHECK_NSR(nsr_create(&cfg, &h));
  CHECK_NSR(nsr_upload_weights(h, 0, p32c, NSR_PACKED_FLOATS));
  CHECK_NSR(nsr_upload_weights(h, 1, p32f, NSR_PACKED_FLOATS));
  CHECK_NSR(nsr_upload_weights16(h, 0, p16c, NSR_PACKED_FLOATS));
  CHECK_NSR(nsr_upload_weights16(h, 1, p16f, NSR_PACKED_FLOATS));
  if (h2) {
    CHECK_NSR(nsr_upload_weights_h2(h, 0, ph2c, NSR_PACKED_FLOATS));
    CHECK_NSR(nsr_upload_weights_h2(h, 1, ph2f, NSR_PACKED_FLOATS));
  }
  CHECK_NSR(nsr_upload_tables(h, t64, 64, u128, 128));
  CHECK_NSR(nsr_selftest(h, NULL));

  const size_t n = (size_t)H * W;
  float *d_c2w, *d_out;                       /* one allocation: rgb 3n | disp n | acc n | rgb0 3n | disp0 n | acc0 n | z_std n */
  CHECK_HIP(hipMalloc((void**)&d_c2w, 12 * sizeof(float)));
  CHECK_HIP(hipMalloc((void**)&d_out, 11 * n * sizeof(float)));
  CHECK_HIP(hipMemcpy(d_c2w, c2w, 12 * sizeof(float), hipMemcpyHostToDevice));
  NsrRenderOut out = {d_out, d_out + 3 * n, d_out + 4 * n, d_out + 5 * n, d_out + 8 * n, d_out + 9 * n, d_out + 10 * n};
  hipStream_t stream;
  CHECK_HIP(hipStreamCreate(&stream));
  CHECK_NSR(nsr_render_views(h, d_c2w, 1, H, W, K9, near_, far_, &out, NULL, stream));
  float ms = 0.0f;
  CHECK_NSR(nsr_last_kernel_ms(h, &ms));
  float* host = (float*)malloc(11 * n * sizeof(float));
  CHECK_HIP(hipMemcpy(host, d_out, 11 * n * sizeof(float), hipMemcpyDeviceToHost));
  f = fopen(argv[2], "wb");
  if (!f || fwrite(host, sizeof(float), 11 * n, f) != 11 * n) { fprintf(stderr, "cannot write %s\n", argv[2]); return 1; }
  fclose(f);
  printf("c_host: %dx%d view, kernel %.3f ms\n", H, W, ms);
  CHECK_NSR(nsr_destroy(h));
  hipFree(d_c2w); hipFree(d_out); hipStreamDestroy(stream);
  free(in); free(host);
  return 0;
}
