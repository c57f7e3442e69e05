/* c_host_wide.c -- a torch-free, Python-free host of the LAYERED renderer in libnsr.so: plain C, the HIP runtime C API and
 * include/nsr_wide.h.
 *
 * A C program describes two NeRFs of any shape (RH:70-97), uploads their parameters in the modules' own layout, asks how much
 * workspace a chunk of rays needs, allocates it with hipMalloc and renders given rays forward and with the input gradient
 * (RN:168-178).  tests/test_gpu_wide.py::test_c_host_of_the_layered_renderer compiles it with gcc, runs it and compares every
 * output with what the Python mirror (wide.WideModel) gets from the same inputs, bit for bit.
 *
 *   gcc -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/c_host_wide.c \
 *       -Lneural_sim_nerf_amd/csrc -lnsr -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/neural_sim_nerf_amd/csrc -o c_host_wide
 *   ./c_host_wide inputs.bin out.bin
 *
 * inputs.bin: int32 header[32] = D, W, multires, multires_views, use_viewdirs, output_ch, n_skips, skips[16], N_samples,
 * N_importance, n_rays, workspace_rays (rays per chunk the workspace is sized for), 0...; then float32: parameters of the coarse
 * network | of the fine network (nsrw_network_floats each) | t[N_samples] | u[N_importance] | rays_o[n,3] | rays_d[n,3] |
 * cotangent[n,3] | near | far.
 * out.bin (float32): rgb 3n | disp n | acc n | rgb0 3n | disp0 n | acc0 n | z_std n | grad_o 3n | grad_d 3n.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime_api.h>
#include "nsr_wide.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_NSRW(x) do { if ((x) != 0) { fprintf(stderr, "%s: %s\n", #x, nsrw_last_error()); return 3; } } while (0)

int main(int argc, char** argv) {
  if (argc != 3) { fprintf(stderr, "usage: %s inputs.bin out.bin\n", argv[0]); return 1; }
  FILE* f = fopen(argv[1], "rb");
  int32_t hd[32];
  if (!f || fread(hd, sizeof(int32_t), 32, f) != 32) { fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
  NsrwNet net;
  memset(&net, 0, sizeof net);
  net.D = hd[0]; net.W = hd[1]; net.multires = hd[2]; net.multires_views = hd[3]; net.use_viewdirs = hd[4];
  net.output_ch = hd[5]; net.n_skips = hd[6];
  for (int i = 0; i < NSRW_MAX_SKIPS; ++i) net.skips[i] = hd[7 + i];
  const int ns = hd[23], ni = hd[24];
  const size_t n = (size_t)hd[25];
  const int64_t ws_rays = hd[26];
  const int flags = hd[27];                 /* NSRW_FLAG_* (0 = fp32 MFMAs, NSRW_FLAG_MLP_BF16X3 / NSRW_FLAG_MLP_F16X2 = the split-precision GEMMs) */
  const size_t nw = nsrw_network_floats(&net);
  if (nw == 0) { fprintf(stderr, "network description: %s\n", nsrw_last_error()); return 1; }
  const size_t n_in = 2 * nw + ns + ni + 9 * n + 2;
  float* in = (float*)malloc(n_in * sizeof(float));
  if (fread(in, sizeof(float), n_in, f) != n_in) { fprintf(stderr, "short read of %s\n", argv[1]); return 1; }
  fclose(f);
  const float* w_c = in;
  const float* w_f = w_c + nw;
  const float* t = w_f + nw;
  const float* u = t + ns;
  const float* ro = u + ni;
  const float* rd = ro + 3 * n;
  const float* cot = rd + 3 * n;
  const float near_ = cot[3 * n], far_ = cot[3 * n + 1];

  NsrwConfig cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.device = 0; cfg.n_samples = ns; cfg.n_importance = ni; cfg.flags = flags;
  nsrw_handle h = NULL;
  CHECK_NSRW(nsrw_create(&cfg, &h));
  CHECK_NSRW(nsrw_upload_network(h, 0, &net, w_c, nw));
  CHECK_NSRW(nsrw_upload_network(h, 1, &net, w_f, nw));
  CHECK_NSRW(nsrw_upload_tables(h, t, ns, u, ni));
  size_t ws_bytes = 0;
  CHECK_NSRW(nsrw_workspace_bytes(h, ws_rays, 1, &ws_bytes));          /* with_grad = 1 covers the forward call too */

  float *d_in, *d_out;
  void* d_ws;
  CHECK_HIP(hipMalloc((void**)&d_in, 9 * n * sizeof(float)));
  CHECK_HIP(hipMalloc((void**)&d_out, 17 * n * sizeof(float)));
  CHECK_HIP(hipMalloc(&d_ws, ws_bytes));
  CHECK_HIP(hipMemcpy(d_in, ro, 9 * n * sizeof(float), hipMemcpyHostToDevice));
  const float *d_ro = d_in, *d_rd = d_in + 3 * n, *d_cot = d_in + 6 * n;
  NsrwOut out;
  memset(&out, 0, sizeof out);
  out.d_rgb = d_out; out.d_disp = d_out + 3 * n; out.d_acc = d_out + 4 * n; out.d_rgb0 = d_out + 5 * n;
  out.d_disp0 = d_out + 8 * n; out.d_acc0 = d_out + 9 * n; out.d_z_std = d_out + 10 * n;
  hipStream_t stream;
  CHECK_HIP(hipStreamCreate(&stream));
  CHECK_NSRW(nsrw_render_rays(h, d_ro, d_rd, (int64_t)n, near_, far_, NULL, &out, d_ws, ws_bytes, stream));
  float ms_f = 0.0f, ms_g = 0.0f;
  int chunks_f = 0, chunks_g = 0;
  CHECK_NSRW(nsrw_last_ms(h, &ms_f, &chunks_f));
  CHECK_NSRW(nsrw_render_rays_vjp(h, d_ro, d_rd, (int64_t)n, near_, far_, NULL, d_cot, NULL, d_out + 11 * n, d_out + 14 * n, NULL,
                                  d_ws, ws_bytes, stream));
  CHECK_NSRW(nsrw_last_ms(h, &ms_g, &chunks_g));
  float* host = (float*)malloc(17 * n * sizeof(float));
  CHECK_HIP(hipMemcpy(host, d_out, 17 * n * sizeof(float), hipMemcpyDeviceToHost));
  f = fopen(argv[2], "wb");
  if (!f || fwrite(host, sizeof(float), 17 * n, f) != 17 * n) { fprintf(stderr, "cannot write %s\n", argv[2]); return 1; }
  fclose(f);
  printf("c_host_wide: %zu rays of a %d x %d network, %d + %d samples: forward %.3f ms in %d chunk(s), with gradient %.3f ms in %d\n",
         n, net.D, net.W, ns, ni, ms_f, chunks_f, ms_g, chunks_g);
  CHECK_NSRW(nsrw_destroy(h));
  hipFree(d_in); hipFree(d_out); hipFree(d_ws); hipStreamDestroy(stream);
  free(in); free(host);
  return 0;
}
