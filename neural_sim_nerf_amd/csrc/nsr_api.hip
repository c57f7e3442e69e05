// nsr_api.hip -- C ABI (include/nsr.h) over the kernels in nsr_kernels.hip.  Host side only: argument
// checking, weight/table residency, launches, HIP-event timing.  No torch types.
// Launch entry points (nsr_render_*, nsr_pose_grad, nsr_to8b, nsr_find_bbox, stage kernels) only enqueue work on the
// caller's stream: every scratch buffer they need is allocated by the setup calls (nsr_create, nsr_upload_weights_bwd,
// nsr_reserve_bbox), they read no environment variables, they leave the calling thread's current device as they found
// it, and they can be captured into a hipGraph.
#define NSR_UNIT_F32 1          // this unit: the host API, the stage kernels, the x32 fp32 kernels (nsr_kernels.hip: "Translation units")
#include "nsr_kernels.hip"
extern "C" int nsr_unit_bounds_h2(unsigned* line);
extern "C" int nsr_unit_bounds_b3(unsigned* line);
extern "C" int nsr_unit_bounds_x16(unsigned* line);
#include "nsr_handoff.hip"
#include "nsr_pose.hip"

#include <string>
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/nsr.h"

static_assert(NSR_SLAB_FLOATS == nsr::kSlabFloats, "header/kernels out of sync");
static_assert(NSR_STREAM_SLABS == nsr::kStreamSlabs, "header/kernels out of sync");
static_assert(NSR_AUX_FLOATS == nsr::kAuxFloats, "header/kernels out of sync");
static_assert(NSR_STREAM_SLABS_B3 == nsr::kStreamSlabsB3, "header/kernels out of sync");
static_assert(NSR_STREAM_SLABS_B3_BWD == nsr::kStreamSlabsB3Bwd, "header/kernels out of sync");
static_assert(NSR_STREAM_SLABS_H2_BWD == nsr::kStreamSlabsH2Bwd, "header/kernels out of sync");

namespace {

thread_local std::string g_err;

int fail(const std::string& m) {
  g_err = m;
  return 1;
}

#define NSR_HIP(expr)                                                                       \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess) return fail(std::string(#expr) + ": " + hipGetErrorString(e_));   \
  } while (0)

// current device := `dev` for the lifetime of the guard; restored on exit (libnsr shares the caller's HIP runtime, so
// a bare hipSetDevice would silently move e.g. PyTorch's current device)
struct DeviceGuard {
  int prev = -1, dev;
  hipError_t err = hipSuccess;
  explicit DeviceGuard(int d) : dev(d) {
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != dev) err = hipSetDevice(dev);
  }
  ~DeviceGuard() { if (prev >= 0 && prev != dev) (void)hipSetDevice(prev); }
};
#define NSR_DEVICE(h) DeviceGuard guard_((h)->cfg.device); NSR_HIP(guard_.err)

constexpr unsigned kOvfCap = 1u << 17;      // items (2 rays each) the safety net's list holds at first: 1 MiB per handle; every launch
                                            // call grows it to its own size first (ensure_range), so no item is ever dropped
constexpr long long kMaxRaysH2 = 2 * 0xFFFFFFFFll;   // f16x2 handles: the launch's item count is a 32-bit device word
static int kSuperLg = 12;              // k_render16p: 4096 rays per super-chunk (8 rounds of the 512-workgroup grid)

constexpr size_t kRenderLds = nsr::kLdsState + sizeof(nsr::ItemState);
constexpr size_t kRenderLdsBig = nsr::kLdsState + sizeof(nsr::ItemStateBig);      // N_samples = 128 (r05)
static_assert(kRenderLdsBig <= 163840, "the N_samples = 128 item state fits the 160 KiB LDS next to the ring and the aux blocks");
constexpr size_t kRender16Lds = nsr::kLds16State + sizeof(nsr::ItemState16);
constexpr size_t kVjp16Lds = nsr::kLds16State + sizeof(nsr::ItemStateV16);
constexpr size_t kNetLds = nsr::kLdsAux + nsr::kAuxFloats * 4;
// bf16x3 images in d_nets_b3: coarse | fine | fine transposed, kB3Stride floats apart (the transposed stream is the longest)
constexpr size_t kB3Stride = (size_t)NSR_STREAM_SLABS_B3_BWD * NSR_SLAB_FLOATS + NSR_AUX_FLOATS;
static_assert(kB3Stride >= (size_t)NSR_PACKED_B3_FLOATS, "stride covers the forward images");
// f16x2 images in d_nets_h2: coarse | fine | fine transposed, kH2Stride floats apart (the transposed stream is one slab longer)
constexpr size_t kH2Stride = (size_t)NSR_STREAM_SLABS_H2_BWD * NSR_SLAB_FLOATS + NSR_AUX_FLOATS;
static_assert(kH2Stride >= (size_t)NSR_PACKED_FLOATS, "stride covers the forward images");


// The sample counts the x32-structured kernels are instantiated for (RN:439 N_samples, RN:474 N_importance).  N_samples = 64:
// N_importance 128 on every handle; 0 (coarse only); 96 / 64 / 32 on f16x2 handles.  r05: N_samples = 32 with N_importance 64 (or
// 0) and N_samples = 128 with N_importance 128 (or 0), f16x2 handles only.
struct Counts { int ns, ni; };
typedef void (*RenderKernel)(const nsr::RenderArgs*);
typedef void (*VjpKernel)(const nsr::VjpArgs*);
struct KernelSet { RenderKernel h2, b3; VjpKernel vjp_h2, vjp_b3; };
inline bool special_counts(int ns, int ni) { return ns != NSR_N_SAMPLES || (ni != NSR_N_IMPORTANCE && ni != 0); }
// kernels of an f16x2 handle with `special_counts`: the coarse-only form of N_samples 32 / 128 runs the kernel of its fine partner
inline KernelSet special_kernels(int ns, int ni) {
  if (ns == 32) return {nsr::k_render_h2_c32_n64, nsr::k_render_b3_c32_n64, nsr::k_render_vjp_h2_c32_n64, nsr::k_render_vjp_b3_c32_n64};
  if (ns == 128) return {nsr::k_render_h2_c128_n128, nsr::k_render_b3_c128_n128, nsr::k_render_vjp_h2_c128_n128, nsr::k_render_vjp_b3_c128_n128};
  if (ni == 96) return {nsr::k_render_h2_n96, nsr::k_render_b3_n96, nsr::k_render_vjp_h2_n96, nsr::k_render_vjp_b3_n96};
  if (ni == 64) return {nsr::k_render_h2_n64, nsr::k_render_b3_n64, nsr::k_render_vjp_h2_n64, nsr::k_render_vjp_b3_n64};
  return {nsr::k_render_h2_n32, nsr::k_render_b3_n32, nsr::k_render_vjp_h2_n32, nsr::k_render_vjp_b3_n32};
}

}  // namespace

struct nsr_handle_s {
  NsrConfig cfg;
  int n_cu = 0;
  int chunk = 1;                                        // k_render16: rays per chunk of the work queue
  float* d_nets = nullptr;
  float* d_packed[3] = {nullptr, nullptr, nullptr};   // views into d_nets: coarse, fine, fine transposed
  bool have_net[3] = {false, false, false};
  float* d_nets16 = nullptr;                            // coarse | fine | fine transposed in the x16 layout
  bool have_net16[3] = {false, false, false};
  float* d_nets_b3 = nullptr;                           // coarse | fine | fine transposed in the bf16x3 layout
  bool have_net_b3[3] = {false, false, false};          // (NSR_FLAG_MLP_BF16X3)
  float* d_nets_h2 = nullptr;                           // coarse | fine | fine transposed in the f16x2 layout, kH2Stride apart
  bool have_net_h2[3] = {false, false, false};          // (NSR_FLAG_MLP_F16X2)
  float* d_tables = nullptr;  // [64] + [128]
  bool have_tables = false;
  float* d_scratch = nullptr;  // selftest
  nsr::RenderArgs* d_args = nullptr;  // kernel argument block (device), written stream-ordered by k_set_args
  nsr::VjpArgs* d_vjp_args = nullptr;
  uint4* d_mask_scratch = nullptr;    // relu patterns of the fine forward passes: x32 [n_cu][3][9][256] uint4 = x16
  int mask_grid = 0;                  // [2 n_cu][3][9][256] uint2 (same bytes); allocated by nsr_upload_weights_bwd*
  unsigned long long* d_work_counter = nullptr;  // work-queue heads: [0] the launch, [1] its fp32 fallback launch (f16x2)
  nsr::RenderArgs* d_args_fb = nullptr;          // f16x2 range safety net: argument blocks of the fallback launches,
  nsr::VjpArgs* d_vjp_args_fb = nullptr;
  unsigned long long* d_ovf_items = nullptr;     // ... the items (2 rays) the f16x2 kernel reported, [ovf_cap]
  unsigned ovf_cap = 0;
  std::vector<void*> retired;                    // ... lists the handle has outgrown: a launch still in flight, or a hipGraph
                                                 // captured earlier, holds their address in its argument block (by value),
                                                 // so they live as long as the handle does
  unsigned* d_ovf_stat = nullptr;                // ... [0] items of the last launch, [1] points, [2] rays, [3] items beyond the cap
  float* d_zf_scratch = nullptr;      // k_render16: sorted fine z values between the two phases of a chunk, [2 n_cu][chunk][192]
  int zf_grid = 0;
  unsigned* d_sched_flags = nullptr;  // global phases: ready / taken generations of the 3 * 2^kSuperLg hand-off slots
  unsigned* d_status = nullptr;       // global phases: rays whose fine task recomputed its coarse pass (nsr_schedule_stats)
  int* d_box_scratch = nullptr;       // nsr_find_bbox: parent + stats of one batch of images (nsr_reserve_bbox)
  size_t box_scratch_ints = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;   // kernel timing (eager launches only)
  hipEvent_t ev_busy = nullptr;              // completion of the last launch that used the per-handle scratch
  hipStream_t last_stream = nullptr;
  bool launched = false;
  bool timed = false;
};

// One handle = one argument block, one work counter, one set of scratch buffers: launches on ONE stream are ordered
// by the stream; a launch on a different stream while the previous one is still running would race on them and
// is refused (include/nsr.h: one handle per (model, stream)).  Never blocks, never queried while capturing.
static int claim_stream(nsr_handle h, hipStream_t s, bool capturing) {
  if (h->launched && s != h->last_stream && !capturing) {
    const hipError_t q = hipEventQuery(h->ev_busy);
    if (q == hipErrorNotReady)
      return fail("handle is busy on another stream (one handle per (model, stream); use a second handle or order the "
                  "streams with an event)");
    if (q != hipSuccess) return fail(std::string("hipEventQuery: ") + hipGetErrorString(q));
  }
  h->last_stream = s;
  return 0;
}

static int stream_capturing(hipStream_t s, bool* capturing) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  NSR_HIP(hipStreamIsCapturing(s, &st));
  *capturing = st != hipStreamCaptureStatusNone;
  return 0;
}

// every device allocation, event and kernel attribute a handle needs for its launch calls (setup time only)
static int allocate_handle(nsr_handle h) {
  const NsrConfig* cfg = &h->cfg;
  // one allocation: coarse | fine | fine^T (backward stream), NSR_PACKED_FLOATS apart
  NSR_HIP(hipMalloc(&h->d_nets, sizeof(float) * 3 * NSR_PACKED_FLOATS));
  for (int i = 0; i < 3; ++i) h->d_packed[i] = h->d_nets + (size_t)i * NSR_PACKED_FLOATS;
  // d_nets16 (the x16 images) is allocated by the first nsr_upload_weights16 / _bwd16: handles whose kernels never read
  // it (f16x2 / bf16x3 handles run k_render_h2 / _b3, their fallback and the stage kernels read the x32 image) do not pay for it
  NSR_HIP(hipFuncSetAttribute((const void*)nsr::k_render_vjp16, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kVjp16Lds));
  NSR_HIP(hipFuncSetAttribute((const void*)nsr::k_render16, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRender16Lds));
  if (cfg->flags & NSR_FLAG_MLP_BF16X3) {
    NSR_HIP(hipMalloc(&h->d_nets_b3, sizeof(float) * 3 * kB3Stride));
    NSR_HIP(hipFuncSetAttribute((const void*)nsr::k_render_b3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRenderLds));
    NSR_HIP(hipFuncSetAttribute((const void*)nsr::k_render_vjp_b3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRenderLds));
  }
  if (cfg->flags & NSR_FLAG_MLP_F16X2) {
    NSR_HIP(hipMalloc(&h->d_nets_h2, sizeof(float) * 3 * kH2Stride));
    NSR_HIP(hipMalloc(&h->d_args_fb, sizeof(nsr::RenderArgs)));
    NSR_HIP(hipMalloc(&h->d_vjp_args_fb, sizeof(nsr::VjpArgs)));
    NSR_HIP(hipMalloc(&h->d_ovf_items, sizeof(unsigned long long) * kOvfCap));
    h->ovf_cap = kOvfCap;
    NSR_HIP(hipMalloc(&h->d_ovf_stat, 4 * sizeof(unsigned)));
    NSR_HIP(hipMemset(h->d_ovf_stat, 0, 4 * sizeof(unsigned)));
    NSR_HIP(hipFuncSetAttribute((const void*)nsr::k_render_h2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRenderLds));
    NSR_HIP(hipFuncSetAttribute((const void*)nsr::k_render_vjp_h2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRenderLds));
    // the safety net's fallback kernels: bf16x3 once nsr_upload_weights_b3 has been called on this handle
    NSR_HIP(hipFuncSetAttribute((const void*)nsr::k_render_b3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRenderLds));
    NSR_HIP(hipFuncSetAttribute((const void*)nsr::k_render_vjp_b3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRenderLds));
    if (special_counts(cfg->n_samples, cfg->n_importance)) {
      const KernelSet ks = special_kernels(cfg->n_samples, cfg->n_importance);
      const int lds = (int)(cfg->n_samples == 128 ? kRenderLdsBig : kRenderLds);
      for (const void* k : {(const void*)ks.h2, (const void*)ks.b3, (const void*)ks.vjp_h2, (const void*)ks.vjp_b3})
        NSR_HIP(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    }
  }
  NSR_HIP(hipMalloc(&h->d_tables, sizeof(float) * 256));      // [N_samples <= 128] + [128]
  NSR_HIP(hipMalloc(&h->d_scratch, sizeof(float) * 4096));
  NSR_HIP(hipMalloc(&h->d_args, sizeof(nsr::RenderArgs)));
  NSR_HIP(hipMalloc(&h->d_work_counter, 2 * sizeof(unsigned long long)));
  // k_render16's inter-phase scratch: bounded by the grid (2 workgroups per CU, or max_workgroups) x chunk
  h->zf_grid = cfg->max_workgroups > 0 ? cfg->max_workgroups : 2 * h->n_cu;
  size_t zf_rays = (size_t)h->zf_grid * h->chunk;
  if (cfg->flags & NSR_FLAG_SCHED_PHASES) {                // the z hand-off ring of the global-phases schedule
    if (const char* e = getenv("NSR_EXP_SUPER_LG")) {      // experiment knob (super-chunk size), read at setup only
      const int v = atoi(e);
      if (v >= 6 && v <= 20) kSuperLg = v;
    }
    zf_rays = (size_t)3 << kSuperLg;
    NSR_HIP(hipMalloc(&h->d_sched_flags, sizeof(unsigned) * 2 * zf_rays));
    NSR_HIP(hipMalloc(&h->d_status, 2 * sizeof(unsigned)));      // [0] recomputed rays, [1] launch epoch (device-side counter)
    NSR_HIP(hipMemset(h->d_status, 0, 2 * sizeof(unsigned)));
    NSR_HIP(hipFuncSetAttribute((const void*)nsr::k_render16p, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRender16Lds));
    NSR_HIP(hipFuncSetAttribute((const void*)nsr::k_render_vjp16p, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kVjp16Lds));
  }
  // (global phases: 8-byte {value, tag} granules, hence twice the floats)
  NSR_HIP(hipMalloc(&h->d_zf_scratch, sizeof(float) * 192 * zf_rays * ((cfg->flags & NSR_FLAG_SCHED_PHASES) ? 2 : 1)));
  NSR_HIP(hipEventCreateWithFlags(&h->ev_busy, hipEventDisableTiming));
  NSR_HIP(hipMalloc(&h->d_vjp_args, sizeof(nsr::VjpArgs)));
  NSR_HIP(hipFuncSetAttribute((const void*)nsr::k_render_vjp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRenderLds));
  NSR_HIP(hipEventCreate(&h->ev0));
  NSR_HIP(hipEventCreate(&h->ev1));
  NSR_HIP(hipFuncSetAttribute((const void*)nsr::k_render, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRenderLds));
  NSR_HIP(hipFuncSetAttribute((const void*)nsr::k_run_network, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kNetLds));
  return 0;
}

extern "C" {

const char* nsr_last_error(void) { return g_err.c_str(); }
int nsr_abi_version(void) { return NSR_ABI_VERSION; }

int nsr_create(const NsrConfig* cfg, nsr_handle* out) {
  if (!cfg || !out) return fail("nsr_create: null argument");
  if (cfg->abi_version != NSR_ABI_VERSION) return fail("nsr_create: ABI version mismatch");
  if ((cfg->flags & NSR_FLAG_MLP_BF16X3) && (cfg->flags & NSR_FLAG_MLP_F16X2))
    return fail("nsr_create: NSR_FLAG_MLP_BF16X3 and NSR_FLAG_MLP_F16X2 are mutually exclusive");
  if (cfg->flags & ~(NSR_FLAG_WHITE_BKGD | NSR_FLAG_LINDISP | NSR_FLAG_SCHED_PHASES | NSR_FLAG_MLP_BF16X3 | NSR_FLAG_MLP_F16X2))
    return fail("nsr_create: unknown bits in flags");
  {
    const int ns = cfg->n_samples, ni = cfg->n_importance;
    const bool h2cfg = (cfg->flags & NSR_FLAG_MLP_F16X2) != 0;
    const bool ok = (ns == NSR_N_SAMPLES && (ni == NSR_N_IMPORTANCE || ni == 0)) ||
                    (h2cfg && ns == NSR_N_SAMPLES && (ni == 96 || ni == 64 || ni == 32)) ||
                    (h2cfg && ns == 32 && (ni == 64 || ni == 0)) || (h2cfg && ns == 128 && (ni == 128 || ni == 0));
    if (!ok)
      return fail("nsr_create: unsupported (N_samples, N_importance): (64, 128) and (64, 0) on every handle; on NSR_FLAG_MLP_F16X2 "
                  "handles also (64, 96 / 64 / 32), (32, 64 / 0) and (128, 128 / 0) -- other handles render fewer importance samples "
                  "with N_importance = 128 and a uniforms table of repeated values (engine._host_tables)");
  }
  if ((cfg->flags & NSR_FLAG_SCHED_PHASES) && (cfg->variant == 32 || cfg->n_importance == 0))
    return fail("nsr_create: NSR_FLAG_SCHED_PHASES applies to the x16 coarse+fine forward kernel only");
  if (cfg->chunk < 0 || cfg->chunk > 256) return fail("nsr_create: chunk must be 0 (default) or 1..256");
  if (cfg->variant != 0 && cfg->variant != 16 && cfg->variant != 32)
    return fail("nsr_create: variant must be 0 (default), 16 or 32");
  int ndev = 0;
  NSR_HIP(hipGetDeviceCount(&ndev));
  if (cfg->device < 0 || cfg->device >= ndev) return fail("nsr_create: no such HIP device");
  DeviceGuard guard_(cfg->device); NSR_HIP(guard_.err);
  hipDeviceProp_t prop;
  NSR_HIP(hipGetDeviceProperties(&prop, cfg->device));
  if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
    return fail(std::string("nsr_create: this library is built for gfx950 (MI355X), device is ") + prop.gcnArchName);
  nsr_handle h = new nsr_handle_s();
  h->cfg = *cfg;
  h->n_cu = prop.multiProcessorCount;
  h->chunk = (cfg->chunk > 0 && !(cfg->flags & NSR_FLAG_SCHED_PHASES)) ? cfg->chunk : 1;
  if (int e = allocate_handle(h)) {          // nothing half-built escapes: free whatever was allocated
    const std::string msg = g_err;
    nsr_destroy(h);
    return fail(msg);
  }
  *out = h;
  return 0;
}

int nsr_destroy(nsr_handle h) {
  if (!h) return 0;
  DeviceGuard guard_(h->cfg.device);
  hipDeviceSynchronize();
  hipFree(h->d_nets);
  hipFree(h->d_nets16);
  hipFree(h->d_nets_b3);
  hipFree(h->d_nets_h2);
  hipFree(h->d_tables);
  hipFree(h->d_scratch);
  hipFree(h->d_args);
  hipFree(h->d_vjp_args);
  hipFree(h->d_mask_scratch);
  hipFree(h->d_box_scratch);
  hipFree(h->d_zf_scratch);
  hipFree(h->d_work_counter);
  hipFree(h->d_args_fb);
  hipFree(h->d_vjp_args_fb);
  hipFree(h->d_ovf_items);
  for (void* p : h->retired) hipFree(p);
  hipFree(h->d_ovf_stat);
  hipFree(h->d_sched_flags);
  hipFree(h->d_status);
  if (h->ev0) hipEventDestroy(h->ev0);
  if (h->ev1) hipEventDestroy(h->ev1);
  if (h->ev_busy) hipEventDestroy(h->ev_busy);
  delete h;
  return 0;
}

int nsr_upload_weights(nsr_handle h, int net_id, const float* packed, size_t n_floats) {
  if (!h || !packed) return fail("nsr_upload_weights: null argument");
  if (net_id < 0 || net_id > 1) return fail("nsr_upload_weights: net_id must be 0 (coarse) or 1 (fine)");
  if (n_floats != (size_t)NSR_PACKED_FLOATS) return fail("nsr_upload_weights: wrong packed size");
  NSR_DEVICE(h);
  NSR_HIP(hipMemcpy(h->d_packed[net_id], packed, sizeof(float) * n_floats, hipMemcpyHostToDevice));
  h->have_net[net_id] = true;
  return 0;
}

int nsr_upload_weights16(nsr_handle h, int net_id, const float* packed, size_t n_floats) {
  if (!h || !packed) return fail("nsr_upload_weights16: null argument");
  if (net_id < 0 || net_id > 1) return fail("nsr_upload_weights16: net_id must be 0 (coarse) or 1 (fine)");
  if (n_floats != (size_t)NSR_PACKED_FLOATS) return fail("nsr_upload_weights16: wrong packed size");
  NSR_DEVICE(h);
  if (!h->d_nets16) NSR_HIP(hipMalloc(&h->d_nets16, sizeof(float) * 3 * NSR_PACKED_FLOATS));      // setup call
  NSR_HIP(hipMemcpy(h->d_nets16 + (size_t)net_id * NSR_PACKED_FLOATS, packed, sizeof(float) * n_floats,
                    hipMemcpyHostToDevice));
  h->have_net16[net_id] = true;
  return 0;
}

int nsr_upload_weights_b3(nsr_handle h, int net_id, const float* packed, size_t n_floats) {
  if (!h || !packed) return fail("nsr_upload_weights_b3: null argument");
  if (!(h->cfg.flags & (NSR_FLAG_MLP_BF16X3 | NSR_FLAG_MLP_F16X2)))
    return fail("nsr_upload_weights_b3: the handle was created with neither NSR_FLAG_MLP_BF16X3 nor NSR_FLAG_MLP_F16X2");
  if (net_id < 0 || net_id > 1) return fail("nsr_upload_weights_b3: net_id must be 0 (coarse) or 1 (fine)");
  if (n_floats != (size_t)NSR_PACKED_B3_FLOATS) return fail("nsr_upload_weights_b3: wrong packed size");
  NSR_DEVICE(h);
  if (!h->d_nets_b3) NSR_HIP(hipMalloc(&h->d_nets_b3, sizeof(float) * 3 * kB3Stride));      // f16x2 handle: the fallback's images (setup call)
  NSR_HIP(hipMemcpy(h->d_nets_b3 + (size_t)net_id * kB3Stride, packed, sizeof(float) * n_floats, hipMemcpyHostToDevice));
  h->have_net_b3[net_id] = true;
  return 0;
}

static int alloc_mask_scratch(nsr_handle h) {
  if (h->d_mask_scratch) return 0;   // setup call: the VJP kernels' relu-pattern scratch, one block per workgroup
  h->mask_grid = h->cfg.max_workgroups > 0 ? h->cfg.max_workgroups : h->n_cu;     // in x32 workgroups (16 B entries)
  const int passes = h->cfg.n_samples == 128 ? 4 : 3;      // fine forward passes per item (render_vjp32_body: kMaskPasses)
  NSR_HIP(hipMalloc(&h->d_mask_scratch, sizeof(uint4) * (size_t)h->mask_grid * passes * 9 * 256));
  return 0;
}

int nsr_upload_weights_bwd(nsr_handle h, const float* stream, size_t n_floats) {
  if (!h || !stream) return fail("nsr_upload_weights_bwd: null argument");
  if (n_floats != (size_t)NSR_STREAM_SLABS * NSR_SLAB_FLOATS) return fail("nsr_upload_weights_bwd: wrong stream size");
  NSR_DEVICE(h);
  NSR_HIP(hipMemcpy(h->d_packed[2], stream, sizeof(float) * n_floats, hipMemcpyHostToDevice));
  if (int e = alloc_mask_scratch(h)) return e;
  h->have_net[2] = true;
  return 0;
}

int nsr_upload_weights_h2(nsr_handle h, int net_id, const float* packed, size_t n_floats) {
  if (!h || !packed) return fail("nsr_upload_weights_h2: null argument");
  if (!(h->cfg.flags & NSR_FLAG_MLP_F16X2)) return fail("nsr_upload_weights_h2: the handle was not created with NSR_FLAG_MLP_F16X2");
  if (net_id < 0 || net_id > 1) return fail("nsr_upload_weights_h2: net_id must be 0 (coarse) or 1 (fine)");
  if (n_floats != (size_t)NSR_PACKED_FLOATS) return fail("nsr_upload_weights_h2: wrong packed size");
  NSR_DEVICE(h);
  NSR_HIP(hipMemcpy(h->d_nets_h2 + (size_t)net_id * kH2Stride, packed, sizeof(float) * n_floats, hipMemcpyHostToDevice));
  h->have_net_h2[net_id] = true;
  return 0;
}

int nsr_upload_weights_bwd_h2(nsr_handle h, const float* stream, size_t n_floats) {
  if (!h || !stream) return fail("nsr_upload_weights_bwd_h2: null argument");
  if (!(h->cfg.flags & NSR_FLAG_MLP_F16X2)) return fail("nsr_upload_weights_bwd_h2: the handle was not created with NSR_FLAG_MLP_F16X2");
  if (n_floats != (size_t)NSR_STREAM_SLABS_H2_BWD * NSR_SLAB_FLOATS) return fail("nsr_upload_weights_bwd_h2: wrong stream size");
  NSR_DEVICE(h);
  NSR_HIP(hipMemcpy(h->d_nets_h2 + 2 * kH2Stride, stream, sizeof(float) * n_floats, hipMemcpyHostToDevice));
  if (int e = alloc_mask_scratch(h)) return e;
  h->have_net_h2[2] = true;
  return 0;
}

int nsr_upload_weights_bwd_b3(nsr_handle h, const float* stream, size_t n_floats) {
  if (!h || !stream) return fail("nsr_upload_weights_bwd_b3: null argument");
  if (!(h->cfg.flags & (NSR_FLAG_MLP_BF16X3 | NSR_FLAG_MLP_F16X2)))
    return fail("nsr_upload_weights_bwd_b3: the handle was created with neither NSR_FLAG_MLP_BF16X3 nor NSR_FLAG_MLP_F16X2");
  if (n_floats != (size_t)NSR_STREAM_SLABS_B3_BWD * NSR_SLAB_FLOATS) return fail("nsr_upload_weights_bwd_b3: wrong stream size");
  NSR_DEVICE(h);
  if (!h->d_nets_b3) NSR_HIP(hipMalloc(&h->d_nets_b3, sizeof(float) * 3 * kB3Stride));      // (setup call)
  NSR_HIP(hipMemcpy(h->d_nets_b3 + 2 * kB3Stride, stream, sizeof(float) * n_floats, hipMemcpyHostToDevice));
  if (int e = alloc_mask_scratch(h)) return e;
  h->have_net_b3[2] = true;
  return 0;
}

int nsr_upload_weights_bwd16(nsr_handle h, const float* stream, size_t n_floats) {
  if (!h || !stream) return fail("nsr_upload_weights_bwd16: null argument");
  if (n_floats != (size_t)NSR_STREAM_SLABS * NSR_SLAB_FLOATS) return fail("nsr_upload_weights_bwd16: wrong stream size");
  NSR_DEVICE(h);
  if (!h->d_nets16) NSR_HIP(hipMalloc(&h->d_nets16, sizeof(float) * 3 * NSR_PACKED_FLOATS));      // setup call
  NSR_HIP(hipMemcpy(h->d_nets16 + (size_t)2 * NSR_PACKED_FLOATS, stream, sizeof(float) * n_floats, hipMemcpyHostToDevice));
  if (int e = alloc_mask_scratch(h)) return e;
  h->have_net16[2] = true;
  return 0;
}

int nsr_upload_tables(nsr_handle h, const float* t_coarse, int n_coarse, const float* u_fine, int n_fine) {
  if (!h || !t_coarse || !u_fine) return fail("nsr_upload_tables: null argument");
  if (n_coarse != h->cfg.n_samples || n_fine != 128)
    return fail("nsr_upload_tables: tables must have N_samples (the handle's) and 128 entries");
  NSR_DEVICE(h);
  NSR_HIP(hipMemcpy(h->d_tables, t_coarse, sizeof(float) * n_coarse, hipMemcpyHostToDevice));
  NSR_HIP(hipMemcpy(h->d_tables + 128, u_fine, sizeof(float) * 128, hipMemcpyHostToDevice));
  h->have_tables = true;
  return 0;
}

static int check_ready(nsr_handle h, bool need_fine) {
  if (!h) return fail("null handle");
  if (!h->have_tables) return fail("tables not uploaded (nsr_upload_tables)");
  if (!h->have_net[0]) return fail("coarse network not uploaded (nsr_upload_weights net_id 0)");
  if (need_fine && !h->have_net[1]) return fail("fine network not uploaded (nsr_upload_weights net_id 1)");
  return 0;
}

static int grid_for(nsr_handle h, long long n_items) {
  long long g = h->cfg.max_workgroups > 0 ? h->cfg.max_workgroups : h->n_cu;
  if (g > n_items) g = n_items;
  return (int)(g < 1 ? 1 : g);
}

// library default (variant 0) = x16: measured 145.0 vs 141.5 TFLOP/s for x32 on one 400x400 view (tools/compare_variants.py)
static bool use_x16(nsr_handle h) { return h->cfg.variant != 32; }

// f16x2 range safety net: the list must hold every item of the launch (each may overflow), so that none is dropped.
// Grows geometrically; the outgrown list is RETIRED, never freed while the handle lives (see nsr_handle_s::retired), and
// nothing synchronises.  Not while the stream is capturing (no allocation inside a capture): a captured launch larger
// than the list keeps the present capacity, and the kernel stores NaN into the rays of the items it cannot list
// (range_poison) -- call nsr_reserve_range before capturing to avoid that.  The allocation itself runs in the thread's RELAXED
// capture mode (hipThreadExchangeStreamCaptureMode, as torch's caching allocator does): a capture in progress on ANOTHER stream
// or thread in global mode is not invalidated by it.  An allocation failure IS an error of the launch call.
static int ensure_range(nsr_handle h, long long n_rays, bool capturing) {
  if (!h->d_ovf_items) return 0;
  if (n_rays > kMaxRaysH2) return fail("f16x2 handles take at most 2^33 - 2 rays per launch");
  const unsigned long long items = (unsigned long long)(n_rays + 1) / 2;
  if (items <= h->ovf_cap || capturing) return 0;
  unsigned long long want = 2ull * h->ovf_cap;
  if (want < items) want = items;
  if (want > 0xFFFFFFFFull) want = 0xFFFFFFFFull;
  unsigned long long* grown = nullptr;
  hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
  (void)hipThreadExchangeStreamCaptureMode(&mode);
  hipError_t e = hipMalloc(&grown, sizeof(unsigned long long) * want);
  if (e != hipSuccess && want != items) {
    (void)hipGetLastError();
    want = items;
    e = hipMalloc(&grown, sizeof(unsigned long long) * want);
  }
  (void)hipThreadExchangeStreamCaptureMode(&mode);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail("out of device memory for the f16x2 range list of this launch (nsr_reserve_range)");
  }
  h->retired.push_back(h->d_ovf_items);
  h->d_ovf_items = grown;
  h->ovf_cap = (unsigned)want;
  return 0;
}

static int launch_render(nsr_handle h, nsr::RenderArgs& a, const NsrRenderOut* out, const NsrDebugOut* dbg,
                         void* stream) {
  const bool fine = h->cfg.n_importance > 0;
  const int ni = h->cfg.n_importance;                      // 64 / 32: the kernels specialised to that many importance samples
  const bool b3 = (h->cfg.flags & NSR_FLAG_MLP_BF16X3) != 0;
  const bool h2 = (h->cfg.flags & NSR_FLAG_MLP_F16X2) != 0;
  // the per-ray extras (NsrRayExtras) are read by the x32-structured kernels: an fp32 handle serves them with k_render
  // (its x32 stream is always uploaded, see check_ready) whatever its `variant`
  const bool extras = a.viewdirs || a.t_rand || a.u_rays || a.noise0 || a.noise1 || a.near_rays;
  const bool x16 = use_x16(h) && !b3 && !h2 && !extras;
  if (int e = check_ready(h, fine)) return e;
  if (h2 && (!h->have_net_h2[0] || (fine && !h->have_net_h2[1])))
    return fail("NSR_FLAG_MLP_F16X2 needs nsr_upload_weights_h2 for every network");
  if (x16 && (!h->have_net16[0] || (fine && !h->have_net16[1])))
    return fail("variant 16 needs nsr_upload_weights16 for every network");
  if (b3 && (!h->have_net_b3[0] || (fine && !h->have_net_b3[1])))
    return fail("NSR_FLAG_MLP_BF16X3 needs nsr_upload_weights_b3 for every network");
  if (!out || !out->d_rgb || !out->d_disp || !out->d_acc) return fail("render: rgb/disp/acc outputs are required");
  if (a.n_rays <= 0) return 0;
  NSR_DEVICE(h);
  float* nets = h2 ? h->d_nets_h2 : (b3 ? h->d_nets_b3 : (x16 ? h->d_nets16 : h->d_nets));
  const size_t net_floats = b3 ? kB3Stride : (h2 ? kH2Stride : (size_t)NSR_PACKED_FLOATS);
  const size_t stream_floats = (size_t)(b3 ? NSR_STREAM_SLABS_B3 : NSR_STREAM_SLABS) * NSR_SLAB_FLOATS;
  a.nets = nets;
  a.net_stride = (long long)sizeof(float) * (long long)net_floats;
  a.aux[0] = nets + stream_floats;
  a.aux[1] = nets + (fine ? net_floats : 0) + stream_floats;
  a.tcoarse = h->d_tables;
  a.ufine = h->d_tables + 128;
  a.fine = fine ? 1 : 0;
  a.white_bkgd = (h->cfg.flags & NSR_FLAG_WHITE_BKGD) ? 1 : 0;
  a.lindisp = (h->cfg.flags & NSR_FLAG_LINDISP) ? 1 : 0;
  a.rgb = out->d_rgb; a.disp = out->d_disp; a.acc = out->d_acc;
  a.rgb0 = out->d_rgb0; a.disp0 = out->d_disp0; a.acc0 = out->d_acc0; a.z_std = out->d_z_std;
  a.dbg_w0 = dbg ? dbg->d_weights0 : nullptr;
  a.dbg_zs = dbg ? dbg->d_z_samples : nullptr;
  a.dbg_zf = dbg ? dbg->d_z_fine : nullptr;
  a.dbg_raw0 = dbg ? dbg->d_raw0 : nullptr;
  a.dbg_raw = dbg ? dbg->d_raw : nullptr;
  a.dbg_inds = dbg ? (long long*)dbg->d_inds : nullptr;
  hipStream_t s = (hipStream_t)stream;
  bool capturing = false;
  if (int e = stream_capturing(s, &capturing)) return e;
  if (int e = claim_stream(h, s, capturing)) return e;
  if (h2) { if (int e = ensure_range(h, a.n_rays, capturing)) return e; }
  long long g = 0;
  const bool phases = x16 && fine && (h->cfg.flags & NSR_FLAG_SCHED_PHASES);
  if (phases) {                                            // global-phases schedule: k_render16p
    g = h->zf_grid;
    if (g > 2 * a.n_rays) g = 2 * a.n_rays;
    a.zf_scratch = h->d_zf_scratch;
    a.sched_flags = h->d_sched_flags;
    a.status = h->d_status;
    a.super_lg = kSuperLg;
    a.spin_max = h->cfg.chunk > 0 ? h->cfg.chunk - 1 : 64;   // looks at the ready flag before recomputing locally
    a.epoch_counter = h->d_status + 1;                     // k_set_args advances it in stream order (also under graph replay)
    a.chunk = 1;
    NSR_HIP(hipMemsetAsync(h->d_sched_flags, 0, sizeof(unsigned) * 2 * ((size_t)3 << kSuperLg), s));
  } else if (x16) {
    g = h->zf_grid;                                        // two workgroups per CU (or max_workgroups)
    // rays per chunk (see k_render16).  Larger chunks keep one network per L2 for longer (less fabric traffic), but
    // the chunk is also the granularity of the dynamic load balance between the unevenly progressing workgroups:
    // measured 1 -> 148.2, 2 -> 148.0, 4 -> 148.0, 8 -> 147.9, 16 -> 145.9 TFLOP/s.  Speed wins: the default is 1
    // (NsrConfig.chunk).
    int chunk = h->chunk;
    if ((long long)chunk * g > a.n_rays) chunk = (int)(a.n_rays / g > 1 ? a.n_rays / g : 1);   // small batches: keep every CU busy
    const long long n_chunks = (a.n_rays + chunk - 1) / chunk;
    if (g > n_chunks) g = n_chunks;
    a.zf_scratch = h->d_zf_scratch;
    a.chunk = chunk;
  } else {
    g = grid_for(h, (a.n_rays + 1) / 2);
  }
  a.work_counter = h->d_work_counter;
  // (the kernels specialised to other sample counts have a bf16x3 fallback only: without nsr_upload_weights_b3 every reported item is
  // dropped -- NaN outputs, counted)
  const int ns = h->cfg.n_samples;
  const bool special = special_counts(ns, ni);            // f16x2 handles only (nsr_create): kernels specialised to the counts
  const size_t lds32 = ns == 128 ? kRenderLdsBig : kRenderLds;
  const bool have_fb = h2 && ((h->have_net_b3[0] && (!fine || h->have_net_b3[1])) || !special);
  if (h2) { a.ovf_items = h->d_ovf_items; a.ovf_stat = h->d_ovf_stat; a.ovf_cap = have_fb ? h->ovf_cap : 0u; }
#ifdef NSR_EXP_SAMENET       // timing experiment: every pass streams the SAME weight image (L2-resident); results are wrong
  a.net_stride = 0;
#endif
  hipLaunchKernelGGL(nsr::k_set_args, dim3(1), dim3(1), 0, s, a, h->d_args);     // also zeroes the work counter
  if (!capturing) NSR_HIP(hipEventRecord(h->ev0, s));
  if (phases)
    hipLaunchKernelGGL(nsr::k_render16p, dim3((int)g), dim3(256), kRender16Lds, s, (const nsr::RenderArgs*)h->d_args);
  else if (x16)
    hipLaunchKernelGGL(nsr::k_render16, dim3((int)g), dim3(256), kRender16Lds, s, (const nsr::RenderArgs*)h->d_args);
  else if (b3)
    hipLaunchKernelGGL(nsr::k_render_b3, dim3((int)g), dim3(256), kRenderLds, s, (const nsr::RenderArgs*)h->d_args);
  else if (h2 && special)
    hipLaunchKernelGGL(special_kernels(ns, ni).h2, dim3((int)g), dim3(256), lds32, s, (const nsr::RenderArgs*)h->d_args);
  else if (h2)
    hipLaunchKernelGGL(nsr::k_render_h2, dim3((int)g), dim3(256), kRenderLds, s, (const nsr::RenderArgs*)h->d_args);
  else
    hipLaunchKernelGGL(nsr::k_render, dim3((int)g), dim3(256), kRenderLds, s, (const nsr::RenderArgs*)h->d_args);
#ifndef NSR_EXP_NO_RANGE
  if (have_fb) {
    // f16x2 range safety net: the items k_render_h2 reported (a NaN network output: a scaled activation beyond the fp16
    // range) are rendered again by the fp32 kernel of the same template, which overwrites their outputs.  The list and
    // its length stay on the device; with an empty list every workgroup of this launch returns at once.
    nsr::RenderArgs f = a;
    // ... on bf16 MFMAs (three-way split: fp32's exponent range, no failure domain, 1.7x the fp32-MFMA kernel) once the
    // handle holds the bf16x3 images (nsr_upload_weights_b3; the Python engine uploads them), else on fp32 MFMAs
    const bool fb3 = h->have_net_b3[0] && (!fine || h->have_net_b3[1]);
    float* fnets = fb3 ? h->d_nets_b3 : h->d_nets;
    const size_t fnet_floats = fb3 ? kB3Stride : (size_t)NSR_PACKED_FLOATS;
    const size_t fstream_floats = (size_t)(fb3 ? NSR_STREAM_SLABS_B3 : NSR_STREAM_SLABS) * NSR_SLAB_FLOATS;
    f.nets = fnets;
    f.net_stride = (long long)sizeof(float) * (long long)fnet_floats;
    f.aux[0] = fnets + fstream_floats;
    f.aux[1] = fnets + (fine ? fnet_floats : 0) + fstream_floats;
    f.ovf_items = nullptr; f.ovf_stat = nullptr; f.ovf_cap = 0;
    f.item_list = h->d_ovf_items; f.item_count = h->d_ovf_stat; f.item_cap = h->ovf_cap;
    f.work_counter = h->d_work_counter + 1;
    hipLaunchKernelGGL(nsr::k_set_args, dim3(1), dim3(1), 0, s, f, h->d_args_fb);
    RenderKernel fk = fb3 ? (special ? special_kernels(ns, ni).b3 : nsr::k_render_b3) : nsr::k_render;
    hipLaunchKernelGGL(fk, dim3((int)g), dim3(256), lds32, s, (const nsr::RenderArgs*)h->d_args_fb);
  }
#endif
  NSR_HIP(hipGetLastError());
  if (!capturing) {
    NSR_HIP(hipEventRecord(h->ev1, s));
    NSR_HIP(hipEventRecord(h->ev_busy, s));
  }
  h->launched = true;
  h->timed = !capturing;
  return 0;
}

static void set_extras(nsr::RenderArgs& a, const NsrRayExtras* ex) {
  if (!ex) return;
  a.viewdirs = ex->d_viewdirs; a.t_rand = ex->d_t_rand; a.u_rays = ex->d_u; a.noise0 = ex->d_noise0; a.noise1 = ex->d_noise1;
  a.near_rays = ex->d_near; a.far_rays = ex->d_far;
}

int nsr_render_rays_ex(nsr_handle h, const float* d_rays_o, const float* d_rays_d, int64_t n_rays, float near_,
                       float far_, const NsrRayExtras* ex, const NsrRenderOut* out, const NsrDebugOut* dbg, void* stream) {
  if (h && n_rays == 0) return 0;      // an empty batch is a valid no-op (buffers may be null)
  if (!d_rays_o || !d_rays_d) return fail("nsr_render_rays: null rays");
  if (n_rays < 0) return fail("nsr_render_rays: negative ray count");
  if (ex && ((ex->d_near == nullptr) != (ex->d_far == nullptr))) return fail("nsr_render_rays_ex: d_near and d_far come together");
  if (h && ex && h->cfg.n_importance == 0 && (ex->d_u || ex->d_noise1))
    return fail("nsr_render_rays_ex: d_u / d_noise1 belong to the fine pass (this handle is coarse only)");
  nsr::RenderArgs a;
  memset(&a, 0, sizeof(a));
  a.rays_o = d_rays_o; a.rays_d = d_rays_d; a.n_rays = n_rays; a.near_ = near_; a.far_ = far_; a.camera = 0;
  set_extras(a, ex);
  return launch_render(h, a, out, dbg, stream);
}

int nsr_render_rays(nsr_handle h, const float* d_rays_o, const float* d_rays_d, int64_t n_rays, float near_,
                    float far_, const NsrRenderOut* out, const NsrDebugOut* dbg, void* stream) {
  return nsr_render_rays_ex(h, d_rays_o, d_rays_d, n_rays, near_, far_, nullptr, out, dbg, stream);
}

int nsr_render_views(nsr_handle h, const float* d_c2w, int n_views, int H, int W, const double* K9, float near_,
                     float far_, const NsrRenderOut* out, const NsrDebugOut* dbg, void* stream) {
  if (h && n_views == 0) return 0;      // an empty batch is a valid no-op (buffers may be null)
  if (!d_c2w || !K9) return fail("nsr_render_views: null argument");
  if (n_views < 0 || H <= 0 || W <= 0) return fail("nsr_render_views: bad image geometry");
  nsr::RenderArgs a;
  memset(&a, 0, sizeof(a));
  a.c2w = d_c2w; a.H = H; a.W = W; a.n_rays = (long long)n_views * H * W; a.near_ = near_; a.far_ = far_;
  a.camera = 1;
  a.fx = (float)K9[0]; a.cx = (float)K9[2]; a.fy = (float)K9[4]; a.cy = (float)K9[5];   // RH:160 (fp32 tensor op)
  return launch_render(h, a, out, dbg, stream);
}

int nsr_render_rays_vjp(nsr_handle h, const float* d_rays_o, const float* d_rays_d, int64_t n_rays, float near_,
                        float far_, const float* d_grad_rgb, float* d_grad_o, float* d_grad_d,
                        const float* d_z_fine, const NsrRenderOut* out, void* stream) {
  return nsr_render_rays_vjp_ex(h, d_rays_o, d_rays_d, n_rays, near_, far_, nullptr, d_grad_rgb, d_grad_o, d_grad_d, nullptr,
                                d_z_fine, out, stream);
}

int nsr_render_rays_vjp_ex(nsr_handle h, const float* d_rays_o, const float* d_rays_d, int64_t n_rays, float near_,
                           float far_, const NsrRayExtras* ex, const float* d_grad_rgb, float* d_grad_o, float* d_grad_d,
                           float* d_grad_viewdirs, const float* d_z_fine, const NsrRenderOut* out, void* stream) {
  return nsr_render_rays_vjp_dbg(h, d_rays_o, d_rays_d, n_rays, near_, far_, ex, d_grad_rgb, d_grad_o, d_grad_d,
                                 d_grad_viewdirs, d_z_fine, out, nullptr, stream);
}

int nsr_render_rays_vjp_dbg(nsr_handle h, const float* d_rays_o, const float* d_rays_d, int64_t n_rays, float near_,
                            float far_, const NsrRayExtras* ex, const float* d_grad_rgb, float* d_grad_o, float* d_grad_d,
                            float* d_grad_viewdirs, const float* d_z_fine, const NsrRenderOut* out,
                            const NsrVjpDebugOut* dbg, void* stream) {
  if (h && n_rays == 0) return 0;      // an empty batch is a valid no-op (buffers may be null)
  if (!h) return fail("nsr_render_rays_vjp: null handle");
  if (h->cfg.n_importance == 0) return fail("nsr_render_rays_vjp: needs the coarse+fine configuration (N_importance > 0)");
  const int ni = h->cfg.n_importance;
  if (int e = check_ready(h, true)) return e;
  const bool b3 = (h->cfg.flags & NSR_FLAG_MLP_BF16X3) != 0;
  // an f16x2 handle runs its input gradients on fp16 MFMAs too once the transposed stream is there (nsr_upload_weights_bwd_h2);
  // without it the fp32 kernels of `variant` serve (they need their own uploads)
  const bool h2 = (h->cfg.flags & NSR_FLAG_MLP_F16X2) && h->have_net_h2[0] && h->have_net_h2[1] && h->have_net_h2[2];
  const int ns = h->cfg.n_samples;
  const bool special = special_counts(ns, ni);
  const size_t lds32 = ns == 128 ? kRenderLdsBig : kRenderLds;
  if (special && !h2)
    return fail("nsr_render_rays_vjp: a handle of these sample counts needs nsr_upload_weights_bwd_h2 (only the f16x2 kernels are specialised to them)");
  const bool extras = ex && (ex->d_viewdirs || ex->d_t_rand || ex->d_u || ex->d_noise0 || ex->d_noise1 || ex->d_near);
  if (ex && ((ex->d_near == nullptr) != (ex->d_far == nullptr))) return fail("nsr_render_rays_vjp_ex: d_near and d_far come together");
  if (d_grad_viewdirs && !(ex && ex->d_viewdirs))
    return fail("nsr_render_rays_vjp_ex: d_grad_viewdirs without d_viewdirs (the view directions are rays_d / |rays_d| then, "
                "and their gradient is part of d_grad_d)");
  const bool taps = dbg && (dbg->d_relu_masks || dbg->d_grad_raw || dbg->d_grad_pts);
  const bool x16 = use_x16(h) && !b3 && !h2 && !extras && !taps;   // extras and debug taps: the x32-structured kernels
  if (b3 && !(h->have_net_b3[0] && h->have_net_b3[1] && h->have_net_b3[2]))
    return fail("nsr_render_rays_vjp: NSR_FLAG_MLP_BF16X3 needs nsr_upload_weights_b3 (both networks) and nsr_upload_weights_bwd_b3");
  if (x16 && !(h->have_net16[0] && h->have_net16[1] && h->have_net16[2]))
    return fail("nsr_render_rays_vjp: variant 16 needs nsr_upload_weights16 (both networks) and nsr_upload_weights_bwd16");
  if (!x16 && !b3 && !h2 && !h->have_net[2]) return fail("nsr_render_rays_vjp: backward stream not uploaded (nsr_upload_weights_bwd)");
  if (!d_rays_o || !d_rays_d || !d_grad_rgb || !d_grad_o || !d_grad_d) return fail("nsr_render_rays_vjp: null argument");
  if (n_rays <= 0) return n_rays == 0 ? 0 : fail("nsr_render_rays_vjp: negative ray count");
  NSR_DEVICE(h);
  hipStream_t s = (hipStream_t)stream;
  bool capturing = false;
  if (int e = stream_capturing(s, &capturing)) return e;
  if (int e = claim_stream(h, s, capturing)) return e;
  if (h2) { if (int e = ensure_range(h, n_rays, capturing)) return e; }
  long long grid;
  if (x16) {                                               // one ray per item, two workgroups per CU
    grid = h->cfg.max_workgroups > 0 ? h->cfg.max_workgroups : 2LL * h->n_cu;
    if (grid > 2LL * h->mask_grid) grid = 2LL * h->mask_grid;   // scratch entries are half the size (uint2)
    if (grid > n_rays) grid = n_rays;
  } else {
    grid = grid_for(h, (n_rays + 1) / 2);
    if (grid > h->mask_grid) grid = h->mask_grid;          // the relu-pattern scratch was sized at upload time
  }
  nsr::VjpArgs v;
  memset(&v, 0, sizeof(v));
  nsr::RenderArgs& a = v.r;
  a.rays_o = d_rays_o; a.rays_d = d_rays_d; a.n_rays = n_rays; a.near_ = near_; a.far_ = far_; a.camera = 0;
  set_extras(a, ex);
  v.grad_viewdirs = d_grad_viewdirs;
  float* nets = h2 ? h->d_nets_h2 : (b3 ? h->d_nets_b3 : (x16 ? h->d_nets16 : h->d_nets));
  const size_t net_floats = b3 ? kB3Stride : (h2 ? kH2Stride : (size_t)NSR_PACKED_FLOATS);
  const size_t stream_floats = (size_t)(b3 ? NSR_STREAM_SLABS_B3 : NSR_STREAM_SLABS) * NSR_SLAB_FLOATS;
  a.nets = nets;
  a.net_stride = (long long)sizeof(float) * (long long)net_floats;
  a.aux[0] = nets + stream_floats;
  a.aux[1] = nets + net_floats + stream_floats;
  a.tcoarse = h->d_tables;
  a.ufine = h->d_tables + 128;
  a.fine = 1;
  a.work_counter = h->d_work_counter;
  a.white_bkgd = (h->cfg.flags & NSR_FLAG_WHITE_BKGD) ? 1 : 0;
  a.lindisp = (h->cfg.flags & NSR_FLAG_LINDISP) ? 1 : 0;
  if (out) { a.rgb = out->d_rgb; a.disp = out->d_disp; a.acc = out->d_acc; }
  v.grad_rgb = d_grad_rgb; v.grad_o = d_grad_o; v.grad_d = d_grad_d; v.mask_scratch = h->d_mask_scratch;
  v.z_fine = d_z_fine;
  if (dbg) { v.dbg_masks = (uint4*)dbg->d_relu_masks; v.dbg_graw = dbg->d_grad_raw; v.dbg_gpts = dbg->d_grad_pts; }
  // f16x2 range safety net (see launch_render): the fallback runs on bf16 MFMAs when the handle holds the bf16x3 images
  // and their transposed stream (nsr_upload_weights_b3 / _bwd_b3), else on fp32 MFMAs (nsr_upload_weights_bwd; N_importance
  // 128 only); with neither, the reported items are dropped (NaN, counted)
  const bool fb3 = h2 && h->have_net_b3[0] && h->have_net_b3[1] && h->have_net_b3[2];
  const bool fallback = fb3 || (h2 && h->have_net[2] && !special);
  if (h2) { a.ovf_items = h->d_ovf_items; a.ovf_stat = h->d_ovf_stat; a.ovf_cap = fallback ? h->ovf_cap : 0u; }
  // global-phases schedule (k_render_vjp16p) unless the caller supplies the depths itself (then nothing is handed over)
  const bool phases = x16 && (h->cfg.flags & NSR_FLAG_SCHED_PHASES) && !d_z_fine;
  if (phases) {
    a.zf_scratch = h->d_zf_scratch;
    a.sched_flags = h->d_sched_flags;
    a.status = h->d_status;
    a.super_lg = kSuperLg;
    a.spin_max = h->cfg.chunk > 0 ? h->cfg.chunk - 1 : 64;
    a.epoch_counter = h->d_status + 1;
    NSR_HIP(hipMemsetAsync(h->d_sched_flags, 0, sizeof(unsigned) * 2 * ((size_t)3 << kSuperLg), s));
  }
  hipLaunchKernelGGL(nsr::k_set_vjp_args, dim3(1), dim3(1), 0, s, v, h->d_vjp_args);   // also zeroes the work counter
  if (!capturing) NSR_HIP(hipEventRecord(h->ev0, s));
  if (phases)
    hipLaunchKernelGGL(nsr::k_render_vjp16p, dim3((int)grid), dim3(256), kVjp16Lds, s, (const nsr::VjpArgs*)h->d_vjp_args);
  else if (x16)
    hipLaunchKernelGGL(nsr::k_render_vjp16, dim3((int)grid), dim3(256), kVjp16Lds, s, (const nsr::VjpArgs*)h->d_vjp_args);
  else if (b3)
    hipLaunchKernelGGL(nsr::k_render_vjp_b3, dim3((int)grid), dim3(256), kRenderLds, s, (const nsr::VjpArgs*)h->d_vjp_args);
  else if (h2 && special)
    hipLaunchKernelGGL(special_kernels(ns, ni).vjp_h2, dim3((int)grid), dim3(256), lds32, s, (const nsr::VjpArgs*)h->d_vjp_args);
  else if (h2)
    hipLaunchKernelGGL(nsr::k_render_vjp_h2, dim3((int)grid), dim3(256), kRenderLds, s, (const nsr::VjpArgs*)h->d_vjp_args);
  else
    hipLaunchKernelGGL(nsr::k_render_vjp, dim3((int)grid), dim3(256), kRenderLds, s, (const nsr::VjpArgs*)h->d_vjp_args);
  if (fallback) {          // the reported items again, forward and backward, on the fp32 kernel of the same template
    nsr::VjpArgs f = v;
    nsr::RenderArgs& fa = f.r;
    float* fnets = fb3 ? h->d_nets_b3 : h->d_nets;
    const size_t fnet_floats = fb3 ? kB3Stride : (size_t)NSR_PACKED_FLOATS;
    const size_t fstream_floats = (size_t)(fb3 ? NSR_STREAM_SLABS_B3 : NSR_STREAM_SLABS) * NSR_SLAB_FLOATS;
    fa.nets = fnets;
    fa.net_stride = (long long)sizeof(float) * (long long)fnet_floats;
    fa.aux[0] = fnets + fstream_floats;
    fa.aux[1] = fnets + fnet_floats + fstream_floats;
    fa.ovf_items = nullptr; fa.ovf_stat = nullptr; fa.ovf_cap = 0;
    fa.item_list = h->d_ovf_items; fa.item_count = h->d_ovf_stat; fa.item_cap = h->ovf_cap;
    fa.work_counter = h->d_work_counter + 1;
    hipLaunchKernelGGL(nsr::k_set_vjp_args, dim3(1), dim3(1), 0, s, f, h->d_vjp_args_fb);
    VjpKernel fk = fb3 ? (special ? special_kernels(ns, ni).vjp_b3 : nsr::k_render_vjp_b3) : nsr::k_render_vjp;
    hipLaunchKernelGGL(fk, dim3((int)grid), dim3(256), lds32, s, (const nsr::VjpArgs*)h->d_vjp_args_fb);
  }
  NSR_HIP(hipGetLastError());
  if (!capturing) {
    NSR_HIP(hipEventRecord(h->ev1, s));
    NSR_HIP(hipEventRecord(h->ev_busy, s));
  }
  h->launched = true;
  h->timed = !capturing;
  return 0;
}

int nsr_pose_grad(nsr_handle h, const float* d_grad_o, const float* d_grad_d, int H, int W, const double* K9,
                  int patch, float* d_out, void* stream) {
  if (!h || !d_grad_o || !d_grad_d || !K9 || !d_out) return fail("nsr_pose_grad: null argument");
  if (H <= 0 || W <= 0 || patch <= 0) return fail("nsr_pose_grad: bad geometry");
  NSR_DEVICE(h);
  const int n = H * W, n_patches = (n + patch - 1) / patch;
  hipLaunchKernelGGL(nsr::k_pose_grad, dim3(n_patches), dim3(256), 0, (hipStream_t)stream, d_grad_o, d_grad_d,
                     (float)K9[0], (float)K9[4], (float)K9[2], (float)K9[5], W, n, patch, d_out);
  NSR_HIP(hipGetLastError());
  return 0;
}

int nsr_get_rays(nsr_handle h, const float* d_c2w, int H, int W, const double* K9, float* d_rays_o,
                 float* d_rays_d, void* stream) {
  if (!h || !d_c2w || !K9 || !d_rays_o || !d_rays_d) return fail("nsr_get_rays: null argument");
  if (H <= 0 || W <= 0) return fail("nsr_get_rays: bad image geometry");
  NSR_DEVICE(h);
  const int n = H * W;
  hipLaunchKernelGGL(nsr::k_get_rays, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_c2w, (float)K9[0],
                     (float)K9[4], (float)K9[2], (float)K9[5], H, W, d_rays_o, d_rays_d);
  NSR_HIP(hipGetLastError());
  return 0;
}

int nsr_get_rays_views(nsr_handle h, const float* d_c2w, int n_views, int H, int W, const double* K9, float* d_rays_o,
                       float* d_rays_d, void* stream) {
  if (h && n_views == 0) return 0;
  if (!h || !d_c2w || !K9 || !d_rays_o || !d_rays_d) return fail("nsr_get_rays_views: null argument");
  if (H <= 0 || W <= 0 || n_views < 0 || n_views > 65535) return fail("nsr_get_rays_views: bad image geometry / view count");
  NSR_DEVICE(h);
  const int n = H * W;
  hipLaunchKernelGGL(nsr::k_get_rays, dim3((n + 255) / 256, n_views), dim3(256), 0, (hipStream_t)stream, d_c2w, (float)K9[0],
                     (float)K9[4], (float)K9[2], (float)K9[5], H, W, d_rays_o, d_rays_d);
  NSR_HIP(hipGetLastError());
  return 0;
}

int nsr_ndc_rays(nsr_handle h, const float* d_rays_o, const float* d_rays_d, int64_t n_rays, int H, int W, double focal,
                 double near_, float* d_o_out, float* d_d_out, void* stream) {
  if (h && n_rays == 0) return 0;
  if (!h || !d_rays_o || !d_rays_d || !d_o_out || !d_d_out) return fail("nsr_ndc_rays: null argument");
  if (n_rays < 0 || H <= 0 || W <= 0 || !(focal > 0.0)) return fail("nsr_ndc_rays: bad geometry");
  NSR_DEVICE(h);
  hipLaunchKernelGGL(nsr::k_ndc_rays, dim3((unsigned)((n_rays + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_rays_o,
                     d_rays_d, (long long)n_rays, (float)near_, (float)(-1.0 / (W / (2.0 * focal))),
                     (float)(-1.0 / (H / (2.0 * focal))), (float)(2.0 * near_), (float)(-2.0 * near_), d_o_out, d_d_out);
  NSR_HIP(hipGetLastError());
  return 0;
}

int nsr_ndc_rays_vjp(nsr_handle h, const float* d_rays_o, const float* d_rays_d, int64_t n_rays, int H, int W, double focal,
                     double near_, const float* d_grad_o_ndc, const float* d_grad_d_ndc, float* d_grad_o, float* d_grad_d,
                     void* stream) {
  if (h && n_rays == 0) return 0;
  if (!h || !d_rays_o || !d_rays_d || !d_grad_o_ndc || !d_grad_d_ndc || !d_grad_o || !d_grad_d)
    return fail("nsr_ndc_rays_vjp: null argument");
  if (n_rays < 0 || H <= 0 || W <= 0 || !(focal > 0.0)) return fail("nsr_ndc_rays_vjp: bad geometry");
  NSR_DEVICE(h);
  hipLaunchKernelGGL(nsr::k_ndc_rays_vjp, dim3((unsigned)((n_rays + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     d_rays_o, d_rays_d, (long long)n_rays, (float)near_, (float)(-1.0 / (W / (2.0 * focal))),
                     (float)(-1.0 / (H / (2.0 * focal))), (float)(2.0 * near_), d_grad_o_ndc, d_grad_d_ndc, d_grad_o,
                     d_grad_d);
  NSR_HIP(hipGetLastError());
  return 0;
}

int nsr_to8b(nsr_handle h, const float* d_x, int64_t n, uint8_t* d_out, void* stream) {
  if (h && n == 0) return 0;      // an empty batch is a valid no-op (buffers may be null)
  if (!h || !d_x || !d_out) return fail("nsr_to8b: null argument");
  if (n <= 0) return n == 0 ? 0 : fail("nsr_to8b: negative element count");
  NSR_DEVICE(h);
  const long long groups = (n + 3) / 4;
  hipLaunchKernelGGL(nsr::k_to8b, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_x,
                     (long long)n, d_out);
  NSR_HIP(hipGetLastError());
  return 0;
}

int nsr_fingerprint(nsr_handle h, const void* const* d_ptrs, const int64_t* d_n_words, int n_tensors, uint64_t* d_out,
                    void* stream) {
  if (!h || !d_ptrs || !d_n_words || !d_out) return fail("nsr_fingerprint: null argument");
  if (n_tensors <= 0 || n_tensors > 65535) return fail("nsr_fingerprint: 1..65535 tensors");
  NSR_DEVICE(h);
  NSR_HIP(hipMemsetAsync(d_out, 0, sizeof(uint64_t), (hipStream_t)stream));
  hipLaunchKernelGGL(nsr::k_fingerprint, dim3(16, (unsigned)n_tensors), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned* const*)d_ptrs, (const long long*)d_n_words, (unsigned long long*)d_out);
  NSR_HIP(hipGetLastError());
  return 0;
}

int nsr_find_bbox(nsr_handle h, const uint8_t* d_rgb8, int n_images, int H, int W, int32_t* d_bbox,
                  int32_t* d_count, uint8_t* d_mask, void* stream) {
  if (h && n_images == 0) return 0;      // an empty batch is a valid no-op (buffers may be null)
  if (!h || !d_rgb8 || !d_bbox || !d_count) return fail("nsr_find_bbox: null argument");
  if (H <= 0 || W <= 0 || (long long)H * W > (1 << 20)) return fail("nsr_find_bbox: image must have 1..2^20 pixels");
  if (n_images <= 0) return n_images == 0 ? 0 : fail("nsr_find_bbox: negative image count");
  NSR_DEVICE(h);
  const long long hw = (long long)H * W;
  const int batch = (int)(h->box_scratch_ints / (size_t)(hw + (hw + 1) * 5));   // images per round of the 5 kernels
  if (batch < 1)
    return fail("nsr_find_bbox: scratch not reserved for this image size (call nsr_reserve_bbox(h, H, W) once, at setup)");
  bool capturing = false;
  if (int e = stream_capturing((hipStream_t)stream, &capturing)) return e;
  if (int e = claim_stream(h, (hipStream_t)stream, capturing)) return e;
  hipStream_t s = (hipStream_t)stream;
  for (int i0 = 0; i0 < n_images; i0 += batch) {
    nsr::BoxArgs a;
    a.K = (n_images - i0) < batch ? (n_images - i0) : batch;
    a.H = H; a.W = W;
    a.rgb8 = d_rgb8 + (size_t)i0 * hw * 3;
    a.mask = d_mask ? d_mask + (size_t)i0 * hw : nullptr;
    a.parent = h->d_box_scratch;
    a.stats = h->d_box_scratch + (size_t)batch * hw;
    a.bbox = d_bbox + (size_t)i0 * 4;
    a.count = d_count + i0;
    const long long n_init = (long long)a.K * (hw + 1), n_pix = (long long)a.K * hw;
    hipLaunchKernelGGL(nsr::k_box_init, dim3((unsigned)((n_init + 255) / 256)), dim3(256), 0, s, a);
    hipLaunchKernelGGL(nsr::k_box_union, dim3((unsigned)((n_pix + 255) / 256)), dim3(256), 0, s, a);
    hipLaunchKernelGGL(nsr::k_box_flatten, dim3((unsigned)((n_pix + 255) / 256)), dim3(256), 0, s, a);
    hipLaunchKernelGGL(nsr::k_box_stats, dim3((unsigned)((n_pix + 255) / 256)), dim3(256), 0, s, a);
    hipLaunchKernelGGL(nsr::k_box_select, dim3(a.K), dim3(256), 0, s, a);
  }
  NSR_HIP(hipGetLastError());
  if (!capturing) NSR_HIP(hipEventRecord(h->ev_busy, s));
  h->launched = true;
  return 0;
}

int nsr_reserve_bbox(nsr_handle h, int H, int W) {
  if (!h) return fail("nsr_reserve_bbox: null handle");
  if (H <= 0 || W <= 0 || (long long)H * W > (1 << 20)) return fail("nsr_reserve_bbox: image must have 1..2^20 pixels");
  NSR_DEVICE(h);
  const long long hw = (long long)H * W;
  const size_t need = (size_t)16 * (hw + (hw + 1) * 5);     // 16 images in flight: 24 B per pixel each
  if (need <= h->box_scratch_ints) return 0;
  if (h->d_box_scratch) { NSR_HIP(hipDeviceSynchronize()); NSR_HIP(hipFree(h->d_box_scratch)); }   // setup call
  h->d_box_scratch = nullptr; h->box_scratch_ints = 0;
  NSR_HIP(hipMalloc(&h->d_box_scratch, need * sizeof(int)));
  h->box_scratch_ints = need;
  return 0;
}

int nsr_sample_pose(nsr_handle h, const float* d_prob, const double* d_gumbel, const double* d_uniform,
                    const double* d_theta, int K, int n_cat, double gumbel_T, double radius, float* d_poses44,
                    float* d_c2w34, float* d_jac, void* stream) {
  if (h && K == 0) return 0;
  if (!h || !d_prob || !d_gumbel || !d_uniform || !d_theta) return fail("nsr_sample_pose: null argument");
  if (K < 0 || n_cat < 1 || n_cat > nsr::kMaxCat) return fail("nsr_sample_pose: K >= 0, n_cat in 1..16");
  if (!(gumbel_T > 0.0)) return fail("nsr_sample_pose: gumbel_T must be positive");
  NSR_DEVICE(h);
  nsr::PoseArgs a{d_prob, nullptr, d_gumbel, d_uniform, d_theta, K, n_cat, gumbel_T, radius, d_poses44, d_c2w34, d_jac};
  hipLaunchKernelGGL(nsr::k_sample_pose, dim3((K + 63) / 64), dim3(64), 0, (hipStream_t)stream, a);
  NSR_HIP(hipGetLastError());
  return 0;
}

int nsr_sample_pose_nograd(nsr_handle h, const double* d_logits, const double* d_gumbel, const double* d_uniform,
                           const double* d_theta, int K, int n_cat, double gumbel_T, double radius, float* d_poses44,
                           float* d_c2w34, void* stream) {
  if (h && K == 0) return 0;
  if (!h || !d_logits || !d_gumbel || !d_uniform || !d_theta) return fail("nsr_sample_pose_nograd: null argument");
  if (K < 0 || n_cat < 1 || n_cat > nsr::kMaxCat) return fail("nsr_sample_pose_nograd: K >= 0, n_cat in 1..16");
  if (!(gumbel_T > 0.0)) return fail("nsr_sample_pose_nograd: gumbel_T must be positive");
  NSR_DEVICE(h);
  nsr::PoseArgs a{nullptr, d_logits, d_gumbel, d_uniform, d_theta, K, n_cat, gumbel_T, radius, d_poses44, d_c2w34, nullptr};
  hipLaunchKernelGGL(nsr::k_sample_pose_nograd, dim3((K + 63) / 64), dim3(64), 0, (hipStream_t)stream, a);
  NSR_HIP(hipGetLastError());
  return 0;
}

int nsr_embed(nsr_handle h, const float* d_x, int64_t n, int multires, float* d_out, void* stream) {
  if (h && n == 0) return 0;      // an empty batch is a valid no-op (buffers may be null)
  if (!h || !d_x || !d_out) return fail("nsr_embed: null argument");
  if (multires < 1 || multires > 16) return fail("nsr_embed: multires in 1..16");
  if (n <= 0) return n == 0 ? 0 : fail("nsr_embed: negative point count");
  NSR_DEVICE(h);
  const long long work = (long long)n * (multires + 1);
  hipLaunchKernelGGL(nsr::k_embed, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_x,
                     (long long)n, multires, d_out);
  NSR_HIP(hipGetLastError());
  return 0;
}

int nsr_run_network(nsr_handle h, int net_id, const float* d_pts, const float* d_viewdirs, int64_t n_pts,
                    float* d_raw, void* stream) {
  if (h && n_pts == 0) return 0;      // an empty batch is a valid no-op (buffers may be null)
  if (!h || !d_pts || !d_viewdirs || !d_raw) return fail("nsr_run_network: null argument");
  if (net_id < 0 || net_id > 1 || !h->have_net[net_id]) return fail("nsr_run_network: network not uploaded");
  if (n_pts <= 0) return n_pts == 0 ? 0 : fail("nsr_run_network: negative point count");
  NSR_DEVICE(h);
  nsr::NetArgs a;
  a.stream = h->d_packed[net_id];
  a.aux = h->d_packed[net_id] + (size_t)NSR_STREAM_SLABS * NSR_SLAB_FLOATS;
  a.pts = d_pts; a.dirs = d_viewdirs; a.raw = d_raw; a.n_pts = n_pts;
  const long long tiles = (n_pts + 127) / 128;
  hipLaunchKernelGGL(nsr::k_run_network, dim3(grid_for(h, tiles)), dim3(256), kNetLds, (hipStream_t)stream, a);
  NSR_HIP(hipGetLastError());
  return 0;
}

int nsr_raw2outputs(nsr_handle h, const float* d_raw, const float* d_z, const float* d_rays_d, int64_t n_rays,
                    int n_samples, float* d_rgb, float* d_disp, float* d_acc, float* d_weights, float* d_depth,
                    void* stream) {
  if (h && n_rays == 0) return 0;      // an empty batch is a valid no-op (buffers may be null)
  if (!h || !d_raw || !d_z || !d_rays_d || !d_rgb || !d_disp || !d_acc || !d_weights || !d_depth)
    return fail("nsr_raw2outputs: null argument");
  if (n_samples != 64 && n_samples != 192) return fail("nsr_raw2outputs: n_samples must be 64 or 192");
  if (n_rays <= 0) return n_rays == 0 ? 0 : fail("nsr_raw2outputs: negative ray count");
  NSR_DEVICE(h);
  nsr::R2OArgs a{d_raw, d_z, d_rays_d, n_rays, (h->cfg.flags & NSR_FLAG_WHITE_BKGD) ? 1 : 0,
                 d_rgb, d_disp, d_acc, d_weights, d_depth};
  const long long items = (n_rays + 1) / 2;
  const int grid = (int)(items < 4096 ? items : 4096);
  if (n_samples == 64)
    hipLaunchKernelGGL(nsr::k_raw2outputs<64>, dim3(grid), dim3(256), sizeof(nsr::ItemState), (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(nsr::k_raw2outputs<192>, dim3(grid), dim3(256), sizeof(nsr::ItemState), (hipStream_t)stream, a);
  NSR_HIP(hipGetLastError());
  return 0;
}

int nsr_sample_pdf(nsr_handle h, const float* d_bins, const float* d_weights, int64_t n_rays, float* d_samples,
                   int64_t* d_inds, void* stream) {
  if (h && n_rays == 0) return 0;      // an empty batch is a valid no-op (buffers may be null)
  if (!h || !d_bins || !d_weights || !d_samples) return fail("nsr_sample_pdf: null argument");
  if (!h->have_tables) return fail("nsr_sample_pdf: tables not uploaded");
  if (n_rays <= 0) return n_rays == 0 ? 0 : fail("nsr_sample_pdf: negative ray count");
  NSR_DEVICE(h);
  if (h->cfg.n_samples != NSR_N_SAMPLES) return fail("nsr_sample_pdf: the stage kernel is specialised to N_samples = 64");
  nsr::PdfArgs a{d_bins, d_weights, h->d_tables + 128, n_rays, d_samples, (long long*)d_inds};
  const long long items = (n_rays + 1) / 2;
  const int grid = (int)(items < 4096 ? items : 4096);
  hipLaunchKernelGGL(nsr::k_sample_pdf, dim3(grid), dim3(256), sizeof(nsr::ItemState), (hipStream_t)stream, a);
  NSR_HIP(hipGetLastError());
  return 0;
}

int nsr_sort_merge(nsr_handle h, const float* d_z_coarse, const float* d_z_samples, int64_t n_rays, float* d_z_sorted,
                   void* stream) {
  if (h && n_rays == 0) return 0;      // an empty batch is a valid no-op (buffers may be null)
  if (!h || !d_z_coarse || !d_z_samples || !d_z_sorted) return fail("nsr_sort_merge: null argument");
  if (n_rays < 0) return fail("nsr_sort_merge: negative ray count");
  NSR_DEVICE(h);
  const long long items = (n_rays + 1) / 2;
  const int grid = (int)(items < 4096 ? items : 4096);
  hipLaunchKernelGGL(nsr::k_sort_merge, dim3(grid), dim3(256), sizeof(nsr::ItemState), (hipStream_t)stream,
                     d_z_coarse, d_z_samples, (long long)n_rays, d_z_sorted);
  NSR_HIP(hipGetLastError());
  return 0;
}

int nsr_selftest(nsr_handle h, void* stream) {
  if (!h) return fail("nsr_selftest: null handle");
  NSR_DEVICE(h);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(nsr::k_selftest, dim3(1), dim3(64), 0, s, h->d_scratch);
  NSR_HIP(hipGetLastError());
  std::vector<float> host(64 * 16);
  NSR_HIP(hipMemcpyAsync(host.data(), h->d_scratch, sizeof(float) * host.size(), hipMemcpyDeviceToHost, s));
  NSR_HIP(hipStreamSynchronize(s));
  // expected D[i][j] = sum_k A[i][k] B[k][j], lane l holds col j = l&31, rows (r&3)+8*(r>>2)+4*(l>>5)
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 16; ++r) {
      const int j = l & 31, i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
      double want = 0;
      for (int k = 0; k < 2; ++k) want += (1.0 + i + 100.0 * k) * (3.0 + 7.0 * j - 1000.0 * k);
      if ((double)host[l * 16 + r] != want) {
        char buf[160];
        snprintf(buf, sizeof buf, "nsr_selftest: MFMA 32x32x2 layout mismatch at lane %d reg %d: got %g want %g", l, r,
                 (double)host[l * 16 + r], want);
        return fail(buf);
      }
    }
  return 0;
}

int nsr_schedule_stats(nsr_handle h, unsigned* recomputed_rays) {
  if (!h || !recomputed_rays) return fail("nsr_schedule_stats: null argument");
  *recomputed_rays = 0u;
  if (!h->d_status) return 0;                              // per-ray queue: nothing is ever handed over
  NSR_DEVICE(h);
  NSR_HIP(hipDeviceSynchronize());
  NSR_HIP(hipMemcpy(recomputed_rays, h->d_status, sizeof(unsigned), hipMemcpyDeviceToHost));
  return 0;
}

int nsr_reserve_range(nsr_handle h, int64_t n_rays) {
  if (!h) return fail("nsr_reserve_range: null handle");
  if (n_rays < 0 || n_rays > kMaxRaysH2) return fail("nsr_reserve_range: 0 .. 2^33 - 2 rays");
  if (!h->d_ovf_items) return 0;                           // not an f16x2 handle: nothing to reserve
  NSR_DEVICE(h);
  bool capturing = false;                                  // a setup call: it allocates, so not while the handle's stream captures
  if (h->launched && h->last_stream) { if (int e = stream_capturing(h->last_stream, &capturing)) return e; }
  if (capturing) return fail("nsr_reserve_range: the handle's stream is being captured (reserve before the capture begins)");
  const unsigned long long items = (unsigned long long)(n_rays + 1) / 2;
  if (int e = ensure_range(h, n_rays, false)) return e;
  if (items > h->ovf_cap) return fail("nsr_reserve_range: out of device memory for the list");
  return 0;
}

int nsr_range_status(nsr_handle h, unsigned* last_items, unsigned* points, unsigned* rays, unsigned* dropped_items) {
  if (!h) return fail("nsr_range_status: null handle");
  unsigned st[4] = {0u, 0u, 0u, 0u};
  if (h->d_ovf_stat) {                                     // handles without NSR_FLAG_MLP_F16X2 have nothing to report
    NSR_DEVICE(h);
    NSR_HIP(hipDeviceSynchronize());
    NSR_HIP(hipMemcpy(st, h->d_ovf_stat, sizeof(st), hipMemcpyDeviceToHost));
  }
  if (last_items) *last_items = st[0];
  if (points) *points = st[1];
  if (rays) *rays = st[2];
  if (dropped_items) *dropped_items = st[3];
  return 0;
}

int nsr_debug_bounds_status(nsr_handle h, int* built_with_checks, unsigned* first_bad_line) {
  if (!h || !built_with_checks || !first_bad_line) return fail("nsr_debug_bounds_status: null argument");
  *first_bad_line = 0u;
#ifdef NSR_DEBUG_BOUNDS
  NSR_DEVICE(h);
  *built_with_checks = 1;
  NSR_HIP(hipDeviceSynchronize());
  NSR_HIP(hipMemcpyFromSymbol(first_bad_line, HIP_SYMBOL(nsr::g_bounds_violation), sizeof(unsigned)));
  for (auto unit : {nsr_unit_bounds_h2, nsr_unit_bounds_b3, nsr_unit_bounds_x16}) {      // the other units' copies of the word
    unsigned line = 0;
    if (unit(&line) != 0) return fail("nsr_debug_bounds_status: reading a unit's violation word failed");
    if (line > *first_bad_line) *first_bad_line = line;
  }
#else
  *built_with_checks = 0;
#endif
  return 0;
}

int nsr_last_kernel_ms(nsr_handle h, float* ms) {
  if (!h || !ms) return fail("nsr_last_kernel_ms: null argument");
  if (!h->timed) return fail("nsr_last_kernel_ms: no timed render launch yet (launches captured into a graph are not timed)");
  NSR_DEVICE(h);
  NSR_HIP(hipEventSynchronize(h->ev1));
  NSR_HIP(hipEventElapsedTime(ms, h->ev0, h->ev1));
  return 0;
}

}  // extern "C"
