// nsr_handoff.hip -- the image hand-off between the renderer and the detector's loader, on device.
//
// The reference writes every rendered view to a PNG (RN:245-250, to8b RH:14), and the detector side reads the
// files back to derive its annotations (NM:786-797 get_annotation / find_bbox):
//     img  = cv2.imread(path)                         -> uint8 BGR
//     gray = cv2.cvtColor(img, cv2.COLOR_RGB2GRAY)    -> the BGR data is *treated* as RGB (a reference quirk that is
//                                                        kept: channel 0 = blue gets the red coefficient)
//     mask = gray > 1 ? 255 : 0                       (cv2.threshold(gray, 1, 255, THRESH_BINARY), NM:795)
//     stats = connectedComponentsWithStats(mask)      (8-connectivity; row 0 = all zero pixels)
//     stats = stats[stats[:, 4].argsort()][:-1]       (drop the largest-area row, NM:788-789)
//     bbox  = the row with the largest w*h            (NM:691-692 / NM:817-818)
// These kernels produce the same uint8 image, mask and XYWH box from the float render without leaving the GPU.
// Integer work throughout: bit-exact against oracle/handoff_oracle.py (tests/test_gpu_parity.py).
//
// Connected components: lock-free union-find over the 8-neighbourhood (every foreground pixel unions with its W,
// NW, N, NE neighbours; roots are the smallest pixel index of a component = cv2's raster label order), then one
// flattening pass, one pass of per-root statistics (wave-aggregated integer atomics), then a one-workgroup
// selection per image.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nsr {

// to8b (RH:14): (255 * clip(x, 0, 1)).astype(uint8) -- fp32 product, truncation; NaN -> 0 (numpy on x86).
__global__ void k_to8b(const float* __restrict__ x, long long n, uint8_t* __restrict__ out) {
  const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n) return;
  auto cv = [](float v) -> uint8_t {
    const float c = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);       // np.clip; NaN falls through both compares
    const float s = 255.0f * c;
    return s == s ? (uint8_t)(int)s : (uint8_t)0;
  };
  if (i + 3 < n && (((uintptr_t)(x + i)) & 15) == 0 && (((uintptr_t)(out + i)) & 3) == 0) {
    const float4 v = *(const float4*)(x + i);
    *(uchar4*)(out + i) = make_uchar4(cv(v.x), cv(v.y), cv(v.z), cv(v.w));
  } else {
    for (long long k = i; k < n && k < i + 4; ++k) out[k] = cv(x[k]);
  }
}

struct BoxArgs {
  const uint8_t* rgb8;   // [K,H,W,3], RGB as rendered
  uint8_t* mask;         // [K,H,W] 0/255 (nullable)
  int* parent;           // [K,H*W] scratch
  int* stats;            // [K,H*W+1,5] scratch: area, minx, miny, maxx, maxy; slot H*W = the zero-pixel row
  int* bbox;             // [K,4] x, y, w, h
  int* count;            // [K] number of candidate rows (components incl. background, minus the dropped one)
  int K, H, W;
};

// OpenCV 4.x RGB2Gray<uchar> (imgproc/src/color_rgb.simd.hpp): (c0*RY15 + c1*GY15 + c2*BY15 + (1<<14)) >> 15 with
// RY15 = 9798, GY15 = 19235, BY15 = 3735; the reference hands it BGR data, so c0 = blue.
__device__ __forceinline__ int gray_of(const uint8_t* px) {
  const int r = px[0], g = px[1], b = px[2];
  return (b * 9798 + g * 19235 + r * 3735 + (1 << 14)) >> 15;
}

__global__ void k_box_init(BoxArgs a) {
  const long long hw = (long long)a.H * a.W;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)a.K * (hw + 1)) return;
  const long long img = i / (hw + 1);
  const long long p = i - img * (hw + 1);
  int* st = a.stats + (img * (hw + 1) + p) * 5;
  st[0] = 0; st[1] = a.W; st[2] = a.H; st[3] = -1; st[4] = -1;
  if (p == hw) return;
  const bool fg = gray_of(a.rgb8 + (img * hw + p) * 3) > 1;
  a.parent[img * hw + p] = fg ? (int)p : -1;
  if (a.mask) a.mask[img * hw + p] = fg ? 255 : 0;
}

// find with path halving: every visited node is re-pointed at its grandparent (atomicMin: labels only ever decrease
// towards the root, so a concurrent union on the same node cannot be undone).  Without it the chains left by the
// raster-order unions are hundreds of links long and the later passes crawl.
__device__ __forceinline__ int uf_find(int* parent, int i) {
  while (true) {
    const int p = __hip_atomic_load(parent + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (p == i) return i;
    const int gp = __hip_atomic_load(parent + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (gp == p) return p;
    atomicMin(parent + i, gp);
    i = gp;
  }
}

__device__ __forceinline__ void uf_union(int* parent, int x, int y) {
  while (true) {
    x = uf_find(parent, x);
    y = uf_find(parent, y);
    if (x == y) return;
    if (x < y) { const int t = x; x = y; y = t; }          // hang the larger root under the smaller index
    const int old = atomicMin(parent + x, y);
    if (old == x) return;
    x = old;
  }
}

__global__ void k_box_union(BoxArgs a) {
  const long long hw = (long long)a.H * a.W;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)a.K * hw) return;
  const long long img = i / hw;
  const int p = (int)(i - img * hw);
  int* parent = a.parent + img * hw;
  if (parent[p] < 0) return;
  const int y = p / a.W, x = p - y * a.W;
  if (x > 0 && parent[p - 1] >= 0) uf_union(parent, p, p - 1);
  if (y > 0) {
    if (parent[p - a.W] >= 0) uf_union(parent, p, p - a.W);
    if (x > 0 && parent[p - a.W - 1] >= 0) uf_union(parent, p, p - a.W - 1);
    if (x + 1 < a.W && parent[p - a.W + 1] >= 0) uf_union(parent, p, p - a.W + 1);
  }
}

// After the unions: point every foreground pixel straight at its root.
__global__ void k_box_flatten(BoxArgs a) {
  const long long hw = (long long)a.H * a.W;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)a.K * hw) return;
  const long long img = i / hw;
  const int p = (int)(i - img * hw);
  int* parent = a.parent + img * hw;
  if (parent[p] < 0) return;
  const int root = uf_find(parent, p);
  if (root != p) __hip_atomic_store(parent + p, root, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Per-root statistics.  Neighbouring pixels almost always share a root (and ~70 % of an image is the zero-pixel
// row), so each wave first combines its lanes per distinct root -- typically one or two rounds -- and only the
// round's leader touches memory: 5 atomics per (wave, root) instead of 5 per pixel on a handful of hot addresses.
__global__ void k_box_stats(BoxArgs a) {
  const long long hw = (long long)a.H * a.W;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = i < (long long)a.K * hw;
  long long slot = -1;
  int x = 0, y = 0;
  if (valid) {
    const long long img = i / hw;
    const int p = (int)(i - img * hw);
    const int par = a.parent[img * hw + p];                 // flattened: the root itself, or -1 for a zero pixel
    slot = img * (hw + 1) + (par < 0 ? (int)hw : par);
    y = p / a.W; x = p - y * a.W;
  }
  const int lane = threadIdx.x & 63;
  unsigned long long todo = __ballot(valid);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const long long s0 = __shfl(slot, leader);
    const bool m = valid && slot == s0;
    const unsigned long long mm = __ballot(m);
    int mnx = m ? x : 0x7fffffff, mny = m ? y : 0x7fffffff, mxx = m ? x : -1, mxy = m ? y : -1;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      mnx = min(mnx, __shfl_xor(mnx, d)); mny = min(mny, __shfl_xor(mny, d));
      mxx = max(mxx, __shfl_xor(mxx, d)); mxy = max(mxy, __shfl_xor(mxy, d));
    }
    if (lane == leader) {
      int* st = a.stats + s0 * 5;
      atomicAdd(st + 0, __popcll(mm));
      atomicMin(st + 1, mnx);
      atomicMin(st + 2, mny);
      atomicMax(st + 3, mxx);
      atomicMax(st + 4, mxy);
    }
    todo &= ~mm;
  }
}

// One workgroup per image.  Rows: the zero-pixel row (label 0) and every root (label = root index + 1, raster
// order).  Drop the largest area (ties: the highest label, what a stable argsort leaves last), then take the
// largest w*h among the rest (ties: the smallest area, then the smallest label = the first row after the sort).
__global__ void __launch_bounds__(256) k_box_select(BoxArgs a) {
  __shared__ unsigned long long best;
  __shared__ int n_rows;
  const int hw = a.H * a.W;
  const int img = blockIdx.x;
  const int* parent = a.parent + (long long)img * hw;
  const int* stats = a.stats + (long long)img * (hw + 1) * 5;
  if (threadIdx.x == 0) { best = 0ull; n_rows = 0; }
  __syncthreads();
  // pass 1: the row to drop
  unsigned long long mine = 0ull;
  int rows = 0;
  for (int p = threadIdx.x; p <= hw; p += blockDim.x) {
    const bool is_row = (p == hw) ? true : (parent[p] == p);      // cv2 always reports the zero-pixel row
    if (!is_row) continue;
    ++rows;
    const unsigned long long label = (p == hw) ? 0ull : (unsigned long long)(p + 1);
    const unsigned long long key = ((unsigned long long)stats[p * 5] << 32) | label;
    mine = key > mine ? key : mine;
  }
  atomicMax(&best, mine);
  atomicAdd(&n_rows, rows);
  __syncthreads();
  const unsigned long long dropped = best & 0xffffffffull;
  const int total_rows = n_rows;
  __syncthreads();
  if (threadIdx.x == 0) best = 0ull;
  __syncthreads();
  // pass 2: the largest w*h among the remaining rows
  mine = 0ull;
  for (int p = threadIdx.x; p <= hw; p += blockDim.x) {
    const bool is_row = (p == hw) ? true : (parent[p] == p);
    if (!is_row) continue;
    const unsigned long long label = (p == hw) ? 0ull : (unsigned long long)(p + 1);
    if (label == dropped) continue;
    const int* st = stats + p * 5;
    const unsigned long long area = (unsigned long long)st[0];
    // an empty zero-pixel row has no extent; cv2 reports a degenerate box for it, which never wins
    const unsigned long long wh = st[3] >= st[1] ? (unsigned long long)(st[3] - st[1] + 1) * (st[4] - st[2] + 1) : 0ull;
    const unsigned long long key = ((wh + 1) << 42) | ((0x1fffffull - area) << 21) | (0x1fffffull - label);
    mine = key > mine ? key : mine;
  }
  atomicMax(&best, mine);
  __syncthreads();
  if (threadIdx.x == 0) {
    int* bb = a.bbox + img * 4;
    a.count[img] = total_rows - 1;
    if (best == 0ull) { bb[0] = bb[1] = bb[2] = bb[3] = 0; return; }
    const unsigned long long label = 0x1fffffull - (best & 0x1fffffull);
    const int p = label == 0ull ? hw : (int)(label - 1);
    const int* st = stats + p * 5;
    if (st[3] < st[1]) { bb[0] = bb[1] = bb[2] = bb[3] = 0; return; }
    bb[0] = st[1]; bb[1] = st[2]; bb[2] = st[3] - st[1] + 1; bb[3] = st[4] - st[2] + 1;
  }
}

// ------------------------------------------------------------------------------------------------------
// Content fingerprint of a set of device tensors (the drop-in API's packed-weight cache key, RH:70-122 parameters): an
// order-independent 64-bit sum of per-word hashes h(tensor, index, bits) -- one pass over the bytes (HBM-bound: 2.4 MB per
// network), one 8-byte result, no temporaries.  Catches in-place writes that bump no autograd version (p.data.copy_()).
// ------------------------------------------------------------------------------------------------------
__global__ void k_fingerprint(const unsigned* const* __restrict__ ptrs, const long long* __restrict__ n_words,
                              unsigned long long* __restrict__ out) {
  const int t = blockIdx.y;
  const unsigned* p = ptrs[t];
  const long long n = n_words[t];
  unsigned long long acc = 0ull;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    unsigned long long x = ((unsigned long long)(unsigned)t << 40) ^ ((unsigned long long)i << 1) ^ ((unsigned long long)p[i] << 32 | p[i]);
    x ^= x >> 29; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 32; x *= 0x94d049bb133111ebull; x ^= x >> 29;   // splitmix64 finaliser
    acc += x;
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const unsigned lo = __shfl_xor((unsigned)acc, m), hi = __shfl_xor((unsigned)(acc >> 32), m);
    acc += ((unsigned long long)hi << 32) | lo;
  }
  if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out, acc);
}

}  // namespace nsr
