// distCUDA2 for gfx950: mean squared distance of every point to its 3 nearest other points.
//
// Replaces SimpleKNN::knn (submodules/simple-knn/simple_knn.cu:184-219) behind the torch op
// distCUDA2 (spatial.cu:14-25), which the Street-Gaussians model files import at module level
// (lib/models/gaussian_model.py:5,63) to initialise the Gaussian scales.
//
// Same outline as the reference -- Morton order, boxes of 1024 consecutive points, prune a box when
// it is farther than the running 3rd-best -- and therefore the same result: pruning only removes
// boxes that cannot hold one of the 3 nearest neighbours, so the output is the exact 3-NN multiset
// and does not depend on codes, order or box layout.  What differs is the mapping:
//   * the reference gives every point its own thread that walks ALL boxes and then the points of the
//     surviving boxes straight from global memory through an index indirection;
//   * here a workgroup owns 256 consecutive Morton-sorted query points.  Boxes are visited
//     outwards from the tile's own box; a box is first tested against the TILE (box-to-box
//     distance against the largest running 3rd-best of the tile, one test per workgroup instead
//     of 256), its 1024 points are then staged once into LDS (coalesced 16-byte loads of the
//     gathered, sorted copy) and every lane that still needs the box scans it from LDS (broadcast
//     reads).  The running bound of the tile is refreshed after every staged box.
// Arithmetic: fp32 in the reference's order, no FMA contraction (-ffp-contract=off for this file),
// so the result is bit-identical to oracle/knn_oracle.py.
#include <cfloat>

#include "common.h"

namespace grpg {

constexpr int KNN_BOX = 1024;      // simple_knn.cu:12
constexpr int KNN_TILE = 256;      // query points per workgroup

// order-preserving float <-> uint mapping for atomicMin / atomicMax
__device__ __forceinline__ uint32_t enc_f(const float f) {
  const uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float dec_f(const uint32_t e) {
  return __uint_as_float((e & 0x80000000u) ? (e & 0x7FFFFFFFu) : ~e);
}

// bounds[0..2] = min, bounds[3..5] = max, both reduced FROM ZERO like the reference
// (cub::DeviceReduce with init {0,0,0}, simple_knn.cu:190-196)
__global__ void __launch_bounds__(256)
knn_bounds_init_kernel(uint32_t* __restrict__ bounds) {
  if (threadIdx.x < 6) bounds[threadIdx.x] = enc_f(0.0f);
}

__global__ void __launch_bounds__(256)
knn_bounds_kernel(const int P, const float* __restrict__ pts, uint32_t* __restrict__ bounds) {
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int i = blockIdx.x * 256 + threadIdx.x; i < P; i += gridDim.x * 256) {
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const float v = pts[3 * (size_t)i + c];
      mn[c] = fminf(mn[c], v);
      mx[c] = fmaxf(mx[c], v);
    }
  }
#pragma unroll
  for (int c = 0; c < 3; c++) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      mn[c] = fminf(mn[c], __shfl_xor(mn[c], d, 64));
      mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], d, 64));
    }
  }
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int c = 0; c < 3; c++) {
      atomicMin(&bounds[c], enc_f(mn[c]));
      atomicMax(&bounds[3 + c], enc_f(mx[c]));
    }
  }
}

__device__ __forceinline__ uint32_t prep_morton(uint32_t x) {   // simple_knn.cu:45-52
  x = (x | (x << 16)) & 0x030000FFu;
  x = (x | (x << 8)) & 0x0300F00Fu;
  x = (x | (x << 4)) & 0x030C30C3u;
  x = (x | (x << 2)) & 0x09249249u;
  return x;
}

__global__ void __launch_bounds__(256)
knn_morton_kernel(const int P, const float* __restrict__ pts, const uint32_t* __restrict__ bounds,
                  uint32_t* __restrict__ codes) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  uint32_t m[3];
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const float lo = dec_f(bounds[c]), hi = dec_f(bounds[3 + c]);
    const float span = hi - lo;
    // simple_knn.cu:56-58; a degenerate axis (span 0) would divide by zero there -- any code is
    // as good as another for the result, use 0
    const float t = span > 0.0f ? ((pts[3 * (size_t)i + c] - lo) / span) * 1023.0f : 0.0f;
    m[c] = prep_morton((uint32_t)fminf(fmaxf(t, 0.0f), 1023.0f));
  }
  codes[i] = m[0] | (m[1] << 1) | (m[2] << 2);
}

// sorted copy {x, y, z, original index} + per-box bounds
__global__ void __launch_bounds__(256)
knn_gather_boxes_kernel(const int P, const float* __restrict__ pts,
                        const uint32_t* __restrict__ order, float4* __restrict__ spts,
                        float* __restrict__ boxes) {
  __shared__ float s_red[4][6];
  const int b = blockIdx.x;
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
#pragma unroll
  for (int r = 0; r < KNN_BOX / 256; r++) {
    const int i = b * KNN_BOX + r * 256 + threadIdx.x;
    if (i < P) {
      const uint32_t o = order[i];
      const float x = pts[3 * (size_t)o], y = pts[3 * (size_t)o + 1], z = pts[3 * (size_t)o + 2];
      spts[i] = make_float4(x, y, z, __uint_as_float(o));
      mn[0] = fminf(mn[0], x); mn[1] = fminf(mn[1], y); mn[2] = fminf(mn[2], z);
      mx[0] = fmaxf(mx[0], x); mx[1] = fmaxf(mx[1], y); mx[2] = fmaxf(mx[2], z);
    }
  }
#pragma unroll
  for (int c = 0; c < 3; c++) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      mn[c] = fminf(mn[c], __shfl_xor(mn[c], d, 64));
      mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], d, 64));
    }
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int c = 0; c < 3; c++) { s_red[wave][c] = mn[c]; s_red[wave][3 + c] = mx[c]; }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int c = threadIdx.x;
    float v = s_red[0][c];
    for (int w = 1; w < 4; w++) v = c < 3 ? fminf(v, s_red[w][c]) : fmaxf(v, s_red[w][c]);
    boxes[6 * (size_t)b + c] = v;
  }
}

__device__ __forceinline__ void update_best(float dist, float (&best)[3]) {   // simple_knn.cu:120-134
#pragma unroll
  for (int j = 0; j < 3; j++) {
    if (best[j] > dist) { const float t = best[j]; best[j] = dist; dist = t; }
  }
}

__device__ __forceinline__ float dist2(const float4 q, const float4 p) {
  const float dx = p.x - q.x, dy = p.y - q.y, dz = p.z - q.z;
  return dx * dx + dy * dy + dz * dz;
}

// squared distance from point q to the box (simple_knn.cu:108-118)
__device__ __forceinline__ float dist_box_point(const float* __restrict__ bx, const float4 q) {
  float dx = 0.f, dy = 0.f, dz = 0.f;
  if (q.x < bx[0] || q.x > bx[3]) dx = fminf(fabsf(q.x - bx[0]), fabsf(q.x - bx[3]));
  if (q.y < bx[1] || q.y > bx[4]) dy = fminf(fabsf(q.y - bx[1]), fabsf(q.y - bx[4]));
  if (q.z < bx[2] || q.z > bx[5]) dz = fminf(fabsf(q.z - bx[2]), fabsf(q.z - bx[5]));
  return dx * dx + dy * dy + dz * dz;
}

__global__ void __launch_bounds__(KNN_TILE)
knn_search_kernel(const int P, const float4* __restrict__ spts, const float* __restrict__ boxes,
                  const int nboxes, float* __restrict__ out) {
  __shared__ float4 s_pts[KNN_BOX];
  __shared__ float s_tile[4][7];     // per wave: tile bounds (6) + largest running 3rd-best
  __shared__ float s_box[6];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = blockIdx.x * KNN_TILE + tid;
  const bool live = i < P;
  const float4 q = live ? spts[i] : make_float4(0.f, 0.f, 0.f, 0.f);

  // Seed the pruning bound with the 3rd-best among the +-3 neighbours in Morton order
  // (simple_knn.cu:146-153); the search below starts again from FLT_MAX like the reference
  // (:155-158), the seed only prunes.
  float bound = FLT_MAX;
  if (live) {
    float nb[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
    const int lo = max(0, i - 3), hi = min(P - 1, i + 3);
    for (int j = lo; j <= hi; j++)
      if (j != i) update_best(dist2(q, spts[j]), nb);
    bound = nb[2];
  }
  float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};

  // tile bounding box (over live lanes)
  float mn[3] = {live ? q.x : FLT_MAX, live ? q.y : FLT_MAX, live ? q.z : FLT_MAX};
  float mx[3] = {live ? q.x : -FLT_MAX, live ? q.y : -FLT_MAX, live ? q.z : -FLT_MAX};
#pragma unroll
  for (int c = 0; c < 3; c++) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      mn[c] = fminf(mn[c], __shfl_xor(mn[c], d, 64));
      mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], d, 64));
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < 3; c++) { s_tile[wave][c] = mn[c]; s_tile[wave][3 + c] = mx[c]; }
  }
  __syncthreads();
  float tb[6];
#pragma unroll
  for (int c = 0; c < 3; c++) {
    tb[c] = fminf(fminf(s_tile[0][c], s_tile[1][c]), fminf(s_tile[2][c], s_tile[3][c]));
    tb[3 + c] = fmaxf(fmaxf(s_tile[0][3 + c], s_tile[1][3 + c]),
                      fmaxf(s_tile[2][3 + c], s_tile[3][3 + c]));
  }

  const int own = (blockIdx.x * KNN_TILE) / KNN_BOX;
  // visit boxes outwards from the tile's own one: own, own+1, own-1, own+2, ...
  for (int step = 0; step < 2 * nboxes; step++) {
    const int off = (step + 1) >> 1;
    const int b = (step & 1) ? own + off : own - off;
    if (b < 0 || b >= nboxes) continue;   // workgroup-uniform
    // the lane's pruning bound: the reference's `dist > reject || dist > best[2]`
    const float mine = live ? fminf(bound, best[2]) : -1.0f;
    float wmax = mine;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, d, 64));
    __syncthreads();   // previous iteration's readers of s_tile / s_pts / s_box are done
    if (lane == 0) s_tile[wave][6] = wmax;
    if (tid < 6) s_box[tid] = boxes[6 * (size_t)b + tid];
    __syncthreads();
    const float rmax = fmaxf(fmaxf(s_tile[0][6], s_tile[1][6]), fmaxf(s_tile[2][6], s_tile[3][6]));
    float bx[6];
#pragma unroll
    for (int c = 0; c < 6; c++) bx[c] = s_box[c];
    // box-to-tile distance (lower bound of every point-to-box distance of the tile)
    float gap2 = 0.f;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const float g = fmaxf(0.f, fmaxf(bx[c] - tb[3 + c], tb[c] - bx[3 + c]));
      gap2 += g * g;
    }
    // 1e-6 relative slack: gap2 is computed from other operands than the per-point distance,
    // the per-point test below is the exact one
    if (gap2 > rmax * 1.000001f) continue;   // workgroup-uniform
    const int base = b * KNN_BOX;
    const int n = min(KNN_BOX, P - base);
#pragma unroll
    for (int r = 0; r < KNN_BOX / KNN_TILE; r++) {
      const int j = r * KNN_TILE + tid;
      if (j < n) s_pts[j] = spts[base + j];
    }
    __syncthreads();
    if (live) {
      const float d = dist_box_point(bx, q);
      if (!(d > bound || d > best[2])) {   // simple_knn.cu:165-167
        for (int j = 0; j < n; j++) {
          if (base + j == i) continue;
          update_best(dist2(q, s_pts[j]), best);
        }
      }
    }
  }
  if (live) out[__float_as_uint(q.w)] = (best[0] + best[1] + best[2]) / 3.0f;
}

size_t knn_workspace_bytes(const int P) {
  const size_t n = (size_t)(P > 0 ? P : 1);
  const size_t nchunks = (n + RS_CHUNK - 1) / RS_CHUNK;
  const size_t nboxes = (n + KNN_BOX - 1) / KNN_BOX;
  size_t o = 256;                                             // bounds
  o += 4 * align_up(n * 4, 256);                              // codes a/b, order a/b
  o += align_up((size_t)RS_MAX_RADIX * nchunks * 8, 256);     // radix table
  o += align_up(4 * RS_MAX_RADIX * 4, 256);                   // radix totals
  o += align_up(n * 16, 256);                                 // sorted points
  o += align_up(nboxes * 6 * 4, 256);                         // boxes
  return o;
}

void launch_knn(hipStream_t s, const int P, const float* points, float* mean_dists, char* ws) {
  if (P <= 0) return;
  const size_t n = (size_t)P;
  const size_t nchunks = (n + RS_CHUNK - 1) / RS_CHUNK;
  const int nboxes = (int)((n + KNN_BOX - 1) / KNN_BOX);
  size_t o = 0;
  auto take = [&](size_t bytes) { char* p = ws + o; o += align_up(bytes, 256); return p; };
  uint32_t* bounds = (uint32_t*)take(256);
  uint32_t* key_a = (uint32_t*)take(n * 4);
  uint32_t* key_b = (uint32_t*)take(n * 4);
  uint32_t* val_a = (uint32_t*)take(n * 4);
  uint32_t* val_b = (uint32_t*)take(n * 4);
  uint32_t* table = (uint32_t*)take((size_t)RS_MAX_RADIX * nchunks * 8);
  uint32_t* totals = (uint32_t*)take(4 * RS_MAX_RADIX * 4);
  float4* spts = (float4*)take(n * 16);
  float* boxes = (float*)take((size_t)nboxes * 6 * 4);

  knn_bounds_init_kernel<<<1, 256, 0, s>>>(bounds);
  const size_t gb_all = (n + 255) / 256;
  const int gb = (int)(gb_all < 2048 ? gb_all : 2048);
  knn_bounds_kernel<<<gb, 256, 0, s>>>(P, points, bounds);
  knn_morton_kernel<<<(P + 255) / 256, 256, 0, s>>>(P, points, bounds, key_a);
  const bool in_b = radix_sort_pairs(s, (uint32_t)P, nullptr, key_a, val_a, key_b, val_b, true, 0, 30,
                                     table, totals, (uint32_t)nchunks, false);
  const uint32_t* order = in_b ? val_b : val_a;
  knn_gather_boxes_kernel<<<nboxes, 256, 0, s>>>(P, points, order, spts, boxes);
  knn_search_kernel<<<(P + KNN_TILE - 1) / KNN_TILE, KNN_TILE, 0, s>>>(P, spts, boxes, nboxes,
                                                                      mean_dists);
}

}  // namespace grpg
