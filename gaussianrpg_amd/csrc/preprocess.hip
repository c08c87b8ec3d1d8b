// Per-Gaussian preprocess for gfx950: near cull, projection, Sigma = R S^2 R^T, EWA 2x2
// covariance (+0.3 low-pass), conic, radius, tile rectangle, SH -> RGB.
//
// Replaces preprocessCUDA / filter_preprocessCUDA / checkFrustum of the reference
// (cuda_rasterizer/forward.cu:155-256, :259-334; rasterizer_impl.cu:54-66) and the device
// helpers they call (forward.cu:20-152, auxiliary.h:41-164).
//
// ARITHMETIC CONTRACT: this translation unit is compiled with -ffp-contract=off and every
// expression below is written in the evaluation order of the reference source (glm 0.9.9.9
// matrix products term by term), so that radii, tile rectangles, tiles_touched and the depth
// sort keys are BIT-EXACT against oracle/gs_oracle.c.  Division and sqrt are IEEE-correct in
// HIP by default (-fhip-fp32-correctly-rounded-divide-sqrt).  Do not "simplify" anything here.
//
// Memory: one thread per Gaussian, 256-thread workgroups.  Inputs arrive AoS ([P,3], [P,4],
// [P,M,3]) straight from the caller's torch.cat; a wave's 64 lanes read 64 consecutive
// records, so every cache line fetched is fully used.  Output is one 48-byte record per
// visible Gaussian (common.h) + radii + depth key + tiles_touched.  The kernel is HBM-bound:
// algorithmic bytes = P*(44+12M) read + 12 B (+48 B if visible) written per Gaussian.
#include "common.h"

#pragma clang fp contract(off)

#include "gaussian_math.h"
#include "compose_math.h"

namespace grpg {

// Everything up to the tile rectangle.  Returns false if the Gaussian is culled
// (near plane, det==0, empty rectangle) -- forward.cu:192-237.
__device__ __forceinline__ bool project_and_bound(const int idx, const float mx, const float my,
                                                  const float mz, const float s0, const float s1,
                                                  const float s2, const float scale_modifier,
                                                  const float* __restrict__ rotations,
                                                  const float* __restrict__ cov3D_precomp,
                                                  const float* __restrict__ view,
                                                  const float* __restrict__ proj, const int W,
                                                  const int H, const int gx, const int gy,
                                                  const float tan_fovx, const float tan_fovy,
                                                  const float focal_x, const float focal_y,
                                                  Projected& o, const bool q_given = false,
                                                  const float4 q_val = make_float4(1.f, 0.f, 0.f, 0.f)) {
  // in_frustum, auxiliary.h:139-164
  const float hx = proj[0] * mx + proj[4] * my + proj[8] * mz + proj[12];
  const float hy = proj[1] * mx + proj[5] * my + proj[9] * mz + proj[13];
  const float hw = proj[3] * mx + proj[7] * my + proj[11] * mz + proj[15];
  const float p_w = 1.0f / (hw + 0.0000001f);
  const float projx = hx * p_w, projy = hy * p_w;
  const float vz = view[2] * mx + view[6] * my + view[10] * mz + view[14];
  if (vz <= 0.2f) return false;
  if (cov3D_precomp != nullptr) {
#pragma unroll
    for (int i = 0; i < 6; i++) o.cov3d[i] = cov3D_precomp[6 * idx + i];
  } else {
    const float4 q = q_given ? q_val : load_quat(rotations, idx);
    cov3d_from_scale_rot(s0, s1, s2, scale_modifier, q, o.cov3d);
  }
  Cov2D cv;
  cov2d_project(mx, my, mz, view, focal_x, focal_y, tan_fovx, tan_fovy, o.cov3d, cv);
  const float cov_x = cv.a + 0.3f, cov_y = cv.b, cov_z = cv.c + 0.3f;
  const float det = (cov_x * cov_z - cov_y * cov_y);
  if (det == 0.0f) return false;
  const float det_inv = 1.f / det;
  o.conic[0] = cov_z * det_inv;
  o.conic[1] = -cov_y * det_inv;
  o.conic[2] = cov_x * det_inv;
  const float mid = 0.5f * (cov_x + cov_z);
  const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
  const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
  const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
  o.px = ndc2pix(projx, W);
  o.py = ndc2pix(projy, H);
  o.radius = f2i_rz(my_radius);
  get_rect(o.px, o.py, o.radius, gx, gy, o.minx, o.miny, o.maxx, o.maxy);
  if ((uint32_t)(o.maxx - o.minx) * (uint32_t)(o.maxy - o.miny) == 0) return false;
  o.depth = vz;
  return true;
}

// SH -> RGB, forward.cu:20-71.  `sh` points at this Gaussian's M coefficients (x3 channels).
__device__ __forceinline__ void sh_to_rgb(const int deg, const float* __restrict__ sh,
                                          const float dx, const float dy, const float dz,
                                          float* rgb, uint32_t& clamped) {
  const float len = sqrtf(dx * dx + dy * dy + dz * dz);
  const float x = dx / len, y = dy / len, z = dz / len;
  clamped = 0;
#pragma unroll
  for (int c = 0; c < 3; c++) {
#define SH(k) sh[3 * (k) + c]
    float result = SH_C0 * SH(0);
    if (deg > 0) {
      result = result - SH_C1 * y * SH(1) + SH_C1 * z * SH(2) - SH_C1 * x * SH(3);
      if (deg > 1) {
        const float xx = x * x, yy = y * y, zz = z * z;
        const float xy = x * y, yz = y * z, xz = x * z;
        result = result + SH_C2[0] * xy * SH(4) + SH_C2[1] * yz * SH(5) +
                 SH_C2[2] * (2.0f * zz - xx - yy) * SH(6) + SH_C2[3] * xz * SH(7) +
                 SH_C2[4] * (xx - yy) * SH(8);
        if (deg > 2) {
          result = result + SH_C3[0] * y * (3.0f * xx - yy) * SH(9) + SH_C3[1] * xy * z * SH(10) +
                   SH_C3[2] * y * (4.0f * zz - xx - yy) * SH(11) +
                   SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * SH(12) +
                   SH_C3[4] * x * (4.0f * zz - xx - yy) * SH(13) +
                   SH_C3[5] * z * (xx - yy) * SH(14) + SH_C3[6] * x * (xx - 3.0f * yy) * SH(15);
        }
      }
    }
#undef SH
    result += 0.5f;
    if (result < 0) clamped |= (1u << c);
    rgb[c] = fmaxf(result, 0.0f);
  }
}

// Per-workgroup sums of the tile counts (instances, and (Gaussian, super-tile) pairs of the
// hierarchical binning) and min / max of the visible depth keys -> pre_counts[blockIdx].  They are
// added up right behind the preprocess launch (sort.hip: by the first pass of the depth sort, or by
// publish_counts_kernel), so num_rendered reaches the host ~100 us into the frame -- long before the
// host has finished enqueuing the frame's remaining launches -- instead of after the whole binning
// chain (rasterizer_impl.cu:284 reads it back after its InclusiveSum), and the depth sort learns the
// range of its keys.  No global atomics.
__device__ __forceinline__ void block_count_sums(const uint32_t my_tiles, const uint32_t my_st,
                                                 const uint32_t my_key /* CULLED_KEY: not visible */,
                                                 uint4* __restrict__ pre_counts) {
  __shared__ uint32_t s_cnt[16];
  uint32_t a = my_tiles, b = my_st, mn = my_key, mx = my_key == CULLED_KEY ? 0u : my_key;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    a += __shfl_xor(a, o);
    b += __shfl_xor(b, o);
    mn = min(mn, (uint32_t)__shfl_xor((int)mn, o));
    mx = max(mx, (uint32_t)__shfl_xor((int)mx, o));
  }
  if ((threadIdx.x & 63) == 0) {
    const uint32_t w = threadIdx.x >> 6;
    s_cnt[4 * w] = a; s_cnt[4 * w + 1] = b; s_cnt[4 * w + 2] = mn; s_cnt[4 * w + 3] = mx;
  }
  __syncthreads();
  if (threadIdx.x == 0)
    pre_counts[blockIdx.x] = make_uint4(s_cnt[0] + s_cnt[4] + s_cnt[8] + s_cnt[12],
                                        s_cnt[1] + s_cnt[5] + s_cnt[9] + s_cnt[13],
                                        min(min(s_cnt[2], s_cnt[6]), min(s_cnt[10], s_cnt[14])),
                                        max(max(s_cnt[3], s_cnt[7]), max(s_cnt[11], s_cnt[15])));
}

// Layered frames (common.h TileObjBits): an OBJECT-class Gaussian flags every tile of its rectangle (plain byte
// stores of 1: every writer stores the same value, nothing is read back inside this launch).  A rectangle of up to
// MARK_SMALL tiles is flagged by its own thread; larger ones (a metre-sized splat near the camera covers thousands
// of tiles: one thread looping over them held its whole workgroup for 40 us) are handed to the wave.
constexpr int MARK_SMALL = 16;
struct MarkRect { int minx, miny, maxx, maxy; };
__device__ __forceinline__ void mark_object_tiles(const LayerMarks lm, const int minx, const int miny,
                                                  const int maxx, const int maxy) {
  for (int ty = miny; ty < maxy; ty++)
    for (int tx = minx; tx < maxx; tx++) lm.tile_obj[(size_t)ty * lm.gx + tx] = 1;
}
// all lanes of the wave (reconverged): the rectangles of the lanes with `big`, 64 tiles per step
__device__ __forceinline__ void mark_object_tiles_wave(const LayerMarks lm, const bool big, const MarkRect r) {
  const int lane = (int)(threadIdx.x & 63);
  for (uint64_t todo = __ballot(big); todo != 0ull; todo &= todo - 1ull) {
    const int l = (int)__builtin_ctzll(todo);
    const int x0 = __builtin_amdgcn_readlane(r.minx, l), y0 = __builtin_amdgcn_readlane(r.miny, l);
    const int w = __builtin_amdgcn_readlane(r.maxx, l) - x0, h = __builtin_amdgcn_readlane(r.maxy, l) - y0;
    if (w >= 64) {          // rows of >= 64 tiles: lanes along the row
      for (int ty = 0; ty < h; ty++)
        for (int tx = lane; tx < w; tx += 64) lm.tile_obj[(size_t)(y0 + ty) * lm.gx + x0 + tx] = 1;
    } else {                // narrow: 64 / w rows per step (w < 64: float division is exact enough, corrected)
      const int n = w * h;
      for (int i = lane; i < n; i += 64) {
        int ty = (int)((float)i * __builtin_amdgcn_rcpf((float)w));
        ty -= (ty * w > i) ? 1 : 0;
        ty += ((ty + 1) * w <= i) ? 1 : 0;
        lm.tile_obj[(size_t)(y0 + ty) * lm.gx + x0 + (i - ty * w)] = 1;
      }
    }
  }
}

__global__ void __launch_bounds__(256)
preprocess_kernel(const int P, const int D, const int M, const float* __restrict__ means3D,
                  const float* __restrict__ scales, const float scale_modifier,
                  const float* __restrict__ rotations, const float* __restrict__ opacities,
                  const float* __restrict__ shs, const float* __restrict__ cov3D_precomp,
                  const float* __restrict__ colors_precomp, const float* __restrict__ view,
                  const float* __restrict__ proj, const float* __restrict__ campos, const int W,
                  const int H, const int gx, const int gy, const float tan_fovx,
                  const float tan_fovy, const float focal_x, const float focal_y,
                  int* __restrict__ radii, float4* __restrict__ rec,
                  uint32_t* __restrict__ depth_key, uint32_t* __restrict__ tiles,
                  uint2* __restrict__ rects /* packed tile rectangles (hierarchical binning), or NULL */,
                  uint32_t* __restrict__ ds_table0 /* [chunk][DS_RADIX] pass-0 counts of the fat depth sort, or NULL */,
                  uint4* __restrict__ pre_counts /* [workgroups] (instances, coarse pairs, min key, max key) */,
                  const int vec_ok /* means3D, scales, shs are 16-byte aligned */, const LayerMarks lm) {
  // The 64-byte records leave through LDS: every thread stages its record, then the workgroup
  // stores the 16 KB slab with consecutive lanes on consecutive 16-byte pieces (whole sectors, fully
  // coalesced; a lane-per-record store would touch 64 sectors per instruction).  Culled Gaussians
  // get an all-zero record (radius 0, opacity 0): nothing ever gathers it.
  __shared__ float4 s_out[256 * REC_STRIDE];
  // means3D / scales arrive as [P,3] fp32: a lane-per-Gaussian read is three stride-12-byte dword
  // loads.  Stage the workgroup's 256 x 12 B = 3 KB per array through LDS with 16-byte loads
  // instead (the slab of workgroup b starts at byte 3072*b, so it is 16-byte aligned).  The staging
  // area lives in the record slab's space (dead until the records are staged, a barrier in between):
  // 18.5 KB of LDS per workgroup = 8 workgroups per CU instead of 6.
  float* const s_mean = reinterpret_cast<float*>(s_out);
  float* const s_scale = s_mean + 256 * 3;
  const int base = blockIdx.x * 256;
  const int nval = min(256, P - base) * 3;   // floats in this workgroup's slab
  {
    const int t4 = threadIdx.x * 4;
    if (vec_ok && t4 + 3 < nval) {
      reinterpret_cast<float4*>(s_mean)[threadIdx.x] =
          reinterpret_cast<const float4*>(means3D + (size_t)base * 3)[threadIdx.x];
      if (scales != nullptr)
        reinterpret_cast<float4*>(s_scale)[threadIdx.x] =
            reinterpret_cast<const float4*>(scales + (size_t)base * 3)[threadIdx.x];
    } else if (t4 < nval) {
      for (int e = t4; e < min(t4 + 4, nval); e++) {
        s_mean[e] = means3D[(size_t)base * 3 + e];
        if (scales != nullptr) s_scale[e] = scales[(size_t)base * 3 + e];
      }
    }
  }
  __syncthreads();
  // Pass-0 digit histogram of the depth sort (sort.hip, fat passes): the low DS_BITS bits of the
  // depth key of every VISIBLE Gaussian, aggregated per workgroup in LDS; the 256 Gaussians of a
  // workgroup lie in one 8192-key chunk, so it costs one global atomic per non-empty digit.
  __shared__ uint32_t s_dh[DS_RADIX];
#pragma unroll
  for (int k = 0; k < DS_RADIX / 256; k++) s_dh[k * 256 + threadIdx.x] = 0u;
  const int idx = base + threadIdx.x;
  float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0;
  uint32_t my_key = CULLED_KEY, my_tiles = 0u, my_st = 0u;
  bool mark_big = false;   // layered frame: an object Gaussian whose rectangle the wave flags together
  MarkRect mark_rect = {0, 0, 0, 0};
  float mx = 0.f, my = 0.f, mz = 0.f, s0 = 0.f, s1 = 0.f, s2 = 0.f;
  if (idx < P) {
    mx = s_mean[3 * threadIdx.x]; my = s_mean[3 * threadIdx.x + 1]; mz = s_mean[3 * threadIdx.x + 2];
    if (scales != nullptr) {
      s0 = s_scale[3 * threadIdx.x]; s1 = s_scale[3 * threadIdx.x + 1]; s2 = s_scale[3 * threadIdx.x + 2];
    }
  }
  __syncthreads();   // every thread holds its inputs: the staging area may become the record slab
  if (idx < P) {
    // degree <= 1 (every shipped config): the 48 B of SH coefficients as three 16-byte loads, issued
    // BEFORE the projection arithmetic so that their latency hides behind it (a culled Gaussian's
    // coefficients are then loaded for nothing: + 7 % of the kernel's bytes)
    const bool sh_fast = colors_precomp == nullptr && M == 4 && vec_ok;
    float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0, q2 = q0;
    if (sh_fast) {
      const float4* sp = reinterpret_cast<const float4*>(shs + (size_t)idx * 12);
      q0 = sp[0]; q1 = sp[1]; q2 = sp[2];
    }
    const float opac = opacities[idx];
    const bool q_early = cov3D_precomp == nullptr;   // the quaternion too: one 16-byte load in flight early
    const float4 quat = q_early ? load_quat(rotations, idx) : make_float4(1.f, 0.f, 0.f, 0.f);
    Projected o;
    const bool vis = project_and_bound(idx, mx, my, mz, s0, s1, s2, scale_modifier, rotations,
                                       cov3D_precomp, view, proj, W, H, gx, gy, tan_fovx, tan_fovy,
                                       focal_x, focal_y, o, q_early, quat);
    if (!vis) {
      radii[idx] = 0;
      tiles[idx] = 0;
      depth_key[idx] = CULLED_KEY;
    } else {
      float rgb[3];
      uint32_t clamped = 0;
      if (colors_precomp == nullptr) {
        const float dx = mx - campos[0];
        const float dy = my - campos[1];
        const float dz = mz - campos[2];
        if (sh_fast) {
          const float sh12[12] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};
          sh_to_rgb(D > 1 ? 1 : D, sh12, dx, dy, dz, rgb, clamped);
        } else {
          sh_to_rgb(D, shs + (size_t)idx * M * 3, dx, dy, dz, rgb, clamped);
        }
      } else {
        rgb[0] = colors_precomp[3 * idx];
        rgb[1] = colors_precomp[3 * idx + 1];
        rgb[2] = colors_precomp[3 * idx + 2];
      }
      radii[idx] = o.radius;
      my_tiles = (uint32_t)(o.maxy - o.miny) * (uint32_t)(o.maxx - o.minx);
      tiles[idx] = my_tiles;
      if (rects) {   // x and y tile ranges, half-open, 16 bits each (hier_binning.hip)
        const uint2 rc = make_uint2((uint32_t)o.minx | ((uint32_t)o.maxx << 16),
                                    (uint32_t)o.miny | ((uint32_t)o.maxy << 16));
        rects[idx] = rc;
        my_st = rect_super_tiles(rc.x, rc.y);
      }
      uint32_t cls_bit = 0u;   // layered frame: the class rides in the record (common.h REC_CLASS_BIT)
      if (lm.tile_obj != nullptr && lm.layer_class[idx] != 0) {
        if (my_tiles <= (uint32_t)MARK_SMALL) mark_object_tiles(lm, o.minx, o.miny, o.maxx, o.maxy);
        else { mark_big = true; mark_rect = MarkRect{o.minx, o.miny, o.maxx, o.maxy}; }
        cls_bit = REC_CLASS_BIT;
      }
      depth_key[idx] = __float_as_uint(o.depth);
      my_key = __float_as_uint(o.depth);
      r0 = make_float4(o.px, o.py, opac, __uint_as_float((uint32_t)o.radius | cls_bit));
      r1 = make_float4(o.conic[0], o.conic[1], o.conic[2], o.depth);
      r2 = make_float4(rgb[0], rgb[1], rgb[2], __uint_as_float(clamped));
    }
  }
  if (lm.tile_obj != nullptr) mark_object_tiles_wave(lm, mark_big, mark_rect);
  s_out[REC_STRIDE * threadIdx.x + 0] = r0;
  s_out[REC_STRIDE * threadIdx.x + 1] = r1;
  s_out[REC_STRIDE * threadIdx.x + 2] = r2;
  s_out[REC_STRIDE * threadIdx.x + 3] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();   // also orders the zeroing of s_dh before the atomics below
  if (ds_table0 != nullptr && my_key != CULLED_KEY) atomicAdd(&s_dh[my_key & (DS_RADIX - 1)], 1u);
  const int nrec4 = min(256, P - base) * REC_STRIDE;
  float4* dst = rec + (size_t)REC_STRIDE * base;
#pragma unroll
  for (int k = 0; k < REC_STRIDE; k++) {
    const int e = k * 256 + threadIdx.x;
    // a culled Gaussian's record (radius bits 0) is never read: its sector is not written
    if (e < nrec4 && __float_as_int(s_out[e & ~(REC_STRIDE - 1)].w) != 0) dst[e] = s_out[e];
  }
  if (ds_table0 != nullptr) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < DS_RADIX / 256; k++) {
      const uint32_t cnt = s_dh[k * 256 + threadIdx.x];
      if (cnt != 0u)
        atomicAdd(&ds_table0[(size_t)(blockIdx.x / (DS_CHUNK / 256)) * DS_RADIX + k * 256 + threadIdx.x], cnt);
    }
  }
  if (pre_counts != nullptr) block_count_sums(my_tiles, my_st, my_key, pre_counts);
}

__global__ void __launch_bounds__(256)
visible_filter_kernel(const int P, const float* __restrict__ means3D,
                      const float* __restrict__ scales, const float scale_modifier,
                      const float* __restrict__ rotations,
                      const float* __restrict__ cov3D_precomp, const float* __restrict__ view,
                      const float* __restrict__ proj, const int W, const int H, const int gx,
                      const int gy, const float tan_fovx, const float tan_fovy,
                      const float focal_x, const float focal_y, int* __restrict__ radii,
                      float* __restrict__ means2D) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= P) return;
  Projected o;
  const bool has_s = scales != nullptr;
  const bool vis = project_and_bound(idx, means3D[3 * idx], means3D[3 * idx + 1],
                                     means3D[3 * idx + 2], has_s ? scales[3 * idx] : 0.f,
                                     has_s ? scales[3 * idx + 1] : 0.f,
                                     has_s ? scales[3 * idx + 2] : 0.f, scale_modifier, rotations,
                                     cov3D_precomp, view, proj, W, H, gx, gy, tan_fovx, tan_fovy,
                                     focal_x, focal_y, o);
  radii[idx] = vis ? o.radius : 0;
  if (vis) {  // forward.cu:331-333: only visible Gaussians get a screen position
    means2D[2 * idx] = o.px;
    means2D[2 * idx + 1] = o.py;
  }
}

__global__ void __launch_bounds__(256)
mark_visible_kernel(const int P, const float* __restrict__ means3D,
                    const float* __restrict__ view, unsigned char* __restrict__ present) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= P) return;
  const float mx = means3D[3 * idx], my = means3D[3 * idx + 1], mz = means3D[3 * idx + 2];
  const float vz = view[2] * mx + view[6] * my + view[10] * mz + view[14];
  present[idx] = (vz <= 0.2f) ? 0 : 1;
}

__global__ void __launch_bounds__(256)
compose_kernel(const int P, const int M, const SegmentDev* __restrict__ segs, const int nseg,
               float* __restrict__ means3D, float* __restrict__ scales,
               float* __restrict__ rotations, float* __restrict__ opacities,
               float* __restrict__ shs) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= P) return;
  const SegmentDev& sg = *find_segment(segs, nseg, (uint32_t)idx);
  const uint32_t j = (uint32_t)idx - sg.start;
  const Activated a = compose_one(sg, j);
  means3D[3 * idx] = a.mx; means3D[3 * idx + 1] = a.my; means3D[3 * idx + 2] = a.mz;
  scales[3 * idx] = a.s0; scales[3 * idx + 1] = a.s1; scales[3 * idx + 2] = a.s2;
  rotations[4 * idx] = a.q.x; rotations[4 * idx + 1] = a.q.y; rotations[4 * idx + 2] = a.q.z;
  rotations[4 * idx + 3] = a.q.w;
  opacities[idx] = a.opacity;
  float* o = shs + (size_t)idx * M * 3;
  o[0] = a.dc[0]; o[1] = a.dc[1]; o[2] = a.dc[2];
  if (M > 1) {
    const float* fr = sg.frest + (size_t)j * (M - 1) * 3;
    for (int k = 0; k < (M - 1) * 3; k++) o[3 + k] = fr[k];
  }
}

// preprocess_kernel on composed inputs: same projection / EWA / SH / record code, the activated
// per-Gaussian values come out of compose_one() instead of the flat tensors.  M4: four SH
// coefficients per Gaussian (degree <= 1, every shipped config) stay in registers; the generic
// variant stages up to 16 in a per-lane array.
template <bool M4>
__global__ void __launch_bounds__(256)
preprocess_composed_kernel(const int P, const int D, const int M,
                           const SegmentDev* __restrict__ segs, const int nseg,
                           const float scale_modifier, const float* __restrict__ view,
                           const float* __restrict__ proj, const float* __restrict__ campos,
                           const int W, const int H, const int gx, const int gy,
                           const float tan_fovx, const float tan_fovy, const float focal_x,
                           const float focal_y, int* __restrict__ radii, float4* __restrict__ rec,
                           uint32_t* __restrict__ depth_key, uint32_t* __restrict__ tiles,
                           uint2* __restrict__ rects, uint32_t* __restrict__ ds_table0,
                           uint4* __restrict__ pre_counts, const LayerMarks lm) {
  __shared__ float4 s_out[256 * REC_STRIDE];
  __shared__ uint32_t s_dh[DS_RADIX];
#pragma unroll
  for (int k = 0; k < DS_RADIX / 256; k++) s_dh[k * 256 + threadIdx.x] = 0u;
  const int base = blockIdx.x * 256;
  const int idx = base + threadIdx.x;
  float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0;
  uint32_t my_key = CULLED_KEY, my_tiles = 0u, my_st = 0u;
  bool mark_big = false;   // layered frame: an object Gaussian whose rectangle the wave flags together
  MarkRect mark_rect = {0, 0, 0, 0};
  if (idx < P) {
    const SegmentDev& sg = *find_segment(segs, nseg, (uint32_t)idx);
    const uint32_t j = (uint32_t)idx - sg.start;
    const Activated a = compose_one(sg, j);
    Projected o;
    const bool vis = project_and_bound(idx, a.mx, a.my, a.mz, a.s0, a.s1, a.s2, scale_modifier,
                                       nullptr, nullptr, view, proj, W, H, gx, gy, tan_fovx, tan_fovy,
                                       focal_x, focal_y, o, true, a.q);
    if (!vis) {
      radii[idx] = 0;
      tiles[idx] = 0;
      depth_key[idx] = CULLED_KEY;
    } else {
      float rgb[3];
      uint32_t clamped = 0;
      const float dx = a.mx - campos[0];
      const float dy = a.my - campos[1];
      const float dz = a.mz - campos[2];
      if (M4) {
        const float* fr = sg.frest + (size_t)j * 9;
        const float sh12[12] = {a.dc[0], a.dc[1], a.dc[2], fr[0], fr[1], fr[2], fr[3], fr[4], fr[5],
                                fr[6], fr[7], fr[8]};
        sh_to_rgb(D > 1 ? 1 : D, sh12, dx, dy, dz, rgb, clamped);
      } else {
        float sh[48];
        compose_features(sg, j, a, M, sh);
        sh_to_rgb(D, sh, dx, dy, dz, rgb, clamped);
      }
      radii[idx] = o.radius;
      my_tiles = (uint32_t)(o.maxy - o.miny) * (uint32_t)(o.maxx - o.minx);
      tiles[idx] = my_tiles;
      if (rects) {   // x and y tile ranges, half-open, 16 bits each (hier_binning.hip)
        const uint2 rc = make_uint2((uint32_t)o.minx | ((uint32_t)o.maxx << 16),
                                    (uint32_t)o.miny | ((uint32_t)o.maxy << 16));
        rects[idx] = rc;
        my_st = rect_super_tiles(rc.x, rc.y);
      }
      uint32_t cls_bit = 0u;   // layered frame: the class rides in the record (common.h REC_CLASS_BIT)
      if (lm.tile_obj != nullptr && ((uint32_t)(uintptr_t)sg.pad1 & 1u)) {
        if (my_tiles <= (uint32_t)MARK_SMALL) mark_object_tiles(lm, o.minx, o.miny, o.maxx, o.maxy);
        else { mark_big = true; mark_rect = MarkRect{o.minx, o.miny, o.maxx, o.maxy}; }
        cls_bit = REC_CLASS_BIT;
      }
      depth_key[idx] = __float_as_uint(o.depth);
      my_key = __float_as_uint(o.depth);
      r0 = make_float4(o.px, o.py, a.opacity, __uint_as_float((uint32_t)o.radius | cls_bit));
      r1 = make_float4(o.conic[0], o.conic[1], o.conic[2], o.depth);
      r2 = make_float4(rgb[0], rgb[1], rgb[2], __uint_as_float(clamped));
    }
  }
  if (lm.tile_obj != nullptr) mark_object_tiles_wave(lm, mark_big, mark_rect);
  s_out[REC_STRIDE * threadIdx.x + 0] = r0;
  s_out[REC_STRIDE * threadIdx.x + 1] = r1;
  s_out[REC_STRIDE * threadIdx.x + 2] = r2;
  s_out[REC_STRIDE * threadIdx.x + 3] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  if (ds_table0 != nullptr && my_key != CULLED_KEY) atomicAdd(&s_dh[my_key & (DS_RADIX - 1)], 1u);
  const int nrec4 = min(256, P - base) * REC_STRIDE;
  float4* dst = rec + (size_t)REC_STRIDE * base;
#pragma unroll
  for (int k = 0; k < REC_STRIDE; k++) {
    const int e = k * 256 + threadIdx.x;
    // a culled Gaussian's record (radius bits 0) is never read: its sector is not written
    if (e < nrec4 && __float_as_int(s_out[e & ~(REC_STRIDE - 1)].w) != 0) dst[e] = s_out[e];
  }
  if (ds_table0 != nullptr) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < DS_RADIX / 256; k++) {
      const uint32_t cnt = s_dh[k * 256 + threadIdx.x];
      if (cnt != 0u)
        atomicAdd(&ds_table0[(size_t)(blockIdx.x / (DS_CHUNK / 256)) * DS_RADIX + k * 256 + threadIdx.x], cnt);
    }
  }
  if (pre_counts != nullptr) block_count_sums(my_tiles, my_st, my_key, pre_counts);
}

void launch_preprocess_composed(hipStream_t s, int P, int D, int M, const SegmentDev* segs, int nseg,
                                float scale_modifier, const CameraArgs& cam, int* radii, float4* rec,
                                uint32_t* depth_key, uint32_t* tiles, uint2* rects,
                                uint32_t* ds_table0, uint4* pre_counts, LayerMarks lm) {
  if (P <= 0) return;
#define PC_LAUNCH(M4)                                                                           \
  preprocess_composed_kernel<M4><<<(P + 255) / 256, 256, 0, s>>>(                                \
      P, D, M, segs, nseg, scale_modifier, cam.view, cam.proj, cam.campos, cam.W, cam.H, cam.gx,  \
      cam.gy, cam.tan_fovx, cam.tan_fovy, cam.focal_x, cam.focal_y, radii, rec, depth_key, tiles, \
      rects, ds_table0, pre_counts, lm)
  if (M == 4) PC_LAUNCH(true); else PC_LAUNCH(false);
#undef PC_LAUNCH
}

void launch_compose(hipStream_t s, int P, int M, const SegmentDev* segs, int nseg, float* means3D,
                    float* scales, float* rotations, float* opacities, float* shs) {
  if (P <= 0) return;
  compose_kernel<<<(P + 255) / 256, 256, 0, s>>>(P, M, segs, nseg, means3D, scales, rotations,
                                                  opacities, shs);
}

void launch_preprocess(hipStream_t s, int P, int D, int M, const float* means3D,
                       const float* scales, float scale_modifier, const float* rotations,
                       const float* opacities, const float* shs, const float* cov3D_precomp,
                       const float* colors_precomp, const CameraArgs& cam, int* radii,
                       float4* rec, uint32_t* depth_key, uint32_t* tiles, uint2* rects,
                       uint32_t* ds_table0, uint4* pre_counts, LayerMarks lm) {
  if (P <= 0) return;
  preprocess_kernel<<<(P + 255) / 256, 256, 0, s>>>(
      P, D, M, means3D, scales, scale_modifier, rotations, opacities, shs, cov3D_precomp,
      colors_precomp, cam.view, cam.proj, cam.campos, cam.W, cam.H, cam.gx, cam.gy, cam.tan_fovx,
      cam.tan_fovy, cam.focal_x, cam.focal_y, radii, rec, depth_key, tiles, rects, ds_table0,
      pre_counts, ((((uintptr_t)means3D | (uintptr_t)scales | (uintptr_t)shs) & 15) == 0) ? 1 : 0, lm);
}


void launch_visible_filter(hipStream_t s, int P, const float* means3D, const float* scales,
                           float scale_modifier, const float* rotations,
                           const float* cov3D_precomp, const CameraArgs& cam, int* radii,
                           float* means2D) {
  if (P <= 0) return;
  visible_filter_kernel<<<(P + 255) / 256, 256, 0, s>>>(
      P, means3D, scales, scale_modifier, rotations, cov3D_precomp, cam.view, cam.proj, cam.W,
      cam.H, cam.gx, cam.gy, cam.tan_fovx, cam.tan_fovy, cam.focal_x, cam.focal_y, radii, means2D);
}

void launch_mark_visible(hipStream_t s, int P, const float* means3D, const float* view,
                         unsigned char* present) {
  if (P <= 0) return;
  mark_visible_kernel<<<(P + 255) / 256, 256, 0, s>>>(P, means3D, view, present);
}

}  // namespace grpg
