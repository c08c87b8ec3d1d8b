// Per-tile front-to-back alpha blend for gfx950.
//
// Replaces renderCUDA<3> (cuda_rasterizer/forward.cu:340-467).  Same per-pixel semantics:
//   power = -1/2 (a dx^2 + c dy^2) - b dx dy ; skip if power > 0
//   alpha = min(0.99, opacity * exp(power))   ; skip if alpha < 1/255
//   stop  if T (1 - alpha) < 1e-4             ; else C += c alpha T, D += z alpha T, W += alpha T
//   out_color = C + T bg, out_alpha = W, out_depth = D, n_contrib = index of last blended splat.
//
// CDNA4 mapping (NOT the reference's 256-thread / one-pixel-per-thread / LDS-broadcast layout):
//   * one wave64 owns one 16x16 tile, 4 vertically adjacent pixels per lane.  Broadcasting a
//     splat's 40 bytes from LDS costs the same ~10 LDS cycles per wave whether a lane shades 1
//     pixel or 4, so 4 px/lane cuts LDS traffic per pixel-pair 4x, shares dx and the
//     a*dx^2 / b*dx terms across the 4 pixels, and gives 4 independent exp chains per lane.
//   * no __syncthreads anywhere: a workgroup is 4 independent waves (4 tiles); each wave stages
//     batches of 64 splat records (one coalesced-ish 48-byte gather per lane) into its private
//     3 KB LDS slice and then walks them with uniform (broadcast) ds_read_b128.
//   * wave-uniform early exit via ballot when all 256 pixels of the tile are saturated.
// The kernel is VALU-bound (about 20 flop-equivalents per pixel-splat pair), not HBM-bound;
// its compulsory HBM traffic is 4 B (id) + 48 B (record) per tile instance + 24 B per pixel.
#include "common.h"

namespace grpg {

constexpr int RW_WAVES = 4;  // tiles per workgroup

struct PixelState {
  float T, Cr, Cg, Cb, D, Wt;
  uint32_t last;
  bool done;
};

template <bool WRITE_AUX>
__global__ void __launch_bounds__(256)
render_forward_kernel(const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                      const float4* __restrict__ rec, const int W, const int H, const int gx,
                      const int ntiles, const float* __restrict__ bg,
                      float* __restrict__ out_color, float* __restrict__ out_depth,
                      float* __restrict__ out_alpha, uint32_t* __restrict__ n_contrib) {
  __shared__ float4 s_rec[RW_WAVES][WAVE * REC_F4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tile = blockIdx.x * RW_WAVES + wave;
  if (tile >= ntiles) return;  // whole wave exits together; no workgroup barriers are used
  const int ty = tile / gx, tx = tile - ty * gx;
  const int px = tx * TILE + (lane & 15);
  const int py0 = ty * TILE + (lane >> 4) * 4;
  const float pxf = (float)px;
  const uint2 range = ranges[tile];

  PixelState p[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    p[k].T = 1.0f; p[k].Cr = p[k].Cg = p[k].Cb = 0.f; p[k].D = 0.f; p[k].Wt = 0.f;
    p[k].last = 0;
    p[k].done = !(px < W && (py0 + k) < H);
  }
  float4* my = s_rec[wave];

  for (uint32_t base = range.x; base < range.y; base += WAVE) {
    const bool alldone = p[0].done && p[1].done && p[2].done && p[3].done;
    if (__ballot(!alldone) == 0ull) break;
    const uint32_t n = min((uint32_t)WAVE, range.y - base);
    if ((uint32_t)lane < n) {
      const uint32_t id = point_list[base + lane];
      const float4* r = rec + (size_t)id * REC_F4;
      const float4 a = r[0], b = r[1], c = r[2];
      my[lane * REC_F4 + 0] = a;
      my[lane * REC_F4 + 1] = b;
      my[lane * REC_F4 + 2] = c;
    }
    __builtin_amdgcn_wave_barrier();
    const uint32_t idx0 = base - range.x + 1;  // 1-based position in the tile's list
    for (uint32_t j = 0; j < n; j++) {
      const float4 a = my[j * REC_F4 + 0];   // px, py, depth, opacity
      const float4 b = my[j * REC_F4 + 1];   // conic.x, conic.y, conic.z, R
      const float4 c = my[j * REC_F4 + 2];   // G, B, -, -
      const float dx = a.x - pxf;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const float dy = a.y - (float)(py0 + k);
        const float power = -0.5f * (b.x * dx * dx + b.z * dy * dy) - b.y * dx * dy;
        const float alpha = fminf(0.99f, a.w * __expf(power));
        bool valid = !p[k].done && !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
        const float test_T = p[k].T * (1.0f - alpha);
        const bool term = valid && (test_T < 0.0001f);
        p[k].done = p[k].done || term;
        valid = valid && !term;
        const float w = valid ? alpha * p[k].T : 0.0f;
        p[k].Cr += b.w * w;
        p[k].Cg += c.x * w;
        p[k].Cb += c.y * w;
        p[k].D += a.z * w;
        p[k].Wt += w;
        p[k].T = valid ? test_T : p[k].T;
        p[k].last = valid ? (idx0 + j) : p[k].last;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }

  const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
  const size_t HW = (size_t)H * W;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int py = py0 + k;
    if (px < W && py < H) {
      const size_t pix = (size_t)py * W + px;
      out_color[pix] = p[k].Cr + p[k].T * bg0;
      out_color[HW + pix] = p[k].Cg + p[k].T * bg1;
      out_color[2 * HW + pix] = p[k].Cb + p[k].T * bg2;
      out_alpha[pix] = p[k].Wt;
      out_depth[pix] = p[k].D;
      if (WRITE_AUX) n_contrib[pix] = p[k].last;
    }
  }
}

// N-channel "semantic" planes (forward.cu:442-444): same traversal, NCH channels per launch
// accumulated in registers instead of the reference's per-contribution global read-modify-write.
template <int NCH>
__global__ void __launch_bounds__(256)
render_semantic_kernel(const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                       const float4* __restrict__ rec, const float* __restrict__ semantics,
                       const int S, const int c0, const int W, const int H, const int gx,
                       const int ntiles, float* __restrict__ out_semantic) {
  __shared__ float4 s_rec[RW_WAVES][WAVE * 2];
  __shared__ float s_sem[RW_WAVES][WAVE * NCH];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tile = blockIdx.x * RW_WAVES + wave;
  if (tile >= ntiles) return;
  const int ty = tile / gx, tx = tile - ty * gx;
  const int px = tx * TILE + (lane & 15);
  const int py0 = ty * TILE + (lane >> 4) * 4;
  const float pxf = (float)px;
  const uint2 range = ranges[tile];
  float T[4], acc[4][NCH];
  bool done[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    T[k] = 1.0f;
    done[k] = !(px < W && (py0 + k) < H);
#pragma unroll
    for (int c = 0; c < NCH; c++) acc[k][c] = 0.f;
  }
  float4* my = s_rec[wave];
  float* mysem = s_sem[wave];
  for (uint32_t base = range.x; base < range.y; base += WAVE) {
    const bool alldone = done[0] && done[1] && done[2] && done[3];
    if (__ballot(!alldone) == 0ull) break;
    const uint32_t n = min((uint32_t)WAVE, range.y - base);
    if ((uint32_t)lane < n) {
      const uint32_t id = point_list[base + lane];
      const float4* r = rec + (size_t)id * REC_F4;
      my[lane * 2 + 0] = r[0];
      my[lane * 2 + 1] = r[1];
#pragma unroll
      for (int c = 0; c < NCH; c++)
        mysem[lane * NCH + c] = (c0 + c < S) ? semantics[(size_t)id * S + c0 + c] : 0.f;
    }
    __builtin_amdgcn_wave_barrier();
    for (uint32_t j = 0; j < n; j++) {
      const float4 a = my[j * 2 + 0];
      const float4 b = my[j * 2 + 1];
      float sv[NCH];
#pragma unroll
      for (int c = 0; c < NCH; c++) sv[c] = mysem[j * NCH + c];
      const float dx = a.x - pxf;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const float dy = a.y - (float)(py0 + k);
        const float power = -0.5f * (b.x * dx * dx + b.z * dy * dy) - b.y * dx * dy;
        const float alpha = fminf(0.99f, a.w * __expf(power));
        bool valid = !done[k] && !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
        const float test_T = T[k] * (1.0f - alpha);
        const bool term = valid && (test_T < 0.0001f);
        done[k] = done[k] || term;
        valid = valid && !term;
        const float w = valid ? alpha * T[k] : 0.0f;
#pragma unroll
        for (int c = 0; c < NCH; c++) acc[k][c] += sv[c] * w;
        T[k] = valid ? test_T : T[k];
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  const size_t HW = (size_t)H * W;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int py = py0 + k;
    if (px < W && py < H) {
      const size_t pix = (size_t)py * W + px;
#pragma unroll
      for (int c = 0; c < NCH; c++)
        if (c0 + c < S) out_semantic[(size_t)(c0 + c) * HW + pix] = acc[k][c];
    }
  }
}

void launch_render_forward(hipStream_t s, const uint2* ranges, const uint32_t* point_list,
                           const float4* rec, int W, int H, int gx, int gy, const float* bg,
                           float* out_color, float* out_depth, float* out_alpha,
                           uint32_t* n_contrib) {
  const int ntiles = gx * gy;
  if (ntiles <= 0) return;
  const int blocks = (ntiles + RW_WAVES - 1) / RW_WAVES;
  render_forward_kernel<true><<<blocks, 256, 0, s>>>(ranges, point_list, rec, W, H, gx, ntiles,
                                                     bg, out_color, out_depth, out_alpha,
                                                     n_contrib);
}

void launch_render_semantic(hipStream_t s, const uint2* ranges, const uint32_t* point_list,
                            const float4* rec, const float* semantics, int S, int W, int H, int gx,
                            int gy, float* out_semantic) {
  const int ntiles = gx * gy;
  if (ntiles <= 0 || S <= 0) return;
  const int blocks = (ntiles + RW_WAVES - 1) / RW_WAVES;
  constexpr int NCH = 4;
  for (int c0 = 0; c0 < S; c0 += NCH)
    render_semantic_kernel<NCH><<<blocks, 256, 0, s>>>(ranges, point_list, rec, semantics, S, c0,
                                                       W, H, gx, ntiles, out_semantic);
}

}  // namespace grpg
