// Backward of the per-tile blend for gfx950.
//
// Replaces renderCUDA<3,20> backward (cuda_rasterizer/backward.cu:415-641): back-to-front
// re-traversal of each tile's list starting from n_contrib, T recovered by division (:547),
// gradients w.r.t. colour / depth / semantic / accumulated alpha / background (:556-614) pushed to
// (mean2D, conic, opacity, colour, depth, semantic) of every contributing Gaussian (:618-638).
//
// CDNA4 mapping: the reference issues 11+S float atomics per (pixel, Gaussian) pair.  Here one
// wave64 owns a 16x16 tile with 4 pixels per lane (same layout as the forward); the 11+S partial
// gradients of a splat are summed over the lane's 4 pixels in registers, reduced across the wave
// with 6 DPP adds each (quad_perm, row_half_mirror, row_mirror, row_bcast15, row_bcast31 -- no LDS
// traffic) and lane 63 issues ONE hardware global_atomic_add_f32 per field per (tile, Gaussian):
// 256x fewer atomics than the reference.  Summation order differs from the reference's
// (unspecified) atomic order, so gradients agree to rounding, not bitwise -- exactly as two runs of
// the reference differ from each other.
#include "blend_math.h"
#include "common.h"

namespace grpg {

constexpr int RB_WAVES = 4;

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_step(float v) {
  const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false);
  return v + __int_as_float(t);
}

// Sum over the 64 lanes; the total is valid in lane 63.
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
  v = dpp_step<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
  v = dpp_step<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
  v = dpp_step<0x141, 0xF>(v);  // row_half_mirror
  v = dpp_step<0x140, 0xF>(v);  // row_mirror      -> every lane holds its row's (16-lane) sum
  v = dpp_step<0x142, 0xA>(v);  // row_bcast:15 into rows 1,3
  v = dpp_step<0x143, 0xC>(v);  // row_bcast:31 into rows 2,3 -> lane 63 = total
  return v;
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d, 64));
  return v;
}

template <int SMAX>
__global__ void __launch_bounds__(256)
render_backward_kernel(const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                       const float4* __restrict__ rec, const float* __restrict__ semantics,
                       const int S, const int W, const int H, const int gx, const int ntiles,
                       const float* __restrict__ bg, const float* __restrict__ alphas,
                       const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpix,
                       const float* __restrict__ dL_dpix_depth,
                       const float* __restrict__ dL_dalphas,
                       const float* __restrict__ dL_dpix_semantic, float* __restrict__ dL_dmean2D,
                       float* __restrict__ dL_dconic, float* __restrict__ dL_dopacity,
                       float* __restrict__ dL_dcolor, float* __restrict__ dL_ddepth,
                       float* __restrict__ dL_dsemantic) {
  __shared__ float4 s_rec[RB_WAVES][WAVE * REC_F4];
  __shared__ uint32_t s_id[RB_WAVES][WAVE];
  constexpr int SM = SMAX > 0 ? SMAX : 1;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tile = blockIdx.x * RB_WAVES + wave;
  if (tile >= ntiles) return;
  const int ty = tile / gx, tx = tile - ty * gx;
  const int px = tx * TILE + (lane & 15);
  const int py0 = ty * TILE + (lane >> 4) * 4;
  const float pxf = (float)px;
  const uint2 range = ranges[tile];
  const size_t HW = (size_t)H * W;
  const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];

  float T[4], T_final[4], dLr[4], dLg[4], dLb[4], dLd[4], dLa[4], bgdot[4];
  float acc_r[4], acc_g[4], acc_b[4], acc_d[4], acc_a[4], last_alpha[4], last_r[4], last_g[4],
      last_b[4], last_d[4];
  float dLs[4][SM], acc_s[4][SM], last_s[4][SM];
  uint32_t lastc[4];
  uint32_t maxlast = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int py = py0 + k;
    const bool inside = px < W && py < H;
    const size_t pix = inside ? (size_t)py * W + px : 0;
    T_final[k] = inside ? 1.0f - alphas[pix] : 0.f;
    T[k] = T_final[k];
    lastc[k] = inside ? n_contrib[pix] : 0u;
    dLr[k] = inside ? dL_dpix[pix] : 0.f;
    dLg[k] = inside ? dL_dpix[HW + pix] : 0.f;
    dLb[k] = inside ? dL_dpix[2 * HW + pix] : 0.f;
    dLd[k] = inside ? dL_dpix_depth[pix] : 0.f;
    dLa[k] = inside ? dL_dalphas[pix] : 0.f;
    bgdot[k] = bg0 * dLr[k] + bg1 * dLg[k] + bg2 * dLb[k];
    acc_r[k] = acc_g[k] = acc_b[k] = acc_d[k] = acc_a[k] = 0.f;
    last_alpha[k] = last_r[k] = last_g[k] = last_b[k] = last_d[k] = 0.f;
#pragma unroll
    for (int c = 0; c < SM; c++) {
      dLs[k][c] = (SMAX > 0 && c < S && inside) ? dL_dpix_semantic[(size_t)c * HW + pix] : 0.f;
      acc_s[k][c] = 0.f;
      last_s[k][c] = 0.f;
    }
    maxlast = max(maxlast, lastc[k]);
  }
  maxlast = wave_max_u32(maxlast);   // nothing behind the tile's deepest contributor matters
  const float ddelx_dx = (float)(0.5 * W), ddely_dy = (float)(0.5 * H);   // backward.cu:501-502

  float4* my = s_rec[wave];
  uint32_t* myid = s_id[wave];
  const uint32_t count = min(range.y - range.x, maxlast);
  for (uint32_t hi = count; hi > 0;) {
    const uint32_t n = hi >= (uint32_t)WAVE ? (uint32_t)WAVE : hi;
    const uint32_t lo = hi - n;   // batch covers list positions [lo, hi)
    if ((uint32_t)lane < n) {
      const uint32_t id = point_list[range.x + lo + lane] & ID_MASK;
      const float4* r = rec + (size_t)id * REC_F4;
      my[lane * REC_F4 + 0] = r[0];
      my[lane * REC_F4 + 1] = r[1];
      my[lane * REC_F4 + 2] = r[2];
      myid[lane] = id;
    }
    __builtin_amdgcn_wave_barrier();
    for (int j = (int)n - 1; j >= 0; j--) {
      const float4 a = my[j * REC_F4 + 0];   // px, py, depth, opacity
      const float4 b = my[j * REC_F4 + 1];   // conic.x, conic.y, conic.z, R
      const float4 c = my[j * REC_F4 + 2];   // G, B
      const uint32_t gid = myid[j];
      const uint32_t pos = lo + (uint32_t)j;   // 0-based position == reference's `contributor`
      const float dx = a.x - pxf;
      const SplatTerms st = splat_terms(dx, b.x, b.y, b.z);
      float g_mx = 0.f, g_my = 0.f, g_mabs = 0.f, g_cx = 0.f, g_cy = 0.f, g_cw = 0.f, g_op = 0.f;
      float g_r = 0.f, g_g = 0.f, g_b = 0.f, g_d = 0.f;
      float g_s[SM];
#pragma unroll
      for (int cc = 0; cc < SM; cc++) g_s[cc] = 0.f;
      bool any = false;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const float dy = a.y - (float)(py0 + k);
        float G, alpha;   // identical arithmetic to the forward (blend_math.h)
        const bool valid = pair_alpha(pair_power(st, dy), a.w, G, alpha) && (pos < lastc[k]);
        if (valid) {
          any = true;
          T[k] = T[k] / (1.f - alpha);
          const float dch = alpha * T[k];
          float dL_dopa = 0.f;
          acc_r[k] = last_alpha[k] * last_r[k] + (1.f - last_alpha[k]) * acc_r[k];
          last_r[k] = b.w;
          dL_dopa += (b.w - acc_r[k]) * dLr[k];
          g_r += dch * dLr[k];
          acc_g[k] = last_alpha[k] * last_g[k] + (1.f - last_alpha[k]) * acc_g[k];
          last_g[k] = c.x;
          dL_dopa += (c.x - acc_g[k]) * dLg[k];
          g_g += dch * dLg[k];
          acc_b[k] = last_alpha[k] * last_b[k] + (1.f - last_alpha[k]) * acc_b[k];
          last_b[k] = c.y;
          dL_dopa += (c.y - acc_b[k]) * dLb[k];
          g_b += dch * dLb[k];
          if (SMAX > 0) {
#pragma unroll
            for (int cc = 0; cc < SM; cc++) {
              if (cc < S) {
                const float sv = semantics[(size_t)gid * S + cc];
                acc_s[k][cc] = last_alpha[k] * last_s[k][cc] + (1.f - last_alpha[k]) * acc_s[k][cc];
                last_s[k][cc] = sv;
                dL_dopa += (sv - acc_s[k][cc]) * dLs[k][cc];
                g_s[cc] += dch * dLs[k][cc];
              }
            }
          }
          acc_d[k] = last_alpha[k] * last_d[k] + (1.f - last_alpha[k]) * acc_d[k];
          last_d[k] = a.z;
          dL_dopa += (a.z - acc_d[k]) * dLd[k];
          g_d += dch * dLd[k];
          acc_a[k] = last_alpha[k] + (1.f - last_alpha[k]) * acc_a[k];
          dL_dopa += (1.f - acc_a[k]) * dLa[k];
          dL_dopa *= T[k];
          last_alpha[k] = alpha;
          dL_dopa += (-T_final[k] / (1.f - alpha)) * bgdot[k];
          const float dL_dG = a.w * dL_dopa;
          const float gdx = G * dx, gdy = G * dy;
          const float dG_ddelx = -gdx * b.x - gdy * b.y;
          const float dG_ddely = -gdy * b.z - gdx * b.y;
          const float mx_ = dL_dG * dG_ddelx * ddelx_dx;
          const float my_ = dL_dG * dG_ddely * ddely_dy;
          g_mx += mx_;
          g_my += my_;
          g_mabs += fabsf(mx_) + fabsf(my_);
          g_cx += -0.5f * gdx * dx * dL_dG;
          g_cy += -0.5f * gdx * dy * dL_dG;
          g_cw += -0.5f * gdy * dy * dL_dG;
          g_op += G * dL_dopa;
        }
      }
      if (__ballot(any) == 0ull) continue;   // wave-uniform: nobody in the tile used this splat
      g_mx = wave_sum_to_lane63(g_mx);
      g_my = wave_sum_to_lane63(g_my);
      g_mabs = wave_sum_to_lane63(g_mabs);
      g_cx = wave_sum_to_lane63(g_cx);
      g_cy = wave_sum_to_lane63(g_cy);
      g_cw = wave_sum_to_lane63(g_cw);
      g_op = wave_sum_to_lane63(g_op);
      g_r = wave_sum_to_lane63(g_r);
      g_g = wave_sum_to_lane63(g_g);
      g_b = wave_sum_to_lane63(g_b);
      g_d = wave_sum_to_lane63(g_d);
      if (SMAX > 0) {
#pragma unroll
        for (int cc = 0; cc < SM; cc++) g_s[cc] = wave_sum_to_lane63(g_s[cc]);
      }
      if (lane == 63) {
        atomicAdd(&dL_dmean2D[3 * (size_t)gid + 0], g_mx);
        atomicAdd(&dL_dmean2D[3 * (size_t)gid + 1], g_my);
        atomicAdd(&dL_dmean2D[3 * (size_t)gid + 2], g_mabs);
        atomicAdd(&dL_dconic[4 * (size_t)gid + 0], g_cx);
        atomicAdd(&dL_dconic[4 * (size_t)gid + 1], g_cy);
        atomicAdd(&dL_dconic[4 * (size_t)gid + 3], g_cw);
        atomicAdd(&dL_dopacity[gid], g_op);
        atomicAdd(&dL_dcolor[3 * (size_t)gid + 0], g_r);
        atomicAdd(&dL_dcolor[3 * (size_t)gid + 1], g_g);
        atomicAdd(&dL_dcolor[3 * (size_t)gid + 2], g_b);
        atomicAdd(&dL_ddepth[gid], g_d);
        if (SMAX > 0) {
#pragma unroll
          for (int cc = 0; cc < SM; cc++)
            if (cc < S) atomicAdd(&dL_dsemantic[(size_t)gid * S + cc], g_s[cc]);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    hi = lo;
  }
}

void launch_render_backward(hipStream_t s, const uint2* ranges, const uint32_t* point_list,
                            const float4* rec, const float* semantics, int S, int W, int H, int gx,
                            int gy, const float* bg, const float* alphas,
                            const uint32_t* n_contrib, const float* dL_dpix,
                            const float* dL_dpix_depth, const float* dL_dalphas,
                            const float* dL_dpix_semantic, float* dL_dmean2D, float* dL_dconic,
                            float* dL_dopacity, float* dL_dcolor, float* dL_ddepth,
                            float* dL_dsemantic) {
  const int ntiles = gx * gy;
  if (ntiles <= 0) return;
  const int blocks = (ntiles + RB_WAVES - 1) / RB_WAVES;
#define RB_ARGS                                                                                 \
  ranges, point_list, rec, semantics, S, W, H, gx, ntiles, bg, alphas, n_contrib, dL_dpix,      \
      dL_dpix_depth, dL_dalphas, dL_dpix_semantic, dL_dmean2D, dL_dconic, dL_dopacity,          \
      dL_dcolor, dL_ddepth, dL_dsemantic
  if (S <= 0)
    render_backward_kernel<0><<<blocks, 256, 0, s>>>(RB_ARGS);
  else if (S <= 4)
    render_backward_kernel<4><<<blocks, 256, 0, s>>>(RB_ARGS);
  else
    render_backward_kernel<32><<<blocks, 256, 0, s>>>(RB_ARGS);   // S <= 32 (reference: 20)
#undef RB_ARGS
}

}  // namespace grpg
