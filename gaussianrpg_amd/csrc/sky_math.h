// Sky cube-map lookup and the frame epilogue's arithmetic for gfx950 -- shared by sky.hip (the stand-alone composite
// launch) and render_fwd.hip (the same composite inside the render's epilogue, grpg_forward_frame): ONE body, and
// floating-point contraction switched OFF inside it, so that both launches produce the same bits whatever the
// surrounding code looks like (tests/test_gpu_frame.py compares bytes).  Semantics: csrc/sky.hip's header comment
// (lib/models/sky_cubemap.py:77-122, lib/utils/graphics_utils.py:186-207).
#pragma once
#include "common.h"

namespace grpg {

struct SkyArgs {
  const float* cube;   // [6, res, res, 3]
  int res;
  float m[9];          // row-major R^T K^-1: ray = m * (x + 0.5, y + 0.5, 1)
  float fill;          // sky colour where the mask is off (0, or 1 for a white background)
  int clamp_out;       // evaluation mode: clamp the composite to [0,1]
  // train-mode extras (sky_cubemap.py:80-82,91-92), all optional:
  const float* m_dev;          // the same 9 floats in DEVICE memory (no host read of K / w2c)
  const unsigned char* mask;   // [H,W] != 0 -> fetch the texture there (camera.original_sky_mask with
                               // its top rows set); replaces the (1 - acc) > 1e-3 rule
  const float* jitter;         // [2,H,W]: per-pixel (x, y) offsets replacing the +0.5 pixel centre
                               // (get_rays_torch(perturb=True): two torch.rand(H, W) planes)
};

// does this pixel fetch the texture?  sky_cubemap.py:80-87
__device__ __forceinline__ bool sky_fetches(const SkyArgs& s, const size_t pix, const bool have_acc,
                                            const float tr) {
#pragma clang fp contract(off)
  if (s.mask) return s.mask[pix] != 0;
  return !have_acc || tr > 1e-3f;
}

// direction -> face, (u, v) in [0,1]; returns -1 for a degenerate direction
__device__ __forceinline__ int cube_face_uv(const float x, const float y, const float z, float& u,
                                            float& v) {
#pragma clang fp contract(off)
  const float ax = fabsf(x), ay = fabsf(y), az = fabsf(z);
  int f;
  float mj, a, b;
  if (az > fmaxf(ax, ay)) { f = z < 0.f ? 5 : 4; mj = az; a = z < 0.f ? -x : x; b = -y; }
  else if (ay > ax)       { f = y < 0.f ? 3 : 2; mj = ay; a = x; b = y < 0.f ? -z : z; }
  else                    { f = x < 0.f ? 1 : 0; mj = ax; a = x < 0.f ? z : -z; b = -y; }
  const float h = 0.5f / mj;
  u = a * h + 0.5f;
  v = b * h + 0.5f;
  if (!(fabsf(u) < 3e38f) || !(fabsf(v) < 3e38f)) return -1;
  u = fminf(fmaxf(u, 0.f), 1.f);
  v = fminf(fmaxf(v, 0.f), 1.f);
  return f;
}

// face + face coordinates (a, b) in [-1,1]^2 (may exceed it by a little) -> the 3-D point on the
// cube, unfolded about the edge when ONE coordinate is outside: cube_to_dir (sky_cubemap.py:139-146)
// plus "walk round the edge".
__device__ __forceinline__ void cube_point(const int f, float a, float b, float& x, float& y, float& z) {
#pragma clang fp contract(off)
  float d = 0.f;   // distance walked beyond the edge, taken inward along -normal
  if (a > 1.f) { d = a - 1.f; a = 1.f; } else if (a < -1.f) { d = -1.f - a; a = -1.f; }
  if (b > 1.f) { d = b - 1.f; b = 1.f; } else if (b < -1.f) { d = -1.f - b; b = -1.f; }
  const float n = 1.f - d;
  switch (f) {
    case 0: x = n; y = -b; z = -a; break;
    case 1: x = -n; y = -b; z = a; break;
    case 2: x = a; y = n; z = b; break;
    case 3: x = a; y = -n; z = -b; break;
    case 4: x = a; y = -b; z = n; break;
    default: x = -a; y = -b; z = -n; break;
  }
}

struct CubeTaps {
  int idx[4];      // texel index (face * res + row) * res + col, -1 = no texel (corner)
  float w[4];      // renormalised bilinear weights
};

__device__ __forceinline__ CubeTaps cube_taps(const float dx, const float dy, const float dz,
                                              const int res) {
#pragma clang fp contract(off)
  CubeTaps t;
  float u, v;
  const int f = cube_face_uv(dx, dy, dz, u, v);
#pragma unroll
  for (int k = 0; k < 4; k++) { t.idx[k] = -1; t.w[k] = 0.f; }
  if (f < 0) return t;
  const float fu = u * (float)res - 0.5f, fv = v * (float)res - 0.5f;
  const float flu = floorf(fu), flv = floorf(fv);
  const int iu0 = (int)flu, iv0 = (int)flv;
  const float wu = fu - flu, wv = fv - flv;
  float wsum = 0.f;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int iu = iu0 + (k & 1), iv = iv0 + (k >> 1);
    const float w = ((k & 1) ? wu : 1.f - wu) * ((k >> 1) ? wv : 1.f - wv);
    const bool ou = iu < 0 || iu >= res, ov = iv < 0 || iv >= res;
    if (ou && ov) continue;                       // across a corner: no texel there
    int face = f, cu = iu, cv = iv;
    if (ou || ov) {                               // across one edge: the neighbouring face
      const float a = ((float)iu + 0.5f) * (2.f / (float)res) - 1.f;
      const float b = ((float)iv + 0.5f) * (2.f / (float)res) - 1.f;
      float px, py, pz, nu, nv;
      cube_point(f, a, b, px, py, pz);
      face = cube_face_uv(px, py, pz, nu, nv);
      if (face < 0) continue;
      cu = min(res - 1, max(0, (int)floorf(nu * (float)res)));
      cv = min(res - 1, max(0, (int)floorf(nv * (float)res)));
    }
    t.idx[k] = (face * res + cv) * res + cu;
    t.w[k] = w;
    wsum += w;
  }
  if (wsum > 0.f) {
    const float inv = 1.f / wsum;
#pragma unroll
    for (int k = 0; k < 4; k++) t.w[k] *= inv;
  }
  return t;
}

__device__ __forceinline__ void pixel_ray(const SkyArgs& s, const int px, const int py, const size_t pix,
                                          const size_t HW, float& dx, float& dy, float& dz) {
#pragma clang fp contract(off)
  const float ox = s.jitter ? s.jitter[pix] : 0.5f, oy = s.jitter ? s.jitter[HW + pix] : 0.5f;
  const float fx = (float)px + ox, fy = (float)py + oy;
  float m[9];
#pragma unroll
  for (int i = 0; i < 9; i++) m[i] = s.m_dev ? s.m_dev[i] : s.m[i];
  const float x = m[0] * fx + m[1] * fy + m[2];
  const float y = m[3] * fx + m[4] * fy + m[5];
  const float z = m[6] * fx + m[7] * fy + m[8];
  const float n = sqrtf(x * x + y * y + z * z);
  dx = x / n; dy = y / n; dz = z / n;
}


// colour of the sky behind pixel (px, py): the clamped cube-map sample where the pixel fetches it, `fill` elsewhere
// (sky_cubemap.py:84-87,100-120).  tr = 1 - acc.
__device__ __forceinline__ void sky_pixel(const SkyArgs& s, const int px, const int py, const size_t pix,
                                          const size_t HW, const bool have_acc, const float tr, float (&sky)[3]) {
#pragma clang fp contract(off)
  sky[0] = sky[1] = sky[2] = s.fill;
  if (sky_fetches(s, pix, have_acc, tr)) {
    float dx, dy, dz;
    pixel_ray(s, px, py, pix, HW, dx, dy, dz);
    const CubeTaps t = cube_taps(dx, dy, dz, s.res);
    sky[0] = sky[1] = sky[2] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (t.idx[k] >= 0) {
        const float* c = s.cube + 3 * (size_t)t.idx[k];
        sky[0] += t.w[k] * c[0]; sky[1] += t.w[k] * c[1]; sky[2] += t.w[k] * c[2];
      }
    }
  }
#pragma unroll
  for (int c = 0; c < 3; c++) sky[c] = clamp01(sky[c]);   // sky_cubemap.py:103,120
}

// rgb + sky * (1 - acc), clamped to [0, 1] outside train mode (street_gaussian_renderer.py:110,116)
__device__ __forceinline__ float sky_over(const float rgb, const float sky, const float tr, const int clamp_out) {
#pragma clang fp contract(off)
  const float v = rgb + sky * tr;
  return clamp_out ? clamp01(v) : v;
}

}  // namespace grpg
