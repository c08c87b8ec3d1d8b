// Pixel/splat pair arithmetic shared by the forward blend, the semantic blend and the backward
// (render_fwd.hip, render_bwd.hip), so that all three make IDENTICAL accept/reject decisions.
//
// Reference semantics (cuda_rasterizer/forward.cu:410-432, backward.cu:526-544):
//   power = -0.5 (a dx^2 + c dy^2) - b dx dy ;  reject if power > 0
//   alpha = min(0.99, opacity * exp(power))  ;  reject if alpha < 1/255
// Evaluated here as  power = hA - dy * (hc*dy + bdx)  with hA = -0.5 a dx^2, bdx = b dx,
// hc = 0.5 c  (two FMAs per pixel, dx-only terms shared by the pixels of a lane) and
// exp(x) = v_exp_f32(x * log2 e).  Differences to the oracle's unfused fp32 evaluation are a few
// ulp of the largest term; the parity tests allow the other branch only on pixels the oracle
// flags as threshold-fragile.
#pragma once
#include "common.h"

namespace grpg {

struct SplatTerms {   // per (lane, splat): everything that does not depend on the pixel row
  float hA, bdx, hc;
};

__device__ __forceinline__ SplatTerms splat_terms(const float dx, const float cx, const float cy,
                                                  const float cz) {
  SplatTerms t;
  t.hA = -0.5f * cx * dx * dx;
  t.bdx = cy * dx;
  t.hc = 0.5f * cz;
  return t;
}

__device__ __forceinline__ float pair_power(const SplatTerms& t, const float dy) {
  return fmaf(-dy, fmaf(t.hc, dy, t.bdx), t.hA);
}

// alpha and G = exp(power); returns true if the pair passes both reference tests.
__device__ __forceinline__ bool pair_alpha(const float power, const float opacity, float& G,
                                           float& alpha) {
  G = __expf(power);
  alpha = fminf(0.99f, opacity * G);
  return !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
}

// Conservative whole-rectangle cull.  Pixels px in [x0,x1], py in [y0,y1] (inclusive, pixel
// centres on integers).  Returns true when NO pixel of the rectangle can pass the alpha test:
//   max over the rectangle of opacity*exp(power)  <  1/255   (with a safety margin),
// using the exact minimum of the convex quadratic q = a dx^2 + 2b dx dy + c dy^2 over the
// continuous rectangle (a superset of the pixel centres).  q is minimised on the rectangle's
// boundary when the splat centre lies outside: four 1-D clamped minimisations.
// Never rejects on doubt: degenerate / non-finite conics are kept.
__device__ __forceinline__ bool splat_misses_rect(const float gx, const float gy, const float a,
                                                  const float b, const float c,
                                                  const float opacity, const float x0,
                                                  const float x1, const float y0, const float y1) {
  // Branch-free on purpose (predicates only): this runs once per list entry per wave.
  const float dxl = gx - x1, dxh = gx - x0, dyl = gy - y1, dyh = gy - y0;
  const bool centre_inside = dxl <= 0.0f && dxh >= 0.0f && dyl <= 0.0f && dyh >= 0.0f;
  // 1 ulp reciprocals are fine: the 1-D minimiser only has to be accurate to the safety margin
  const float ia = __builtin_amdgcn_rcpf(a), ic = __builtin_amdgcn_rcpf(c);
  const float nb = -b;
  float qmin;
  {
    const float dy0 = fminf(dyh, fmaxf(dyl, nb * dxl * ic));
    const float dy1 = fminf(dyh, fmaxf(dyl, nb * dxh * ic));
    const float dx0 = fminf(dxh, fmaxf(dxl, nb * dyl * ia));
    const float dx1 = fminf(dxh, fmaxf(dxl, nb * dyh * ia));
    const float q0 = a * dxl * dxl + (2.0f * b * dxl + c * dy0) * dy0;
    const float q1 = a * dxh * dxh + (2.0f * b * dxh + c * dy1) * dy1;
    const float q2 = c * dyl * dyl + (2.0f * b * dyl + a * dx0) * dx0;
    const float q3 = c * dyh * dyh + (2.0f * b * dyh + a * dx1) * dx1;
    qmin = fminf(fminf(q0, q1), fminf(q2, q3));
  }
  const float thr = 2.0f * __logf(255.0f * opacity);
  const bool far = qmin > thr + 1e-4f * fabsf(thr) + 1e-3f * (1.0f + fabsf(qmin) * 1e-3f);
  // alpha <= opacity * exp(power <= 0) <= opacity, so a splat below 1/255 never passes anywhere
  const bool transparent = opacity < (1.0f / 255.0f) * 0.999f;
  // never reject on doubt: NaNs, non-positive diagonal (degenerate conic), centre inside
  const bool sane = (a > 0.0f) && (c > 0.0f) && (qmin == qmin) && (opacity == opacity);
  return transparent || (sane && !centre_inside && far);
}

}  // namespace grpg
