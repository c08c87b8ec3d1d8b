// Pixel/splat pair arithmetic shared by the forward blend, the semantic blend and the backward
// (render_fwd.hip, render_bwd.hip), so that all three make IDENTICAL accept/reject decisions.
//
// Reference semantics (cuda_rasterizer/forward.cu:410-432, backward.cu:526-544):
//   power = -0.5 (a dx^2 + c dy^2) - b dx dy ;  reject if power > 0
//   alpha = min(0.99, opacity * exp(power))  ;  reject if alpha < 1/255
// Evaluated here in base 2 with the conic pre-scaled once per splat,
//   A = -0.5 log2(e) a,  B = -log2(e) b,  C = -0.5 log2(e) c          (splat_q, 3 multiplies)
//   power2 = log2(e) power = (A dx) dx + dy (C dy + B dx)             (3 mul + 2 fma per pair)
//   exp(power) = v_exp_f32(power2)                                     (no per-pair scaling)
// and power > 0  <=>  power2 > 0.  Every step is an explicit multiply or fma (nothing the compiler
// may contract differently in two kernels), so the forward, the semantic pass and the backward
// get bit-identical power2 / G / alpha whether they run the scalar or the packed (v_pk_*) form.
// Differences to the oracle's unfused natural-base fp32 evaluation are a few ulp of the largest
// term; the parity tests allow the other branch only on pixels the oracle flags as
// threshold-fragile.
#pragma once
#include "common.h"

namespace grpg {

typedef float v2f __attribute__((ext_vector_type(2)));

constexpr float LOG2E = 1.4426950408889634f;
constexpr float ALPHA_MIN = 1.0f / 255.0f;
constexpr float ALPHA_MAX = 0.99f;

struct SplatQ {       // pre-scaled conic of one splat
  float A, B, C;
};

__device__ __forceinline__ SplatQ splat_q(const float cx, const float cy, const float cz) {
  SplatQ q;
  q.A = (-0.5f * LOG2E) * cx;
  q.B = (-LOG2E) * cy;
  q.C = (-0.5f * LOG2E) * cz;
  return q;
}

// True when power2 as computed below cannot come out positive for ANY pixel offset, so that the reference's
// `power > 0` test (forward.cu:420) can be skipped for this splat without changing a single accept decision.
//   power2 = fma(dy, fma(C, dy, fl(B dx)), fl(fl(A dx) dx)),  A, C < 0;  its sign is the sign of
//   -(a dx^2 (1+d1)(1+d2) - B dx dy (1+d3)(1+d4) + c dy^2 (1+d4)),  a = -A, c = -C, |d_i| <= 2^-24,
// and the exact form a dx^2 - B dx dy + c dy^2 is >= lambda_min (dx^2 + dy^2), while the rounding terms are
// below 2.01 * 2^-24 (max(a, c) + |B| / 2) (dx^2 + dy^2).  The test asks for lambda_min > k = 2^-17 (a + c + |B|)
// -- sixty times that bound -- in its sqrt-free form: a > k, c > k, (a - k)(c - k) > B^2 / 4.  The floor 1e-12
// keeps the products far from the denormal range for pixel offsets >= 1e-4 (pixel centres are integers, splat
// centres floats of magnitude <= a few thousand); non-finite conics fail every comparison.  An exact zero offset
// gives power2 = +-0, which is not > 0 either.  Fails only for 2-D Gaussians of axis ratio beyond ~300 : 1.
__device__ __forceinline__ bool splat_power_never_positive(const SplatQ q) {
  const float a = -q.A, c = -q.C;
  const float k = fmaxf(7.62939453125e-06f * (a + c + fabsf(q.B)), 1e-12f);
  return a > k && c > k && (a - k) * (c - k) > 0.25f * q.B * q.B;
}

struct SplatTerms {   // per (lane, splat): everything that does not depend on the pixel row
  float hA, bdx, hc;
};

__device__ __forceinline__ SplatTerms splat_terms_q(const float dx, const SplatQ q) {
  SplatTerms t;
  t.hA = (q.A * dx) * dx;
  t.bdx = q.B * dx;
  t.hc = q.C;
  return t;
}

__device__ __forceinline__ SplatTerms splat_terms(const float dx, const float cx, const float cy,
                                                  const float cz) {
  return splat_terms_q(dx, splat_q(cx, cy, cz));
}

// power2 = log2(e) * power
__device__ __forceinline__ float pair_power(const SplatTerms& t, const float dy) {
  return fmaf(dy, fmaf(t.hc, dy, t.bdx), t.hA);
}

// alpha and G = exp(power) = 2^power2; returns true if the pair passes both reference tests.
__device__ __forceinline__ bool pair_alpha(const float power2, const float opacity, float& G,
                                           float& alpha) {
  G = __builtin_amdgcn_exp2f(power2);
  alpha = fminf(ALPHA_MAX, opacity * G);
  return !(power2 > 0.0f) && !(alpha < ALPHA_MIN);
}

// Two splats at once for one pixel (packed fp32: v_pk_mul_f32 / v_pk_fma_f32); lane-wise the same
// operations in the same order as the scalar form above, hence the same bits.
__device__ __forceinline__ v2f pair_power_x2(const v2f dx, const v2f dy, const v2f A, const v2f B,
                                             const v2f C) {
  const v2f hA = (A * dx) * dx;
  const v2f bdx = B * dx;
  return __builtin_elementwise_fma(dy, __builtin_elementwise_fma(C, dy, bdx), hA);
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

// Which of the four 16x4 quarters (rows y0+4q .. y0+4q+3, columns x0 .. x0+15) of a tile can the
// splat reach at all?  Bit q set = quarter q may hold a pixel that passes the alpha test.
// CONSERVATIVE (never clears a bit on doubt) and tight (tests/mask_check.py: 35.6 % of the
// (quarter, instance) pairs of the bench frame survive, brute force over pixel centres: 35.6 %).
//
// alpha >= 1/255  <=>  Q(u) = a ux^2 + 2 b ux uy + c uy^2 <= t = 2 ln(255 opacity), u = pixel - centre:
// an ellipse.  Its intersection with the tile's column strip [xl, xh] is convex, so a quarter
// (strip x row band) is reached iff the band overlaps the y-extent [ylo, yup] of that intersection.
// The upper boundary yup(x) is concave with its maximum at x_top = -(b/a) y_top,
// y_top = sqrt(t a / det): over the strip the maximum sits at clamp(x_top, xl, xh) (same for the
// lower boundary at -x_top); at abscissa x the boundary is (-b x +- sqrt(c t - det x^2)) / c, and a
// negative discriminant at the clamped abscissa means the strip misses the ellipse altogether.
// One evaluation serves all four quarters (about 45 instructions instead of four rectangle
// minimisations).  t is inflated by 1e-4 relative + 2e-3 and the extent by 1e-3 + 1e-5 |y|, far
// above the fp32 error of these expressions; det uses Kahan's fma form, because a c and b^2
// nearly cancel for elongated splats.
// The evaluation comes in three pieces, so that a caller looping over the tiles of one splat
// (hier_binning.hip) pays for each piece once: quarter_pre depends on the splat alone, strip_extent
// on the splat and the tile COLUMN, quarter_bits on the tile row.
struct QuarterPre {
  float b, ic, det, ct, xtop;   // conic b, 1/c, determinant, c t, abscissa of the upper extreme
  bool sane, transparent;
};
__device__ __forceinline__ QuarterPre quarter_pre(const float a, const float b, const float c,
                                                  const float opacity) {
  QuarterPre q;
  float t = 2.0f * __logf(255.0f * opacity);
  t = t + 1e-4f * fabsf(t) + 2e-3f;
  const float p = b * b;
  const float det = fmaf(a, c, -p) - fmaf(b, b, -p);
  q.sane = (a > 0.0f) && (c > 0.0f) && (det > 0.0f) && (fabsf(t) < 3e38f) && (fabsf(det) < 3e38f);
  const float ytop = __builtin_amdgcn_sqrtf(fmaxf(t * a * __builtin_amdgcn_rcpf(det), 0.0f));
  q.xtop = -(b * __builtin_amdgcn_rcpf(a)) * ytop;
  q.ct = c * t;
  q.ic = __builtin_amdgcn_rcpf(c);
  q.b = b;
  q.det = det;
  // alpha <= opacity, so a splat below 1/255 never passes anywhere
  q.transparent = opacity < (1.0f / 255.0f) * 0.999f;
  return q;
}
// y-extent [ylo, yup] (relative to the centre) of the ellipse over the column strip starting at x0
__device__ __forceinline__ void strip_extent(const QuarterPre& q, const float gx, const float x0,
                                             float& ylo, float& yup, bool& empty) {
  const float xl = x0 - gx, xh = xl + 15.0f;
  const float xet = fminf(fmaxf(q.xtop, xl), xh), xeb = fminf(fmaxf(-q.xtop, xl), xh);
  const float dt = fmaf(-q.det * xet, xet, q.ct), db = fmaf(-q.det * xeb, xeb, q.ct);
  yup = (__builtin_amdgcn_sqrtf(fmaxf(dt, 0.0f)) - q.b * xet) * q.ic;
  ylo = (-__builtin_amdgcn_sqrtf(fmaxf(db, 0.0f)) - q.b * xeb) * q.ic;
  yup += 1e-3f + 1e-5f * fabsf(yup);
  ylo -= 1e-3f + 1e-5f * fabsf(ylo);
  empty = (dt < 0.0f) && (db < 0.0f);
}
__device__ __forceinline__ uint32_t quarter_bits(const bool sane, const bool transparent,
                                                 const float ylo, const float yup, const bool empty,
                                                 const float gy, const float y0) {
  uint32_t m = 0;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const float yl_q = y0 + (float)(4 * q) - gy, yh_q = yl_q + 3.0f;
    const bool hit = sane ? ((ylo <= yh_q) && (yup >= yl_q) && !empty) : true;
    m |= (hit && !transparent) ? (1u << q) : 0u;
  }
  return m;
}
// All 32 quarter rows of a COLUMN of 8 tiles at once (hier_binning.hip): bit k = the splat may
// reach pixel rows Y0 + 4k .. Y0 + 4k + 3 of the strip, i.e. quarter k & 3 of tile k >> 2.  The same
// test as quarter_bits -- band k is reached iff ylo + gy <= Y0 + 4k + 3 and yup + gy >= Y0 + 4k --
// solved for k; the float rounding of the division is far inside the 1e-3 inflation of the extent.
__device__ __forceinline__ uint32_t quarter_column_mask(const bool sane, const bool transparent,
                                                        const float ylo, const float yup,
                                                        const bool empty, const float gy,
                                                        const float Y0) {
  const float ya = ylo + (gy - Y0), yb = yup + (gy - Y0);
  const int klo = max(0, (int)ceilf((ya - 3.0f) * 0.25f));
  const int khi = min(31, (int)floorf(yb * 0.25f));
  uint32_t m = (klo <= khi && !empty) ? (0xFFFFFFFFu >> (31 - (khi - klo))) << klo : 0u;
  if (!sane || ya != ya || yb != yb) m = 0xFFFFFFFFu;   // never reject on doubt
  return transparent ? 0u : m;
}
__device__ __forceinline__ uint32_t quarter_reach_mask(const float gx, const float gy,
                                                       const float a, const float b, const float c,
                                                       const float opacity, const float x0,
                                                       const float y0) {
  const QuarterPre q = quarter_pre(a, b, c, opacity);
  float ylo, yup;
  bool empty;
  strip_extent(q, gx, x0, ylo, yup, empty);
  return quarter_bits(q.sane, q.transparent, ylo, yup, empty, gy, y0);
}

}  // namespace grpg
