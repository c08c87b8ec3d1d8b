// Backward of the per-Gaussian preprocess for gfx950: one fused kernel for what the reference
// runs as computeCov2DCUDA + preprocessCUDA (cuda_rasterizer/backward.cu:144-274, :346-412) with
// their helpers (SH backward :20-139, Sigma backward :278-341, dnormvdv auxiliary.h:107-117).
//
// The gradients are derived here in matrix form (the reference spells every entry out):
//   conic K = S^-1, S = [[a b][b c]] the 2x2 screen covariance:  dL/dS = -K G K
//   S = A V A^T, A = J W (2x3), V the 3x3 world covariance:       dL/dV = A^T H A,  dL/dA = 2 H A V
//   A -> the four non-trivial entries of the Jacobian J -> the view-space mean t -> the mean
//   V = R D R^T, D = diag(s^2):   dL/dR = 2 Hv R D,  dL/ds_k = 2 s_k (R^T Hv R)_kk,
//   and dL/dq from the antisymmetric / symmetric parts of dL/dR
//   colour = sum_k sh_k Y_k(d):   dL/dsh_k = Y_k g,  dL/dd = sum_k (sh_k . g) grad Y_k,
//   d = v / |v|:                  dL/dv = (g_d - d (d . g_d)) / |v|
// Results agree with the reference's to rounding (tests/test_gpu_backward.py: relative L2 <= 1e-3
// against the oracle, which follows the reference entry by entry); its observable quirks are
// kept: the 1/(det^2 + 1e-7) regulariser, gradients through tx / ty cut where the forward clamped
// them, the depth term that treats view row 3 as a homogeneous divisor, and dL/dscale taken with
// respect to the MODIFIED scale.
//
// Sigma and A are RECOMPUTED with the forward's own functions (gaussian_math.h, same
// -ffp-contract=off arithmetic), so nothing but the 48-byte record survives from the forward;
// the SH clamp mask comes from the record (r[2].w).  The blend backward's sums arrive as one 64-byte
// record per Gaussian; this kernel writes ALL of the op's per-Gaussian gradient arrays, for every
// Gaussian (zeros for culled ones), so none of them needs a zero-fill.  HBM-bound streaming kernel.
#include "common.h"

#pragma clang fp contract(off)

#include "gaussian_math.h"
#include "compose_math.h"

namespace grpg {

struct Vec3 { float x, y, z; };
__device__ __forceinline__ float dot3(const Vec3 a, const Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// Colour -> SH coefficients and view direction.  g = dL/dRGB with the clamped channels zeroed
// (forward.cu:103-105 clamps at 0 after the +0.5 shift; the mask is in the record).
// Writes dL_dsh[0 .. (deg+1)^2) and returns dL/d(mean - campos).
__device__ __forceinline__ Vec3 sh_backward(const int deg, const float* __restrict__ sh, const Vec3 v,
                                            const Vec3 g, float* __restrict__ dL_dsh) {
  const float rlen = 1.0f / sqrtf(dot3(v, v));
  const Vec3 d = {v.x * rlen, v.y * rlen, v.z * rlen};
  Vec3 gd = {0.f, 0.f, 0.f};   // dL/dd, d treated as three free variables like the reference does
  // one basis function: value Yk, gradient (yx, yy, yz)
  auto term = [&](const int k, const float Yk, const float yx, const float yy, const float yz) {
    dL_dsh[3 * k + 0] = Yk * g.x;
    dL_dsh[3 * k + 1] = Yk * g.y;
    dL_dsh[3 * k + 2] = Yk * g.z;
    const float w = sh[3 * k] * g.x + sh[3 * k + 1] * g.y + sh[3 * k + 2] * g.z;
    gd.x += w * yx; gd.y += w * yy; gd.z += w * yz;
  };
  const float x = d.x, y = d.y, z = d.z;
  term(0, SH_C0, 0.f, 0.f, 0.f);
  if (deg > 0) {
    term(1, -SH_C1 * y, 0.f, -SH_C1, 0.f);
    term(2, SH_C1 * z, 0.f, 0.f, SH_C1);
    term(3, -SH_C1 * x, -SH_C1, 0.f, 0.f);
  }
  if (deg > 1) {
    const float xx = x * x, yy = y * y, zz = z * z;
    term(4, SH_C2[0] * x * y, SH_C2[0] * y, SH_C2[0] * x, 0.f);
    term(5, SH_C2[1] * y * z, 0.f, SH_C2[1] * z, SH_C2[1] * y);
    term(6, SH_C2[2] * (2.f * zz - xx - yy), SH_C2[2] * -2.f * x, SH_C2[2] * -2.f * y, SH_C2[2] * 4.f * z);
    term(7, SH_C2[3] * x * z, SH_C2[3] * z, 0.f, SH_C2[3] * x);
    term(8, SH_C2[4] * (xx - yy), SH_C2[4] * 2.f * x, SH_C2[4] * -2.f * y, 0.f);
    if (deg > 2) {
      const float xy = x * y, yz = y * z, xz = x * z;
      term(9, SH_C3[0] * y * (3.f * xx - yy), SH_C3[0] * 6.f * xy, SH_C3[0] * 3.f * (xx - yy), 0.f);
      term(10, SH_C3[1] * xy * z, SH_C3[1] * yz, SH_C3[1] * xz, SH_C3[1] * xy);
      term(11, SH_C3[2] * y * (4.f * zz - xx - yy), SH_C3[2] * -2.f * xy,
           SH_C3[2] * (4.f * zz - xx - 3.f * yy), SH_C3[2] * 8.f * yz);
      term(12, SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy), SH_C3[3] * -6.f * xz, SH_C3[3] * -6.f * yz,
           SH_C3[3] * (6.f * zz - 3.f * xx - 3.f * yy));
      term(13, SH_C3[4] * x * (4.f * zz - xx - yy), SH_C3[4] * (4.f * zz - 3.f * xx - yy),
           SH_C3[4] * -2.f * xy, SH_C3[4] * 8.f * xz);
      term(14, SH_C3[5] * z * (xx - yy), SH_C3[5] * 2.f * xz, SH_C3[5] * -2.f * yz, SH_C3[5] * (xx - yy));
      term(15, SH_C3[6] * x * (xx - 3.f * yy), SH_C3[6] * 3.f * (xx - yy), SH_C3[6] * -6.f * xy, 0.f);
    }
  }
  // through the normalisation d = v / |v|: the component of gd orthogonal to d, over |v|
  const float along = dot3(d, gd);
  return Vec3{(gd.x - d.x * along) * rlen, (gd.y - d.y * along) * rlen, (gd.z - d.z * along) * rlen};
}

// World covariance V = R D R^T (R from the unnormalised quaternion q = (r, x, y, z), D = diag(s^2),
// s = modifier * scale) -> dL/ds (3) and dL/dq (4).  h = dL/dV as the 6 upper-triangle values with
// the off-diagonal ones already counted twice (what dL_dcov3D holds).
__device__ __forceinline__ void cov3d_backward(const float s0, const float s1, const float s2,
                                               const float mod, const float4 q,
                                               const float* __restrict__ h, float* dscale,
                                               float* drot) {
  const float r = q.x, x = q.y, y = q.z, z = q.w;
  const float R[3][3] = {
      {1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},
      {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
      {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)}};
  const float s[3] = {mod * s0, mod * s1, mod * s2};
  const float Hv[3][3] = {{h[0], 0.5f * h[1], 0.5f * h[2]},
                          {0.5f * h[1], h[3], 0.5f * h[4]},
                          {0.5f * h[2], 0.5f * h[4], h[5]}};
  float HR[3][3];   // Hv R
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) HR[i][j] = Hv[i][0] * R[0][j] + Hv[i][1] * R[1][j] + Hv[i][2] * R[2][j];
  float G[3][3];    // dL/dR = 2 Hv R D
#pragma unroll
  for (int j = 0; j < 3; j++) {
    const float rhr = R[0][j] * HR[0][j] + R[1][j] * HR[1][j] + R[2][j] * HR[2][j];   // (R^T Hv R)_jj
    dscale[j] = 2.f * s[j] * rhr;
    const float d2 = 2.f * s[j] * s[j];
#pragma unroll
    for (int i = 0; i < 3; i++) G[i][j] = d2 * HR[i][j];
  }
  // R is linear in the products of quaternion components: with A = G - G^T, S = G + G^T
  const float A21 = G[2][1] - G[1][2], A02 = G[0][2] - G[2][0], A10 = G[1][0] - G[0][1];
  const float S01 = G[0][1] + G[1][0], S02 = G[0][2] + G[2][0], S12 = G[1][2] + G[2][1];
  drot[0] = 2.f * (x * A21 + y * A02 + z * A10);
  drot[1] = 2.f * (r * A21 + y * S01 + z * S02) - 4.f * x * (G[1][1] + G[2][2]);
  drot[2] = 2.f * (r * A02 + x * S01 + z * S12) - 4.f * y * (G[0][0] + G[2][2]);
  drot[3] = 2.f * (r * A10 + x * S02 + y * S12) - 4.f * z * (G[0][0] + G[1][1]);
}

// Gradient of one visible Gaussian given its ACTIVATED parameters (mean m, 3-D covariance c3 --
// recomputed by the caller with the forward's own function -- and, when it came from scale /
// rotation, s and q), the blend backward's sums for it, and its SH coefficients.  Outputs: gm
// (dL/dmean), dcov (dL/dSigma, 6 values), ds / dr (dL/dscale, dL/dq; only when from_sr) and, through
// dsh, dL/dsh (M x 3; NULL = colours were precomputed).  Shared by the flat kernel and the
// composed one (raw scene-graph parameters).
struct BlendGrad {   // one Gaussian's 64-byte gradient record (common.h GRAD_*)
  float g2x, g2y, gabs, gxx, gxy, gyy, gcr, gcg, gcb, gop, gdep;
};
__device__ __forceinline__ BlendGrad load_blend_grad(const float4* __restrict__ grad_rec, const int idx) {
  const float4 gr0 = grad_rec[(GRAD_STRIDE / 4) * (size_t)idx], gr1 = grad_rec[(GRAD_STRIDE / 4) * (size_t)idx + 1],
               gr2 = grad_rec[(GRAD_STRIDE / 4) * (size_t)idx + 2];
  return BlendGrad{gr0.x, gr0.y, gr0.z, gr0.w, gr1.x, gr1.y, gr1.z, gr1.w, gr2.x, gr2.y, gr2.z};
}

__device__ __forceinline__ void preprocess_backward_core(
    const Vec3 m, const float* c3, const bool from_sr, const float s0, const float s1, const float s2,
    const float scale_modifier, const float4 q, const float* __restrict__ view,
    const float* __restrict__ proj, const float* __restrict__ campos, const float h_x, const float h_y,
    const float tan_fovx, const float tan_fovy, const BlendGrad& bg, const uint32_t clamped, const int D,
    const float* __restrict__ sh, float* __restrict__ dsh, Vec3& gm_out, float* dcov, float* ds, float* dr) {
  Cov2D cv;
  cov2d_project(m.x, m.y, m.z, view, h_x, h_y, tan_fovx, tan_fovy, c3, cv);

  // ---- conic -> screen covariance S (with the +0.3 dilation):  dL/dS = -K G K, K = adj(S) / det ----
  // the render backward stores HALF of the off-diagonal conic derivative (backward.cu:619-621),
  // so G carries that stored value in both off-diagonal entries
  const float gxx = bg.gxx, gxy = bg.gxy, gyy = bg.gyy;
  const float a = cv.a + 0.3f, b = cv.b, c = cv.c + 0.3f;
  const float det = a * c - b * b;
  const float inv_det2 = 1.0f / ((det * det) + 0.0000001f);
  // adj(S) G adj(S), adj(S) = [[c, -b], [-b, a]]
  const float p0 = c * gxx - b * gxy, p1 = c * gxy - b * gyy;   // first row of adj G
  const float q0 = a * gxy - b * gxx, q1 = a * gyy - b * gxy;   // second row of adj G
  float Haa = -(p0 * c - p1 * b) * inv_det2;      // dL/da
  float Hcc = -(q1 * a - q0 * b) * inv_det2;      // dL/dc
  float Hab = -(p1 * a - p0 * b) * inv_det2;      // dL/d(one off-diagonal entry); dL/db = 2 Hab
  if (!(inv_det2 != 0.f)) { Haa = 0.f; Hcc = 0.f; Hab = 0.f; }

  // ---- S = A V A^T:  dL/dV = A^T H A (off-diagonals counted twice),  dL/dA = 2 H A V ----
  const float A[2][3] = {{cv.T00, cv.T01, cv.T02}, {cv.T10, cv.T11, cv.T12}};
  float HA[2][3];
#pragma unroll
  for (int j = 0; j < 3; j++) {
    HA[0][j] = Haa * A[0][j] + Hab * A[1][j];
    HA[1][j] = Hab * A[0][j] + Hcc * A[1][j];
  }
  auto aha = [&](const int r, const int cc) { return A[0][r] * HA[0][cc] + A[1][r] * HA[1][cc]; };
  dcov[0] = aha(0, 0); dcov[1] = 2.f * aha(0, 1); dcov[2] = 2.f * aha(0, 2);
  dcov[3] = aha(1, 1); dcov[4] = 2.f * aha(1, 2); dcov[5] = aha(2, 2);
  const float V[3][3] = {{c3[0], c3[1], c3[2]}, {c3[1], c3[3], c3[4]}, {c3[2], c3[4], c3[5]}};
  float dA[2][3];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 3; j++)
      dA[i][j] = 2.f * (HA[i][0] * V[0][j] + HA[i][1] * V[1][j] + HA[i][2] * V[2][j]);

  // ---- A = J W: the view rotation's rows are (v0 v4 v8), (v1 v5 v9), (v2 v6 v10) ----
  const Vec3 W0 = {view[0], view[4], view[8]}, W1 = {view[1], view[5], view[9]}, W2 = {view[2], view[6], view[10]};
  const Vec3 dA0 = {dA[0][0], dA[0][1], dA[0][2]}, dA1 = {dA[1][0], dA[1][1], dA[1][2]};
  const float dJ00 = dot3(W0, dA0), dJ02 = dot3(W2, dA0);
  const float dJ11 = dot3(W1, dA1), dJ12 = dot3(W2, dA1);
  // J00 = hx / tz, J02 = -hx tx / tz^2, J11 = hy / tz, J12 = -hy ty / tz^2; no gradient through a
  // tx / ty the forward clamped to the frustum guard band
  const float itz = 1.f / cv.tz, itz2 = itz * itz, itz3 = itz2 * itz;
  const float pass_x = (cv.txtz < -cv.limx || cv.txtz > cv.limx) ? 0.f : 1.f;
  const float pass_y = (cv.tytz < -cv.limy || cv.tytz > cv.limy) ? 0.f : 1.f;
  const Vec3 dt = {pass_x * (-h_x * itz2) * dJ02, pass_y * (-h_y * itz2) * dJ12,
                   -h_x * itz2 * dJ00 - h_y * itz2 * dJ11 + (2.f * h_x * cv.tx) * itz3 * dJ02 +
                       (2.f * h_y * cv.ty) * itz3 * dJ12};
  // t = W m + translation
  Vec3 gm = {W0.x * dt.x + W1.x * dt.y + W2.x * dt.z, W0.y * dt.x + W1.y * dt.y + W2.y * dt.z,
             W0.z * dt.x + W1.z * dt.y + W2.z * dt.z};

  // ---- mean2D = ndc2pix(proj m / w):  d(p_k / w)/dm_j = (P_kj - P_3j p_k / w) / w ----
  {
    const float w = proj[3] * m.x + proj[7] * m.y + proj[11] * m.z + proj[15];
    const float iw = 1.0f / (w + 0.0000001f);
    const float px_w2 = (proj[0] * m.x + proj[4] * m.y + proj[8] * m.z + proj[12]) * iw * iw;
    const float py_w2 = (proj[1] * m.x + proj[5] * m.y + proj[9] * m.z + proj[13]) * iw * iw;
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const float add = (proj[4 * j] * iw - proj[4 * j + 3] * px_w2) * bg.g2x +
                        (proj[4 * j + 1] * iw - proj[4 * j + 3] * py_w2) * bg.g2y;
      if (j == 0) gm.x += add; else if (j == 1) gm.y += add; else gm.z += add;
    }
  }
  // ---- depth = (view m)_z, with view row 3 treated as a divisor of weight depth (backward.cu:384-391) ----
  {
    const float depth = view[2] * m.x + view[6] * m.y + view[10] * m.z + view[14];
    gm.x += (view[2] - view[3] * depth) * bg.gdep;
    gm.y += (view[6] - view[7] * depth) * bg.gdep;
    gm.z += (view[10] - view[11] * depth) * bg.gdep;
  }
  if (sh != nullptr) {
    const Vec3 g = {(clamped & 1u) ? 0.f : bg.gcr, (clamped & 2u) ? 0.f : bg.gcg, (clamped & 4u) ? 0.f : bg.gcb};
    const Vec3 gs = sh_backward(D, sh, Vec3{m.x - campos[0], m.y - campos[1], m.z - campos[2]}, g, dsh);
    gm.x += gs.x; gm.y += gs.y; gm.z += gs.z;
  }
  gm_out = gm;
  if (from_sr) cov3d_backward(s0, s1, s2, scale_modifier, q, dcov, ds, dr);
}

__global__ void __launch_bounds__(256)
preprocess_backward_kernel(const int P, const int D, const int M, const float* __restrict__ means3D,
                           const int* __restrict__ radii, const float* __restrict__ shs,
                           const RecView rec, const float* __restrict__ scales,
                           const float* __restrict__ rotations, const float scale_modifier,
                           const float* __restrict__ cov3D_precomp, const float* __restrict__ view,
                           const float* __restrict__ proj, const float* __restrict__ campos,
                           const float h_x, const float h_y, const float tan_fovx,
                           const float tan_fovy, const float4* __restrict__ grad_rec,
                           float* __restrict__ dL_dmean2D, float* __restrict__ dL_dconics,
                           float* __restrict__ dL_dopacity, float* __restrict__ dL_dmeans,
                           float* __restrict__ dL_dcolor, float* __restrict__ dL_ddepth,
                           float* __restrict__ dL_dcov, float* __restrict__ dL_dsh,
                           float* __restrict__ dL_dscale, float* __restrict__ dL_drot) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= P) return;
  // Every output array is written for EVERY Gaussian (zeros for a culled one): nothing has to
  // arrive zero-filled (the reference zero-fills eleven arrays per call, rasterize_points.cu:166-176).
  if (!(radii[idx] > 0)) {
    dL_dmean2D[3 * idx] = 0.f; dL_dmean2D[3 * idx + 1] = 0.f; dL_dmean2D[3 * idx + 2] = 0.f;
    if (dL_dconics != nullptr) reinterpret_cast<float4*>(dL_dconics)[idx] = make_float4(0.f, 0.f, 0.f, 0.f);
    dL_dopacity[idx] = 0.f;
    if (dL_dcolor != nullptr) { dL_dcolor[3 * idx] = 0.f; dL_dcolor[3 * idx + 1] = 0.f; dL_dcolor[3 * idx + 2] = 0.f; }
    if (dL_ddepth != nullptr) dL_ddepth[idx] = 0.f;
    dL_dmeans[3 * idx] = 0.f; dL_dmeans[3 * idx + 1] = 0.f; dL_dmeans[3 * idx + 2] = 0.f;
    if (dL_dcov != nullptr) {
#pragma unroll
      for (int e = 0; e < 6; e++) dL_dcov[6 * idx + e] = 0.f;
    }
    if (dL_dsh != nullptr)
      for (int e = 0; e < 3 * M; e++) dL_dsh[(size_t)idx * M * 3 + e] = 0.f;
    if (dL_dscale != nullptr) { dL_dscale[3 * idx] = 0.f; dL_dscale[3 * idx + 1] = 0.f; dL_dscale[3 * idx + 2] = 0.f; }
    if (dL_drot != nullptr) reinterpret_cast<float4*>(dL_drot)[idx] = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  // the blend backward's per-Gaussian sums: one 64-byte record (common.h GRAD_*), fanned out into the
  // reference's separate arrays here (dL_dmean2D and dL_dopacity are outputs of the op, the conic /
  // colour / depth gradients its intermediate results, rasterize_points.cu:166-219); the
  // intermediate arrays are optional (NULL: the caller does not read them)
  const BlendGrad bgr = load_blend_grad(grad_rec, idx);
  dL_dmean2D[3 * idx] = bgr.g2x; dL_dmean2D[3 * idx + 1] = bgr.g2y; dL_dmean2D[3 * idx + 2] = bgr.gabs;
  if (dL_dconics != nullptr) reinterpret_cast<float4*>(dL_dconics)[idx] = make_float4(bgr.gxx, bgr.gxy, 0.f, bgr.gyy);
  dL_dopacity[idx] = bgr.gop;
  if (dL_dcolor != nullptr) { dL_dcolor[3 * idx] = bgr.gcr; dL_dcolor[3 * idx + 1] = bgr.gcg; dL_dcolor[3 * idx + 2] = bgr.gcb; }
  if (dL_ddepth != nullptr) dL_ddepth[idx] = bgr.gdep;

  const Vec3 m = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
  float c3[6];
  float4 q = make_float4(0, 0, 0, 0);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  const bool from_sr = cov3D_precomp == nullptr;
  if (!from_sr) {
#pragma unroll
    for (int i = 0; i < 6; i++) c3[i] = cov3D_precomp[6 * idx + i];
  } else {
    q = load_quat(rotations, idx);
    s0 = scales[3 * idx]; s1 = scales[3 * idx + 1]; s2 = scales[3 * idx + 2];
    cov3d_from_scale_rot(s0, s1, s2, scale_modifier, q, c3);
  }
  const uint32_t clamped = shs != nullptr ? __float_as_uint(rec.colour(idx).w) : 0u;
  Vec3 gm;
  float dcov[6], ds[3] = {0.f, 0.f, 0.f}, dr[4] = {0.f, 0.f, 0.f, 0.f};
  preprocess_backward_core(m, c3, from_sr, s0, s1, s2, scale_modifier, q, view, proj, campos, h_x, h_y,
                           tan_fovx, tan_fovy, bgr, clamped, D,
                           shs != nullptr ? shs + (size_t)idx * M * 3 : nullptr,
                           shs != nullptr ? dL_dsh + (size_t)idx * M * 3 : nullptr, gm, dcov, ds, dr);
  // coefficients above the active degree get no gradient (the training schedule raises sh_degree
  // step by step while M stays (max_degree + 1)^2): written as zeros, nothing arrives zero-filled
  if (shs != nullptr)
    for (int e = 3 * (D + 1) * (D + 1); e < 3 * M; e++) dL_dsh[(size_t)idx * M * 3 + e] = 0.f;
  if (dL_dcov != nullptr) {
#pragma unroll
    for (int e = 0; e < 6; e++) dL_dcov[6 * idx + e] = dcov[e];
  }
  dL_dmeans[3 * idx] = gm.x;
  dL_dmeans[3 * idx + 1] = gm.y;
  dL_dmeans[3 * idx + 2] = gm.z;
  if (dL_dscale != nullptr) { dL_dscale[3 * idx] = ds[0]; dL_dscale[3 * idx + 1] = ds[1]; dL_dscale[3 * idx + 2] = ds[2]; }
  if (dL_drot != nullptr) reinterpret_cast<float4*>(dL_drot)[idx] = make_float4(dr[0], dr[1], dr[2], dr[3]);
}

// ------------------------------------------------------------------------------------------
// Training backward of the FUSED scene-graph composition: the same
// per-Gaussian core on values recomputed by compose_one(), then the chain rule through the
// activations and the rigid actor transform down to the models' RAW parameters and the actors'
// tracked poses -- what autograd would do through lib/models/street_gaussian_model.py:296-453:
//   _scaling   s = exp(r)                 dL/dr = dL/ds * s
//   _opacity   o = sigmoid(r)             dL/dr = dL/do * o (1 - o)
//   _rotation  q = r / |r|                dL/dr = (g - q (q.g)) / |r|
//     actor:   q = p / |p|, p = a (x) ql  (a = obj_rot as given, ql = r / |r|): through the
//              normalisation, then the Hamilton product (bilinear) to ql and to a
//   _xyz       actor: m = R(a / |a|) x + t     dL/dx = R^T g,  dL/dt = sum g,  dL/dR = sum g x^T
//   _features_dc  sh_0 = sum_c dc_c idft_c     dL/ddc_c = dL/dsh_0 * idft_c;  _features_rest = sh_1..
// Per actor the sums over its Gaussians (dL/dt 3, dL/dR 9, dL/da of the product 4) go to
// pose_acc[segment][16] (LDS-aggregated per workgroup, then one float atomic per value);
// pose_finish_kernel turns them into dL/d obj_rot (through R(a/|a|) and the normalisation) and
// dL/d obj_trans.
// ------------------------------------------------------------------------------------------
struct SegmentGradDev {
  float* xyz;
  float* scaling;
  float* rotation;
  float* opacity;
  float* fdc;
  float* frest;
};

constexpr int POSE_ACC = 16;   // per segment: dL/dt (3), dL/dR row-major (9), dL/da product path (4)

// One Gaussian of the composed training backward.  sg / go refer either to the workgroup's copies in
// LDS (all 256 Gaussians in one segment: the common case) or to the tables in global memory; the
// function is inlined into both call sites, so each knows its address space and the stores to the
// gradient arrays cannot force the table fields to be reloaded.
template <bool M4>
__device__ __forceinline__ void composed_backward_one(
    const SegmentDev& sg, const SegmentGradDev& go, const int idx, const int D, const int M,
    const int* __restrict__ radii, const RecView rec, const float scale_modifier,
    const float* __restrict__ view, const float* __restrict__ proj, const float* __restrict__ campos,
    const float h_x, const float h_y, const float tan_fovx, const float tan_fovy,
    const float4* __restrict__ grad_rec, float* __restrict__ dL_dmean2D, float (&pv)[POSE_ACC]) {
  const int Mc = M4 ? 4 : M;   // compile-time trip counts in the common (SH degree 1) instance
  const uint32_t j = (uint32_t)idx - sg.start;
  const int F = sg.fourier_dim;
  const bool vis = radii[idx] > 0;
  if (!vis) {
    dL_dmean2D[3 * idx] = 0.f; dL_dmean2D[3 * idx + 1] = 0.f; dL_dmean2D[3 * idx + 2] = 0.f;
    go.xyz[3 * j] = 0.f; go.xyz[3 * j + 1] = 0.f; go.xyz[3 * j + 2] = 0.f;
    go.scaling[3 * j] = 0.f; go.scaling[3 * j + 1] = 0.f; go.scaling[3 * j + 2] = 0.f;
    reinterpret_cast<float4*>(go.rotation)[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    go.opacity[j] = 0.f;
    for (int e = 0; e < 3 * F; e++) go.fdc[(size_t)j * F * 3 + e] = 0.f;
    for (int e = 0; e < 3 * (Mc - 1); e++) go.frest[(size_t)j * (Mc - 1) * 3 + e] = 0.f;
  } else {
    const Activated a = compose_one(sg, j);
    const BlendGrad bgr = load_blend_grad(grad_rec, idx);
    dL_dmean2D[3 * idx] = bgr.g2x; dL_dmean2D[3 * idx + 1] = bgr.g2y; dL_dmean2D[3 * idx + 2] = bgr.gabs;
    float c3[6];
    cov3d_from_scale_rot(a.s0, a.s1, a.s2, scale_modifier, a.q, c3);
    float sh[M4 ? 12 : 48], dsh[M4 ? 12 : 48];
    compose_features(sg, j, a, Mc, sh);
#pragma unroll
    for (int e = 0; e < (M4 ? 12 : 48); e++) dsh[e] = 0.f;
    const uint32_t clamped = __float_as_uint(rec.colour(idx).w);
    Vec3 gm;
    float dcov[6], ds[3], dr[4];
    preprocess_backward_core(Vec3{a.mx, a.my, a.mz}, c3, true, a.s0, a.s1, a.s2, scale_modifier, a.q, view,
                             proj, campos, h_x, h_y, tan_fovx, tan_fovy, bgr, clamped, D, sh, dsh, gm, dcov,
                             ds, dr);
    // ---- activations ----
    go.scaling[3 * j] = ds[0] * a.s0; go.scaling[3 * j + 1] = ds[1] * a.s1; go.scaling[3 * j + 2] = ds[2] * a.s2;
    go.opacity[j] = bgr.gop * (a.opacity * (1.0f - a.opacity));
    for (int c = 0; c < F; c++) {
      go.fdc[((size_t)j * F + c) * 3 + 0] = dsh[0] * sg.idft[c];
      go.fdc[((size_t)j * F + c) * 3 + 1] = dsh[1] * sg.idft[c];
      go.fdc[((size_t)j * F + c) * 3 + 2] = dsh[2] * sg.idft[c];
    }
    for (int e = 0; e < 3 * (Mc - 1); e++) go.frest[(size_t)j * (Mc - 1) * 3 + e] = dsh[3 + e];
    // ---- rotation: raw r -> ql = r / |r| [-> p = a (x) ql -> q = p / |p|] ----
    const bool flip = sg.flip != nullptr && sg.flip[j] != 0;
    const float4 rq = load_quat(sg.rotation, (int)j);
    const float rn = fmaxf(sqrtf(rq.x * rq.x + rq.y * rq.y + rq.z * rq.z + rq.w * rq.w), 1e-12f);
    const float4 qn_ = make_float4(rq.x / rn, rq.y / rn, rq.z / rn, rq.w / rn);
    const float4 ql = flip ? make_float4(-qn_.z, qn_.w, qn_.x, -qn_.y) : qn_;   // as in compose_one
    float4 gql = make_float4(dr[0], dr[1], dr[2], dr[3]);   // dL/dq of the quaternion the op used
    if (sg.rigid) {
      const float aw = sg.rot[0], ax = sg.rot[1], ay = sg.rot[2], az = sg.rot[3];
      const float ow = aw * ql.x - ax * ql.y - ay * ql.z - az * ql.w;
      const float ox = aw * ql.y + ax * ql.x + ay * ql.w - az * ql.z;
      const float oy = aw * ql.z - ax * ql.w + ay * ql.x + az * ql.y;
      const float oz = aw * ql.w + ax * ql.z - ay * ql.y + az * ql.x;
      const float pn = fmaxf(sqrtf(ow * ow + ox * ox + oy * oy + oz * oz), 1e-12f);
      // through q = p / |p| (a.q is that q)
      const float along = a.q.x * gql.x + a.q.y * gql.y + a.q.z * gql.z + a.q.w * gql.w;
      const float ipn = 1.0f / pn;
      const float gw = (gql.x - a.q.x * along) * ipn, gx = (gql.y - a.q.y * along) * ipn,
                  gy = (gql.z - a.q.z * along) * ipn, gz = (gql.w - a.q.w * along) * ipn;
      // through the Hamilton product (general_utils.py:220-238): p = a (x) b, b = ql
      gql = make_float4(aw * gw + ax * gx + ay * gy + az * gz, -ax * gw + aw * gx + az * gy - ay * gz,
                        -ay * gw - az * gx + aw * gy + ax * gz, -az * gw + ay * gx - ax * gy + aw * gz);
      pv[12] = ql.x * gw + ql.y * gx + ql.z * gy + ql.w * gz;
      pv[13] = -ql.y * gw + ql.x * gx - ql.w * gy + ql.z * gz;
      pv[14] = -ql.z * gw + ql.w * gx + ql.x * gy - ql.y * gz;
      pv[15] = -ql.w * gw - ql.z * gx + ql.y * gy + ql.x * gz;
    }
    {
      // back through the flip's signed permutation ql = (-n.y', n.z', n.w', -n.x') of the normalised
      // local quaternion n = (w, x, y, z): dL/dn = (g_y, -g_z, -g_w, g_x) in (w, x, y, z) order
      const float4 gn = flip ? make_float4(gql.z, -gql.w, -gql.x, gql.y) : gql;
      const float along = qn_.x * gn.x + qn_.y * gn.y + qn_.z * gn.z + qn_.w * gn.w;
      const float irn = 1.0f / rn;
      reinterpret_cast<float4*>(go.rotation)[j] =
          make_float4((gn.x - qn_.x * along) * irn, (gn.y - qn_.y * along) * irn, (gn.z - qn_.z * along) * irn,
                      (gn.w - qn_.w * along) * irn);
    }
    // ---- mean: world == local, or m = R(a / |a|) x + t ----
    if (sg.rigid) {
      const float x = sg.xyz[3 * j], y = flip ? -sg.xyz[3 * j + 1] : sg.xyz[3 * j + 1], z = sg.xyz[3 * j + 2];
      const float on = sqrtf(sg.rot[0] * sg.rot[0] + sg.rot[1] * sg.rot[1] + sg.rot[2] * sg.rot[2] +
                             sg.rot[3] * sg.rot[3]);
      const float r = sg.rot[0] / on, qx = sg.rot[1] / on, qy = sg.rot[2] / on, qz = sg.rot[3] / on;
      const float R00 = 1.f - 2.f * (qy * qy + qz * qz), R01 = 2.f * (qx * qy - r * qz), R02 = 2.f * (qx * qz + r * qy);
      const float R10 = 2.f * (qx * qy + r * qz), R11 = 1.f - 2.f * (qx * qx + qz * qz), R12 = 2.f * (qy * qz - r * qx);
      const float R20 = 2.f * (qx * qz - r * qy), R21 = 2.f * (qy * qz + r * qx), R22 = 1.f - 2.f * (qx * qx + qy * qy);
      go.xyz[3 * j] = R00 * gm.x + R10 * gm.y + R20 * gm.z;
      go.xyz[3 * j + 1] = (flip ? -1.f : 1.f) * (R01 * gm.x + R11 * gm.y + R21 * gm.z);
      go.xyz[3 * j + 2] = R02 * gm.x + R12 * gm.y + R22 * gm.z;
      pv[0] = gm.x; pv[1] = gm.y; pv[2] = gm.z;
      pv[3] = gm.x * x; pv[4] = gm.x * y; pv[5] = gm.x * z;
      pv[6] = gm.y * x; pv[7] = gm.y * y; pv[8] = gm.y * z;
      pv[9] = gm.z * x; pv[10] = gm.z * y; pv[11] = gm.z * z;
    } else {
      go.xyz[3 * j] = gm.x; go.xyz[3 * j + 1] = flip ? -gm.y : gm.y; go.xyz[3 * j + 2] = gm.z;
    }
  }
}

template <bool M4>
__global__ void __launch_bounds__(256)
preprocess_backward_composed_kernel(const int P, const int D, const int M,
                                    const SegmentDev* __restrict__ segs,
                                    const SegmentGradDev* __restrict__ gsegs, const int nseg,
                                    const int* __restrict__ radii, const RecView rec,
                                    const float scale_modifier, const float* __restrict__ view,
                                    const float* __restrict__ proj, const float* __restrict__ campos,
                                    const float h_x, const float h_y, const float tan_fovx,
                                    const float tan_fovy, const float4* __restrict__ grad_rec,
                                    float* __restrict__ dL_dmean2D, float* __restrict__ pose_acc) {
  // one actor's sums, aggregated over the workgroup when all of its Gaussians belong to one segment
  __shared__ float s_pose[POSE_ACC];
  __shared__ int s_seg0;
  __shared__ SegmentDev s_sg;          // the workgroup's segment (uniform case): one copy in LDS instead
  __shared__ SegmentGradDev s_go;      // of a binary search and dependent global loads per thread
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (threadIdx.x < POSE_ACC) s_pose[threadIdx.x] = 0.f;
  const int last = min(P, (int)(blockIdx.x + 1) * 256) - 1;
  const SegmentDev* sg_first = find_segment(segs, nseg, (uint32_t)(blockIdx.x * 256));
  const bool uniform = sg_first == find_segment(segs, nseg, (uint32_t)last);
  if (threadIdx.x == 0) s_seg0 = (int)(sg_first - segs);
  if (uniform) {
    constexpr int SW = sizeof(SegmentDev) / 4, GW = sizeof(SegmentGradDev) / 4;
    if (threadIdx.x < SW)
      reinterpret_cast<uint32_t*>(&s_sg)[threadIdx.x] = reinterpret_cast<const uint32_t*>(sg_first)[threadIdx.x];
    else if (threadIdx.x >= 64 && threadIdx.x < 64 + GW)
      reinterpret_cast<uint32_t*>(&s_go)[threadIdx.x - 64] =
          reinterpret_cast<const uint32_t*>(gsegs + (sg_first - segs))[threadIdx.x - 64];
  }
  __syncthreads();
  float pv[POSE_ACC];
#pragma unroll
  for (int i = 0; i < POSE_ACC; i++) pv[i] = 0.f;
  bool rigid_lane = false;
  int seg_index = 0;
  if (idx < P) {
    const SegmentDev* sgp = uniform ? sg_first : find_segment(segs, nseg, (uint32_t)idx);
    seg_index = (int)(sgp - segs);
    if (uniform) {
      rigid_lane = s_sg.rigid != 0;
      composed_backward_one<M4>(s_sg, s_go, idx, D, M, radii, rec, scale_modifier, view, proj, campos, h_x, h_y,
                                tan_fovx, tan_fovy, grad_rec, dL_dmean2D, pv);
    } else {
      rigid_lane = sgp->rigid != 0;
      composed_backward_one<M4>(*sgp, gsegs[seg_index], idx, D, M, radii, rec, scale_modifier, view, proj,
                                campos, h_x, h_y, tan_fovx, tan_fovy, grad_rec, dL_dmean2D, pv);
    }
  }
  // ---- the actor's sums ----
  if (uniform) {
    if (s_sg.rigid) {
#pragma unroll
      for (int i = 0; i < POSE_ACC; i++) {
        float v = pv[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if ((threadIdx.x & 63) == 0 && v != 0.f) atomicAdd(&s_pose[i], v);
      }
      __syncthreads();
      if (threadIdx.x < POSE_ACC && s_pose[threadIdx.x] != 0.f)
        atomicAdd(&pose_acc[(size_t)s_seg0 * POSE_ACC + threadIdx.x], s_pose[threadIdx.x]);
    }
  } else if (rigid_lane) {   // a workgroup across a model boundary (at most nseg - 1 of them)
    const size_t sidx = (size_t)seg_index;
#pragma unroll
    for (int i = 0; i < POSE_ACC; i++)
      if (pv[i] != 0.f) atomicAdd(&pose_acc[sidx * POSE_ACC + i], pv[i]);
  }
}

// pose_acc -> dL/d obj_rot (w, x, y, z), dL/d obj_trans: [segment][8] (last value unused)
__global__ void pose_finish_kernel(const SegmentDev* __restrict__ segs, const int nseg,
                                   const float* __restrict__ pose_acc, float* __restrict__ dL_dposes) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nseg) return;
  float* o = dL_dposes + 8 * (size_t)i;
  if (!segs[i].rigid) {
#pragma unroll
    for (int k = 0; k < 8; k++) o[k] = 0.f;
    return;
  }
  const float* p = pose_acc + (size_t)i * POSE_ACC;
  const float G00 = p[3], G01 = p[4], G02 = p[5], G10 = p[6], G11 = p[7], G12 = p[8], G20 = p[9],
              G21 = p[10], G22 = p[11];
  const float aw = segs[i].rot[0], ax = segs[i].rot[1], ay = segs[i].rot[2], az = segs[i].rot[3];
  const float on = sqrtf(aw * aw + ax * ax + ay * ay + az * az);
  const float r = aw / on, x = ax / on, y = ay / on, z = az / on;
  // dL/d(normalised quaternion) through R (general_utils.py:125-146)
  const float gr = 2.f * (-z * G01 + y * G02 + z * G10 - x * G12 - y * G20 + x * G21);
  const float gx = 2.f * (y * G01 + z * G02 + y * G10 - 2.f * x * G11 - r * G12 + z * G20 + r * G21 - 2.f * x * G22);
  const float gy = 2.f * (-2.f * y * G00 + x * G01 + r * G02 + x * G10 + z * G12 - r * G20 + z * G21 - 2.f * y * G22);
  const float gz = 2.f * (-2.f * z * G00 - r * G01 + x * G02 + r * G10 - 2.f * z * G11 + y * G12 + x * G20 + y * G21);
  const float along = r * gr + x * gx + y * gy + z * gz;
  o[0] = (gr - r * along) / on + p[12];
  o[1] = (gx - x * along) / on + p[13];
  o[2] = (gy - y * along) / on + p[14];
  o[3] = (gz - z * along) / on + p[15];
  o[4] = p[0]; o[5] = p[1]; o[6] = p[2];
  o[7] = 0.f;
}

void launch_preprocess_backward_composed(hipStream_t s, int P, int D, int M, const SegmentDev* segs,
                                         const void* seg_grads, int nseg, const int* radii,
                                         const RecView rec, float scale_modifier, const CameraArgs& cam,
                                         const float* grad_rec, float* dL_dmean2D, float* pose_acc,
                                         float* dL_dposes) {
  if (P <= 0) return;
  (void)hipMemsetAsync(pose_acc, 0, (size_t)nseg * POSE_ACC * sizeof(float), s);
#define PBC_LAUNCH(M4)                                                                              \
  preprocess_backward_composed_kernel<M4><<<(P + 255) / 256, 256, 0, s>>>(                           \
      P, D, M, segs, reinterpret_cast<const SegmentGradDev*>(seg_grads), nseg, radii, rec,            \
      scale_modifier, cam.view, cam.proj, cam.campos, cam.focal_x, cam.focal_y, cam.tan_fovx,         \
      cam.tan_fovy, reinterpret_cast<const float4*>(grad_rec), dL_dmean2D, pose_acc)
  if (M == 4) PBC_LAUNCH(true); else PBC_LAUNCH(false);
#undef PBC_LAUNCH
  pose_finish_kernel<<<(nseg + 63) / 64, 64, 0, s>>>(segs, nseg, pose_acc, dL_dposes);
}

void launch_preprocess_backward(hipStream_t s, int P, int D, int M, const float* means3D,
                                const int* radii, const float* shs, const RecView rec,
                                const float* scales, const float* rotations, float scale_modifier,
                                const float* cov3D_precomp, const CameraArgs& cam,
                                const float* grad_rec, float* dL_dmean2D, float* dL_dconic,
                                float* dL_dopacity, float* dL_dmean3D, float* dL_dcolor,
                                float* dL_ddepth, float* dL_dcov3D, float* dL_dsh, float* dL_dscale,
                                float* dL_drot) {
  if (P <= 0) return;
  preprocess_backward_kernel<<<(P + 255) / 256, 256, 0, s>>>(
      P, D, M, means3D, radii, shs, rec, scales, rotations, scale_modifier, cov3D_precomp, cam.view,
      cam.proj, cam.campos, cam.focal_x, cam.focal_y, cam.tan_fovx, cam.tan_fovy,
      reinterpret_cast<const float4*>(grad_rec), dL_dmean2D, dL_dconic, dL_dopacity, dL_dmean3D,
      dL_dcolor, dL_ddepth, dL_dcov3D, dL_dsh, dL_dscale, dL_drot);
}

}  // namespace grpg
