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
  // colour / depth gradients its intermediate results, rasterize_points.cu:166-219)
  const float4 gr0 = grad_rec[(GRAD_STRIDE / 4) * (size_t)idx], gr1 = grad_rec[(GRAD_STRIDE / 4) * (size_t)idx + 1],
               gr2 = grad_rec[(GRAD_STRIDE / 4) * (size_t)idx + 2];
  const float g2x = gr0.x, g2y = gr0.y;
  const float gxx = gr0.w, gxy = gr1.x, gyy = gr1.y;
  const float gcr = gr1.z, gcg = gr1.w, gcb = gr2.x;
  const float gdep = gr2.z;
  dL_dmean2D[3 * idx] = g2x; dL_dmean2D[3 * idx + 1] = g2y; dL_dmean2D[3 * idx + 2] = gr0.z;
  // the intermediate arrays are optional (NULL: the caller does not read them)
  if (dL_dconics != nullptr) reinterpret_cast<float4*>(dL_dconics)[idx] = make_float4(gxx, gxy, 0.f, gyy);
  dL_dopacity[idx] = gr2.y;
  if (dL_dcolor != nullptr) { dL_dcolor[3 * idx] = gcr; dL_dcolor[3 * idx + 1] = gcg; dL_dcolor[3 * idx + 2] = gcb; }
  if (dL_ddepth != nullptr) dL_ddepth[idx] = gdep;
  const Vec3 m = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
  float c3[6];
  float4 q = make_float4(0, 0, 0, 0);
  if (cov3D_precomp != nullptr) {
#pragma unroll
    for (int i = 0; i < 6; i++) c3[i] = cov3D_precomp[6 * idx + i];
  } else {
    q = load_quat(rotations, idx);
    cov3d_from_scale_rot(scales[3 * idx], scales[3 * idx + 1], scales[3 * idx + 2],
                         scale_modifier, q, c3);
  }
  Cov2D cv;
  cov2d_project(m.x, m.y, m.z, view, h_x, h_y, tan_fovx, tan_fovy, c3, cv);

  // ---- conic -> screen covariance S (with the +0.3 dilation):  dL/dS = -K G K, K = adj(S) / det ----
  // the render backward stores HALF of the off-diagonal conic derivative (backward.cu:619-621),
  // so G carries that stored value in both off-diagonal entries
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
  auto aha = [&](const int r, const int c) { return A[0][r] * HA[0][c] + A[1][r] * HA[1][c]; };
  const float dcov[6] = {aha(0, 0), 2.f * aha(0, 1), 2.f * aha(0, 2), aha(1, 1), 2.f * aha(1, 2), aha(2, 2)};
  if (dL_dcov != nullptr) {
#pragma unroll
    for (int e = 0; e < 6; e++) dL_dcov[6 * idx + e] = dcov[e];
  }
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
      const float add = (proj[4 * j] * iw - proj[4 * j + 3] * px_w2) * g2x +
                        (proj[4 * j + 1] * iw - proj[4 * j + 3] * py_w2) * g2y;
      if (j == 0) gm.x += add; else if (j == 1) gm.y += add; else gm.z += add;
    }
  }
  // ---- depth = (view m)_z, with view row 3 treated as a divisor of weight depth (backward.cu:384-391) ----
  {
    const float depth = view[2] * m.x + view[6] * m.y + view[10] * m.z + view[14];
    gm.x += (view[2] - view[3] * depth) * gdep;
    gm.y += (view[6] - view[7] * depth) * gdep;
    gm.z += (view[10] - view[11] * depth) * gdep;
  }
  if (shs != nullptr) {
    const uint32_t clamped = __float_as_uint(rec.colour(idx).w);
    const Vec3 g = {(clamped & 1u) ? 0.f : gcr, (clamped & 2u) ? 0.f : gcg, (clamped & 4u) ? 0.f : gcb};
    const Vec3 gs = sh_backward(D, shs + (size_t)idx * M * 3,
                                Vec3{m.x - campos[0], m.y - campos[1], m.z - campos[2]}, g,
                                dL_dsh + (size_t)idx * M * 3);
    gm.x += gs.x; gm.y += gs.y; gm.z += gs.z;
  }
  dL_dmeans[3 * idx] = gm.x;
  dL_dmeans[3 * idx + 1] = gm.y;
  dL_dmeans[3 * idx + 2] = gm.z;
  if (scales == nullptr) {   // cov3D_precomp: the binding still returns (zero) scale / rotation gradients
    if (dL_dscale != nullptr) { dL_dscale[3 * idx] = 0.f; dL_dscale[3 * idx + 1] = 0.f; dL_dscale[3 * idx + 2] = 0.f; }
    if (dL_drot != nullptr) reinterpret_cast<float4*>(dL_drot)[idx] = make_float4(0.f, 0.f, 0.f, 0.f);
  } else {
    float ds[3], dr[4];
    cov3d_backward(scales[3 * idx], scales[3 * idx + 1], scales[3 * idx + 2], scale_modifier, q,
                   dcov, ds, dr);
    dL_dscale[3 * idx] = ds[0]; dL_dscale[3 * idx + 1] = ds[1]; dL_dscale[3 * idx + 2] = ds[2];
    reinterpret_cast<float4*>(dL_drot)[idx] = make_float4(dr[0], dr[1], dr[2], dr[3]);
  }
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
