// Backward of the per-Gaussian preprocess for gfx950: one fused kernel for what the reference
// runs as computeCov2DCUDA + preprocessCUDA (cuda_rasterizer/backward.cu:144-274, :346-412) with
// their helpers (SH backward :20-139, Sigma backward :278-341, dnormvdv auxiliary.h:107-117).
//
// Per visible Gaussian: d conic -> d cov2D -> d Sigma (6) and d mean (through J); d mean2D ->
// d mean (through the projection); d depth -> d mean; d colour -> d SH and d mean (view
// direction); d Sigma -> d scale, d (unnormalised) quaternion.  The mean-gradient parts are
// summed in the reference's order (cov2D part, projection part, depth part, SH part).
//
// Sigma and T are RECOMPUTED with the forward's own functions (gaussian_math.h, same
// -ffp-contract=off arithmetic), so nothing but the 48-byte record survives from the forward;
// the SH clamp mask comes from the record (r[2].w).  HBM-bound streaming kernel: reads (71+12M) B and
// writes up to (64+12M) B per Gaussian.
#include "common.h"

#pragma clang fp contract(off)

#include "gaussian_math.h"

namespace grpg {

// auxiliary.h:107-117
__device__ __forceinline__ void dnormvdv3(const float vx, const float vy, const float vz,
                                          const float dx, const float dy, const float dz,
                                          float& ox, float& oy, float& oz) {
  const float sum2 = vx * vx + vy * vy + vz * vz;
  const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
  ox = ((+sum2 - vx * vx) * dx - vy * vx * dy - vz * vx * dz) * invsum32;
  oy = (-vx * vy * dx + (sum2 - vy * vy) * dy - vz * vy * dz) * invsum32;
  oz = (-vx * vz * dx - vy * vz * dy + (sum2 - vz * vz) * dz) * invsum32;
}

// backward.cu:20-139.  Writes dL_dsh[0..(deg+1)^2) and returns the mean gradient part.
__device__ __forceinline__ void sh_backward(const int deg, const float* __restrict__ sh,
                                            const float dox, const float doy, const float doz,
                                            const uint32_t clamped, const float* dL_dcolor3,
                                            float* __restrict__ dL_dsh, float& gmx, float& gmy,
                                            float& gmz) {
  const float len = sqrtf(dox * dox + doy * doy + doz * doz);
  const float x = dox / len, y = doy / len, z = doz / len;
  float dL_dRGB[3];
#pragma unroll
  for (int c = 0; c < 3; c++) dL_dRGB[c] = dL_dcolor3[c] * (((clamped >> c) & 1u) ? 0.f : 1.f);
  float dRGBdx[3] = {0, 0, 0}, dRGBdy[3] = {0, 0, 0}, dRGBdz[3] = {0, 0, 0};
#define SH(k, c) sh[3 * (k) + (c)]
#define DSH(k, v)                                                    \
  {                                                                  \
    const float v_ = (v);                                            \
    _Pragma("unroll") for (int c = 0; c < 3; c++) dL_dsh[3 * (k) + c] = v_ * dL_dRGB[c]; \
  }
  DSH(0, SH_C0);
  if (deg > 0) {
    DSH(1, -SH_C1 * y);
    DSH(2, SH_C1 * z);
    DSH(3, -SH_C1 * x);
#pragma unroll
    for (int c = 0; c < 3; c++) {
      dRGBdx[c] = -SH_C1 * SH(3, c);
      dRGBdy[c] = -SH_C1 * SH(1, c);
      dRGBdz[c] = SH_C1 * SH(2, c);
    }
    if (deg > 1) {
      const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      DSH(4, SH_C2[0] * xy);
      DSH(5, SH_C2[1] * yz);
      DSH(6, SH_C2[2] * (2.f * zz - xx - yy));
      DSH(7, SH_C2[3] * xz);
      DSH(8, SH_C2[4] * (xx - yy));
#pragma unroll
      for (int c = 0; c < 3; c++) {
        dRGBdx[c] += SH_C2[0] * y * SH(4, c) + SH_C2[2] * 2.f * -x * SH(6, c) + SH_C2[3] * z * SH(7, c) + SH_C2[4] * 2.f * x * SH(8, c);
        dRGBdy[c] += SH_C2[0] * x * SH(4, c) + SH_C2[1] * z * SH(5, c) + SH_C2[2] * 2.f * -y * SH(6, c) + SH_C2[4] * 2.f * -y * SH(8, c);
        dRGBdz[c] += SH_C2[1] * y * SH(5, c) + SH_C2[2] * 2.f * 2.f * z * SH(6, c) + SH_C2[3] * x * SH(7, c);
      }
      if (deg > 2) {
        DSH(9, SH_C3[0] * y * (3.f * xx - yy));
        DSH(10, SH_C3[1] * xy * z);
        DSH(11, SH_C3[2] * y * (4.f * zz - xx - yy));
        DSH(12, SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy));
        DSH(13, SH_C3[4] * x * (4.f * zz - xx - yy));
        DSH(14, SH_C3[5] * z * (xx - yy));
        DSH(15, SH_C3[6] * x * (xx - 3.f * yy));
#pragma unroll
        for (int c = 0; c < 3; c++) {
          dRGBdx[c] += (SH_C3[0] * SH(9, c) * 3.f * 2.f * xy + SH_C3[1] * SH(10, c) * yz +
                        SH_C3[2] * SH(11, c) * -2.f * xy + SH_C3[3] * SH(12, c) * -3.f * 2.f * xz +
                        SH_C3[4] * SH(13, c) * (-3.f * xx + 4.f * zz - yy) +
                        SH_C3[5] * SH(14, c) * 2.f * xz + SH_C3[6] * SH(15, c) * 3.f * (xx - yy));
          dRGBdy[c] += (SH_C3[0] * SH(9, c) * 3.f * (xx - yy) + SH_C3[1] * SH(10, c) * xz +
                        SH_C3[2] * SH(11, c) * (-3.f * yy + 4.f * zz - xx) +
                        SH_C3[3] * SH(12, c) * -3.f * 2.f * yz + SH_C3[4] * SH(13, c) * -2.f * xy +
                        SH_C3[5] * SH(14, c) * -2.f * yz + SH_C3[6] * SH(15, c) * -3.f * 2.f * xy);
          dRGBdz[c] += (SH_C3[1] * SH(10, c) * xy + SH_C3[2] * SH(11, c) * 4.f * 2.f * yz +
                        SH_C3[3] * SH(12, c) * 3.f * (2.f * zz - xx - yy) +
                        SH_C3[4] * SH(13, c) * 4.f * 2.f * xz + SH_C3[5] * SH(14, c) * (xx - yy));
        }
      }
    }
  }
#undef SH
#undef DSH
  const float ddx = dRGBdx[0] * dL_dRGB[0] + dRGBdx[1] * dL_dRGB[1] + dRGBdx[2] * dL_dRGB[2];
  const float ddy = dRGBdy[0] * dL_dRGB[0] + dRGBdy[1] * dL_dRGB[1] + dRGBdy[2] * dL_dRGB[2];
  const float ddz = dRGBdz[0] * dL_dRGB[0] + dRGBdz[1] * dL_dRGB[1] + dRGBdz[2] * dL_dRGB[2];
  dnormvdv3(dox, doy, doz, ddx, ddy, ddz, gmx, gmy, gmz);
}

// backward.cu:278-341.  glm index convention [column][row] throughout.
__device__ __forceinline__ void cov3d_backward(const float s0, const float s1, const float s2,
                                               const float mod, const float4 q,
                                               const float* __restrict__ d, float* dscale,
                                               float* drot) {
  const float r = q.x, x = q.y, y = q.z, z = q.w;
  const float R[3][3] = {
      {1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},
      {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
      {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)}};
  const float s[3] = {mod * s0, mod * s1, mod * s2};
  float M[3][3];
#pragma unroll
  for (int c = 0; c < 3; c++) {
    M[c][0] = s[0] * R[c][0] + 0.f * R[c][1] + 0.f * R[c][2];
    M[c][1] = 0.f * R[c][0] + s[1] * R[c][1] + 0.f * R[c][2];
    M[c][2] = 0.f * R[c][0] + 0.f * R[c][1] + s[2] * R[c][2];
  }
  const float dS[3][3] = {{d[0], 0.5f * d[1], 0.5f * d[2]},
                          {0.5f * d[1], d[3], 0.5f * d[4]},
                          {0.5f * d[2], 0.5f * d[4], d[5]}};
  float dL_dM[3][3];   // (2.0f * M) * dL_dSigma
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int rr = 0; rr < 3; rr++)
      dL_dM[c][rr] = (2.0f * M[0][rr]) * dS[c][0] + (2.0f * M[1][rr]) * dS[c][1] + (2.0f * M[2][rr]) * dS[c][2];
  // Rt[c][r] = R[r][c];  dL_dMt[c][r] = dL_dM[r][c]
  float dMt[3][3];
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int rr = 0; rr < 3; rr++) dMt[c][rr] = dL_dM[rr][c];
#pragma unroll
  for (int k = 0; k < 3; k++)
    dscale[k] = R[0][k] * dMt[k][0] + R[1][k] * dMt[k][1] + R[2][k] * dMt[k][2];
#pragma unroll
  for (int k = 0; k < 3; k++)
#pragma unroll
    for (int rr = 0; rr < 3; rr++) dMt[k][rr] *= s[k];
  drot[0] = 2 * z * (dMt[0][1] - dMt[1][0]) + 2 * y * (dMt[2][0] - dMt[0][2]) + 2 * x * (dMt[1][2] - dMt[2][1]);
  drot[1] = 2 * y * (dMt[1][0] + dMt[0][1]) + 2 * z * (dMt[2][0] + dMt[0][2]) + 2 * r * (dMt[1][2] - dMt[2][1]) - 4 * x * (dMt[2][2] + dMt[1][1]);
  drot[2] = 2 * x * (dMt[1][0] + dMt[0][1]) + 2 * r * (dMt[2][0] - dMt[0][2]) + 2 * z * (dMt[1][2] + dMt[2][1]) - 4 * y * (dMt[2][2] + dMt[0][0]);
  drot[3] = 2 * r * (dMt[0][1] - dMt[1][0]) + 2 * x * (dMt[2][0] + dMt[0][2]) + 2 * y * (dMt[1][2] + dMt[2][1]) - 4 * z * (dMt[1][1] + dMt[0][0]);
}

__global__ void __launch_bounds__(256)
preprocess_backward_kernel(const int P, const int D, const int M, const float* __restrict__ means3D,
                           const int* __restrict__ radii, const float* __restrict__ shs,
                           const RecView rec, const float* __restrict__ scales,
                           const float* __restrict__ rotations, const float scale_modifier,
                           const float* __restrict__ cov3D_precomp, const float* __restrict__ view,
                           const float* __restrict__ proj, const float* __restrict__ campos,
                           const float h_x, const float h_y, const float tan_fovx,
                           const float tan_fovy, const float* __restrict__ dL_dmean2D,
                           const float* __restrict__ dL_dconics, float* __restrict__ dL_dmeans,
                           const float* __restrict__ dL_dcolor, const float* __restrict__ dL_ddepth,
                           float* __restrict__ dL_dcov, float* __restrict__ dL_dsh,
                           float* __restrict__ dL_dscale, float* __restrict__ dL_drot) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= P || !(radii[idx] > 0)) return;
  const float mx = means3D[3 * idx], my = means3D[3 * idx + 1], mz = means3D[3 * idx + 2];
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
  // ---- computeCov2DCUDA, backward.cu:144-274 ----
  Cov2D cv;
  cov2d_project(mx, my, mz, view, h_x, h_y, tan_fovx, tan_fovy, c3, cv);
  const float dconx = dL_dconics[4 * idx], dcony = dL_dconics[4 * idx + 1], dconz = dL_dconics[4 * idx + 3];
  const float x_grad_mul = (cv.txtz < -cv.limx || cv.txtz > cv.limx) ? 0.f : 1.f;
  const float y_grad_mul = (cv.tytz < -cv.limy || cv.tytz > cv.limy) ? 0.f : 1.f;
  const float a = cv.a + 0.3f, b = cv.b, c = cv.c + 0.3f;
  const float denom = a * c - b * b;
  float dL_da = 0, dL_db = 0, dL_dc = 0;
  const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
  const float T00 = cv.T00, T01 = cv.T01, T02 = cv.T02, T10 = cv.T10, T11 = cv.T11, T12 = cv.T12;
  float dcov[6];
  if (denom2inv != 0) {
    dL_da = denom2inv * (-c * c * dconx + 2 * b * c * dcony + (denom - a * c) * dconz);
    dL_dc = denom2inv * (-a * a * dconz + 2 * a * b * dcony + (denom - a * c) * dconx);
    dL_db = denom2inv * 2 * (b * c * dconx - (denom + 2 * b * b) * dcony + a * b * dconz);
    dcov[0] = (T00 * T00 * dL_da + T00 * T10 * dL_db + T10 * T10 * dL_dc);
    dcov[3] = (T01 * T01 * dL_da + T01 * T11 * dL_db + T11 * T11 * dL_dc);
    dcov[5] = (T02 * T02 * dL_da + T02 * T12 * dL_db + T12 * T12 * dL_dc);
    dcov[1] = 2 * T00 * T01 * dL_da + (T00 * T11 + T01 * T10) * dL_db + 2 * T10 * T11 * dL_dc;
    dcov[2] = 2 * T00 * T02 * dL_da + (T00 * T12 + T02 * T10) * dL_db + 2 * T10 * T12 * dL_dc;
    dcov[4] = 2 * T02 * T01 * dL_da + (T01 * T12 + T02 * T11) * dL_db + 2 * T11 * T12 * dL_dc;
  } else {
#pragma unroll
    for (int i = 0; i < 6; i++) dcov[i] = 0;
  }
#pragma unroll
  for (int i = 0; i < 6; i++) dL_dcov[6 * idx + i] = dcov[i];
  // Vrk[c][r] (symmetric)
  const float V00 = c3[0], V01 = c3[1], V02 = c3[2], V11 = c3[3], V12 = c3[4], V22 = c3[5];
  const float dL_dT00 = 2 * (T00 * V00 + T01 * V01 + T02 * V02) * dL_da + (T10 * V00 + T11 * V01 + T12 * V02) * dL_db;
  const float dL_dT01 = 2 * (T00 * V01 + T01 * V11 + T02 * V12) * dL_da + (T10 * V01 + T11 * V11 + T12 * V12) * dL_db;
  const float dL_dT02 = 2 * (T00 * V02 + T01 * V12 + T02 * V22) * dL_da + (T10 * V02 + T11 * V12 + T12 * V22) * dL_db;
  const float dL_dT10 = 2 * (T10 * V00 + T11 * V01 + T12 * V02) * dL_dc + (T00 * V00 + T01 * V01 + T02 * V02) * dL_db;
  const float dL_dT11 = 2 * (T10 * V01 + T11 * V11 + T12 * V12) * dL_dc + (T00 * V01 + T01 * V11 + T02 * V12) * dL_db;
  const float dL_dT12 = 2 * (T10 * V02 + T11 * V12 + T12 * V22) * dL_dc + (T00 * V02 + T01 * V12 + T02 * V22) * dL_db;
  // W[c][r]: W[0]=(v0,v4,v8) W[1]=(v1,v5,v9) W[2]=(v2,v6,v10)
  const float dL_dJ00 = view[0] * dL_dT00 + view[4] * dL_dT01 + view[8] * dL_dT02;
  const float dL_dJ02 = view[2] * dL_dT00 + view[6] * dL_dT01 + view[10] * dL_dT02;
  const float dL_dJ11 = view[1] * dL_dT10 + view[5] * dL_dT11 + view[9] * dL_dT12;
  const float dL_dJ12 = view[2] * dL_dT10 + view[6] * dL_dT11 + view[10] * dL_dT12;
  const float tz = 1.f / cv.tz;
  const float tz2 = tz * tz;
  const float tz3 = tz2 * tz;
  const float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
  const float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
  const float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * cv.tx) * tz3 * dL_dJ02 + (2 * h_y * cv.ty) * tz3 * dL_dJ12;
  // transformVec4x3Transpose (auxiliary.h:88-96); this is the "overwrite" of backward.cu:273
  float gmx = view[0] * dL_dtx + view[1] * dL_dty + view[2] * dL_dtz;
  float gmy = view[4] * dL_dtx + view[5] * dL_dty + view[6] * dL_dtz;
  float gmz = view[8] * dL_dtx + view[9] * dL_dty + view[10] * dL_dtz;

  // ---- preprocessCUDA (backward), backward.cu:346-412 ----
  const float m_hw = proj[3] * mx + proj[7] * my + proj[11] * mz + proj[15];
  const float m_w = 1.0f / (m_hw + 0.0000001f);
  const float mul1 = (proj[0] * mx + proj[4] * my + proj[8] * mz + proj[12]) * m_w * m_w;
  const float mul2 = (proj[1] * mx + proj[5] * my + proj[9] * mz + proj[13]) * m_w * m_w;
  const float g2x = dL_dmean2D[3 * idx], g2y = dL_dmean2D[3 * idx + 1];
  const float d1x = (proj[0] * m_w - proj[3] * mul1) * g2x + (proj[1] * m_w - proj[3] * mul2) * g2y;
  const float d1y = (proj[4] * m_w - proj[7] * mul1) * g2x + (proj[5] * m_w - proj[7] * mul2) * g2y;
  const float d1z = (proj[8] * m_w - proj[11] * mul1) * g2x + (proj[9] * m_w - proj[11] * mul2) * g2y;
  gmx += d1x; gmy += d1y; gmz += d1z;
  const float mul3 = view[2] * mx + view[6] * my + view[10] * mz + view[14];
  const float gdep = dL_ddepth[idx];
  const float d2x = (view[2] - view[3] * mul3) * gdep;
  const float d2y = (view[6] - view[7] * mul3) * gdep;
  const float d2z = (view[10] - view[11] * mul3) * gdep;
  gmx += d2x; gmy += d2y; gmz += d2z;
  if (shs != nullptr) {
    const uint32_t clamped = __float_as_uint(rec.colour(idx).w);
    const float dc3[3] = {dL_dcolor[3 * idx], dL_dcolor[3 * idx + 1], dL_dcolor[3 * idx + 2]};
    float sx, sy, sz;
    sh_backward(D, shs + (size_t)idx * M * 3, mx - campos[0], my - campos[1], mz - campos[2],
                clamped, dc3, dL_dsh + (size_t)idx * M * 3, sx, sy, sz);
    gmx += sx; gmy += sy; gmz += sz;
  }
  dL_dmeans[3 * idx] = gmx;
  dL_dmeans[3 * idx + 1] = gmy;
  dL_dmeans[3 * idx + 2] = gmz;
  if (scales != nullptr) {
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
                                const float* dL_dmean2D, const float* dL_dconic,
                                float* dL_dmean3D, const float* dL_dcolor, const float* dL_ddepth,
                                float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot) {
  if (P <= 0) return;
  preprocess_backward_kernel<<<(P + 255) / 256, 256, 0, s>>>(
      P, D, M, means3D, radii, shs, rec, scales, rotations, scale_modifier, cov3D_precomp, cam.view,
      cam.proj, cam.campos, cam.focal_x, cam.focal_y, cam.tan_fovx, cam.tan_fovy, dL_dmean2D,
      dL_dconic, dL_dmean3D, dL_dcolor, dL_ddepth, dL_dcov3D, dL_dsh, dL_dscale, dL_drot);
}

}  // namespace grpg
