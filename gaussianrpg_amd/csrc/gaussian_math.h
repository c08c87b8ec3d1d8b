// Device math shared by preprocess.hip (forward) and preprocess_bwd.hip (backward).
// Every expression is written in the evaluation order of the reference source / glm 0.9.9.9
// and both users are compiled with -ffp-contract=off: see the ARITHMETIC CONTRACT note in
// preprocess.hip.  The backward recomputes Sigma and T with these same functions, so it sees
// bit-identical intermediates to the forward without storing them (the reference stores
// cov3D [24 B/Gaussian] and clamped[3] in its geometry blob, rasterizer_impl.cu:155-170).
#pragma once
#include "common.h"

namespace grpg {

__device__ const float SH_C0 = 0.28209479177387814f;
__device__ const float SH_C1 = 0.4886025119029199f;
__device__ const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f,
                                   0.31539156525252005f, -1.0925484305920792f,
                                   0.5462742152960396f};
__device__ const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f,
                                   -0.4570457994644658f, 0.3731763325901154f,
                                   -0.4570457994644658f, 1.445305721320277f,
                                   -0.5900435899266435f};

// auxiliary.h:41-44 -- fp64 intermediate, one rounding to fp32.
__device__ __forceinline__ float ndc2pix(float v, int S) {
  return (float)((((double)v + 1.0) * (double)S - 1.0) * 0.5);
}

// [P,4] quaternion: one 16-byte load when the array is 16-byte aligned (always, for a tensor
// torch allocated), four dword loads for an offset view.
__device__ __forceinline__ float4 load_quat(const float* __restrict__ rotations, const int idx) {
  if ((reinterpret_cast<uintptr_t>(rotations) & 15) == 0)
    return reinterpret_cast<const float4*>(rotations)[idx];
  return make_float4(rotations[4 * idx], rotations[4 * idx + 1], rotations[4 * idx + 2],
                     rotations[4 * idx + 3]);
}

struct Projected {
  float px, py, depth;
  float conic[3];
  float cov3d[6];
  int radius;
  int minx, miny, maxx, maxy;
};

// Sigma from (scale, unnormalised quaternion): forward.cu:118-152.
// glm: M = S*R with M[c][r] = s_r * R[c][r];  Sigma[c][r] = sum_k M[r][k]*M[c][k].
__device__ __forceinline__ void cov3d_from_scale_rot(const float s0, const float s1,
                                                     const float s2, const float mod,
                                                     const float4 q, float* cov) {
  const float r = q.x, x = q.y, y = q.z, z = q.w;
  const float sx = mod * s0, sy = mod * s1, sz = mod * s2;
  // R[c][r] as the column-major glm literal
  const float R00 = 1.f - 2.f * (y * y + z * z), R01 = 2.f * (x * y - r * z), R02 = 2.f * (x * z + r * y);
  const float R10 = 2.f * (x * y + r * z), R11 = 1.f - 2.f * (x * x + z * z), R12 = 2.f * (y * z - r * x);
  const float R20 = 2.f * (x * z - r * y), R21 = 2.f * (y * z + r * x), R22 = 1.f - 2.f * (x * x + y * y);
  // (S*R)[c][r] = S[0][r]*R[c][0] + S[1][r]*R[c][1] + S[2][r]*R[c][2] with S diagonal; the
  // zero terms are kept so that -0/NaN propagate exactly like glm's product.
  const float M00 = sx * R00 + 0.f * R01 + 0.f * R02, M01 = 0.f * R00 + sy * R01 + 0.f * R02, M02 = 0.f * R00 + 0.f * R01 + sz * R02;
  const float M10 = sx * R10 + 0.f * R11 + 0.f * R12, M11 = 0.f * R10 + sy * R11 + 0.f * R12, M12 = 0.f * R10 + 0.f * R11 + sz * R12;
  const float M20 = sx * R20 + 0.f * R21 + 0.f * R22, M21 = 0.f * R20 + sy * R21 + 0.f * R22, M22 = 0.f * R20 + 0.f * R21 + sz * R22;
  // Sigma = transpose(M) * M : Sigma[c][r] = Mt[0][r]*M[c][0] + Mt[1][r]*M[c][1] + Mt[2][r]*M[c][2]
  //                                        = M[r][0]*M[c][0] + M[r][1]*M[c][1] + M[r][2]*M[c][2]
  cov[0] = M00 * M00 + M01 * M01 + M02 * M02;  // Sigma[0][0]
  cov[1] = M10 * M00 + M11 * M01 + M12 * M02;  // Sigma[0][1]
  cov[2] = M20 * M00 + M21 * M01 + M22 * M02;  // Sigma[0][2]
  cov[3] = M10 * M10 + M11 * M11 + M12 * M12;  // Sigma[1][1]
  cov[4] = M20 * M10 + M21 * M11 + M22 * M12;  // Sigma[1][2]
  cov[5] = M20 * M20 + M21 * M21 + M22 * M22;  // Sigma[2][2]
}

// EWA projection (forward.cu:74-113; recomputed identically by backward.cu:163-199).
// Cov2D keeps the intermediates the backward needs.  (a,b,c) is cov2D BEFORE the low-pass.
struct Cov2D {
  float tx, ty, tz;          // view-space mean, x/y after the 1.3*tanfov clamp
  float txtz, tytz, limx, limy;
  float T00, T01, T02, T10, T11, T12;   // T = W*J, glm [column][row]
  float a, b, c;
};

__device__ __forceinline__ void cov2d_project(const float mx, const float my, const float mz,
                                              const float* __restrict__ view, const float focal_x,
                                              const float focal_y, const float tan_fovx,
                                              const float tan_fovy, const float* c3, Cov2D& o) {
  float tx = view[0] * mx + view[4] * my + view[8] * mz + view[12];
  float ty = view[1] * mx + view[5] * my + view[9] * mz + view[13];
  const float tz = view[2] * mx + view[6] * my + view[10] * mz + view[14];
  const float limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
  const float txtz = tx / tz, tytz = ty / tz;
  tx = fminf(limx, fmaxf(-limx, txtz)) * tz;
  ty = fminf(limy, fmaxf(-limy, tytz)) * tz;
  const float J00 = focal_x / tz, J02 = -(focal_x * tx) / (tz * tz);
  const float J11 = focal_y / tz, J12 = -(focal_y * ty) / (tz * tz);
  // W[c][r]: W[0]=(v0,v4,v8) W[1]=(v1,v5,v9) W[2]=(v2,v6,v10);  T = W*J
  // T[c][r] = W[0][r]*J[c][0] + W[1][r]*J[c][1] + W[2][r]*J[c][2]
  const float W00 = view[0], W01 = view[4], W02 = view[8];
  const float W10 = view[1], W11 = view[5], W12 = view[9];
  const float W20 = view[2], W21 = view[6], W22 = view[10];
  const float T00 = W00 * J00 + W10 * 0.0f + W20 * J02;
  const float T01 = W01 * J00 + W11 * 0.0f + W21 * J02;
  const float T02 = W02 * J00 + W12 * 0.0f + W22 * J02;
  const float T10 = W00 * 0.0f + W10 * J11 + W20 * J12;
  const float T11 = W01 * 0.0f + W11 * J11 + W21 * J12;
  const float T12 = W02 * 0.0f + W12 * J11 + W22 * J12;
  // Vrk symmetric: Vrk[0]=(c0,c1,c2) Vrk[1]=(c1,c3,c4) Vrk[2]=(c2,c4,c5)
  // A = Tt*Vt : A[c][r] = T[r][0]*Vrk[0][c] + T[r][1]*Vrk[1][c] + T[r][2]*Vrk[2][c]
  const float V00 = c3[0], V01 = c3[1], V02 = c3[2], V11 = c3[3], V12 = c3[4], V22 = c3[5];
  const float A00 = T00 * V00 + T01 * V01 + T02 * V02;  // A[0][0]
  const float A10 = T00 * V01 + T01 * V11 + T02 * V12;  // A[1][0]
  const float A20 = T00 * V02 + T01 * V12 + T02 * V22;  // A[2][0]
  const float A01 = T10 * V00 + T11 * V01 + T12 * V02;  // A[0][1]
  const float A11 = T10 * V01 + T11 * V11 + T12 * V12;  // A[1][1]
  const float A21 = T10 * V02 + T11 * V12 + T12 * V22;  // A[2][1]
  // cov = A*T : cov[c][r] = A[0][r]*T[c][0] + A[1][r]*T[c][1] + A[2][r]*T[c][2]
  o.a = A00 * T00 + A10 * T01 + A20 * T02;  // cov[0][0]
  o.b = A01 * T00 + A11 * T01 + A21 * T02;  // cov[0][1]
  o.c = A01 * T10 + A11 * T11 + A21 * T12;  // cov[1][1]
  o.tx = tx; o.ty = ty; o.tz = tz; o.txtz = txtz; o.tytz = tytz; o.limx = limx; o.limy = limy;
  o.T00 = T00; o.T01 = T01; o.T02 = T02; o.T10 = T10; o.T11 = T11; o.T12 = T12;
}

}  // namespace grpg
