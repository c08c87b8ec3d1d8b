/*
 * gs_oracle.c -- TEST INFRASTRUCTURE ONLY.  Scalar CPU restatement of the
 * Street-Gaussians tile rasterizer used by GimpelZhang/GaussianRPG.
 *
 * Nothing in the product path (gaussianrpg_amd/, diff_gaussian_rasterization/)
 * may import, link or call this file.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg use it, and only as the checker.
 *
 * Shorthands for the citations below (all under
 * /root/reference/submodules/diff-gaussian-rasterization/):
 *   CR/  = cuda_rasterizer/
 *   GLM/ = third_party/glm/glm/
 *
 * PARITY PINNING STATUS -- "parity unpinned" for the CUDA stages.
 *   The reference path is CUDA only (needs nvcc, cuda_runtime.h,
 *   cooperative_groups.h, cub); none of those exist in this image and the
 *   build rules forbid header stand-ins, so the reference kernels cannot be
 *   executed here, and the reference ships no golden vectors / asserts for
 *   this path (script/test_gaussian_rasterization.py compares nothing).
 *   What IS pinned, by vectors generated from the importable reference
 *   Python (tests/golden/make_golden.py):
 *     - SH->RGB, computeColorFromSH (lib/utils/sh_utils.py eval_sh;
 *       ref_sh.npz, tests/test_oracle_golden.py);
 *     - the camera / projection conventions (lib/utils/graphics_utils.py;
 *       ref_camera.npz);
 *     - the quaternion -> R convention of quat_to_R_glm / computeCov3D
 *       (lib/utils/general_utils.py quaternion_to_matrix_numpy; ref_quat.npz,
 *       tests/test_reference_pins.py) -- round 3;
 *     - computeColorFromSH_bwd: dL_dsh and the direction part of dL_dmeans3D
 *       (torch.autograd through the reference's eval_sh; ref_sh_bwd.npz,
 *       tests/test_reference_pins.py) -- round 3, the first reference-derived
 *       pin of anything in the backward;
 *     - computeCov3D as a whole -- scale-modifier placement, (S R)^T (S R),
 *       order of the six entries -- against the reference's own Python route
 *       (get_covariance = build_covariance_from_scaling_rotation,
 *       lib/models/gaussian_model.py:208-212; ref_cov3d.npz,
 *       tests/test_reference_pins.py) -- round 4.
 *   Everything else (EWA projection, tile binning, blend, the rest of the
 *   backward) is pinned only by agreement of three independent
 *   implementations: this file, oracle/torch_splat.py (+ fp64 autograd) and
 *   the HIP kernels.
 *
 * CANONICAL ARITHMETIC.  IEEE binary32, round-to-nearest, NO fused
 * multiply-add contraction (compile with -ffp-contract=off), every
 * expression evaluated in the order the reference source / glm 0.9.9.9
 * writes it.  nvcc would have contracted some of these into FMAs, so even
 * the real CUDA build differs from this by ulps; the integer outputs
 * (radii, tile rects, keys, ranges) of the HIP preprocess are bit-exact
 * against THIS arithmetic.
 */
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define BLOCK_X 16 /* CR/config.h:17 */
#define BLOCK_Y 16 /* CR/config.h:18 */
#define BLOCK_SIZE (BLOCK_X * BLOCK_Y)

/* CR/auxiliary.h:22-39 */
static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f,
                               0.31539156525252005f, -1.0925484305920792f,
                               0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f,
                               -0.4570457994644658f, 0.3731763325901154f,
                               -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

/* float -> int as the GPU does it (cvt.rzi.s32.f32 / v_cvt_i32_f32):
 * truncate toward zero, saturate, NaN -> 0.  Plain C casts are UB there. */
static int f2i_rz_sat(float x) {
  if (x != x) return 0;
  if (x >= 2147483648.0f) return 2147483647;
  if (x <= -2147483648.0f) return (-2147483647 - 1);
  return (int)x;
}

/* CR/auxiliary.h:41-44.  The literals are doubles: evaluated in fp64 and
 * rounded to fp32 once on return. */
static float ndc2Pix(float v, int S) {
  return (float)((((double)v + 1.0) * (double)S - 1.0) * 0.5);
}

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* CR/auxiliary.h:46-56 */
static void getRect(float px, float py, int max_radius, int gx, int gy,
                    int* minx, int* miny, int* maxx, int* maxy) {
  const float r = (float)max_radius;
  *minx = imin(gx, imax(0, f2i_rz_sat((px - r) / (float)BLOCK_X)));
  *miny = imin(gy, imax(0, f2i_rz_sat((py - r) / (float)BLOCK_Y)));
  *maxx = imin(gx, imax(0, f2i_rz_sat((px + r + (float)(BLOCK_X - 1)) / (float)BLOCK_X)));
  *maxy = imin(gy, imax(0, f2i_rz_sat((py + r + (float)(BLOCK_Y - 1)) / (float)BLOCK_Y)));
}

/* CR/auxiliary.h:58-66 */
static void transformPoint4x3(const float* p, const float* m, float* o) {
  o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
  o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
  o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
/* CR/auxiliary.h:68-77 */
static void transformPoint4x4(const float* p, const float* m, float* o) {
  o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
  o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
  o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
  o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}

/* glm mat3 helpers.  Storage is glm's: m[c][r] = column c, row r.
 * Product follows GLM/detail/type_mat3x3.inl:486-519 term by term:
 * (A*B)[c][r] = A[0][r]*B[c][0] + A[1][r]*B[c][1] + A[2][r]*B[c][2]. */
typedef float mat3[3][3];
static void m3_mul(mat3 A, mat3 B, mat3 out) {
  mat3 t;
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++)
      t[c][r] = A[0][r] * B[c][0] + A[1][r] * B[c][1] + A[2][r] * B[c][2];
  memcpy(out, t, sizeof(mat3));
}
static void m3_transpose(mat3 A, mat3 out) {
  mat3 t;
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++) t[c][r] = A[r][c];
  memcpy(out, t, sizeof(mat3));
}

/* R as the glm literal in CR/forward.cu:134-138 builds it (column-major
 * constructor: first three scalars are column 0). */
static void quat_to_R_glm(const float* q, mat3 R) {
  const float r = q[0], x = q[1], y = q[2], z = q[3];
  R[0][0] = 1.f - 2.f * (y * y + z * z);
  R[0][1] = 2.f * (x * y - r * z);
  R[0][2] = 2.f * (x * z + r * y);
  R[1][0] = 2.f * (x * y + r * z);
  R[1][1] = 1.f - 2.f * (x * x + z * z);
  R[1][2] = 2.f * (y * z - r * x);
  R[2][0] = 2.f * (x * z - r * y);
  R[2][1] = 2.f * (y * z + r * x);
  R[2][2] = 1.f - 2.f * (x * x + y * y);
}

/* CR/forward.cu:118-152.  Quaternion is used as given (no normalisation). */
static void computeCov3D(const float* scale, float mod, const float* rot, float* cov3D) {
  mat3 S = {{0}}, R, M, Mt, Sigma;
  S[0][0] = mod * scale[0];
  S[1][1] = mod * scale[1];
  S[2][2] = mod * scale[2];
  quat_to_R_glm(rot, R);
  m3_mul(S, R, M);
  m3_transpose(M, Mt);
  m3_mul(Mt, M, Sigma);
  cov3D[0] = Sigma[0][0];
  cov3D[1] = Sigma[0][1];
  cov3D[2] = Sigma[0][2];
  cov3D[3] = Sigma[1][1];
  cov3D[4] = Sigma[1][2];
  cov3D[5] = Sigma[2][2];
}

/* Shared between forward (CR/forward.cu:74-113) and backward
 * (CR/backward.cu:163-199): builds t (clamped), J, W, T=W*J, Vrk, cov2D. */
typedef struct {
  float t[3];
  float txtz, tytz, limx, limy;
  mat3 J, W, T, Vrk, cov;
} Cov2DCtx;

static void cov2d_common(const float* mean, float focal_x, float focal_y, float tan_fovx,
                         float tan_fovy, const float* cov3D, const float* view, Cov2DCtx* c) {
  transformPoint4x3(mean, view, c->t);
  c->limx = 1.3f * tan_fovx;
  c->limy = 1.3f * tan_fovy;
  c->txtz = c->t[0] / c->t[2];
  c->tytz = c->t[1] / c->t[2];
  c->t[0] = fminf(c->limx, fmaxf(-c->limx, c->txtz)) * c->t[2];
  c->t[1] = fminf(c->limy, fmaxf(-c->limy, c->tytz)) * c->t[2];
  const float tz = c->t[2];
  /* glm::mat3(a0..a8): columns (a0,a1,a2),(a3,a4,a5),(a6,a7,a8) */
  c->J[0][0] = focal_x / tz; c->J[0][1] = 0.0f; c->J[0][2] = -(focal_x * c->t[0]) / (tz * tz);
  c->J[1][0] = 0.0f; c->J[1][1] = focal_y / tz; c->J[1][2] = -(focal_y * c->t[1]) / (tz * tz);
  c->J[2][0] = 0; c->J[2][1] = 0; c->J[2][2] = 0;
  c->W[0][0] = view[0]; c->W[0][1] = view[4]; c->W[0][2] = view[8];
  c->W[1][0] = view[1]; c->W[1][1] = view[5]; c->W[1][2] = view[9];
  c->W[2][0] = view[2]; c->W[2][1] = view[6]; c->W[2][2] = view[10];
  m3_mul(c->W, c->J, c->T);
  c->Vrk[0][0] = cov3D[0]; c->Vrk[0][1] = cov3D[1]; c->Vrk[0][2] = cov3D[2];
  c->Vrk[1][0] = cov3D[1]; c->Vrk[1][1] = cov3D[3]; c->Vrk[1][2] = cov3D[4];
  c->Vrk[2][0] = cov3D[2]; c->Vrk[2][1] = cov3D[4]; c->Vrk[2][2] = cov3D[5];
  mat3 Tt, Vt, A;
  m3_transpose(c->T, Tt);
  m3_transpose(c->Vrk, Vt);
  m3_mul(Tt, Vt, A);
  m3_mul(A, c->T, c->cov);
}

/* CR/forward.cu:20-71.  Returns clamped flags in clamped[0..2]. */
static void computeColorFromSH(int idx, int deg, int max_coeffs, const float* means,
                               const float* campos, const float* shs, uint8_t* clamped,
                               float* out) {
  const float* pos = means + 3 * idx;
  float dir[3] = {pos[0] - campos[0], pos[1] - campos[1], pos[2] - campos[2]};
  /* glm::length = sqrt(dot), dot = (x*x + y*y) + z*z (GLM/detail/func_geometric.inl:48-55) */
  const float len = sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
  dir[0] = dir[0] / len; dir[1] = dir[1] / len; dir[2] = dir[2] / len;
  const float* sh = shs + (size_t)idx * max_coeffs * 3;
  const float x = dir[0], y = dir[1], z = dir[2];
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
                   SH_C3[4] * x * (4.0f * zz - xx - yy) * SH(13) + SH_C3[5] * z * (xx - yy) * SH(14) +
                   SH_C3[6] * x * (xx - 3.0f * yy) * SH(15);
        }
      }
    }
#undef SH
    result += 0.5f;
    clamped[c] = (result < 0);
    out[c] = fmaxf(result, 0.0f);
  }
}

/* ------------------------------------------------------------------------
 * Stage 1: preprocess (CR/forward.cu:155-256 + CR/auxiliary.h:139-164) and
 * the inclusive scan of tiles_touched (CR/rasterizer_impl.cu:280).
 * All per-Gaussian arrays have P entries; entries of culled Gaussians other
 * than radii/tiles_touched are left untouched (the reference leaves them
 * uninitialised) -- callers pre-fill them with zeros.
 * Returns num_rendered = point_offsets[P-1].
 * ---------------------------------------------------------------------- */
int64_t gso_preprocess(int P, int D, int M, const float* means3D, const float* scales,
                       float scale_modifier, const float* rotations, const float* opacities,
                       const float* shs, const float* cov3D_precomp, const float* colors_precomp,
                       const float* view, const float* proj, const float* campos, int W, int H,
                       float tan_fovx, float tan_fovy, int32_t* radii, float* means2D,
                       float* depths, float* cov3Ds, float* rgb, float* conic_opacity,
                       uint8_t* clamped, uint32_t* tiles_touched, uint32_t* point_offsets) {
  /* CR/rasterizer_impl.cu:225-226 */
  const float focal_y = H / (2.0f * tan_fovy);
  const float focal_x = W / (2.0f * tan_fovx);
  const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
  /* Gaussians are independent: the OpenMP build (libgs_oracle_omp.so, bench.py's cpu_baseline
   * leg) splits them over the host cores; results are identical, the pragma is a comment
   * without -fopenmp. */
#pragma omp parallel for schedule(static)
  for (int idx = 0; idx < P; idx++) {
    radii[idx] = 0;
    tiles_touched[idx] = 0;
    const float* p_orig = means3D + 3 * idx;
    /* in_frustum, CR/auxiliary.h:139-164 */
    float p_hom[4], p_view[3];
    transformPoint4x4(p_orig, proj, p_hom);
    const float p_w = 1.0f / (p_hom[3] + 0.0000001f);
    const float p_proj[3] = {p_hom[0] * p_w, p_hom[1] * p_w, p_hom[2] * p_w};
    transformPoint4x3(p_orig, view, p_view);
    if (p_view[2] <= 0.2f) continue;

    const float* cov3D;
    if (cov3D_precomp) {
      cov3D = cov3D_precomp + 6 * idx;
    } else {
      computeCov3D(scales + 3 * idx, scale_modifier, rotations + 4 * idx, cov3Ds + 6 * idx);
      cov3D = cov3Ds + 6 * idx;
    }
    Cov2DCtx c;
    cov2d_common(p_orig, focal_x, focal_y, tan_fovx, tan_fovy, cov3D, view, &c);
    /* low-pass, CR/forward.cu:110-112 */
    const float cov_x = c.cov[0][0] + 0.3f, cov_y = c.cov[0][1], cov_z = c.cov[1][1] + 0.3f;
    /* CR/forward.cu:219-223 */
    const float det = (cov_x * cov_z - cov_y * cov_y);
    if (det == 0.0f) continue;
    const float det_inv = 1.f / det;
    const float conic[3] = {cov_z * det_inv, -cov_y * det_inv, cov_x * det_inv};
    /* CR/forward.cu:229-237 */
    const float mid = 0.5f * (cov_x + cov_z);
    const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
    const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
    const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
    const float pix[2] = {ndc2Pix(p_proj[0], W), ndc2Pix(p_proj[1], H)};
    int minx, miny, maxx, maxy;
    getRect(pix[0], pix[1], f2i_rz_sat(my_radius), gx, gy, &minx, &miny, &maxx, &maxy);
    if ((uint32_t)(maxx - minx) * (uint32_t)(maxy - miny) == 0) continue;
    /* CR/forward.cu:241-247 */
    if (!colors_precomp) {
      computeColorFromSH(idx, D, M, means3D, campos, shs, clamped + 3 * idx, rgb + 3 * idx);
    }
    /* CR/forward.cu:249-255 */
    depths[idx] = p_view[2];
    radii[idx] = f2i_rz_sat(my_radius);
    means2D[2 * idx] = pix[0];
    means2D[2 * idx + 1] = pix[1];
    conic_opacity[4 * idx + 0] = conic[0];
    conic_opacity[4 * idx + 1] = conic[1];
    conic_opacity[4 * idx + 2] = conic[2];
    conic_opacity[4 * idx + 3] = opacities[idx];
    tiles_touched[idx] = (uint32_t)(maxy - miny) * (uint32_t)(maxx - minx);
  }
  uint32_t run = 0; /* cub::DeviceScan::InclusiveSum, CR/rasterizer_impl.cu:280 */
  for (int i = 0; i < P; i++) {
    run += tiles_touched[i];
    point_offsets[i] = run;
  }
  return P > 0 ? (int64_t)point_offsets[P - 1] : 0;
}

/* CR/rasterizer_impl.cu:35-50 */
static uint32_t getHigherMsb(uint32_t n) {
  uint32_t msb = sizeof(n) * 4;
  uint32_t step = msb;
  while (step > 1) {
    step /= 2;
    if (n >> msb) msb += step; else msb -= step;
  }
  if (n >> msb) msb++;
  return msb;
}

/* ------------------------------------------------------------------------
 * Stage 2: duplicateWithKeys (CR/rasterizer_impl.cu:70-111), stable sort of
 * the 64-bit keys on bits [0, 32+bit) (cub::DeviceRadixSort::SortPairs,
 * CR/rasterizer_impl.cu:303-311) and identifyTileRanges (:116-138, :313).
 * keys/values arrays have R = num_rendered entries; ranges has 2*T entries.
 * ---------------------------------------------------------------------- */
void gso_bin(int P, int64_t R, int W, int H, const int32_t* radii, const float* means2D,
             const float* depths, const uint32_t* point_offsets, uint64_t* keys_sorted,
             uint32_t* point_list, uint32_t* ranges) {
  const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
  uint64_t* keys = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(R > 0 ? R : 1));
  uint32_t* vals = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(R > 0 ? R : 1));
  uint64_t* keys2 = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(R > 0 ? R : 1));
  uint32_t* vals2 = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(R > 0 ? R : 1));
  for (int idx = 0; idx < P; idx++) {
    if (radii[idx] > 0) {
      uint32_t off = (idx == 0) ? 0 : point_offsets[idx - 1];
      int minx, miny, maxx, maxy;
      getRect(means2D[2 * idx], means2D[2 * idx + 1], radii[idx], gx, gy, &minx, &miny, &maxx, &maxy);
      uint32_t dbits;
      memcpy(&dbits, &depths[idx], 4);
      for (int y = miny; y < maxy; y++)
        for (int x = minx; x < maxx; x++) {
          uint64_t key = (uint64_t)(y * gx + x);
          key <<= 32;
          key |= dbits;
          keys[off] = key;
          vals[off] = (uint32_t)idx;
          off++;
        }
    }
  }
  /* stable LSD radix sort, 8-bit digits, over bits [0, 32+bit) */
  const int end_bit = 32 + (int)getHigherMsb((uint32_t)(gx * gy));
  uint64_t *ka = keys, *kb = keys2;
  uint32_t *va = vals, *vb = vals2;
  for (int shift = 0; shift < end_bit; shift += 8) {
    const int nb = (end_bit - shift) < 8 ? (end_bit - shift) : 8;
    const uint64_t mask = ((uint64_t)1 << nb) - 1;
    size_t count[257] = {0};
    for (int64_t i = 0; i < R; i++) count[((ka[i] >> shift) & mask) + 1]++;
    for (int d = 0; d < 256; d++) count[d + 1] += count[d];
    for (int64_t i = 0; i < R; i++) {
      size_t pos = count[(ka[i] >> shift) & mask]++;
      kb[pos] = ka[i];
      vb[pos] = va[i];
    }
    uint64_t* tk = ka; ka = kb; kb = tk;
    uint32_t* tv = va; va = vb; vb = tv;
  }
  memcpy(keys_sorted, ka, sizeof(uint64_t) * (size_t)R);
  memcpy(point_list, va, sizeof(uint32_t) * (size_t)R);
  memset(ranges, 0, sizeof(uint32_t) * 2 * (size_t)(gx * gy));
  for (int64_t idx = 0; idx < R; idx++) {
    const uint32_t currtile = (uint32_t)(keys_sorted[idx] >> 32);
    if (idx == 0) ranges[2 * currtile] = 0;
    else {
      const uint32_t prevtile = (uint32_t)(keys_sorted[idx - 1] >> 32);
      if (currtile != prevtile) {
        ranges[2 * prevtile + 1] = (uint32_t)idx;
        ranges[2 * currtile] = (uint32_t)idx;
      }
    }
    if (idx == R - 1) ranges[2 * currtile + 1] = (uint32_t)R;
  }
  free(keys); free(vals); free(keys2); free(vals2);
}

/* ------------------------------------------------------------------------
 * Stage 3: per-pixel front-to-back blend (CR/forward.cu:340-467).
 * features = colors_precomp or the rgb array of stage 1 ([P,3]).
 * fragile[pix] (optional) is set when any accept/reject decision of that
 * pixel was within a relative margin of its threshold, i.e. an
 * implementation whose exp()/rounding differs by a few ulp may legitimately
 * take the other branch there (SURVEY.md section 7 "Threshold flips").
 * ---------------------------------------------------------------------- */
void gso_render(int W, int H, int S, const uint32_t* ranges, const uint32_t* point_list,
                const float* means2D, const float* features, const float* depths,
                const float* semantics, const float* conic_opacity, const float* bg,
                float* out_color, float* out_depth, float* out_alpha, float* out_semantic,
                uint32_t* n_contrib, uint8_t* fragile) {
  const int gx = (W + BLOCK_X - 1) / BLOCK_X;
  const size_t HW = (size_t)H * W;
  /* pixels are independent: OpenMP build = parallel over pixel rows (dynamic: rows differ in
   * overdraw), identical results */
#pragma omp parallel for schedule(dynamic, 4)
  for (int py = 0; py < H; py++)
    for (int px = 0; px < W; px++) {
      const size_t pix_id = (size_t)W * py + px;
      const float pixf[2] = {(float)px, (float)py};
      const uint32_t r0 = ranges[2 * ((py / BLOCK_Y) * gx + (px / BLOCK_X))];
      const uint32_t r1 = ranges[2 * ((py / BLOCK_Y) * gx + (px / BLOCK_X)) + 1];
      float T = 1.0f, C[3] = {0, 0, 0}, weight = 0, Dd = 0, uT = 0;
      uint32_t contributor = 0, last_contributor = 0;
      uint8_t frag = 0;
      for (int ch = 0; ch < S; ch++) out_semantic[ch * HW + pix_id] = 0.0f;
      for (uint32_t k = r0; k < r1; k++) {
        contributor++;
        const uint32_t id = point_list[k];
        const float dx = means2D[2 * id] - pixf[0], dy = means2D[2 * id + 1] - pixf[1];
        const float* co = conic_opacity + 4 * id;
        const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
        const float mag = fabsf(0.5f * co[0] * dx * dx) + fabsf(0.5f * co[2] * dy * dy) + fabsf(co[1] * dx * dy);
        if (fabsf(power) <= 1e-5f * mag && mag > 0.0f) frag = 1;
        if (power > 0.0f) continue;
        const float alpha = fminf(0.99f, co[3] * expf(power));
        /* upow: what two float32 evaluations of `power` may differ by (see "ill-conditioned power"
         * below) = the relative uncertainty of alpha; it widens the margin of the alpha test */
        const float upow = mag * 2.4e-7f;
        if (fabsf(alpha - 1.0f / 255.0f) <= (2e-4f + upow) * (1.0f / 255.0f)) frag = 1;
        if (alpha < 1.0f / 255.0f) continue;
        /* Ill-conditioned power (round 4, found by tests/test_gpu_sweep.py): for a needle-thin splat seen
         * far along its long axis the three terms of `power` are thousands of times larger than their
         * sum (conic 0.83 / 1.00 / 1.22 at (106, -88) px: 4615 + 4686 - 9300 = -0.745), so float32
         * evaluations that round or contract differently -- this file, nvcc's FMA-contracted
         * reference, the HIP kernels' pre-scaled fma form -- disagree by ~mag * 2^-23 in power, i.e.
         * by that RELATIVE amount in alpha (1.7e-3 there).  When the resulting uncertainty of the
         * blend weight alpha * T exceeds a tenth of the 1e-4 image tolerance, the pixel is fragile. */
        if (upow * alpha * T > 1e-5f) frag = 1;
        const float test_T = T * (1 - alpha);
        uT += upow * alpha / (1 - alpha);   /* relative uncertainty T has collected so far */
        if (fabsf(test_T - 0.0001f) <= (2e-3f + 2.0f * uT) * 0.0001f) frag = 1;
        if (test_T < 0.0001f) break; /* done = true */
        for (int ch = 0; ch < 3; ch++) C[ch] += features[3 * id + ch] * alpha * T;
        for (int ch = 0; ch < S; ch++)
          out_semantic[ch * HW + pix_id] += semantics[(size_t)id * S + ch] * alpha * T;
        weight += alpha * T;
        Dd += depths[id] * alpha * T;
        T = test_T;
        last_contributor = contributor;
      }
      n_contrib[pix_id] = last_contributor;
      for (int ch = 0; ch < 3; ch++) out_color[ch * HW + pix_id] = C[ch] + T * bg[ch];
      out_alpha[pix_id] = weight;
      out_depth[pix_id] = Dd;
      if (fragile) fragile[pix_id] = frag;
    }
}

/* ------------------------------------------------------------------------
 * Stage 4: backward of the blend (CR/backward.cu:415-641).  The reference
 * sums per-(pixel,Gaussian) terms with float atomics in an unspecified
 * order; each TERM here is computed in fp32 exactly as the reference writes
 * it, and the SUM is taken in fp64 and rounded once, so it is the centre
 * of the distribution of results the reference can produce.
 * Outputs (all pre-zeroed by this function): dL_dmean2D[P,3],
 * dL_dconic[P,4] (.x .y .w used, :633-635), dL_dopacity[P], dL_dcolors[P,3],
 * dL_ddepths[P], dL_dsemantics[P,S].
 * ---------------------------------------------------------------------- */
void gso_render_backward(int P, int W, int H, int S, const uint32_t* ranges,
                         const uint32_t* point_list, const float* bg, const float* means2D,
                         const float* conic_opacity, const float* colors, const float* depths,
                         const float* semantics, const float* alphas, const uint32_t* n_contrib,
                         const float* dL_dpixels, const float* dL_dpixel_depths,
                         const float* dL_dalphas, const float* dL_dpixel_semantics,
                         float* dL_dmean2D, float* dL_dconic, float* dL_dopacity,
                         float* dL_dcolors, float* dL_ddepths, float* dL_dsemantics) {
  const int gx = (W + BLOCK_X - 1) / BLOCK_X;
  const size_t HW = (size_t)H * W;
  double* a_mean = (double*)calloc((size_t)P * 3 + 1, sizeof(double));
  double* a_conic = (double*)calloc((size_t)P * 4 + 1, sizeof(double));
  double* a_opa = (double*)calloc((size_t)P + 1, sizeof(double));
  double* a_col = (double*)calloc((size_t)P * 3 + 1, sizeof(double));
  double* a_dep = (double*)calloc((size_t)P + 1, sizeof(double));
  double* a_sem = (double*)calloc((size_t)P * (S > 0 ? S : 1) + 1, sizeof(double));
  const float ddelx_dx = (float)(0.5 * W), ddely_dy = (float)(0.5 * H); /* :501-502 */
  /* OpenMP build: pixel rows in parallel.  Every per-(pixel, Gaussian) term is still the
   * reference's fp32 expression; the terms are added into the fp64 accumulators atomically, so
   * only the ORDER of the fp64 sums differs between runs (~1e-16 relative: far below anything
   * the tests resolve).  The scalar build is the same code with the pragmas as comments. */
#pragma omp parallel for schedule(dynamic, 4)
  for (int py = 0; py < H; py++) {
    float* accum_semantic_rec = (float*)calloc(S > 0 ? S : 1, sizeof(float));
    float* last_semantic = (float*)calloc(S > 0 ? S : 1, sizeof(float));
    for (int px = 0; px < W; px++) {
      const size_t pix_id = (size_t)W * py + px;
      const float pixf[2] = {(float)px, (float)py};
      const uint32_t r0 = ranges[2 * ((py / BLOCK_Y) * gx + (px / BLOCK_X))];
      const uint32_t r1 = ranges[2 * ((py / BLOCK_Y) * gx + (px / BLOCK_X)) + 1];
      const float T_final = 1 - alphas[pix_id];
      float T = T_final;
      uint32_t contributor = r1 - r0;
      const uint32_t last_contributor = n_contrib[pix_id];
      float accum_rec[3] = {0, 0, 0}, dL_dpixel[3], accum_depth_rec = 0, accum_alpha_rec = 0;
      for (int i = 0; i < 3; i++) dL_dpixel[i] = dL_dpixels[i * HW + pix_id];
      const float dL_dpixel_depth = dL_dpixel_depths[pix_id];
      const float dL_dalpha = dL_dalphas[pix_id];
      float last_alpha = 0, last_color[3] = {0, 0, 0}, last_depth = 0;
      for (int ch = 0; ch < S; ch++) { accum_semantic_rec[ch] = 0; last_semantic[ch] = 0; }
      for (uint32_t k = r1; k-- > r0;) { /* back to front, :517-524 */
        contributor--;
        if (contributor >= last_contributor) continue;
        const uint32_t id = point_list[k];
        const float dx = means2D[2 * id] - pixf[0], dy = means2D[2 * id + 1] - pixf[1];
        const float* co = conic_opacity + 4 * id;
        const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
        if (power > 0.0f) continue;
        const float G = expf(power);
        const float alpha = fminf(0.99f, co[3] * G);
        if (alpha < 1.0f / 255.0f) continue;
        T = T / (1.f - alpha);
        const float dchannel_dcolor = alpha * T;
        float dL_dopa = 0.0f;
        for (int ch = 0; ch < 3; ch++) {
          const float c = colors[3 * id + ch];
          accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
          last_color[ch] = c;
          const float dL_dchannel = dL_dpixel[ch];
          dL_dopa += (c - accum_rec[ch]) * dL_dchannel;
          _Pragma("omp atomic")
          a_col[3 * (size_t)id + ch] += (double)(dchannel_dcolor * dL_dchannel);
        }
        for (int ch = 0; ch < S; ch++) { /* :571-587 */
          const float s = semantics[(size_t)id * S + ch];
          accum_semantic_rec[ch] = last_alpha * last_semantic[ch] + (1.f - last_alpha) * accum_semantic_rec[ch];
          last_semantic[ch] = s;
          const float dL_dchannel = dL_dpixel_semantics[ch * HW + pix_id];
          dL_dopa += (s - accum_semantic_rec[ch]) * dL_dchannel;
          _Pragma("omp atomic")
          a_sem[(size_t)id * S + ch] += (double)(dchannel_dcolor * dL_dchannel);
        }
        const float c_d = depths[id]; /* :592-597 */
        accum_depth_rec = last_alpha * last_depth + (1.f - last_alpha) * accum_depth_rec;
        last_depth = c_d;
        dL_dopa += (c_d - accum_depth_rec) * dL_dpixel_depth;
        _Pragma("omp atomic")
        a_dep[id] += (double)(dchannel_dcolor * dL_dpixel_depth);
        accum_alpha_rec = last_alpha + (1.f - last_alpha) * accum_alpha_rec; /* :601-602 */
        dL_dopa += (1 - accum_alpha_rec) * dL_dalpha;
        dL_dopa *= T;
        last_alpha = alpha;
        float bg_dot_dpixel = 0; /* :611-614 */
        for (int i = 0; i < 3; i++) bg_dot_dpixel += bg[i] * dL_dpixel[i];
        dL_dopa += (-T_final / (1.f - alpha)) * bg_dot_dpixel;
        const float dL_dG = co[3] * dL_dopa; /* :618-638 */
        const float gdx = G * dx, gdy = G * dy;
        const float dG_ddelx = -gdx * co[0] - gdy * co[1];
        const float dG_ddely = -gdy * co[2] - gdx * co[1];
        _Pragma("omp atomic")
        a_mean[3 * (size_t)id + 0] += (double)(dL_dG * dG_ddelx * ddelx_dx);
        _Pragma("omp atomic")
        a_mean[3 * (size_t)id + 1] += (double)(dL_dG * dG_ddely * ddely_dy);
        _Pragma("omp atomic")
        a_mean[3 * (size_t)id + 2] += (double)(fabsf(dL_dG * dG_ddelx * ddelx_dx) + fabsf(dL_dG * dG_ddely * ddely_dy));
        _Pragma("omp atomic")
        a_conic[4 * (size_t)id + 0] += (double)(-0.5f * gdx * dx * dL_dG);
        _Pragma("omp atomic")
        a_conic[4 * (size_t)id + 1] += (double)(-0.5f * gdx * dy * dL_dG);
        _Pragma("omp atomic")
        a_conic[4 * (size_t)id + 3] += (double)(-0.5f * gdy * dy * dL_dG);
        _Pragma("omp atomic")
        a_opa[id] += (double)(G * dL_dopa);
      }
    }
    free(accum_semantic_rec); free(last_semantic);
  }
  for (size_t i = 0; i < (size_t)P * 3; i++) dL_dmean2D[i] = (float)a_mean[i];
  for (size_t i = 0; i < (size_t)P * 4; i++) dL_dconic[i] = (float)a_conic[i];
  for (size_t i = 0; i < (size_t)P; i++) dL_dopacity[i] = (float)a_opa[i];
  for (size_t i = 0; i < (size_t)P * 3; i++) dL_dcolors[i] = (float)a_col[i];
  for (size_t i = 0; i < (size_t)P; i++) dL_ddepths[i] = (float)a_dep[i];
  for (size_t i = 0; i < (size_t)P * S; i++) dL_dsemantics[i] = (float)a_sem[i];
  free(a_mean); free(a_conic); free(a_opa); free(a_col); free(a_dep); free(a_sem);
}

/* ------------------------------------------------------------------------
 * Stage 4, ARBITER variant (not a restatement of a reference function): the gradient of the SAME formulas
 * (CR/backward.cu:500-641) evaluated in float64, for one purpose -- settling a disagreement between the HIP
 * backward and gso_render_backward above at sizes where float64 autograd through oracle/torch_splat.py does
 * not fit (tests/test_gpu_smoke_script.py: 13.3 M tile instances).
 *   - accept / reject decisions are the float32 forward's (the fp32 `power > 0`, `alpha < 1/255` tests and
 *     n_contrib), so the set of (pixel, Gaussian) terms is identical to gso_render_backward's;
 *   - every VALUE is float64, and the walk starts from the exact final transmittance -- the float64 product
 *     of (1 - alpha) over the pixel's contributors -- instead of the reference's T_final = 1 - alphas[pix]
 *     (:468), whose float32 cancellation on nearly opaque pixels is the known noise source (DESIGN.md
 *     section 3; tools/experiments/tfinal_noise.py).
 * Same outputs as gso_render_backward.
 * ---------------------------------------------------------------------- */
void gso_render_backward_f64(int P, int W, int H, int S, const uint32_t* ranges,
                             const uint32_t* point_list, const float* bg, const float* means2D,
                             const float* conic_opacity, const float* colors, const float* depths,
                             const float* semantics, const uint32_t* n_contrib,
                             const float* dL_dpixels, const float* dL_dpixel_depths,
                             const float* dL_dalphas, const float* dL_dpixel_semantics,
                             float* dL_dmean2D, float* dL_dconic, float* dL_dopacity,
                             float* dL_dcolors, float* dL_ddepths, float* dL_dsemantics) {
  const int gx = (W + BLOCK_X - 1) / BLOCK_X;
  const size_t HW = (size_t)H * W;
  double* a_mean = (double*)calloc((size_t)P * 3 + 1, sizeof(double));
  double* a_conic = (double*)calloc((size_t)P * 4 + 1, sizeof(double));
  double* a_opa = (double*)calloc((size_t)P + 1, sizeof(double));
  double* a_col = (double*)calloc((size_t)P * 3 + 1, sizeof(double));
  double* a_dep = (double*)calloc((size_t)P + 1, sizeof(double));
  double* a_sem = (double*)calloc((size_t)P * (S > 0 ? S : 1) + 1, sizeof(double));
  const double ddelx_dx = 0.5 * W, ddely_dy = 0.5 * H;
#pragma omp parallel for schedule(dynamic, 4)
  for (int py = 0; py < H; py++) {
    double* acc_s = (double*)calloc(S > 0 ? S : 1, sizeof(double));
    for (int px = 0; px < W; px++) {
      const size_t pix_id = (size_t)W * py + px;
      const float pixf[2] = {(float)px, (float)py};
      const uint32_t r0 = ranges[2 * ((py / BLOCK_Y) * gx + (px / BLOCK_X))];
      const uint32_t r1 = ranges[2 * ((py / BLOCK_Y) * gx + (px / BLOCK_X)) + 1];
      const uint32_t last_contributor = n_contrib[pix_id];
      /* exact final transmittance: product over the contributors, front to back */
      double T = 1.0;
      for (uint32_t k = r0; k < r1 && k - r0 < last_contributor; k++) {
        const uint32_t id = point_list[k];
        const float dx = means2D[2 * id] - pixf[0], dy = means2D[2 * id + 1] - pixf[1];
        const float* co = conic_opacity + 4 * id;
        const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
        if (power > 0.0f) continue;
        if (fminf(0.99f, co[3] * expf(power)) < 1.0f / 255.0f) continue;
        const double pw = -0.5 * ((double)co[0] * dx * dx + (double)co[2] * dy * dy) - (double)co[1] * dx * dy;
        T *= 1.0 - fmin(0.99, (double)co[3] * exp(pw));
      }
      const double T_final = T;
      double acc[3] = {0, 0, 0}, acc_d = 0, acc_a = 0, dL_dpixel[3];
      for (int i = 0; i < 3; i++) dL_dpixel[i] = dL_dpixels[i * HW + pix_id];
      const double dL_dpixel_depth = dL_dpixel_depths[pix_id], dL_dalpha = dL_dalphas[pix_id];
      double bg_dot = 0;
      for (int i = 0; i < 3; i++) bg_dot += (double)bg[i] * dL_dpixel[i];
      for (int ch = 0; ch < S; ch++) acc_s[ch] = 0;
      uint32_t contributor = r1 - r0;
      for (uint32_t k = r1; k-- > r0;) {
        contributor--;
        if (contributor >= last_contributor) continue;
        const uint32_t id = point_list[k];
        const float dxf = means2D[2 * id] - pixf[0], dyf = means2D[2 * id + 1] - pixf[1];
        const float* co = conic_opacity + 4 * id;
        const float power = -0.5f * (co[0] * dxf * dxf + co[2] * dyf * dyf) - co[1] * dxf * dyf;
        if (power > 0.0f) continue;
        if (fminf(0.99f, co[3] * expf(power)) < 1.0f / 255.0f) continue;
        const double dx = dxf, dy = dyf;
        const double G = exp(-0.5 * ((double)co[0] * dx * dx + (double)co[2] * dy * dy) - (double)co[1] * dx * dy);
        const int clamped = (double)co[3] * G > 0.99;
        const double alpha = clamped ? 0.99 : (double)co[3] * G;
        T = T / (1.0 - alpha);
        const double dch = alpha * T;
        /* acc_* = composite of everything BEHIND this splat as seen from here (the reference's accum_rec
         * after its update, :556-561), then this splat is blended into it */
        double dL_dopa = 0;
        for (int ch = 0; ch < 3; ch++) {
          const double c = colors[3 * id + ch];
          dL_dopa += (c - acc[ch]) * dL_dpixel[ch];
          _Pragma("omp atomic")
          a_col[3 * (size_t)id + ch] += dch * dL_dpixel[ch];
          acc[ch] = alpha * c + (1.0 - alpha) * acc[ch];
        }
        for (int ch = 0; ch < S; ch++) {
          const double sv = semantics[(size_t)id * S + ch], g = dL_dpixel_semantics[ch * HW + pix_id];
          dL_dopa += (sv - acc_s[ch]) * g;
          _Pragma("omp atomic")
          a_sem[(size_t)id * S + ch] += dch * g;
          acc_s[ch] = alpha * sv + (1.0 - alpha) * acc_s[ch];
        }
        const double c_d = depths[id];
        dL_dopa += (c_d - acc_d) * dL_dpixel_depth;
        _Pragma("omp atomic")
        a_dep[id] += dch * dL_dpixel_depth;
        acc_d = alpha * c_d + (1.0 - alpha) * acc_d;
        dL_dopa += (1.0 - acc_a) * dL_dalpha;
        acc_a = alpha + (1.0 - alpha) * acc_a;
        dL_dopa *= T;
        dL_dopa += (-T_final / (1.0 - alpha)) * bg_dot;
        /* the reference differentiates alpha = min(0.99, o G) as o G throughout (:618: no clamp case) */
        const double dL_dG = (double)co[3] * dL_dopa;
        const double gdx = G * dx, gdy = G * dy;
        const double dG_ddelx = -gdx * co[0] - gdy * co[1], dG_ddely = -gdy * co[2] - gdx * co[1];
        _Pragma("omp atomic")
        a_mean[3 * (size_t)id + 0] += dL_dG * dG_ddelx * ddelx_dx;
        _Pragma("omp atomic")
        a_mean[3 * (size_t)id + 1] += dL_dG * dG_ddely * ddely_dy;
        _Pragma("omp atomic")
        a_mean[3 * (size_t)id + 2] += fabs(dL_dG * dG_ddelx * ddelx_dx) + fabs(dL_dG * dG_ddely * ddely_dy);
        _Pragma("omp atomic")
        a_conic[4 * (size_t)id + 0] += -0.5 * gdx * dx * dL_dG;
        _Pragma("omp atomic")
        a_conic[4 * (size_t)id + 1] += -0.5 * gdx * dy * dL_dG;
        _Pragma("omp atomic")
        a_conic[4 * (size_t)id + 3] += -0.5 * gdy * dy * dL_dG;
        _Pragma("omp atomic")
        a_opa[id] += G * dL_dopa;
      }
    }
    free(acc_s);
  }
  for (size_t i = 0; i < (size_t)P * 3; i++) dL_dmean2D[i] = (float)a_mean[i];
  for (size_t i = 0; i < (size_t)P * 4; i++) dL_dconic[i] = (float)a_conic[i];
  for (size_t i = 0; i < (size_t)P; i++) dL_dopacity[i] = (float)a_opa[i];
  for (size_t i = 0; i < (size_t)P * 3; i++) dL_dcolors[i] = (float)a_col[i];
  for (size_t i = 0; i < (size_t)P; i++) dL_ddepths[i] = (float)a_dep[i];
  for (size_t i = 0; i < (size_t)P * S; i++) dL_dsemantics[i] = (float)a_sem[i];
  free(a_mean); free(a_conic); free(a_opa); free(a_col); free(a_dep); free(a_sem);
}

/* CR/auxiliary.h:107-117 (float3 overload) */
static void dnormvdv3(const float* v, const float* dv, float* o) {
  const float sum2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
  o[0] = ((+sum2 - v[0] * v[0]) * dv[0] - v[1] * v[0] * dv[1] - v[2] * v[0] * dv[2]) * invsum32;
  o[1] = (-v[0] * v[1] * dv[0] + (sum2 - v[1] * v[1]) * dv[1] - v[2] * v[1] * dv[2]) * invsum32;
  o[2] = (-v[0] * v[2] * dv[0] - v[1] * v[2] * dv[1] + (sum2 - v[2] * v[2]) * dv[2]) * invsum32;
}

/* CR/backward.cu:20-139: SH backward; dL_dmeans += ... */
static void computeColorFromSH_bwd(int idx, int deg, int max_coeffs, const float* means,
                                   const float* campos, const float* shs, const uint8_t* clamped,
                                   const float* dL_dcolor, float* dL_dmeans, float* dL_dshs) {
  const float* pos = means + 3 * idx;
  const float dir_orig[3] = {pos[0] - campos[0], pos[1] - campos[1], pos[2] - campos[2]};
  const float len = sqrtf(dir_orig[0] * dir_orig[0] + dir_orig[1] * dir_orig[1] + dir_orig[2] * dir_orig[2]);
  const float x = dir_orig[0] / len, y = dir_orig[1] / len, z = dir_orig[2] / len;
  const float* sh = shs + (size_t)idx * max_coeffs * 3;
  float* dL_dsh = dL_dshs + (size_t)idx * max_coeffs * 3;
  float dL_dRGB[3];
  for (int c = 0; c < 3; c++) dL_dRGB[c] = dL_dcolor[3 * idx + c] * (clamped[3 * idx + c] ? 0.f : 1.f);
  float dRGBdx[3] = {0, 0, 0}, dRGBdy[3] = {0, 0, 0}, dRGBdz[3] = {0, 0, 0};
#define SH(k, c) sh[3 * (k) + (c)]
#define DSH(k, v) for (int c = 0; c < 3; c++) dL_dsh[3 * (k) + c] = (v) * dL_dRGB[c]
  DSH(0, SH_C0);
  if (deg > 0) {
    const float dRGBdsh1 = -SH_C1 * y, dRGBdsh2 = SH_C1 * z, dRGBdsh3 = -SH_C1 * x;
    DSH(1, dRGBdsh1); DSH(2, dRGBdsh2); DSH(3, dRGBdsh3);
    for (int c = 0; c < 3; c++) {
      dRGBdx[c] = -SH_C1 * SH(3, c);
      dRGBdy[c] = -SH_C1 * SH(1, c);
      dRGBdz[c] = SH_C1 * SH(2, c);
    }
    if (deg > 1) {
      const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      DSH(4, SH_C2[0] * xy); DSH(5, SH_C2[1] * yz); DSH(6, SH_C2[2] * (2.f * zz - xx - yy));
      DSH(7, SH_C2[3] * xz); DSH(8, SH_C2[4] * (xx - yy));
      for (int c = 0; c < 3; c++) {
        dRGBdx[c] += SH_C2[0] * y * SH(4, c) + SH_C2[2] * 2.f * -x * SH(6, c) + SH_C2[3] * z * SH(7, c) + SH_C2[4] * 2.f * x * SH(8, c);
        dRGBdy[c] += SH_C2[0] * x * SH(4, c) + SH_C2[1] * z * SH(5, c) + SH_C2[2] * 2.f * -y * SH(6, c) + SH_C2[4] * 2.f * -y * SH(8, c);
        dRGBdz[c] += SH_C2[1] * y * SH(5, c) + SH_C2[2] * 2.f * 2.f * z * SH(6, c) + SH_C2[3] * x * SH(7, c);
      }
      if (deg > 2) {
        DSH(9, SH_C3[0] * y * (3.f * xx - yy)); DSH(10, SH_C3[1] * xy * z);
        DSH(11, SH_C3[2] * y * (4.f * zz - xx - yy));
        DSH(12, SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy));
        DSH(13, SH_C3[4] * x * (4.f * zz - xx - yy)); DSH(14, SH_C3[5] * z * (xx - yy));
        DSH(15, SH_C3[6] * x * (xx - 3.f * yy));
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
  const float dL_ddir[3] = {
      dRGBdx[0] * dL_dRGB[0] + dRGBdx[1] * dL_dRGB[1] + dRGBdx[2] * dL_dRGB[2],
      dRGBdy[0] * dL_dRGB[0] + dRGBdy[1] * dL_dRGB[1] + dRGBdy[2] * dL_dRGB[2],
      dRGBdz[0] * dL_dRGB[0] + dRGBdz[1] * dL_dRGB[1] + dRGBdz[2] * dL_dRGB[2]};
  float dL_dmean[3];
  dnormvdv3(dir_orig, dL_ddir, dL_dmean);
  for (int c = 0; c < 3; c++) dL_dmeans[3 * idx + c] += dL_dmean[c];
}

/* CR/backward.cu:278-341 */
static void computeCov3D_bwd(int idx, const float* scale, float mod, const float* rot,
                             const float* dL_dcov3Ds, float* dL_dscales, float* dL_drots) {
  const float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
  mat3 R, S = {{0}}, M, dL_dSigma, M2, dL_dM, Rt, dL_dMt;
  quat_to_R_glm(rot, R);
  const float s[3] = {mod * scale[0], mod * scale[1], mod * scale[2]};
  S[0][0] = s[0]; S[1][1] = s[1]; S[2][2] = s[2];
  m3_mul(S, R, M);
  const float* d = dL_dcov3Ds + 6 * idx;
  dL_dSigma[0][0] = d[0]; dL_dSigma[0][1] = 0.5f * d[1]; dL_dSigma[0][2] = 0.5f * d[2];
  dL_dSigma[1][0] = 0.5f * d[1]; dL_dSigma[1][1] = d[3]; dL_dSigma[1][2] = 0.5f * d[4];
  dL_dSigma[2][0] = 0.5f * d[2]; dL_dSigma[2][1] = 0.5f * d[4]; dL_dSigma[2][2] = d[5];
  for (int c = 0; c < 3; c++) for (int rr = 0; rr < 3; rr++) M2[c][rr] = 2.0f * M[c][rr];
  m3_mul(M2, dL_dSigma, dL_dM);
  m3_transpose(R, Rt);
  m3_transpose(dL_dM, dL_dMt);
  float* dL_dscale = dL_dscales + 3 * idx;
  for (int k = 0; k < 3; k++)
    dL_dscale[k] = Rt[k][0] * dL_dMt[k][0] + Rt[k][1] * dL_dMt[k][1] + Rt[k][2] * dL_dMt[k][2];
  for (int k = 0; k < 3; k++) for (int rr = 0; rr < 3; rr++) dL_dMt[k][rr] *= s[k];
  float* q = dL_drots + 4 * idx;
  q[0] = 2 * z * (dL_dMt[0][1] - dL_dMt[1][0]) + 2 * y * (dL_dMt[2][0] - dL_dMt[0][2]) + 2 * x * (dL_dMt[1][2] - dL_dMt[2][1]);
  q[1] = 2 * y * (dL_dMt[1][0] + dL_dMt[0][1]) + 2 * z * (dL_dMt[2][0] + dL_dMt[0][2]) + 2 * r * (dL_dMt[1][2] - dL_dMt[2][1]) - 4 * x * (dL_dMt[2][2] + dL_dMt[1][1]);
  q[2] = 2 * x * (dL_dMt[1][0] + dL_dMt[0][1]) + 2 * r * (dL_dMt[2][0] - dL_dMt[0][2]) + 2 * z * (dL_dMt[1][2] + dL_dMt[2][1]) - 4 * y * (dL_dMt[2][2] + dL_dMt[0][0]);
  q[3] = 2 * r * (dL_dMt[0][1] - dL_dMt[1][0]) + 2 * x * (dL_dMt[2][0] + dL_dMt[0][2]) + 2 * y * (dL_dMt[1][2] + dL_dMt[2][1]) - 4 * z * (dL_dMt[1][1] + dL_dMt[0][0]);
}

/* ------------------------------------------------------------------------
 * Stage 5: BACKWARD::preprocess = computeCov2DCUDA (CR/backward.cu:144-274)
 * then preprocessCUDA (:346-412).  cov3Ds = cov3D_precomp or stage-1 cov3Ds.
 * dL_dmean2D[P,3], dL_dconic[P,4], dL_dcolor[P,3], dL_ddepth[P] are inputs
 * (outputs of stage 4); dL_dmeans[P,3], dL_dcov[P,6], dL_dsh[P,M,3],
 * dL_dscale[P,3], dL_drot[P,4] must arrive zero-filled (as the binding
 * allocates them, rasterize_points.cu:166-176).
 * ---------------------------------------------------------------------- */
void gso_preprocess_backward(int P, int D, int M, const float* means3D, const int32_t* radii,
                             const float* shs, const uint8_t* clamped, const float* scales,
                             const float* rotations, float scale_modifier, const float* cov3Ds,
                             const float* view, const float* proj, int W, int H, float tan_fovx,
                             float tan_fovy, const float* campos, const float* dL_dmean2D,
                             const float* dL_dconics, float* dL_dmeans, const float* dL_dcolor,
                             const float* dL_ddepth, float* dL_dcov, float* dL_dsh,
                             float* dL_dscale, float* dL_drot) {
  const float h_y = H / (2.0f * tan_fovy);
  const float h_x = W / (2.0f * tan_fovx);
  for (int idx = 0; idx < P; idx++) {
    if (!(radii[idx] > 0)) continue;
    const float* mean = means3D + 3 * idx;
    /* ---- computeCov2DCUDA ---- */
    {
      const float dL_dconic[3] = {dL_dconics[4 * idx], dL_dconics[4 * idx + 1], dL_dconics[4 * idx + 3]};
      Cov2DCtx c;
      cov2d_common(mean, h_x, h_y, tan_fovx, tan_fovy, cov3Ds + 6 * idx, view, &c);
      const float x_grad_mul = (c.txtz < -c.limx || c.txtz > c.limx) ? 0.f : 1.f;
      const float y_grad_mul = (c.tytz < -c.limy || c.tytz > c.limy) ? 0.f : 1.f;
      const float a = c.cov[0][0] + 0.3f, b = c.cov[0][1], cc = c.cov[1][1] + 0.3f;
      const float denom = a * cc - b * b;
      float dL_da = 0, dL_db = 0, dL_dc = 0;
      const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
      float (*T)[3] = c.T;
      float (*Vrk)[3] = c.Vrk;
      float (*Wm)[3] = c.W;
      float* dc = dL_dcov + 6 * idx;
      if (denom2inv != 0) {
        dL_da = denom2inv * (-cc * cc * dL_dconic[0] + 2 * b * cc * dL_dconic[1] + (denom - a * cc) * dL_dconic[2]);
        dL_dc = denom2inv * (-a * a * dL_dconic[2] + 2 * a * b * dL_dconic[1] + (denom - a * cc) * dL_dconic[0]);
        dL_db = denom2inv * 2 * (b * cc * dL_dconic[0] - (denom + 2 * b * b) * dL_dconic[1] + a * b * dL_dconic[2]);
        dc[0] = (T[0][0] * T[0][0] * dL_da + T[0][0] * T[1][0] * dL_db + T[1][0] * T[1][0] * dL_dc);
        dc[3] = (T[0][1] * T[0][1] * dL_da + T[0][1] * T[1][1] * dL_db + T[1][1] * T[1][1] * dL_dc);
        dc[5] = (T[0][2] * T[0][2] * dL_da + T[0][2] * T[1][2] * dL_db + T[1][2] * T[1][2] * dL_dc);
        dc[1] = 2 * T[0][0] * T[0][1] * dL_da + (T[0][0] * T[1][1] + T[0][1] * T[1][0]) * dL_db + 2 * T[1][0] * T[1][1] * dL_dc;
        dc[2] = 2 * T[0][0] * T[0][2] * dL_da + (T[0][0] * T[1][2] + T[0][2] * T[1][0]) * dL_db + 2 * T[1][0] * T[1][2] * dL_dc;
        dc[4] = 2 * T[0][2] * T[0][1] * dL_da + (T[0][1] * T[1][2] + T[0][2] * T[1][1]) * dL_db + 2 * T[1][1] * T[1][2] * dL_dc;
      } else {
        for (int i = 0; i < 6; i++) dc[i] = 0;
      }
      const float dL_dT00 = 2 * (T[0][0] * Vrk[0][0] + T[0][1] * Vrk[0][1] + T[0][2] * Vrk[0][2]) * dL_da +
                            (T[1][0] * Vrk[0][0] + T[1][1] * Vrk[0][1] + T[1][2] * Vrk[0][2]) * dL_db;
      const float dL_dT01 = 2 * (T[0][0] * Vrk[1][0] + T[0][1] * Vrk[1][1] + T[0][2] * Vrk[1][2]) * dL_da +
                            (T[1][0] * Vrk[1][0] + T[1][1] * Vrk[1][1] + T[1][2] * Vrk[1][2]) * dL_db;
      const float dL_dT02 = 2 * (T[0][0] * Vrk[2][0] + T[0][1] * Vrk[2][1] + T[0][2] * Vrk[2][2]) * dL_da +
                            (T[1][0] * Vrk[2][0] + T[1][1] * Vrk[2][1] + T[1][2] * Vrk[2][2]) * dL_db;
      const float dL_dT10 = 2 * (T[1][0] * Vrk[0][0] + T[1][1] * Vrk[0][1] + T[1][2] * Vrk[0][2]) * dL_dc +
                            (T[0][0] * Vrk[0][0] + T[0][1] * Vrk[0][1] + T[0][2] * Vrk[0][2]) * dL_db;
      const float dL_dT11 = 2 * (T[1][0] * Vrk[1][0] + T[1][1] * Vrk[1][1] + T[1][2] * Vrk[1][2]) * dL_dc +
                            (T[0][0] * Vrk[1][0] + T[0][1] * Vrk[1][1] + T[0][2] * Vrk[1][2]) * dL_db;
      const float dL_dT12 = 2 * (T[1][0] * Vrk[2][0] + T[1][1] * Vrk[2][1] + T[1][2] * Vrk[2][2]) * dL_dc +
                            (T[0][0] * Vrk[2][0] + T[0][1] * Vrk[2][1] + T[0][2] * Vrk[2][2]) * dL_db;
      const float dL_dJ00 = Wm[0][0] * dL_dT00 + Wm[0][1] * dL_dT01 + Wm[0][2] * dL_dT02;
      const float dL_dJ02 = Wm[2][0] * dL_dT00 + Wm[2][1] * dL_dT01 + Wm[2][2] * dL_dT02;
      const float dL_dJ11 = Wm[1][0] * dL_dT10 + Wm[1][1] * dL_dT11 + Wm[1][2] * dL_dT12;
      const float dL_dJ12 = Wm[2][0] * dL_dT10 + Wm[2][1] * dL_dT11 + Wm[2][2] * dL_dT12;
      const float tz = 1.f / c.t[2];
      const float tz2 = tz * tz;
      const float tz3 = tz2 * tz;
      const float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
      const float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
      const float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * c.t[0]) * tz3 * dL_dJ02 + (2 * h_y * c.t[1]) * tz3 * dL_dJ12;
      /* transformVec4x3Transpose, CR/auxiliary.h:88-96; OVERWRITES (:273) */
      dL_dmeans[3 * idx + 0] = view[0] * dL_dtx + view[1] * dL_dty + view[2] * dL_dtz;
      dL_dmeans[3 * idx + 1] = view[4] * dL_dtx + view[5] * dL_dty + view[6] * dL_dtz;
      dL_dmeans[3 * idx + 2] = view[8] * dL_dtx + view[9] * dL_dty + view[10] * dL_dtz;
    }
    /* ---- preprocessCUDA (backward) ---- */
    {
      const float* m = mean;
      float m_hom[4];
      transformPoint4x4(m, proj, m_hom);
      const float m_w = 1.0f / (m_hom[3] + 0.0000001f);
      const float mul1 = (proj[0] * m[0] + proj[4] * m[1] + proj[8] * m[2] + proj[12]) * m_w * m_w;
      const float mul2 = (proj[1] * m[0] + proj[5] * m[1] + proj[9] * m[2] + proj[13]) * m_w * m_w;
      const float gx2 = dL_dmean2D[3 * idx], gy2 = dL_dmean2D[3 * idx + 1];
      float dL_dmean[3];
      dL_dmean[0] = (proj[0] * m_w - proj[3] * mul1) * gx2 + (proj[1] * m_w - proj[3] * mul2) * gy2;
      dL_dmean[1] = (proj[4] * m_w - proj[7] * mul1) * gx2 + (proj[5] * m_w - proj[7] * mul2) * gy2;
      dL_dmean[2] = (proj[8] * m_w - proj[11] * mul1) * gx2 + (proj[9] * m_w - proj[11] * mul2) * gy2;
      for (int c = 0; c < 3; c++) dL_dmeans[3 * idx + c] += dL_dmean[c];
      const float mul3 = view[2] * m[0] + view[6] * m[1] + view[10] * m[2] + view[14];
      float dL_dmean2[3];
      dL_dmean2[0] = (view[2] - view[3] * mul3) * dL_ddepth[idx];
      dL_dmean2[1] = (view[6] - view[7] * mul3) * dL_ddepth[idx];
      dL_dmean2[2] = (view[10] - view[11] * mul3) * dL_ddepth[idx];
      for (int c = 0; c < 3; c++) dL_dmeans[3 * idx + c] += dL_dmean2[c];
      if (shs)
        computeColorFromSH_bwd(idx, D, M, means3D, campos, shs, clamped, dL_dcolor, dL_dmeans, dL_dsh);
      if (scales)
        computeCov3D_bwd(idx, scales + 3 * idx, scale_modifier, rotations + 4 * idx, dL_dcov, dL_dscale, dL_drot);
    }
  }
}

/* CR/rasterizer_impl.cu:54-66 (checkFrustum -> mark_visible) */
void gso_mark_visible(int P, const float* means3D, const float* view, const float* proj,
                      uint8_t* present) {
  (void)proj;
  for (int idx = 0; idx < P; idx++) {
    float p_view[3];
    transformPoint4x3(means3D + 3 * idx, view, p_view);
    present[idx] = !(p_view[2] <= 0.2f);
  }
}

uint32_t gso_higher_msb(uint32_t n) { return getHigherMsb(n); }

/* threads the OpenMP build will use (1 for the scalar build) */
int gso_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
