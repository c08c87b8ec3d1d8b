// Scene-graph composition of one Gaussian (shared by preprocess.hip -- forward -- and
// preprocess_bwd.hip -- the training backward of the fused composition).  Include after
// gaussian_math.h, inside a translation unit compiled with -ffp-contract=off.
#pragma once
#include "common.h"

namespace grpg {

// ------------------------------------------------------------------------------------------
// Fused scene-graph composition.  compose_one() restates what the
// reference's properties compute in PyTorch for one Gaussian of one model:
//   get_scaling  = exp(_scaling)                         lib/models/gaussian_model.py:224-226
//   get_rotation = F.normalize(_rotation)                 :228-230  (x / max(|x|, 1e-12))
//   get_opacity  = sigmoid(_opacity)                      :248-250
//   actor means  = quaternion_to_matrix(obj_rots) x + obj_trans
//                                                        lib/models/street_gaussian_model.py:341-365,
//                                                        lib/utils/general_utils.py:125-146
//   actor rots   = F.normalize(quaternion_raw_multiply(obj_rots, get_rotation))
//                                                        street_gaussian_model.py:318-336,
//                                                        general_utils.py:220-238
//   actor colour = cat(sum_c _features_dc[:, c] * IDFT(time)[c], _features_rest)
//                                                        lib/models/gaussian_model_actor.py:73-82
// The training-time flip augmentation (street_gaussian_model.py:286-293; flip_prob applies in
// train mode only) is the optional per-Gaussian `flip` mask of a segment.  One rounding per operation, no contraction:
// grpg_compose and grpg_forward_composed see bit-identical values.
// ------------------------------------------------------------------------------------------
struct Activated {
  float mx, my, mz, s0, s1, s2, opacity;
  float4 q;
  float dc[3];
};

__device__ __forceinline__ const SegmentDev* find_segment(const SegmentDev* __restrict__ segs,
                                                          const int nseg, const uint32_t idx) {
  int lo = 0, hi = nseg - 1;   // last segment with start <= idx (segments are non-empty)
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (segs[mid].start <= idx) lo = mid; else hi = mid - 1;
  }
  return segs + lo;
}

__device__ __forceinline__ Activated compose_one(const SegmentDev& sg, const uint32_t j) {
  Activated a;
  // symmetry prior of rigid actors in training (street_gaussian_model.py:57-61,286-293,332,360): a
  // flipped Gaussian has its local y mirrored and its local rotation pre-multiplied by the
  // quaternion of diag(-1, 1, -1) = (0, 0, 1, 0) -- a signed permutation, exact in fp32
  const bool flip = sg.flip != nullptr && sg.flip[j] != 0;
  const float x = sg.xyz[3 * j], y = flip ? -sg.xyz[3 * j + 1] : sg.xyz[3 * j + 1], z = sg.xyz[3 * j + 2];
  a.s0 = expf(sg.scaling[3 * j]);
  a.s1 = expf(sg.scaling[3 * j + 1]);
  a.s2 = expf(sg.scaling[3 * j + 2]);
  a.opacity = 1.0f / (1.0f + expf(-sg.opacity[j]));
  const float4 rq = load_quat(sg.rotation, (int)j);
  const float rn = fmaxf(sqrtf(rq.x * rq.x + rq.y * rq.y + rq.z * rq.z + rq.w * rq.w), 1e-12f);
  const float4 qn_ = make_float4(rq.x / rn, rq.y / rn, rq.z / rn, rq.w / rn);
  // (0,0,1,0) (x) (w,x,y,z) = (-y, z, w, -x)
  const float4 ql = flip ? make_float4(-qn_.z, qn_.w, qn_.x, -qn_.y) : qn_;
  if (sg.rigid) {
    // quaternion_to_matrix normalises obj_rots first (general_utils.py:126-128)
    const float on = sqrtf(sg.rot[0] * sg.rot[0] + sg.rot[1] * sg.rot[1] + sg.rot[2] * sg.rot[2] +
                           sg.rot[3] * sg.rot[3]);
    const float r = sg.rot[0] / on, qx = sg.rot[1] / on, qy = sg.rot[2] / on, qz = sg.rot[3] / on;
    const float R00 = 1.f - 2.f * (qy * qy + qz * qz), R01 = 2.f * (qx * qy - r * qz), R02 = 2.f * (qx * qz + r * qy);
    const float R10 = 2.f * (qx * qy + r * qz), R11 = 1.f - 2.f * (qx * qx + qz * qz), R12 = 2.f * (qy * qz - r * qx);
    const float R20 = 2.f * (qx * qz - r * qy), R21 = 2.f * (qy * qz + r * qx), R22 = 1.f - 2.f * (qx * qx + qy * qy);
    a.mx = (R00 * x + R01 * y + R02 * z) + sg.trans[0];
    a.my = (R10 * x + R11 * y + R12 * z) + sg.trans[1];
    a.mz = (R20 * x + R21 * y + R22 * z) + sg.trans[2];
    // quaternion_raw_multiply(obj_rots, rotations_local): obj_rots as given (not normalised)
    const float aw = sg.rot[0], ax = sg.rot[1], ay = sg.rot[2], az = sg.rot[3];
    const float ow = aw * ql.x - ax * ql.y - ay * ql.z - az * ql.w;
    const float ox = aw * ql.y + ax * ql.x + ay * ql.w - az * ql.z;
    const float oy = aw * ql.z - ax * ql.w + ay * ql.x + az * ql.y;
    const float oz = aw * ql.w + ax * ql.z - ay * ql.y + az * ql.x;
    const float qn = fmaxf(sqrtf(ow * ow + ox * ox + oy * oy + oz * oz), 1e-12f);
    a.q = make_float4(ow / qn, ox / qn, oy / qn, oz / qn);
  } else {
    a.mx = x; a.my = y; a.mz = z;
    a.q = ql;
  }
  const int F = sg.fourier_dim;
#pragma unroll
  for (int ch = 0; ch < 3; ch++) {
    float acc = sg.fdc[(size_t)j * F * 3 + ch] * sg.idft[0];
    for (int c = 1; c < F; c++) acc = acc + sg.fdc[((size_t)j * F + c) * 3 + ch] * sg.idft[c];
    a.dc[ch] = acc;
  }
  return a;
}

// All M coefficients of Gaussian j of a segment into sh[3*M] (DC first, then _features_rest).
__device__ __forceinline__ void compose_features(const SegmentDev& sg, const uint32_t j,
                                                 const Activated& a, const int M, float* sh) {
  sh[0] = a.dc[0]; sh[1] = a.dc[1]; sh[2] = a.dc[2];
  const float* fr = sg.frest + (size_t)j * (M - 1) * 3;
  for (int k = 0; k < (M - 1) * 3; k++) sh[3 + k] = fr[k];
}


}  // namespace grpg
