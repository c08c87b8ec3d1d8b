// Per-tile front-to-back alpha blend for gfx950.
//
// Replaces renderCUDA<3> (cuda_rasterizer/forward.cu:340-467).  Same per-pixel semantics:
//   power = -1/2 (a dx^2 + c dy^2) - b dx dy ; skip if power > 0
//   alpha = min(0.99, opacity * exp(power))   ; skip if alpha < 1/255
//   stop  if T (1 - alpha) < 1e-4             ; else C += c alpha T, D += z alpha T, W += alpha T
//   out_color = C + T bg, out_alpha = W, out_depth = D, n_contrib = index of last blended splat.
//
// CDNA4 mapping (NOT the reference's 256-thread / one-pixel-per-thread / LDS-broadcast layout):
//   * A wave64 is the unit of work and never synchronises with another wave (no __syncthreads).
//     LIGHT tiles (short lists): one wave owns the whole 16x16 tile, 4 vertically adjacent pixels
//     per lane -- a splat's 48-byte record is broadcast from LDS once per wave whether a lane
//     shades 1 pixel or 4, so LDS traffic per pixel-pair drops 4x and the dx-only terms of the
//     quadratic are shared.  HEAVY tiles (>= heavy_min entries; a street scene's horizon tiles
//     hold up to ~35 k splats and would serialise the whole frame behind one wave): four 16x4
//     quarter-tile waves at 1 pixel per lane, fed through an LDS ring (blend_heavy below).
//   * A tile-classification pre-pass builds the work lists; one launch dispatches them longest
//     processing time first: the few tiles with >= 8192 entries (two workgroups each, a PRODUCER
//     and a CONSUMER wave per quarter, see pc_producer / pc_consumer), tiles with >= 2048
//     entries, the light tiles, the remaining heavy tiles.
//   * emit (binning.hip) stores a 4-bit mask in the top bits of every point-list entry: which
//     16x4 quarters of the tile the splat can reach at all (exact minimum of the conic's
//     quadratic over the quarter's rectangle, blend_math.h).  A record is only ever loaded for an
//     entry whose mask concerns the wave.  The reference bins by the 3-sigma bounding SQUARE, so
//     about 2/3 of the (quarter, entry) pairs are dead.
//   * Surviving records are culled once more against the bounding box of the wave's still-LIVE
//     pixels (shrinks as pixels saturate), compacted with ballot + popcount into the wave's
//     private LDS slice, and only those are evaluated, G at a time (independent alpha chains for
//     ILP, then the in-order blend).  Decisions per pixel are unchanged: both culls only remove
//     splats that every pixel of the rectangle would reject.
//   * The blend part of a group is skipped wave-uniformly when no lane accepted a splat;
//     wave-uniform early exit via ballot when every pixel of the wave is saturated.
// The kernel is VALU-issue-bound (about 25 flop-equivalents per surviving pixel-splat pair), not
// HBM-bound; its compulsory HBM traffic is 4 B (id) + 40 B (record fields) per tile instance
// + 20 B per pixel.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "blend_math.h"
#include "common.h"
#include "sky_math.h"

#ifndef GRPG_LAYERS_ABLATE
#define GRPG_LAYERS_ABLATE 0
#endif

namespace grpg {

constexpr int RW_WAVES = 4;

// ------------------------------------------------------------------------------------------
// Tile classification into work lists (order inside a list is irrelevant to results; thresholds:
// common.h TileClasses):
//   class 0: producer / consumer wave pairs   class 1, class 2: four quarter waves
//   class 3: light (one wave per tile)
// Workgroups are handed out class 0 first, so the longest lists start first (LPT order).
// counts[4] pre-zeroed by the launcher; lists[c] = work + 4 + c*T.
// ------------------------------------------------------------------------------------------
constexpr int NUM_CLASSES = 4;

__global__ void __launch_bounds__(256)
classify_tiles_kernel(const uint32_t T, const uint2* __restrict__ ranges, const TileClasses tc,
                      uint32_t* __restrict__ work) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63;
  int cls = -1;
  if (t < T) {
    const uint2 r = ranges[t];
    const uint32_t len = r.y - r.x;
    cls = tile_class_of(t, len, tc);
  }
  // wave-aggregated slot allocation: one atomic per (wave, class) instead of one per tile
  // (9600 same-address atomics serialise at ~11 ns each = 100 us, measured)
#pragma unroll
  for (int c = 0; c < NUM_CLASSES; c++) {
    const uint64_t m = __ballot(cls == c);
    if (m == 0ull) continue;
    uint32_t base = 0;
    if (lane == (uint32_t)(__ffsll((unsigned long long)m) - 1)) base = atomicAdd(&work[c], (uint32_t)__popcll(m));
    base = __shfl(base, __ffsll((unsigned long long)m) - 1, 64);
    if (cls == c) work[NUM_CLASSES + (size_t)c * T + base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = t;
  }
}

// Per-wave blend of one pixel rectangle: 16 columns x (4*PX... see below) rows.
//   PX = 4: rows y0 + (lane>>4)*4 + k, k<4   -> 16x16 pixels (a whole tile)
//   PX = 1: rows y0 + (lane>>4)               -> 16x4 pixels (a quarter tile)
// GPI splats are evaluated per inner iteration (independent alpha chains), then blended in order.
// Per-lane predicates are carried as wave-level 64-bit lane masks (uint64_t in SGPRs): compare
// results are masks already (ballot is free), all boolean algebra runs on the scalar unit, and
// inverse_ballot hands a mask back to v_cndmask at no VALU cost.
__device__ __forceinline__ uint64_t lanes(const bool pred) { return __builtin_amdgcn_ballot_w64(pred); }
__device__ __forceinline__ bool in_mask(const uint64_t m) { return __builtin_amdgcn_inverse_ballot_w64(m); }

// Bounding box of the pixels whose bit is set in a wave-level lane mask (lane l = column l & 15, row
// l >> 4 of the wave's 16 x 4 lane grid), from the mask's BITS: a dozen scalar instructions.  Until round 5
// the box was a 6-step butterfly over four floats per lane (24 ds_bpermute round trips + 24 min / max, every
// time a pixel saturated: ~900 cycles on a wave that runs alone, i.e. on every chain of the launch).
struct MaskBox { int c0, c1, r0, r1; };   // columns / rows of the lane grid, inclusive; mask != 0
__device__ __forceinline__ MaskBox mask_box(const uint64_t m) {
  MaskBox b;
  b.r0 = (int)(__builtin_ctzll(m) >> 4);
  b.r1 = (int)((63 - __builtin_clzll(m)) >> 4);
  uint32_t cols = (uint32_t)m | (uint32_t)(m >> 32);
  cols = (cols | (cols >> 16)) & 0xFFFFu;
  b.c0 = (int)__builtin_ctz(cols);
  b.c1 = 31 - (int)__builtin_clz(cols);
  return b;
}

template <int PX>
struct WavePix {   // per-lane blending state of PX pixels
  float T[PX];   // out_alpha = sum of alpha_i T_i = 1 - T (telescoping): no separate accumulator
  v2f CrCg[PX], CbD[PX];   // (red, green) and (blue, depth) accumulators: one v_pk_fma_f32 each
  uint32_t last[PX];
  uint64_t done[PX];       // lane masks: pixel k of the lane is saturated / outside the image
};

// N-channel "semantic" planes (forward.cu:442-444: out_semantic[ch] += semantic[ch] alpha T).
//
// Per quad of splats this is a small matrix product, Out[64 pixels][16 channels] += W[64][4] Sem[4][16],
// with W the blend weights the colour path computes anyway -- the one contraction on this path, and the
// matrix pipe idles in this kernel.  Round 4 ran it on the vector ALU (16 fma per splat and wave, the row
// fetched through the scalar unit): S = 15 made the render 3.6 x as long.  Round 5: v_mfma_f32_16x16x4_f32
// (fp32 in, fp32 accumulate) per pixel ROW of the quarter:
//   A (16 pixels x 4 splats)  lane 16 k + m must hold w_k of pixel (column m, row y); the wave holds w_0..3 of
//                             pixel (m, y) in lane 16 y + m -- a 4 x 4 transpose between lane rows and
//                             registers: two v_permlane32_swap + two v_permlane16_swap give all four A's;
//   B (4 splats x 16 channels) lane 16 k + n reads Sem[splat k][channel n] from the batch's rows, which the
//                             compaction step staged in LDS in survivor order (one ds_read_b32 per quad);
//   D  four accumulators of 4 registers: pixel columns 4 (l / 16) + i of row y, channel l % 16.
// Vector-ALU cost per quad: 4 swaps + 4 MFMA issues instead of 64 fma + 4 scalar row fetches.  Products are
// exact (w x sem in fp32), the accumulation order inside an MFMA differs from the sequential fma chain by
// rounding only (1e-7 relative; the bar is 1e-4).  A lane that rejects a splat contributes w = 0: semantic
// values must be FINITE (0 x Inf = NaN; the reference skips rejected splats) -- rows of pad slots are zeros.
typedef float v4f __attribute__((ext_vector_type(4)));
template <int NSEM>
struct SemAcc {
  static_assert(NSEM == 0 || NSEM == 16, "the MFMA tile is 16 channels wide");
  v4f acc[NSEM > 0 ? 4 : 1];
  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int y = 0; y < (NSEM > 0 ? 4 : 1); y++) acc[y] = (v4f){0.f, 0.f, 0.f, 0.f};
  }
};
// semantics [P][S] in HBM; rows: the current batch's staged rows in LDS, [slot][16] floats (slots = the
// batch's compacted survivors incl. pads, same numbering as the pair blocks)
struct SemSrc { const float* semantics; int S; const float* rows; };
constexpr int SEM_ROW = 16;

// stages the row of Gaussian `id` (zeros when !valid: a pad) as slot `slot` of the batch
__device__ __forceinline__ void sem_stage(float* __restrict__ rows, const int slot, const float* __restrict__ semantics,
                                          const int S, const uint32_t id, const bool valid) {
  const float* __restrict__ row = semantics + (size_t)id * (size_t)S;
  float v[SEM_ROW];
#pragma unroll
  for (int c = 0; c < SEM_ROW; c++) v[c] = (valid && c < S) ? row[c < S ? c : 0] : 0.f;   // never past the row
  float4* dst = reinterpret_cast<float4*>(rows + slot * SEM_ROW);
#pragma unroll
  for (int c = 0; c < SEM_ROW; c += 4) dst[c >> 2] = make_float4(v[c], v[c + 1], v[c + 2], v[c + 3]);
}

// the quad whose first slot is j0: w[i] = blend weight of slot j0 + i for this lane's pixel
template <int NSEM>
__device__ __forceinline__ void sem_quad(SemAcc<NSEM>& sa, const float (&w)[4], const float* __restrict__ rows,
                                         const int j0, const int lane) {
  if constexpr (NSEM > 0) {
  const float b = rows[j0 * SEM_ROW + lane];
  const auto s02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(w[0]), __float_as_uint(w[2]), false, false);
  const auto s13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(w[1]), __float_as_uint(w[3]), false, false);
  const auto t01 = __builtin_amdgcn_permlane16_swap(s02[0], s13[0], false, false);
  const auto t23 = __builtin_amdgcn_permlane16_swap(s02[1], s13[1], false, false);
  sa.acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(t01[0]), b, sa.acc[0], 0, 0, 0);
  sa.acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(t01[1]), b, sa.acc[1], 0, 0, 0);
  sa.acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(t23[0]), b, sa.acc[2], 0, 0, 0);
  sa.acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(t23[1]), b, sa.acc[3], 0, 0, 0);
  }
}

// In-order blend of one splat into pixel k (forward.cu:425-440); ok_m = lanes that accept it.
template <int PX, bool AUX = true>
__device__ __forceinline__ float blend_one(WavePix<PX>& s, const int k, const uint64_t ok_m,
                                           const float alpha, const float4 col, const uint32_t pos) {
  const v2f tw = (v2f){1.0f - alpha, alpha} * (v2f){s.T[k], s.T[k]};   // T (1 - alpha), alpha T
  const uint64_t lt_m = lanes(tw.x < 0.0001f);
  const uint64_t live = ok_m & ~s.done[k];
  s.done[k] |= live & lt_m;                       // terminated: this and all later splats rejected
  const bool cont = in_mask(live & ~lt_m);
  const float w = cont ? tw.y : 0.0f;
  if (PX == 1) {
    const v2f w2 = {w, w};
    s.CrCg[k] = __builtin_elementwise_fma((v2f){col.x, col.y}, w2, s.CrCg[k]);
    s.CbD[k] = __builtin_elementwise_fma((v2f){col.z, col.w}, w2, s.CbD[k]);
  } else {   // 4 pixels per lane: leave the pairing to the register allocator
    s.CrCg[k].x = fmaf(col.x, w, s.CrCg[k].x);
    s.CrCg[k].y = fmaf(col.y, w, s.CrCg[k].y);
    s.CbD[k].x = fmaf(col.z, w, s.CbD[k].x);
    s.CbD[k].y = fmaf(col.w, w, s.CbD[k].y);
  }
  s.T[k] = cont ? tw.x : s.T[k];
  if (AUX) s.last[k] = cont ? pos : s.last[k];   // n_contrib: only a later backward needs it
  return w;                                       // the blend weight (0 where the lane does not take the splat)
}

// LDS slot of a compacted survivor, light path (REC_F4 = 3 float4):
//   [0] gx, gy, opacity, list position   [1] A, B, C (pre-scaled conic, blend_math.h), -
//   [2] r, g, b, depth
__device__ __forceinline__ void store_slot(float4* __restrict__ my, const int slot, const float4 a,
                                           const float4 b, const float4 c, const uint32_t pos) {
  const SplatQ q = splat_q(b.x, b.y, b.z);
  my[slot * REC_F4 + 0] = make_float4(a.x, a.y, a.w, __uint_as_float(pos));
  my[slot * REC_F4 + 1] = make_float4(q.A, q.B, q.C, 0.f);
  my[slot * REC_F4 + 2] = make_float4(b.w, c.x, c.y, a.z);
}

// Evaluate G consecutive compacted survivors (LDS slots j0 .. j0+G-1, all present) for the lane's
// PX pixels: G*PX independent power/exp/alpha chains, then the in-order blend.  Returns true if
// the blend part ran (some lane accepted some splat).
template <int PX, int G, bool AUX = true, bool SAFE = false>
__device__ __forceinline__ bool blend_group(WavePix<PX>& s, const float4* __restrict__ my,
                                            const int j0, const float pxf, const int py0) {
  float4 ra[G], rq[G], rc[G];
#pragma unroll
  for (int g = 0; g < G; g++) {
    ra[g] = my[(j0 + g) * REC_F4 + 0];
    rq[g] = my[(j0 + g) * REC_F4 + 1];
    rc[g] = my[(j0 + g) * REC_F4 + 2];
  }
  float alpha[G][PX];
  uint64_t ok[G][PX];
  uint64_t any = 0ull;
#pragma unroll
  for (int g = 0; g < G; g++) {
    const SplatQ q = {rq[g].x, rq[g].y, rq[g].z};
    const SplatTerms st = splat_terms_q(ra[g].x - pxf, q);
#pragma unroll
    for (int k = 0; k < PX; k++) {
      const float dy = ra[g].y - (float)(py0 + k);
      const float p = pair_power(st, dy);
      alpha[g][k] = fminf(ALPHA_MAX, ra[g].z * __builtin_amdgcn_exp2f(p));   // == pair_alpha
      // SAFE: every survivor of the batch has splat_power_never_positive (blend_math.h): p > 0 cannot happen
      ok[g][k] = SAFE ? lanes(!(alpha[g][k] < ALPHA_MIN)) : (lanes(!(p > 0.0f)) & lanes(!(alpha[g][k] < ALPHA_MIN)));
      any |= ok[g][k] & ~s.done[k];
    }
  }
  if (any == 0ull) return false;   // nobody in the wave blends any of these splats
  // Transmittance chains first (alpha = 0 where a lane rejects a splat): if no live pixel's T falls
  // below 1e-4 by the END of the group, none terminates inside it (T only decreases) and the
  // per-splat termination tests are not needed -- see blend_quad.  Same products, same order.
  float w[G][PX], Tn[PX];
  uint64_t bad = 0ull;
#pragma unroll
  for (int k = 0; k < PX; k++) {
    const uint64_t live = ~s.done[k];
    float T = s.T[k];
#pragma unroll
    for (int g = 0; g < G; g++) {
      const float am = in_mask(ok[g][k] & live) ? alpha[g][k] : 0.0f;
      w[g][k] = am * T;
      T = (1.0f - am) * T;
    }
    Tn[k] = T;
    bad |= lanes(T < 0.0001f) & live;
  }
  if (bad == 0ull) {
#pragma unroll
    for (int g = 0; g < G; g++) {
#pragma unroll
      for (int k = 0; k < PX; k++) {
        s.CrCg[k].x = fmaf(rc[g].x, w[g][k], s.CrCg[k].x);
        s.CrCg[k].y = fmaf(rc[g].y, w[g][k], s.CrCg[k].y);
        s.CbD[k].x = fmaf(rc[g].z, w[g][k], s.CbD[k].x);
        s.CbD[k].y = fmaf(rc[g].w, w[g][k], s.CbD[k].y);
        if (AUX) s.last[k] = in_mask(ok[g][k] & ~s.done[k]) ? __float_as_uint(ra[g].w) : s.last[k];
      }
    }
#pragma unroll
    for (int k = 0; k < PX; k++) s.T[k] = Tn[k];
    return true;
  }
#pragma unroll
  for (int g = 0; g < G; g++) {   // some pixel terminates inside this group: exact per-splat blend
#pragma unroll
    for (int k = 0; k < PX; k++)
      (void)blend_one<PX, AUX>(s, k, ok[g][k], alpha[g][k], rc[g], __float_as_uint(ra[g].w));
  }
  return true;
}

// Heavy path (1 pixel per lane): survivors are stored in PAIRS so that two splats are evaluated
// per packed instruction.  LDS block of a pair (PAIR_F4 = 6 float4, same 48 B per splat):
//   [0] gx0 gx1 gy0 gy1   [1] A0 A1 B0 B1   [2] C0 C1 op0 op1
//   [3] r0 g0 b0 depth0   [4] r1 g1 b1 depth1   [5] pos0 pos1 - -
constexpr int PAIR_F4 = 2 * REC_F4;

__device__ __forceinline__ void store_pair_half(float4* __restrict__ my, const int slot,
                                                const float gx, const float gy, const SplatQ q,
                                                const float op, const float4 col, const uint32_t pos) {
  float4* blk = my + (slot >> 1) * PAIR_F4;
  float* f = reinterpret_cast<float*>(blk) + (slot & 1);
  f[0] = gx; f[2] = gy; f[4] = q.A; f[6] = q.B; f[8] = q.C; f[10] = op;
  blk[3 + (slot & 1)] = col;
  f[20] = __uint_as_float(pos);
}

// The accept half of a quad: alpha and the lanes that take the splat, for the four slots of blk
// (two pair blocks).  One body for every caller, so that they all get the same bits.
struct QuadGeom {   // centres, pre-scaled conics and opacities of the four slots of a quad (two pair blocks)
  float4 f[6];
  __device__ __forceinline__ void load(const float4* __restrict__ blk) {
    f[0] = blk[0]; f[1] = blk[1]; f[2] = blk[2];
    f[3] = blk[PAIR_F4 + 0]; f[4] = blk[PAIR_F4 + 1]; f[5] = blk[PAIR_F4 + 2];
  }
};
// SAFE: every slot of the quad has splat_power_never_positive (blend_math.h) or is a neutral pad, so the
// `power > 0` compares -- which could never reject anything -- are not executed (4 of the quad's 32 vector
// instructions, at the half rate of v_cmp).
template <bool SAFE = false>
__device__ __forceinline__ void eval_quad(const QuadGeom& g, const float pxf, const float pyf,
                                          float (&alpha)[4], uint64_t (&ok)[4]) {
  const v2f px2 = {pxf, pxf}, py2 = {pyf, pyf};
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const float4 f0 = g.f[3 * h + 0], f1 = g.f[3 * h + 1], f2 = g.f[3 * h + 2];
    const v2f dx = (v2f){f0.x, f0.y} - px2, dy = (v2f){f0.z, f0.w} - py2;
    const v2f p = pair_power_x2(dx, dy, (v2f){f1.x, f1.y}, (v2f){f1.z, f1.w}, (v2f){f2.x, f2.y});
    const v2f G = {__builtin_amdgcn_exp2f(p.x), __builtin_amdgcn_exp2f(p.y)};
    const v2f al = (v2f){f2.z, f2.w} * G;
    alpha[2 * h + 0] = fminf(ALPHA_MAX, al.x);
    alpha[2 * h + 1] = fminf(ALPHA_MAX, al.y);
    if (SAFE) {
      ok[2 * h + 0] = lanes(!(alpha[2 * h + 0] < ALPHA_MIN));
      ok[2 * h + 1] = lanes(!(alpha[2 * h + 1] < ALPHA_MIN));
    } else {
      ok[2 * h + 0] = lanes(!(p.x > 0.0f)) & lanes(!(alpha[2 * h + 0] < ALPHA_MIN));
      ok[2 * h + 1] = lanes(!(p.y > 0.0f)) & lanes(!(alpha[2 * h + 1] < ALPHA_MIN));
    }
  }
}

// Where the blend half of a quad takes the colour / position blocks of its two pairs from: the LDS pair
// blocks themselves, read when they are needed (heavy path), or registers filled a quad ahead (the
// pipeline's blender).  Same values either way.
struct QuadColsLds {
  const float4* __restrict__ blk;
  __device__ __forceinline__ float4 c0(const int h) const { return blk[h * PAIR_F4 + 3]; }
  __device__ __forceinline__ float4 c1(const int h) const { return blk[h * PAIR_F4 + 4]; }
  __device__ __forceinline__ float4 pp(const int h) const { return blk[h * PAIR_F4 + 5]; }
};
template <bool WITH_PP>
struct QuadColsReg {
  float4 c[4], p[2];
  __device__ __forceinline__ void load(const float4* __restrict__ blk) {
    c[0] = blk[3]; c[1] = blk[4]; c[2] = blk[PAIR_F4 + 3]; c[3] = blk[PAIR_F4 + 4];
    if (WITH_PP) { p[0] = blk[5]; p[1] = blk[PAIR_F4 + 5]; }
  }
  __device__ __forceinline__ float4 c0(const int h) const { return c[2 * h]; }
  __device__ __forceinline__ float4 c1(const int h) const { return c[2 * h + 1]; }
  __device__ __forceinline__ float4 pp(const int h) const { return p[h]; }
};

// The blend half of a quad, given alpha[i] and ok[i] of its four slots.
template <bool AUX, int NSEM, class Cols>
__device__ __forceinline__ bool blend_quad_tail(WavePix<1>& s, const Cols& cols,
                                                const float (&alpha)[4], const uint64_t (&ok)[4],
                                                SemAcc<NSEM>* sa, const SemSrc sem, const int j0, const int lane);

// Four consecutive slots j0..j0+3 (j0 % 4 == 0; absent slots are neutral pads: opacity 0).
template <bool AUX = true, int NSEM = 0, bool SAFE = false>
__device__ __forceinline__ bool blend_quad(WavePix<1>& s, const float4* __restrict__ my,
                                           const int j0, const float pxf, const float pyf,
                                           SemAcc<NSEM>* sa = nullptr, const SemSrc sem = SemSrc{nullptr, 0, nullptr},
                                           const int lane = 0) {
  const float4* blk = my + (j0 >> 1) * PAIR_F4;
  float alpha[4];
  uint64_t ok[4];
  QuadGeom g;
  g.load(blk);
  eval_quad<SAFE>(g, pxf, pyf, alpha, ok);
  return blend_quad_tail<AUX, NSEM>(s, QuadColsLds{blk}, alpha, ok, sa, sem, j0, lane);
}

template <bool AUX, int NSEM, class Cols>
__device__ __forceinline__ bool blend_quad_tail(WavePix<1>& s, const Cols& cols,
                                                const float (&alpha)[4], const uint64_t (&ok)[4],
                                                SemAcc<NSEM>* sa, const SemSrc sem, const int j0, const int lane) {
  const uint64_t live = ~s.done[0];
  if (((ok[0] | ok[1] | ok[2] | ok[3]) & live) == 0ull) return false;
  // The transmittance chain of the four splats first, with alpha = 0 in the lanes that reject a
  // splat.  T only ever decreases, so if the LAST T is still >= 1e-4 in every live lane, no pixel
  // terminates inside this quad and the per-splat termination tests (a v_cmp and two v_cndmask
  // each: half-rate instructions, tools/ubench/valu_rate.hip) are not needed: the colours are
  // accumulated with the weights just computed -- the same products in the same order, hence the
  // same bits (a rejecting lane adds c * 0 and keeps T * 1).  A pixel terminates at most once, so the
  // exact per-splat path below runs for at most 64 quads of a wave.
  float w[4];
  float T = s.T[0];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const float am = in_mask(ok[i] & live) ? alpha[i] : 0.0f;
    const v2f tw = (v2f){1.0f - am, am} * (v2f){T, T};
    w[i] = tw.y;
    T = tw.x;
  }
  if ((lanes(T < 0.0001f) & live) == 0ull) {
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const float4 c0 = cols.c0(h), c1 = cols.c1(h);
      const v2f w0 = {w[2 * h], w[2 * h]}, w1 = {w[2 * h + 1], w[2 * h + 1]};
      s.CrCg[0] = __builtin_elementwise_fma((v2f){c0.x, c0.y}, w0, s.CrCg[0]);
      s.CbD[0] = __builtin_elementwise_fma((v2f){c0.z, c0.w}, w0, s.CbD[0]);
      s.CrCg[0] = __builtin_elementwise_fma((v2f){c1.x, c1.y}, w1, s.CrCg[0]);
      s.CbD[0] = __builtin_elementwise_fma((v2f){c1.z, c1.w}, w1, s.CbD[0]);
      if (AUX) {   // n_contrib: position of the last splat the pixel took
        const float4 pp = cols.pp(h);
        s.last[0] = in_mask(ok[2 * h] & live) ? __float_as_uint(pp.x) : s.last[0];
        s.last[0] = in_mask(ok[2 * h + 1] & live) ? __float_as_uint(pp.y) : s.last[0];
      }
    }
    s.T[0] = T;
    if (NSEM > 0) sem_quad<NSEM>(*sa, w, sem.rows, j0, lane);   // the same weights feed the semantic planes
    return true;
  }
#pragma unroll
  for (int h = 0; h < 2; h++) {   // some pixel terminates in this quad: the exact per-splat blend
    const float4 c0 = cols.c0(h), c1 = cols.c1(h);
    const float4 pp = AUX ? cols.pp(h) : make_float4(0.f, 0.f, 0.f, 0.f);
    w[2 * h + 0] = blend_one<1, AUX>(s, 0, ok[2 * h + 0], alpha[2 * h + 0], c0, __float_as_uint(pp.x));
    w[2 * h + 1] = blend_one<1, AUX>(s, 0, ok[2 * h + 1], alpha[2 * h + 1], c1, __float_as_uint(pp.y));
  }
  if (NSEM > 0) sem_quad<NSEM>(*sa, w, sem.rows, j0, lane);
  return true;
}

// ------------------------------------------------------------------------------------------
// Where a wave's final per-pixel state goes.  One body per frame kind, so that every path of the launch (light
// tiles, quarter waves, the pairs' consumers) writes the same planes:
//   PlainOut   the op's planes: colour on the background, depth, alpha (= 1 - T), n_contrib (AUX)
//   LayersOut  a LAYERED frame (grpg_forward_layers, below): the composition's planes plus the two layer pairs.
//              pixel(): a tile WITHOUT object entries was walked with ONE state -- the background layer's chain
//              saw the very same splats (same bits), the object layer saw none (T = 1, C = 0);
//              pixel3(): a tile with object entries, three states.
// ------------------------------------------------------------------------------------------
// What the reference's callers do with the composition's colour behind the op, inside the render's own epilogue
// (grpg_forward_frame; a frame without it: FrameEpi::on == 0 and none of this is compiled in, EPI = false):
//   rgb = clamp(C + T bg)                           render_kernel's evaluation clamp (street_gaussian_renderer.py:236)
//   rgb = clamp(rgb + clamp(sky) (1 - acc))         StreetGaussianRenderer.render (:106-116), acc = 1 - T
//   bytes [H,W,3] = uint8(rgb * 255)                the simulator's frame (simulator.py:313-314)
// The state is in registers when the walk ends: the stand-alone chain (three launches) wrote the colour and
// alpha planes, read them back for the sky composite, wrote the colour again and read it a third time to pack.
struct FrameEpi {
  SkyArgs sky;              // sky.cube == NULL: no sky composite
  int clamp;                // evaluation mode: both clamps
  unsigned char* rgb8;      // [H,W,3] interleaved bytes, or NULL
  float bias;               // 0 = truncate like astype(np.uint8), 0.5 = round to nearest
  int planes;               // the float planes are written too (colour = the FINAL rgb, depth, alpha)
  int host;                 // how rgb8 is stored: 0 plain; 1 rgb8 IS device-mapped pinned host memory
                            // (grpg_frame_epilogue.out_rgb8_on_host), system-scope write-through; 2 rgb8 is the
                            // device STAGING frame of a drained frame (below), agent-scope write-through
  // Drained frame (host destination, W % 16 == 0): the blending waves never touch the link.  They store their bytes
  // into a device staging frame and count their pixels into the counter of their UNIT (64 x 16 pixels = four tiles
  // side by side = 16 rows x three whole 64-byte lines when W % 64 == 0; otherwise rows of 16-byte pieces, and the last
  // unit of a row is narrower); drain_wgs workgroups at the front of the grid watch the counters and carry every
  // complete unit to host8 while the launch is still blending.
  unsigned char* host8;     // the pinned host frame (drained frame), or NULL
  uint32_t* unit_cnt;       // [drain waves][64] pixels arrived per unit (zero at launch, zero again at its end)
  int drain_wgs, units_x;
};
// final rgb of a pixel from its blended colour (background included) and T; optional byte store
template <bool EPI>
__device__ __forceinline__ void frame_finish(const FrameEpi& e, const int px, const int py, const int W, const int H,
                                             const float T, float (&rgb)[3]) {
  if constexpr (EPI) {
    const size_t HW = (size_t)H * W, pix = (size_t)py * W + px;
    if (e.clamp) { rgb[0] = clamp01(rgb[0]); rgb[1] = clamp01(rgb[1]); rgb[2] = clamp01(rgb[2]); }
    if (e.sky.cube != nullptr) {
      const float acc = 1.0f - T, tr = 1.0f - acc;     // (the stand-alone launch reads acc from the alpha plane)
      float sky[3];
      sky_pixel(e.sky, px, py, pix, HW, true, tr, sky);
      rgb[0] = sky_over(rgb[0], sky[0], tr, e.clamp);
      rgb[1] = sky_over(rgb[1], sky[1], tr, e.clamp);
      rgb[2] = sky_over(rgb[2], sky[2], tr, e.clamp);
    }
    if (e.rgb8 != nullptr) {
      const uint32_t r8 = colour_u8(rgb[0], e.bias), g8 = colour_u8(rgb[1], e.bias), b8 = colour_u8(rgb[2], e.bias);
      if (((W & 3) | (int)((uintptr_t)e.rgb8 & 3)) == 0) {
        // Four neighbouring lanes hold four neighbouring pixels of a row (px = x0 + (lane & 15) on every path, x0 a
        // multiple of 16; W % 4 == 0: the quad is inside the image or outside as a whole): 12 bytes = three aligned
        // dwords, assembled in lanes 0-2 of the quad from the lane's own pixel and its right neighbour's.  One store
        // instead of three, and -- what the host destination needs -- whole dwords: a write-through BYTE store
        // crosses the link on its own (tools/ubench/host_store.hip: 176 ms per frame, against 0.21 for dwords).
        const uint32_t pk = r8 | (g8 << 8) | (b8 << 16);
        const uint32_t nx = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pk, 0xF9 /* quad_perm:[1,2,3,3] */, 0xF, 0xF, false);
        const int j = px & 3;
        const uint32_t v = (pk >> (8 * j)) | (nx << (24 - 8 * j));
        if (j < 3) {
          uint32_t* d = (uint32_t*)(e.rgb8 + 3 * (pix - (size_t)j) + 4 * j);
          // host destination: system-scope store (sc0 sc1), the line leaves the L2 now instead of at the end-of-kernel
          // write-back -- the frame crosses the link while the launch is still blending its long tiles
          if (e.host == 1) __hip_atomic_store(d, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          else if (e.host == 2) __hip_atomic_store(d, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // staging: sc1
          else *d = v;
        }
      } else {
        unsigned char* d = e.rgb8 + 3 * pix;
        d[0] = (unsigned char)r8; d[1] = (unsigned char)g8; d[2] = (unsigned char)b8;
      }
    }
  }
}

// Drained frame: a wave has stored all its pixels of one tile (npix of them inside the image; the call is
// wave-uniform, after the wave's last pixel()).  Its write-through stores are complete once vmcnt reaches zero
// (gfx9: stores count there too) -- only then does the unit's counter learn of them.
template <bool EPI>
__device__ __forceinline__ void frame_arrive(const FrameEpi& e, const int x0, const int y0, const uint32_t npix) {
  if constexpr (EPI) {
    if (e.unit_cnt != nullptr && npix != 0u) {
      __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) (expcnt / lgkmcnt left alone)
      if (__lane_id() == 0) {
        const uint32_t u = (uint32_t)(y0 >> 4) * (uint32_t)e.units_x + (uint32_t)(x0 >> 6);
        const uint32_t nd = (uint32_t)e.drain_wgs * RW_WAVES;   // counter of unit u: drain wave u % nd, slot u / nd
        __hip_atomic_fetch_add(e.unit_cnt + (size_t)(u % nd) * 64u + u / nd, npix, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}
// pixels of a wave's rows that lie inside the image (lanes & 15 -> x, the wave-uniform rest -> rows)
__device__ __forceinline__ uint32_t wave_pixels_inside(const bool inside) { return (uint32_t)__popcll(__ballot(inside)); }

__device__ __forceinline__ void pc_fail(const PCErr err, const int lane);   // (below, with the wave pairs)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// One wave of a drain workgroup.  It owns the units  wid + nd * lane  (nd = drain waves of the launch), polls their
// counters and copies a unit the moment it is complete: rows x 192 bytes = whole 64-byte lines, read past the L2
// (the blending waves sit on other XCDs: agent scope) and written through to the host in 16-byte pieces -- four
// neighbouring lanes make a line, so the link carries 64-byte writes (tools/ubench/host_store.hip: 0.145 ms of link
// time per 1920x1280 frame, against 0.21 for the tiles' own 48-byte row segments).  A unit's counter goes back to
// zero when the unit is taken (nobody arrives at a complete unit): a frame that is rendered twice (capacity overflow)
// starts clean.  The wait is bounded like
// the wave pairs' (pc_fail): a lost arrival is reported, not waited for.
__device__ __forceinline__ void drain_units(const FrameEpi& e, const int W, const int H, const uint32_t wid,
                                            const uint32_t nd, const int lane, const PCErr err) {
#ifdef GRPG_DRAIN_IDLE   // experiment build: nobody drains (the blending waves' share of the cost; wrong host bytes)
  return;
#endif
  const uint32_t gy = (uint32_t)(H + 15) >> 4, NU = (uint32_t)e.units_x * gy;
  const size_t pitch = (size_t)3 * W;
  // unit u belongs to wave u % nd, slot u / nd (< 64: the launch has >= NU / 64 drain waves); a wave's 64 counters
  // are contiguous (one 256-byte poll), its units spread over the frame (neighbours finish together: not one
  // wave's burst)
  const uint32_t u = (uint32_t)lane * nd + wid;
  const bool valid = u < NU;
  const uint32_t uy = valid ? u / (uint32_t)e.units_x : 0u, ux = u - uy * (uint32_t)e.units_x;
  // (the last unit of a row is narrower when W is no multiple of 64: W % 16 == 0 keeps its rows whole 16-byte pieces)
  const uint32_t target = (uint32_t)min(64, W - 64 * (int)ux) * (uint32_t)min(16, H - 16 * (int)uy);
  uint32_t* const my_cnt = e.unit_cnt + (size_t)wid * 64u + (uint32_t)lane;
  uint64_t pending = __ballot(valid);
  const uint64_t t0 = wall_clock64();
  while (pending != 0ull) {
    const bool mine = (pending >> lane) & 1ull;
    const uint32_t c = mine ? __hip_atomic_load(my_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    uint64_t ready = __ballot(mine && c >= target);
    if (ready == 0ull) {
      if (wall_clock64() - t0 > 50000000ull) { pc_fail(err, lane); return; }   // 0.5 s of the 100 MHz clock
      __builtin_amdgcn_s_sleep(32);
      continue;
    }
    pending &= ~ready;
    if (mine && c >= target) __hip_atomic_store(my_cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (ready != 0ull) {   // two units per trip: six 16-byte reads in flight per lane
      const int b0 = (int)__builtin_ctzll(ready);
      ready &= ready - 1ull;
      const bool two = ready != 0ull;
      const int b1 = two ? (int)__builtin_ctzll(ready) : b0;
      ready &= ready - 1ull;   // (0 & anything stays 0)
      size_t off[2][3];
      uint64_t v[2][3][2];
      int chunks[2];
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const uint32_t uu = (uint32_t)(k == 0 ? b0 : b1) * nd + wid;
        const uint32_t ry = uu / (uint32_t)e.units_x, rx = uu - ry * (uint32_t)e.units_x;
        const int cpr = 3 * min(64, W - 64 * (int)rx) / 16;                  // 16-byte pieces per row: 12, fewer in a last unit
        const int nch = cpr * min(16, H - 16 * (int)ry);                    // ... of the unit
        chunks[k] = (k == 1 && !two) ? 0 : nch;
        const size_t org = (size_t)16 * ry * pitch + (size_t)192 * rx;
#pragma unroll
        for (int i = 0; i < 3; i++) {
          const int cidx = min(i * 64 + lane, nch - 1);   // (beyond the unit: a harmless second read)
          const int crow = cidx / cpr;
          off[k][i] = org + (size_t)crow * pitch + (size_t)16 * (cidx - crow * cpr);
          const uint64_t* src = (const uint64_t*)(e.rgb8 + off[k][i]);
          v[k][i][0] = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          v[k][i][1] = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
#pragma unroll
      for (int k = 0; k < 2; k++) {
#pragma unroll
        for (int i = 0; i < 3; i++) {
          if (i * 64 + lane < chunks[k]) {
            const u32x4 q = {(uint32_t)v[k][i][0], (uint32_t)(v[k][i][0] >> 32), (uint32_t)v[k][i][1],
                             (uint32_t)(v[k][i][1] >> 32)};
            unsigned char* dst = e.host8 + off[k][i];
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(dst), "v"(q) : "memory");
          }
        }
      }
    }
  }
}

template <bool EPI>
struct PlainOutT {
  const float* bg; float* out_color; float* out_depth; float* out_alpha; uint32_t* n_contrib;
  int W, H;
  FrameEpi epi;
  template <bool AUX>
  __device__ __forceinline__ void pixel(const int px, const int py, const float T, const v2f CrCg, const v2f CbD,
                                        const uint32_t last) const {
    const size_t HW = (size_t)H * W, pix = (size_t)py * W + px;
    float rgb[3] = {CrCg.x + T * bg[0], CrCg.y + T * bg[1], CbD.x + T * bg[2]};
    frame_finish<EPI>(epi, px, py, W, H, T, rgb);
    if (!EPI || epi.planes) {
      out_color[pix] = rgb[0];
      out_color[HW + pix] = rgb[1];
      out_color[2 * HW + pix] = rgb[2];
      out_alpha[pix] = 1.0f - T;
      out_depth[pix] = CbD.y;
    }
    if (AUX) n_contrib[pix] = last;
  }
  // the wave's pixels of tile (x0, y0 / 16) are stored (drained frame: frame_arrive)
  __device__ __forceinline__ void arrive(const int x0, const int y0, const uint32_t npix) const {
    frame_arrive<EPI>(epi, x0, y0, npix);
  }
};
typedef PlainOutT<false> PlainOut;

struct LayerOut {
  const float* bg_layer;     // [3] background of the two layer planes (the reference renders them on white)
  float* color_bg; float* alpha_bg; float* color_obj; float* alpha_obj;
};

template <bool EPI>
struct LayersOutT {
  const float* bg; float* out_color; float* out_depth; float* out_alpha;
  int W, H;
  LayerOut lo;
  FrameEpi epi;
  // the composition's planes / the background layer's / the object layer's, from that layer's final state
  __device__ __forceinline__ void pixel_a(const int px, const int py, const float T, const v2f CrCg, const v2f CbD) const {
    const size_t HW = (size_t)H * W, pix = (size_t)py * W + px;
    float rgb[3] = {CrCg.x + T * bg[0], CrCg.y + T * bg[1], CbD.x + T * bg[2]};
    frame_finish<EPI>(epi, px, py, W, H, T, rgb);
    if (!EPI || epi.planes) {
      out_color[pix] = rgb[0];
      out_color[HW + pix] = rgb[1];
      out_color[2 * HW + pix] = rgb[2];
      out_alpha[pix] = 1.0f - T;
      out_depth[pix] = CbD.y;
    }
  }
  __device__ __forceinline__ void pixel_layer(float* __restrict__ color, float* __restrict__ alpha, const int px,
                                              const int py, const float T, const v2f CrCg, const v2f CbD) const {
    const size_t HW = (size_t)H * W, pix = (size_t)py * W + px;
    color[pix] = CrCg.x + T * lo.bg_layer[0];
    color[HW + pix] = CrCg.y + T * lo.bg_layer[1];
    color[2 * HW + pix] = CbD.x + T * lo.bg_layer[2];
    alpha[pix] = 1.0f - T;
  }
  __device__ __forceinline__ void pixel_b(const int px, const int py, const float T, const v2f CrCg, const v2f CbD) const {
    pixel_layer(lo.color_bg, lo.alpha_bg, px, py, T, CrCg, CbD);
  }
  __device__ __forceinline__ void pixel_o(const int px, const int py, const float T, const v2f CrCg, const v2f CbD) const {
    pixel_layer(lo.color_obj, lo.alpha_obj, px, py, T, CrCg, CbD);
  }
  __device__ __forceinline__ void pixel3(const int px, const int py, const float Ta, const v2f CrCga, const v2f CbDa,
                                         const float Tb, const v2f CrCgb, const v2f CbDb,
                                         const float To, const v2f CrCgo, const v2f CbDo) const {
    pixel_a(px, py, Ta, CrCga, CbDa);
    pixel_b(px, py, Tb, CrCgb, CbDb);
    pixel_o(px, py, To, CrCgo, CbDo);
  }
  template <bool AUX>
  __device__ __forceinline__ void pixel(const int px, const int py, const float T, const v2f CrCg, const v2f CbD,
                                        const uint32_t) const {
    pixel3(px, py, T, CrCg, CbD, T, CrCg, CbD, 1.0f, (v2f){0.f, 0.f}, (v2f){0.f, 0.f});
  }
  // (the epilogue belongs to the composition: every path that writes pixel_a arrives once per wave)
  __device__ __forceinline__ void arrive(const int x0, const int y0, const uint32_t npix) const {
    frame_arrive<EPI>(epi, x0, y0, npix);
  }
};
typedef LayersOutT<false> LayersOut;

struct WaveTrace { uint32_t batches, survivors, blends, t_stage, t_loop; uint32_t it[4], cyc[4]; };

template <int PX, int GPI, bool WRITE_AUX, bool TRACE = false, class Out = PlainOut>
__device__ __forceinline__ void blend_rect(float4* __restrict__ my, const int lane,
                                           const uint32_t r_begin, const uint32_t r_end,
                                           const int x0, const int y0, const int W, const int H,
                                           const uint32_t* __restrict__ point_list,
                                           const RecView rec, const Out& out,
                                           WaveTrace* tr = nullptr, const int ablate = 0) {
  const int px = x0 + (lane & 15);
  const int py0 = y0 + (lane >> 4) * PX;
  const float pxf = (float)px;
  // pixel rectangle of this wave (for the cull), clipped rows/cols do not matter (superset)
  float rx0 = (float)x0, rx1 = (float)(x0 + 15);
  float ry0 = (float)y0, ry1 = (float)(y0 + 4 * PX - 1);
  uint64_t prev_alive = ~0ull;

  WavePix<PX> st;
#pragma unroll
  for (int k = 0; k < PX; k++) {
    st.T[k] = 1.0f; st.CrCg[k] = (v2f){0.f, 0.f}; st.CbD[k] = (v2f){0.f, 0.f};
    st.last[k] = 0;
    st.done[k] = lanes(!(px < W && (py0 + k) < H));
  }

  // Software pipeline over batches of 64 list entries: while batch i is blended, the records of
  // batch i+1 and the ids of batch i+2 are in flight (the id -> record gather is a dependent pair
  // of L2 / Infinity-Cache round trips).  Deeper register queues were tried (3 batches, loop
  // unrolled over fixed slots): hipcc still waits vmcnt(0) at the top of every batch, so they only
  // cost registers (occupancy 4 -> 3) -- kept at one batch.
  // An entry whose sub-tile mask is empty cannot reach any pixel of the tile: its record is
  // never loaded (about half of a light tile's list on the bench scene).
  uint32_t id_n2 = 0;
  float4 a_n = make_float4(0, 0, 0, 0), b_n = a_n, c_n = a_n;
  bool live_n = false;
  if (r_begin + (uint32_t)lane < r_end) {
    const uint32_t v0 = point_list[r_begin + lane];
    live_n = (v0 >> SUBTILE_SHIFT) != 0u;
    if (live_n) {
      rec.load(v0 & ID_MASK, a_n, b_n, c_n);
    }
  }
  if (r_begin + WAVE + (uint32_t)lane < r_end) id_n2 = point_list[r_begin + WAVE + lane];

  for (uint32_t base = r_begin; base < r_end; base += WAVE) {
    uint64_t alldone = ~0ull;
#pragma unroll
    for (int k = 0; k < PX; k++) alldone &= st.done[k];
    const uint64_t alive = ~alldone;
    if (alive == 0ull) break;
    if (alive != prev_alive) {
      // Shrink the cull rectangle to the bounding box of the pixels that are still live.  Exact:
      // a splat is only dropped if no LIVE pixel can accept it.  In the long tail of a heavy
      // sub-tile only a handful of unsaturated pixels keep scanning tens of thousands of entries;
      // with the tight box almost nothing survives the cull and a batch costs little more than
      // its staging.
      prev_alive = alive;
      // rows y0 + PX * (lane >> 4) + k: per pixel slot k the box of its live lanes (scalar unit, mask_box)
      int c0 = 15, c1 = 0, r0 = 4 * PX, r1 = -1;
#pragma unroll
      for (int k = 0; k < PX; k++) {
        const uint64_t m = ~st.done[k];
        if (m != 0ull) {
          const MaskBox b = mask_box(m);
          c0 = min(c0, b.c0); c1 = max(c1, b.c1);
          r0 = min(r0, PX * b.r0 + k); r1 = max(r1, PX * b.r1 + k);
        }
      }
      rx0 = (float)(x0 + c0); rx1 = (float)(x0 + c1); ry0 = (float)(y0 + r0); ry1 = (float)(y0 + r1);
    }
    const uint32_t n = min((uint32_t)WAVE, r_end - base);
    const float4 a = a_n, b = b_n, c = c_n;
    const bool live = live_n;
    live_n = false;
    if (base + WAVE + (uint32_t)lane < r_end) {        // records of the next batch
      live_n = (id_n2 >> SUBTILE_SHIFT) != 0u;
      if (live_n) {
        rec.load(id_n2 & ID_MASK, a_n, b_n, c_n);
      }
    }
    if (base + 2 * WAVE + (uint32_t)lane < r_end)      // ids of the batch after that
      id_n2 = point_list[base + 2 * WAVE + lane];
    const uint64_t tc0 = TRACE ? __builtin_readcyclecounter() : 0;
    const bool keep = ((uint32_t)lane < n) && live &&
                      !splat_misses_rect(a.x, a.y, b.x, b.y, b.z, a.w, rx0, rx1, ry0, ry1);
    const uint64_t mask = __ballot(keep);
    const int cnt = (int)__popcll(mask);
    if (keep)   // list positions are 1-based (n_contrib convention of the reference)
      store_slot(my, (int)__popcll(mask & lanemask_lt()), a, b, c, base - r_begin + 1 + (uint32_t)lane);
    if (TRACE) { tr->batches++; tr->survivors += (uint32_t)cnt; }
    __builtin_amdgcn_wave_barrier();

    const uint64_t tc1 = TRACE ? __builtin_readcyclecounter() : 0;
    if (TRACE) tr->t_stage += (uint32_t)(tc1 - tc0);
    {
      // full groups of GPI survivors, then the remainder one by one: no dummy slots are evaluated
      int j0 = (TRACE && (ablate & 1)) ? cnt : 0;
      for (; j0 + GPI <= cnt; j0 += GPI) {
        const bool blended = blend_group<PX, GPI, WRITE_AUX>(st, my, j0, pxf, py0);
        if (TRACE && blended) tr->blends++;
      }
      if (GPI > 1) {
        for (; j0 < cnt; j0++) {
          const bool blended = blend_group<PX, 1, WRITE_AUX>(st, my, j0, pxf, py0);
          if (TRACE && blended) tr->blends++;
        }
      }
    }
    if (TRACE) tr->t_loop += (uint32_t)(__builtin_readcyclecounter() - tc1);
    __builtin_amdgcn_wave_barrier();
  }

  uint32_t npix = 0u;
#pragma unroll
  for (int k = 0; k < PX; k++) {
    const int py = py0 + k;
    if (px < W && py < H) out.template pixel<WRITE_AUX>(px, py, st.T[k], st.CrCg[k], st.CbD[k], st.last[k]);
    npix += wave_pixels_inside(px < W && py < H);
  }
  out.arrive(x0, y0, npix);
}

// ------------------------------------------------------------------------------------------
// HEAVY path: one wave = one 16x4 quarter of a heavy tile, 1 pixel per lane.
//
// The list of a horizon tile holds tens of thousands of entries of which only ~1/3 can reach a
// given quarter (and far fewer once most of its pixels are saturated).  A lone wave issues about
// one instruction per 5-6 cycles whatever its type (tools/ubench/lone_wave.hip), so the serial
// scan of dead entries -- not arithmetic -- was the frame's critical path.  Here the scan is
// decoupled from the blending through a small LDS ring:
//   FILL : read 256 list entries at a time (4 per lane, contiguous, prefetched one window ahead),
//          test the quarter's bit of the sub-tile mask that emit stored in the entry, and append
//          the survivors' (id, list position) to the ring (ballot + popcount prefix, no atomics);
//   POP  : take up to 64 survivors from the ring and issue the gather of their 48-byte records;
//   BLEND: while that gather is in flight, cull the PREVIOUS batch against the bounding box of the
//          still-live pixels, compact it into LDS and run the blend groups.
// Dead entries cost 1/256 of a FILL step instead of a slot of a 64-entry batch; a record is only
// ever loaded for an entry that can reach the quarter.
// ------------------------------------------------------------------------------------------
constexpr int QCAP = 512;            // ring capacity in entries (>= 64 + 256), power of two
constexpr int FILL_Q = 4;            // list entries per lane per FILL step of the ordinary quarter waves

// ------------------------------------------------------------------------------------------
// A wave's view of its tile list: a STREAM read in large windows.
//
// The ordinary quarter waves (blend_heavy) read the list 256 entries at a time (four predicated 4-byte
// loads per lane, prefetched ONE step ahead): every FILL step waits out a memory latency (the list was
// written by the fill kernel a moment ago: HBM / Infinity Cache, 1 - 2 us).  With thousands of waves in
// flight that is hidden; for ONE wave walking 8 k - 35 k entries -- a class-0 producer, or the quarter wave of
// such a tile in a layered frame, which has no wave pairs -- those 126 steps are most of its life
// (tools/trace_class0.py: 1.9 us per batch for 0.7 us of work).  Here a window is SUB sub-windows of 256
// entries, requested together with 16-byte loads (lane l owns four consecutive entries) and consumed from
// registers: one memory latency per SUB x 256 entries.  SUB = 8 for the producers (a wave of its own with
// registers to spare), 4 for the layered frame's quarter waves.  (The ordinary quarter waves were measured on
// it too: render 0.2095-0.2112 vs 0.2074-0.2104 ms on one box -- the bulk of the launch is bound by the
// vector ALUs, they keep their 256-entry windows.)
// ------------------------------------------------------------------------------------------
template <int SUB>
struct ListStream {
  static_assert(SUB == 4 || SUB == 8, "sub-windows per register window");
  static constexpr uint32_t SPAN = SUB * 256u;   // entries per register window
  const uint32_t* pl;
  uint32_t r_begin, r_end;
  uint32_t wpos;      // list index (absolute, a multiple of 256) of c0's first entry
  uint32_t q;         // next sub-window of the register window
  // Named registers, not an array: a (wave-uniform) dynamic index would send an array to scratch.
  // ONE bank, refilled when it is used up.  A second bank loaded a window ahead was built three ways
  // (moves between banks, two banks taking turns, a scheduling barrier between moves and reloads): hipcc
  // either routes the reloads through temporaries and waits for them on the spot (loop-carried registers
  // are not coalesced with the load destinations) or waits vmcnt(0) at every use -- no better than this,
  // for twice the registers.
  uint4 c0, c1, c2, c3, c4, c5, c6, c7;
  // UNCONDITIONAL load (address clamped into the list; append() drops entries outside [r_begin, r_end) by
  // index): a predicated load makes the compiler copy the result through a temporary.
  __device__ __forceinline__ uint4 load(const uint32_t first, const int lane) const {
    const uint32_t e = min(first + 4u * (uint32_t)lane, (r_end - 1u) & ~3u);   // 16-byte aligned
    return *reinterpret_cast<const uint4*>(pl + e);
  }
  __device__ __forceinline__ void refill(const int lane) {
    c0 = load(wpos, lane); c1 = load(wpos + 256u, lane); c2 = load(wpos + 512u, lane); c3 = load(wpos + 768u, lane);
    if (SUB == 8) {
      c4 = load(wpos + 1024u, lane); c5 = load(wpos + 1280u, lane);
      c6 = load(wpos + 1536u, lane); c7 = load(wpos + 1792u, lane);
    }
  }
  // re > rb (an empty list must not be opened: the clamp needs r_end >= 1)
  __device__ __forceinline__ void open(const uint32_t* point_list, const uint32_t rb, const uint32_t re, const int lane) {
    pl = point_list; r_begin = rb; r_end = re;
    wpos = rb & ~255u; q = 0;
    refill(lane);
  }
  __device__ __forceinline__ bool exhausted() const { return wpos + 256u * q >= r_end; }
  // Appends the entries of sub-window v (first entry e0 for this lane) whose mask has `bit` to the ring, in
  // list order: lane-major, then the lane's four entries.
  // wmask / wval: an entry must additionally satisfy (entry & wmask) == wval -- a layered frame's "objects only"
  // (LAYER_BIT, LAYER_BIT) and "no objects" (LAYER_BIT, 0) phases; (0, 0) keeps everything.
  __device__ __forceinline__ void append(const uint4 v, const uint32_t e0, uint32_t* __restrict__ qid,
                                         uint32_t* __restrict__ qpos, const uint32_t bit, const uint32_t head,
                                         uint32_t& count, const uint64_t lt, const uint32_t wmask,
                                         const uint32_t wval) const {
    const bool k0 = (e0 >= r_begin) && (e0 < r_end) && (v.x & bit) && ((v.x & wmask) == wval);
    const bool k1 = (e0 + 1u >= r_begin) && (e0 + 1u < r_end) && (v.y & bit) && ((v.y & wmask) == wval);
    const bool k2 = (e0 + 2u >= r_begin) && (e0 + 2u < r_end) && (v.z & bit) && ((v.z & wmask) == wval);
    const bool k3 = (e0 + 3u >= r_begin) && (e0 + 3u < r_end) && (v.w & bit) && ((v.w & wmask) == wval);
    const uint64_t m0 = __ballot(k0), m1 = __ballot(k1), m2 = __ballot(k2), m3 = __ballot(k3);
    uint32_t sl = head + count + (uint32_t)__popcll(m0 & lt) + (uint32_t)__popcll(m1 & lt) +
                  (uint32_t)__popcll(m2 & lt) + (uint32_t)__popcll(m3 & lt);
    const uint32_t p0 = e0 - r_begin + 1u;   // 1-based position in the tile's list
    if (k0) { qid[sl & (QCAP - 1)] = v.x & ID_MASK; qpos[sl & (QCAP - 1)] = p0; sl++; }
    if (k1) { qid[sl & (QCAP - 1)] = v.y & ID_MASK; qpos[sl & (QCAP - 1)] = p0 + 1u; sl++; }
    if (k2) { qid[sl & (QCAP - 1)] = v.z & ID_MASK; qpos[sl & (QCAP - 1)] = p0 + 2u; sl++; }
    if (k3) { qid[sl & (QCAP - 1)] = v.w & ID_MASK; qpos[sl & (QCAP - 1)] = p0 + 3u; }
    count += (uint32_t)__popcll(m0) + (uint32_t)__popcll(m1) + (uint32_t)__popcll(m2) + (uint32_t)__popcll(m3);
  }
  // one FILL step: the next sub-window (256 entries); count is the ring's fill state, which must leave
  // room for 256 entries.  q is wave-uniform: scalar branches, each naming its register.
  __device__ __forceinline__ void fill(uint32_t* __restrict__ qid, uint32_t* __restrict__ qpos, const uint32_t bit,
                                       const uint32_t head, uint32_t& count, const int lane, const uint64_t lt,
                                       const uint32_t wmask = 0u, const uint32_t wval = 0u) {
    const uint32_t e0 = wpos + 256u * q + 4u * (uint32_t)lane;
    switch (q) {
      case 0u: append(c0, e0, qid, qpos, bit, head, count, lt, wmask, wval); break;
      case 1u: append(c1, e0, qid, qpos, bit, head, count, lt, wmask, wval); break;
      case 2u: append(c2, e0, qid, qpos, bit, head, count, lt, wmask, wval); break;
      case 3u: append(c3, e0, qid, qpos, bit, head, count, lt, wmask, wval); break;
      case 4u: append(c4, e0, qid, qpos, bit, head, count, lt, wmask, wval); break;
      case 5u: append(c5, e0, qid, qpos, bit, head, count, lt, wmask, wval); break;
      case 6u: append(c6, e0, qid, qpos, bit, head, count, lt, wmask, wval); break;
      default: append(c7, e0, qid, qpos, bit, head, count, lt, wmask, wval); break;
    }
    if (++q == (uint32_t)SUB) { q = 0; wpos += SPAN; if (wpos < r_end) refill(lane); }
  }
};

// Blend checkpoints for the backward (common.h CK_*): per-quarter state of a long tile's walk.
struct CkptWriter {
  float* recs;        // record 0 of this (tile, quarter); record k is 4 records further each; NULL = off
  uint32_t* count;    // receives the number of records written
  uint32_t k, next, cap;
};
__device__ __forceinline__ CkptWriter ckpt_writer(const CkptArgs& ck, const uint32_t tile, const int quarter,
                                                  const uint32_t rb, const uint32_t re) {
  CkptWriter w = {nullptr, nullptr, 0u, CK_SEG, 0u};
  if (ck.recs != nullptr && re - rb >= CK_LONG_MIN) {
    w.recs = ck.recs + ((size_t)ckpt_tile_base(rb) * 4 + (size_t)quarter) * CK_REC_FLOATS;
    w.count = ck.counts + (size_t)tile * 4 + quarter;
    w.cap = ckpt_tile_cap(re - rb);
  }
  return w;
}
__device__ __forceinline__ void ckpt_store(CkptWriter& w, const int lane, const WavePix<1>& st,
                                           const uint32_t pos /* 1-based position the state follows */) {
  float* r = w.recs + (size_t)w.k * 4 * CK_REC_FLOATS + lane;
  r[0] = st.T[0];
  r[64] = st.CrCg[0].x; r[128] = st.CrCg[0].y;
  r[192] = st.CbD[0].x; r[256] = st.CbD[0].y;
  if (lane == 0) r[320] = __uint_as_float(pos);
  w.k++;
  w.next = pos + CK_SEG;
}
// a batch whose last entry sits at list position last_pos has been blended
__device__ __forceinline__ void ckpt_batch_end(CkptWriter& w, const int lane, const WavePix<1>& st,
                                               const uint32_t last_pos) {
  if (w.recs != nullptr && last_pos >= w.next && w.k + 1u < w.cap) ckpt_store(w, lane, st, last_pos);
}
// end of the walk: the final state, tagged with the list length
__device__ __forceinline__ void ckpt_finish(CkptWriter& w, const int lane, const WavePix<1>& st,
                                            const uint32_t len) {
  if (w.recs == nullptr) return;
  ckpt_store(w, lane, st, len);
  if (lane == 0) *w.count = w.k;
}

// Quarter 0 of a long tile leaves the backward's work items behind: (tile, k) for every record index
// a quarter of this tile can have written (a quarter that cut its list into fewer pieces skips the
// surplus items).  One atomic per long tile, at the very end of the wave: nobody waits for it.
__device__ __forceinline__ void ckpt_publish_items(const CkptArgs& ck, const int lane, const uint32_t tile,
                                                   const uint32_t len) {
  if (ck.recs == nullptr || len < CK_LONG_MIN) return;
  const uint32_t cap = ckpt_tile_cap(len);
  uint32_t base = 0;
  if (lane == 0) base = atomicAdd(ck.n_items, cap);
  base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
  uint2* items = (uint2*)((char*)ck.recs + ckpt_items_offset(ck.slots));
  for (uint32_t k = (uint32_t)lane; k < cap; k += WAVE)
    if (base + k < ck.slots) items[base + k] = make_uint2(tile, k);
}

// Writes the wave's semantic accumulators (channels < min(S, NSEM)); semantics get no background.  The
// MFMA accumulators hold, per pixel row y, columns 4 (lane / 16) + i of channel lane % 16: they go through
// LDS ([channel][pixel of the quarter]; two areas of 8 channels: the wave's dead ring) so that the planes
// are written by pixel, 64 bytes per row and channel.  inside: this lane's pixel lies in the image.
template <int NSEM>
__device__ __forceinline__ void sem_write(const SemAcc<NSEM>& sa, const int S, float* __restrict__ out_semantic,
                                          const size_t HW, const size_t pix, const bool inside, const int lane,
                                          float* __restrict__ t_lo, float* __restrict__ t_hi) {
  const int ch = lane & 15, xg = lane >> 4;
  float* t = (ch < 8 ? t_lo : t_hi) + (ch & 7) * WAVE;
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int y = 0; y < 4; y++)
    *reinterpret_cast<float4*>(t + y * 16 + 4 * xg) = make_float4(sa.acc[y].x, sa.acc[y].y, sa.acc[y].z, sa.acc[y].w);
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int c = 0; c < NSEM; c++)
    if (c < S && inside) out_semantic[(size_t)c * HW + pix] = (c < 8 ? t_lo : t_hi)[(c & 7) * WAVE + lane];
}

// Which list entries a walk takes -- everything in an ordinary frame; in a layered frame (further down) a layer's
// own walk of a tile WITH object entries takes the entries of its class: (entry & wmask) == wval, and strips the
// class bit before it indexes the records (id_and).  The defaults fold away at compile time.
struct ClassFilter {
  uint32_t wmask = 0u, wval = 0u, id_and = 0xFFFFFFFFu;
};

template <bool TRACE, bool AUX = true, int NSEM = 0, class Out = PlainOut>
__device__ __forceinline__ void blend_heavy(float4* __restrict__ my, uint32_t* __restrict__ qid,
                                            uint32_t* __restrict__ qpos, const int lane,
                                            const int quarter, const uint32_t r_begin,
                                            const uint32_t r_end, const int x0, const int y0,
                                            const int W, const int H,
                                            const uint32_t* __restrict__ point_list,
                                            const RecView rec, const Out& out, WaveTrace* tr,
                                            CkptWriter ckw, const SemSrc sem = SemSrc{nullptr, 0, nullptr},
                                            float* __restrict__ out_semantic = nullptr,
                                            float* __restrict__ semrows = nullptr /* LDS: WAVE x SEM_ROW */,
                                            const ClassFilter cf = ClassFilter{}) {
  SemAcc<NSEM> sa;
  sa.clear();
  const SemSrc semb = {sem.semantics, sem.S, semrows};
  const int px = x0 + (lane & 15);
  const int py = y0 + (lane >> 4);
  const float pxf = (float)px;
  float rx0 = (float)x0, rx1 = (float)(x0 + 15), ry0 = (float)y0, ry1 = (float)(y0 + 3);
  uint64_t prev_alive = ~0ull;
  const uint32_t bit = 1u << (SUBTILE_SHIFT + quarter);
  const uint64_t lt = lanemask_lt();

  WavePix<1> st;
  st.T[0] = 1.0f; st.CrCg[0] = (v2f){0.f, 0.f}; st.CbD[0] = (v2f){0.f, 0.f};
  st.last[0] = 0;
  st.done[0] = lanes(!(px < W && py < H));

  uint32_t in_pos = r_begin;   // next unread list entry          (wave-uniform)
  uint32_t head = 0, count = 0;   // ring state                   (wave-uniform)
  uint32_t win[FILL_Q];        // the next FILL window's entries, in flight
#pragma unroll
  for (int q = 0; q < FILL_Q; q++) {
    const uint32_t i = in_pos + q * WAVE + lane;
    win[q] = i < r_end ? point_list[i] : 0u;
  }
  float4 a = make_float4(0, 0, 0, 0), b = a, c = a;   // current batch (records arrived)
  uint32_t pos = 0, idc = 0;
  uint32_t ncur = 0;

  for (;;) {
    const uint64_t alive = ~st.done[0];
    if (alive == 0ull) break;
    if (alive != prev_alive) {   // shrink the cull box to the live pixels (exact, see blend_rect)
      prev_alive = alive;
      const MaskBox b = mask_box(alive);
      rx0 = (float)(x0 + b.c0); rx1 = (float)(x0 + b.c1); ry0 = (float)(y0 + b.r0); ry1 = (float)(y0 + b.r1);
    }

    // ---- FILL ----
    while (count < (uint32_t)WAVE && in_pos < r_end) {
      uint32_t v[FILL_Q];
#pragma unroll
      for (int q = 0; q < FILL_Q; q++) v[q] = win[q];
      const uint32_t nxt = in_pos + FILL_Q * WAVE;
#pragma unroll
      for (int q = 0; q < FILL_Q; q++) {
        const uint32_t i = nxt + q * WAVE + lane;
        win[q] = i < r_end ? point_list[i] : 0u;
      }
#pragma unroll
      for (int q = 0; q < FILL_Q; q++) {
        const uint32_t i = in_pos + q * WAVE + lane;
        const bool keep = (i < r_end) && (v[q] & bit) && ((v[q] & cf.wmask) == cf.wval);
        const uint64_t m = __ballot(keep);
        if (keep) {
          const uint32_t slot = (head + count + (uint32_t)__popcll(m & lt)) & (QCAP - 1);
          qid[slot] = v[q] & ID_MASK;
          qpos[slot] = i - r_begin + 1;   // 1-based position in the tile's list
        }
        count += (uint32_t)__popcll(m);
      }
      in_pos = nxt;
      if (TRACE) tr->batches++;
    }

    // the ring entries written by FILL are read by OTHER lanes in POP: keep the compiler from
    // reordering the LDS accesses across this point (costs no instruction)
    __builtin_amdgcn_wave_barrier();
    // ---- POP: next batch of up to 64 survivors, start its record gather ----
    const uint32_t nn = min(count, (uint32_t)WAVE);
    float4 a_n = make_float4(0, 0, 0, 0), b_n = a_n, c_n = a_n;
    uint32_t pos_n = 0, id_n = 0;
    if ((uint32_t)lane < nn) {
      const uint32_t slot = (head + lane) & (QCAP - 1);
      id_n = qid[slot] & cf.id_and;
      pos_n = qpos[slot];
      rec.load(id_n, a_n, b_n, c_n);
    }
    head = (head + nn) & (QCAP - 1);
    count -= nn;

    // ---- BLEND the previous batch while the gather is in flight ----
    if (ncur > 0) {
      const uint64_t tc0 = TRACE ? __builtin_readcyclecounter() : 0;
      const bool keep = ((uint32_t)lane < ncur) &&
                        !splat_misses_rect(a.x, a.y, b.x, b.y, b.z, a.w, rx0, rx1, ry0, ry1);
      const uint64_t mask = __ballot(keep);
      const int cnt = (int)__popcll(mask);
      const SplatQ sq = splat_q(b.x, b.y, b.z);
      // wave-uniform: no survivor of this batch can produce power > 0 (all but pathological conics)
      const bool batch_safe = __ballot(keep && !splat_power_never_positive(sq)) == 0ull;
      if (keep) {
        store_pair_half(my, (int)__popcll(mask & lt), a.x, a.y, sq, a.w,
                        make_float4(b.w, c.x, c.y, a.z), pos);
        if (NSEM > 0) sem_stage(semrows, (int)__popcll(mask & lt), sem.semantics, sem.S, idc, true);
      }
      if (lane < ((4 - (cnt & 3)) & 3)) {   // neutral pads up to a multiple of 4 (opacity 0)
        const SplatQ zq = {0.f, 0.f, 0.f};
        store_pair_half(my, cnt + lane, 0.f, 0.f, zq, 0.f, make_float4(0.f, 0.f, 0.f, 0.f), 0u);
        if (NSEM > 0) sem_stage(semrows, cnt + lane, sem.semantics, sem.S, 0u, false);
      }
      if (TRACE) tr->survivors += (uint32_t)cnt;
      __builtin_amdgcn_wave_barrier();
      const uint64_t tc1 = TRACE ? __builtin_readcyclecounter() : 0;
      if (TRACE) tr->t_stage += (uint32_t)(tc1 - tc0);
      if (batch_safe) {
        for (int j0 = 0; j0 < cnt; j0 += 4) {
          const bool blended = blend_quad<AUX, NSEM, true>(st, my, j0, pxf, (float)py, &sa, semb, lane);
          if (TRACE && blended) tr->blends++;
        }
      } else {
        for (int j0 = 0; j0 < cnt; j0 += 4) {
          const bool blended = blend_quad<AUX, NSEM, false>(st, my, j0, pxf, (float)py, &sa, semb, lane);
          if (TRACE && blended) tr->blends++;
        }
      }
      if (TRACE) {
        const uint64_t tc2 = __builtin_readcyclecounter();
        tr->t_loop += (uint32_t)(tc2 - tc1);
        const int na = (int)__popcll(alive);   // live pixels when this batch was culled
        const int bk = na <= 2 ? 0 : (na <= 8 ? 1 : (na <= 24 ? 2 : 3));
        tr->it[bk]++; tr->cyc[bk] += (uint32_t)(tc2 - tc0);
      }
      // every entry up to the batch's last ring entry is now blended, masked out or culled
      if (AUX) ckpt_batch_end(ckw, lane, st, (uint32_t)__builtin_amdgcn_readlane((int)pos, (int)ncur - 1));
      __builtin_amdgcn_wave_barrier();
    }
    a = a_n; b = b_n; c = c_n; pos = pos_n; idc = id_n; ncur = nn;
    if (ncur == 0 && in_pos >= r_end) break;   // ring empty (count == 0 here) and list exhausted
  }
  if (AUX) ckpt_finish(ckw, lane, st, r_end - r_begin);

  if (px < W && py < H) out.template pixel<AUX>(px, py, st.T[0], st.CrCg[0], st.CbD[0], st.last[0]);
  out.arrive(x0, y0, wave_pixels_inside(px < W && py < H));
  if (NSEM > 0)   // (all lanes: the accumulators are spread over the wave; the ring is dead by now)
    sem_write<NSEM>(sa, sem.S, out_semantic, (size_t)H * W, (size_t)py * W + px, px < W && py < H, lane,
                    reinterpret_cast<float*>(qid), reinterpret_cast<float*>(qpos));
}

// ------------------------------------------------------------------------------------------
// Producer / consumer wave pairs for the LONGEST tiles (class 0 when pc_mode is on).
//
// A horizon quarter is a serial chain (the T recurrence) that outlives the rest of the launch, and
// a wave running alone on its SIMD issues one instruction per ~6 cycles whatever it does.  Half
// of that chain is not the recurrence at all: scanning the list (FILL), gathering records (POP),
// culling and compacting.  Such a tile is rendered by TWO workgroups (quarters 0-1 and 2-3), and
// each quarter gets two waves of its workgroup: the producer runs FILL / POP / cull / compaction
// and hands compacted batches over through two LDS buffers; the consumer only evaluates and
// blends quads.  Hand-over protocol
// (all in LDS, workgroup-scope acquire/release, no workgroup barrier after the start):
//   flag[b] == 0          buffer b is free (consumer -> producer)
//   flag[b] == cnt + 1    buffer b holds cnt compacted survivors (producer -> consumer); + PC_SAFE when none
//                         of them can produce power > 0 (blend_math.h splat_power_never_positive)
//   flag[b] == PC_DONE    end of the list
//   stop                  the consumer saturated all its pixels: the producer may quit
//   box[4]                bounding box of the still-live pixels (consumer -> producer).  The
//                         producer may read a stale or half-updated box: boxes only shrink, so any
//                         mix of old and new bounds is a superset of the current box (the cull
//                         stays conservative, results are unchanged).
// Same arithmetic in the same order as blend_heavy: bit-identical images.
// Spin loops are bounded (PC_SPIN_LIMIT): a protocol error ends the wave instead of hanging -- and
// is REPORTED: the wave raises pc_timeout in the image blob's header and the calling thread's sticky
// error word (pinned host memory), and the next entry point that thread calls fails with
// GRPG_ERR_HIP (api.hip check_async_error; debug = true: the forward itself fails).  The pixels of
// that quarter are wrong, nothing else is.  -DGRPG_PC_SPIN_LIMIT=0 builds the variant the timeout
// test forces the path with (tests/test_gpu_pc_timeout.py).
// ------------------------------------------------------------------------------------------
constexpr uint32_t PC_DONE = 0xFFFFFFFFu;
#ifndef GRPG_PC_SPIN_LIMIT
#define GRPG_PC_SPIN_LIMIT (1u << 24)
#endif
constexpr uint32_t PC_SPIN_LIMIT = GRPG_PC_SPIN_LIMIT;
constexpr uint32_t PC_SAFE = 0x100u;   // above cnt + 1 <= 65

__device__ __forceinline__ void pc_fail(const PCErr err, const int lane) {
  if (lane == 0) {
    if (err.header_word) *err.header_word = 1u;
    if (err.host_word) { *err.host_word = 1u; __threadfence_system(); }
  }
}

struct PCCtrl {
  uint32_t flag[2];
  uint32_t stop;
  uint32_t pad;
  float box[4];
};

__device__ __forceinline__ uint32_t pc_load(const uint32_t* p) {
  return (uint32_t)__builtin_amdgcn_readfirstlane(
      (int)__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
}
__device__ __forceinline__ void pc_store(uint32_t* p, const uint32_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// waits until ready() (false: the consumer stopped, or the hand-over timed out -- reported, see above)
template <class Cond>
__device__ __forceinline__ bool pc_wait(PCCtrl* __restrict__ ctl, const PCErr err, const int lane, const bool poll_stop,
                                        Cond ready, WaveTrace* tr = nullptr /* experiment build: cycles waited */) {
  if (ready()) return true;
  const uint64_t t0 = tr ? __builtin_readcyclecounter() : 0;
  uint32_t spins = 0;
  bool ok = true;
  while (!ready()) {
    if (poll_stop && pc_load(&ctl->stop) != 0u) { ok = false; break; }
    if (++spins > PC_SPIN_LIMIT) { pc_fail(err, lane); ok = false; break; }
    __builtin_amdgcn_s_sleep(1);
  }
  if (tr) { tr->t_stage += (uint32_t)(__builtin_readcyclecounter() - t0); tr->blends++; }
  return ok;
}

// Round 5.  With nothing else on the chip the 96 class-0 tiles of the bench frame take 0.197 ms
// (experiment build `only0pair`): that chain, not throughput, set the render launch's length.  Per-wave
// traces of a three-stage split of the consumer (evaluator / blender waves, built and measured: no gain,
// twice the wave slots -- DESIGN_EXPERIMENTS.md) showed BOTH halves of the pair within 10 % of each other:
//   producer  1.9 us per batch for 0.7 us of work -- one list window and one record gather in flight
//             per iteration, each a full HBM / Infinity Cache round trip;
//   consumer  ~650 cycles per quad: three exposed LDS round trips per quad, and per batch the live box
//             by 24 ds_bpermute round trips, two polls and a batch of only ~26 survivors to amortise them.
// Hence: the list arrives in 2048-entry windows (ListStream), the producer gathers TWO batches' records
// per iteration and publishes them as ONE batch when their survivors fit a buffer, the consumer keeps a
// quad's geometry one quad ahead in registers and requests its colours before the accept arithmetic,
// and the live box comes from the lane mask's bits (mask_box).
template <bool WITH_SEM = false>
__device__ __forceinline__ void pc_producer(float4* __restrict__ buf0, float4* __restrict__ buf1,
                                            uint32_t* __restrict__ qid, uint32_t* __restrict__ qpos,
                                            PCCtrl* __restrict__ ctl, const int lane,
                                            const int quarter, const uint32_t r_begin,
                                            const uint32_t r_end,
                                            const uint32_t* __restrict__ point_list,
                                            const RecView rec, const PCErr err, WaveTrace* tr = nullptr,
                                            const SemSrc sem = SemSrc{nullptr, 0, nullptr},
                                            float* __restrict__ semrows0 = nullptr /* LDS rows of buf0 / buf1 */,
                                            float* __restrict__ semrows1 = nullptr, const ClassFilter cf = ClassFilter{}) {
  const uint32_t bit = 1u << (SUBTILE_SHIFT + quarter);
  const uint64_t lt = lanemask_lt();
  uint32_t head = 0, count = 0;
  ListStream<8> ls;
  ls.open(point_list, r_begin, r_end, lane);
  float4 a0 = make_float4(0, 0, 0, 0), b0 = a0, c0 = a0, a1 = a0, b1 = a0, c1 = a0;
  uint32_t pos0 = 0, id0 = 0, pos1 = 0, id1 = 0, ncur = 0;
  int cur = 0;
  // the survivors `keep` of one gathered half, compacted behind `base` entries already in buffer `my`
  const auto put = [&](float4* __restrict__ my, const bool keep, const uint64_t mask, const int base,
                       const float4 a, const float4 b, const float4 c, const uint32_t pos, const uint32_t idc) {
    if (keep) {
      store_pair_half(my, base + (int)__popcll(mask & lt), a.x, a.y, splat_q(b.x, b.y, b.z), a.w,
                      make_float4(b.w, c.x, c.y, a.z), pos);
      if (WITH_SEM) sem_stage(my == buf0 ? semrows0 : semrows1, base + (int)__popcll(mask & lt), sem.semantics, sem.S, idc, true);
    }
  };
  // neutral pads up to a multiple of 4 (opacity 0), then the hand-over of cnt survivors in buffer `cur`
  const auto publish = [&](float4* __restrict__ my, const int cnt, const bool safe) {
    if (lane < ((4 - (cnt & 3)) & 3)) {
      const SplatQ zq = {0.f, 0.f, 0.f};
      store_pair_half(my, cnt + lane, 0.f, 0.f, zq, 0.f, make_float4(0.f, 0.f, 0.f, 0.f), 0u);
      if (WITH_SEM) sem_stage(my == buf0 ? semrows0 : semrows1, cnt + lane, sem.semantics, sem.S, 0u, false);
    }
    if (tr) { tr->batches++; tr->survivors += (uint32_t)cnt; }
    pc_store(&ctl->flag[cur], ((uint32_t)cnt + 1u) | (safe ? PC_SAFE : 0u));
    cur ^= 1;
  };
  const auto buffer_free = [&]() { return pc_load(&ctl->flag[cur]) == 0u; };
  for (;;) {
    if (pc_load(&ctl->stop) != 0u) return;
    // ---- FILL: until two batches are queued (ring: < 128 + 256 entries <= QCAP) ----
    while (count < 2u * WAVE && !ls.exhausted()) ls.fill(qid, qpos, bit, head, count, lane, lt, cf.wmask, cf.wval);
    // the ring entries written by FILL are read by OTHER lanes in POP: keep the compiler from
    // reordering the LDS accesses across this point (costs no instruction)
    __builtin_amdgcn_wave_barrier();
    // ---- POP: up to 128 entries, two per lane; their records are requested ----
    const uint32_t nn = min(count, 2u * WAVE);
    float4 a0n = make_float4(0, 0, 0, 0), b0n = a0n, c0n = a0n, a1n = a0n, b1n = a0n, c1n = a0n;
    uint32_t pos0n = 0, id0n = 0, pos1n = 0, id1n = 0;
    if ((uint32_t)lane < nn) {
      const uint32_t slot = (head + lane) & (QCAP - 1);
      id0n = qid[slot] & cf.id_and;
      pos0n = qpos[slot];
      rec.load(id0n, a0n, b0n, c0n);
    }
    if ((uint32_t)lane + WAVE < nn) {
      const uint32_t slot = (head + WAVE + lane) & (QCAP - 1);
      id1n = qid[slot] & cf.id_and;
      pos1n = qpos[slot];
      rec.load(id1n, a1n, b1n, c1n);
    }
    head = (head + nn) & (QCAP - 1);
    count -= nn;
    // ---- the previous iteration's records: cull against the consumer's live box, compact, publish ----
    if (ncur > 0) {
      const float rx0 = ctl->box[0], rx1 = ctl->box[1], ry0 = ctl->box[2], ry1 = ctl->box[3];
      const bool k0 = ((uint32_t)lane < ncur) &&
                      !splat_misses_rect(a0.x, a0.y, b0.x, b0.y, b0.z, a0.w, rx0, rx1, ry0, ry1);
      const bool k1 = ((uint32_t)lane + WAVE < ncur) &&
                      !splat_misses_rect(a1.x, a1.y, b1.x, b1.y, b1.z, a1.w, rx0, rx1, ry0, ry1);
      const uint64_t m0 = __ballot(k0), m1 = __ballot(k1);
      const int n0 = (int)__popcll(m0), n1 = (int)__popcll(m1);
      const bool safe0 = __ballot(k0 && !splat_power_never_positive(splat_q(b0.x, b0.y, b0.z))) == 0ull;
      const bool safe1 = __ballot(k1 && !splat_power_never_positive(splat_q(b1.x, b1.y, b1.z))) == 0ull;
      if (n0 + n1 > 0) {
        if (!pc_wait(ctl, err, lane, true, buffer_free, tr)) return;
        float4* my = cur ? buf1 : buf0;
        if (n0 + n1 <= WAVE) {   // one batch (list order: the first half's survivors first)
          put(my, k0, m0, 0, a0, b0, c0, pos0, id0);
          put(my, k1, m1, n0, a1, b1, c1, pos1, id1);
          publish(my, n0 + n1, safe0 && safe1);
        } else {                 // two batches
          put(my, k0, m0, 0, a0, b0, c0, pos0, id0);
          publish(my, n0, safe0);
          if (!pc_wait(ctl, err, lane, true, buffer_free, tr)) return;
          my = cur ? buf1 : buf0;
          put(my, k1, m1, 0, a1, b1, c1, pos1, id1);
          publish(my, n1, safe1);
        }
      }
    }
    a0 = a0n; b0 = b0n; c0 = c0n; pos0 = pos0n; id0 = id0n;
    a1 = a1n; b1 = b1n; c1 = c1n; pos1 = pos1n; id1 = id1n;
    ncur = nn;
    if (ncur == 0 && ls.exhausted()) break;
  }
  if (!pc_wait(ctl, err, lane, true, buffer_free, tr)) return;   // end-of-list marker
  pc_store(&ctl->flag[cur], PC_DONE);
}

// one handed-over batch: a quad's geometry block waits in registers one quad ahead, its colour / position
// blocks are requested before the accept arithmetic (the wave has its SIMD to itself at the end of the
// launch, and every LDS round trip it waits for -- three per quad until round 5 -- is the launch's)
template <bool AUX, int NSEM, bool SAFE>
__device__ __forceinline__ void pc_blend_batch(WavePix<1>& st, const float4* __restrict__ my, const int cnt,
                                               const float pxf, const float pyf, SemAcc<NSEM>* sa,
                                               const SemSrc semb, const int lane) {
  constexpr bool PP = AUX;
  QuadGeom g_n;
  g_n.load(my);
  for (int j0 = 0; j0 < cnt; j0 += 4) {
    const float4* blk = my + (j0 >> 1) * PAIR_F4;
    const QuadGeom g = g_n;
    QuadColsReg<PP> col;
    col.load(blk);
    g_n.load(my + (min(j0 + 4, cnt - 1) >> 1) * PAIR_F4);   // (the last quad twice: no branch around the loads)
    float alpha[4];
    uint64_t ok[4];
    eval_quad<SAFE>(g, pxf, pyf, alpha, ok);
    blend_quad_tail<AUX, NSEM>(st, col, alpha, ok, sa, semb, j0, lane);
  }
}

template <bool AUX = true, int NSEM = 0, class Out = PlainOut>
__device__ __forceinline__ void pc_consumer(const float4* __restrict__ buf0,
                                            const float4* __restrict__ buf1,
                                            PCCtrl* __restrict__ ctl, const int lane,
                                            const int x0, const int y0, const int W, const int H,
                                            const Out& out, CkptWriter ckw,
                                            const uint32_t len, const PCErr err,
                                            const SemSrc sem = SemSrc{nullptr, 0, nullptr},
                                            float* __restrict__ out_semantic = nullptr, WaveTrace* tr = nullptr,
                                            const float* __restrict__ semrows0 = nullptr,
                                            const float* __restrict__ semrows1 = nullptr,
                                            float* __restrict__ t_lo = nullptr /* LDS: 8 x WAVE floats each, for the */,
                                            float* __restrict__ t_hi = nullptr /* final transpose (this wave's idle ring) */) {
  SemAcc<NSEM> sa;
  sa.clear();
  const int px = x0 + (lane & 15), py = y0 + (lane >> 4);
  const float pxf = (float)px, pyf = (float)py;
  WavePix<1> st;
  st.T[0] = 1.0f; st.CrCg[0] = (v2f){0.f, 0.f}; st.CbD[0] = (v2f){0.f, 0.f};
  st.last[0] = 0;
  st.done[0] = lanes(!(px < W && py < H));
  uint64_t prev_alive = ~0ull;
  int cur = 0;
  if (~st.done[0] == 0ull) pc_store(&ctl->stop, 1u);   // nothing to do (quarter outside the image)
  else for (;;) {
    uint32_t f = 0;
    if (!pc_wait(ctl, err, lane, false, [&]() { f = pc_load(&ctl->flag[cur]); return f != 0u; }, tr)) break;
    if (f == PC_DONE) break;
    const int cnt = (int)((f & (PC_SAFE - 1u)) - 1u);
    if (tr) { tr->batches++; tr->survivors += (uint32_t)cnt; }
    const float4* my = cur ? buf1 : buf0;
    const SemSrc semb = {sem.semantics, sem.S, cur ? semrows1 : semrows0};
    if (f & PC_SAFE) pc_blend_batch<AUX, NSEM, true>(st, my, cnt, pxf, pyf, &sa, semb, lane);
    else pc_blend_batch<AUX, NSEM, false>(st, my, cnt, pxf, pyf, &sa, semb, lane);
    if (AUX && ckw.recs != nullptr) {   // list position of the batch's last survivor (pair layout above)
      const float* blk = reinterpret_cast<const float*>(my + ((cnt - 1) >> 1) * PAIR_F4);
      ckpt_batch_end(ckw, lane, st,
                     (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(blk[20 + ((cnt - 1) & 1)])));
    }
    pc_store(&ctl->flag[cur], 0u);   // hand the buffer back
    cur ^= 1;
    const uint64_t alive = ~st.done[0];
    if (alive == 0ull) { pc_store(&ctl->stop, 1u); break; }
    if (alive != prev_alive) {   // shrink the producer's cull box to the live pixels
      prev_alive = alive;
      const MaskBox b = mask_box(alive);
      if (lane == 0) {
        ctl->box[0] = (float)(x0 + b.c0); ctl->box[1] = (float)(x0 + b.c1);
        ctl->box[2] = (float)(y0 + b.r0); ctl->box[3] = (float)(y0 + b.r1);
      }
    }
  }
  if (AUX) ckpt_finish(ckw, lane, st, len);
  if (px < W && py < H) out.template pixel<AUX>(px, py, st.T[0], st.CrCg[0], st.CbD[0], st.last[0]);
  out.arrive(x0, y0, wave_pixels_inside(px < W && py < H));
  if (NSEM > 0)
    sem_write<NSEM>(sa, sem.S, out_semantic, (size_t)H * W, (size_t)py * W + px, px < W && py < H, lane, t_lo, t_hi);
}

// ------------------------------------------------------------------------------------------
// LAYERED frames (additive: grpg_forward_layers).  The reference's evaluation path renders every frame three
// times -- all models, the background model alone, the object models alone
// (lib/models/street_gaussian_renderer.py:13-40: render_all) -- i.e. three preprocess + binning chains and three
// walks for what is ONE list walk with up to three blend states: a (pixel, splat) pair has one alpha; the
// composition takes every splat, a layer the splats of its class.  A layer's transmittance chain sees alpha = 0
// for the other class (T x 1, C + c x 0: exact no-ops), so each plane carries the very bits the op returns
// for that subset of the scene.
//   * every entry of the point list carries its Gaussian's class in bit 27 (set by the fill from the record, where
//     preprocess parked it: common.h REC_CLASS_BIT; ids < 2^27);
//   * preprocess leaves one flag per tile: does its list hold an OBJECT entry at all (common.h TileObjBits).
//     Where it does not -- most of a street frame -- the background layer's chain IS the composition's and the
//     object layer stays empty: the tile is walked with ONE state by the plain frame's own paths (wave pairs for
//     the longest lists, quarter waves, four-pixel light waves: round 6; round 5 walked every tile with three
//     states on quarter waves, 0.46 ms against the plain frame's 0.21) and LayersOut::pixel derives the planes;
//   * a tile WITH object entries carries three states, one pixel per lane.  Short lists: one quarter wave holds
//     all three (blend_heavy_layers), and even there the states FORK LATE: until the first object entry survives a
//     batch's cull the background layer equals the composition bit for bit and is not computed, the object layer
//     is empty -- a wave whose quarter no object reaches never forks.  Long lists (>= LAYERS_PC_MIN entries): one
//     walk per layer on the plain frame's paths (LayerRoleOut below);
//   * a pixel is finished when all three states are: where no object ever saturates the walk goes on to the
//     end of the list -- but once the composition and the background are through, FILL keeps OBJECT entries
//     only (a dead entry costs 1/256 of a FILL step), and the cull box of the non-object entries is that of
//     the pixels still live in the composition or the background.
// ------------------------------------------------------------------------------------------
constexpr uint32_t LAYER_BIT = POINT_CLASS_BIT;
constexpr uint32_t LAYER_ID_MASK = LAYER_BIT - 1u;
static_assert((ID_MASK & LAYER_BIT) != 0u, "the ring entries (id & ID_MASK) must keep the class bit");

__device__ __forceinline__ void fresh_state(WavePix<1>& s, const uint64_t outside) {
  s.T[0] = 1.0f; s.CrCg[0] = (v2f){0.f, 0.f}; s.CbD[0] = (v2f){0.f, 0.f}; s.last[0] = 0; s.done[0] = outside;
}

// class word of slot `slot` of a compacted batch: the spare words of its pair block's position float4
__device__ __forceinline__ void store_slot_class(float4* __restrict__ my, const int slot, const uint32_t cls) {
  reinterpret_cast<uint32_t*>(my + (slot >> 1) * PAIR_F4)[22 + (slot & 1)] = cls;
}

// one quad into the three states: alpha once, then the blend half per state with its own accept masks
__device__ __forceinline__ void blend_quad_layers(WavePix<1>& sa, WavePix<1>& sb, WavePix<1>& so,
                                                  const float4* __restrict__ my, const int j0,
                                                  const float pxf, const float pyf, const int lane) {
  const float4* blk = my + (j0 >> 1) * PAIR_F4;
  QuadGeom g;
  g.load(blk);
  QuadColsReg<false> cols;
  cols.load(blk);
  const float4 p0 = blk[5], p1 = blk[PAIR_F4 + 5];
  float alpha[4];
  uint64_t ok[4];
  eval_quad(g, pxf, pyf, alpha, ok);   // (the batch-level SAFE form costs this path 18 registers)
  const uint32_t cls[4] = {(uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(p0.z)),
                           (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(p0.w)),
                           (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(p1.z)),
                           (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(p1.w))};
  uint64_t okb[4], oko[4];
  uint64_t anyb = 0ull, anyo = 0ull;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const uint64_t m = cls[i] ? ~0ull : 0ull;   // the slot's class, as a lane mask (wave-uniform)
    okb[i] = ok[i] & ~m; oko[i] = ok[i] & m;
    anyb |= okb[i]; anyo |= oko[i];
  }
  const SemSrc nosem = {nullptr, 0, nullptr};
  SemAcc<0>* nosa = nullptr;
  blend_quad_tail<false, 0>(sa, cols, alpha, ok, nosa, nosem, j0, lane);
#if !(GRPG_LAYERS_ABLATE & 2)   // experiment builds (wrong images): what does a state cost?
  if (anyb & ~sb.done[0]) blend_quad_tail<false, 0>(sb, cols, alpha, okb, nosa, nosem, j0, lane);
#endif
#if !(GRPG_LAYERS_ABLATE & 1)
  if (anyo & ~so.done[0]) blend_quad_tail<false, 0>(so, cols, alpha, oko, nosa, nosem, j0, lane);
#endif
}

// the three states of a quarter wave and what the walk needs to know about them
struct Layers3 {
  WavePix<1> a, b, o;   // composition, background layer, object layer; b and o are only kept once `forked`
  bool forked;          // wave-uniform
  uint64_t outside;
  __device__ __forceinline__ void init(const uint64_t out_m) {
    outside = out_m; forked = false;
    fresh_state(a, out_m); b = a; o = a;
  }
  // an object entry is about to be blended for the first time: up to here the background layer has seen every
  // splat the composition has (same bits), the object layer none
  __device__ __forceinline__ void fork() { if (!forked) { b = a; forked = true; } }
  __device__ __forceinline__ uint64_t live_ab() const { return forked ? ~(a.done[0] & b.done[0]) : ~a.done[0]; }
#if GRPG_LAYERS_ABLATE & 1   // experiment build (wrong images): the object layer never keeps a walk alive
  __device__ __forceinline__ uint64_t live_all() const { return live_ab(); }
#else
  __device__ __forceinline__ uint64_t live_all() const { return forked ? (live_ab() | ~o.done[0]) : ~outside; }
#endif
  template <class Out>
  __device__ __forceinline__ void write(const Out& out, const int px, const int py, const int W, const int H) {
    if (!forked) b = a;
    if (px < W && py < H)
      out.pixel3(px, py, a.T[0], a.CrCg[0], a.CbD[0], b.T[0], b.CrCg[0], b.CbD[0], o.T[0], o.CrCg[0], o.CbD[0]);
  }
};

// quarter wave of a tile WITH object entries (no wave pair: lists below RENDER_PC_MIN)
template <class Out>
__device__ __forceinline__ void blend_heavy_layers(float4* __restrict__ my, uint32_t* __restrict__ qid,
                                                   uint32_t* __restrict__ qpos, const int lane, const int quarter,
                                                   const uint32_t r_begin, const uint32_t r_end, const int x0,
                                                   const int y0, const int W, const int H,
                                                   const uint32_t* __restrict__ point_list, const RecView rec,
                                                   const Out& out) {
  const int px = x0 + (lane & 15), py = y0 + (lane >> 4);
  const float pxf = (float)px;
  const uint32_t bit = 1u << (SUBTILE_SHIFT + quarter);
  const uint64_t lt = lanemask_lt();
  Layers3 L;
  L.init(lanes(!(px < W && py < H)));
  // cull boxes: non-object entries against the pixels live in the composition or the background,
  // object entries against those live in any state
  float ax0 = (float)x0, ax1 = (float)(x0 + 15), ay0 = (float)y0, ay1 = (float)(y0 + 3);
  float ox0 = ax0, ox1 = ax1, oy0 = ay0, oy1 = ay1;
  uint64_t prev_ab = ~0ull, prev_all = ~0ull;

  uint32_t head = 0, count = 0;
  ListStream<4> ls;
  ls.open(point_list, r_begin, r_end, lane);
  float4 a = make_float4(0, 0, 0, 0), b = a, c = a;
  uint32_t pos = 0, idc = 0, ncur = 0;

  for (;;) {
    const uint64_t live_ab = L.live_ab();
    const uint64_t live_all = L.live_all();
    if (live_all == 0ull) break;
    const bool need_ab = live_ab != 0ull;   // wave-uniform: the composition or the background still blends
    if (live_ab != prev_ab && need_ab) {
      prev_ab = live_ab;
      const MaskBox m = mask_box(live_ab);
      ax0 = (float)(x0 + m.c0); ax1 = (float)(x0 + m.c1); ay0 = (float)(y0 + m.r0); ay1 = (float)(y0 + m.r1);
    }
    if (live_all != prev_all) {
      prev_all = live_all;
      const MaskBox m = mask_box(live_all);
      ox0 = (float)(x0 + m.c0); ox1 = (float)(x0 + m.c1); oy0 = (float)(y0 + m.r0); oy1 = (float)(y0 + m.r1);
    }
    // ---- FILL (the ring entry keeps the class bit: ID_MASK covers bit 27) ----
    while (count < (uint32_t)WAVE && !ls.exhausted())
      ls.fill(qid, qpos, bit, head, count, lane, lt, need_ab ? 0u : LAYER_BIT, need_ab ? 0u : LAYER_BIT);
    __builtin_amdgcn_wave_barrier();
    // ---- POP ----
    const uint32_t nn = min(count, (uint32_t)WAVE);
    float4 a_n = make_float4(0, 0, 0, 0), b_n = a_n, c_n = a_n;
    uint32_t pos_n = 0, id_n = 0;
    if ((uint32_t)lane < nn) {
      const uint32_t slot = (head + lane) & (QCAP - 1);
      id_n = qid[slot];
      pos_n = qpos[slot];
      rec.load(id_n & LAYER_ID_MASK, a_n, b_n, c_n);
    }
    head = (head + nn) & (QCAP - 1);
    count -= nn;
    // ---- BLEND the previous batch ----
    if (ncur > 0) {
      const bool is_obj = (idc & LAYER_BIT) != 0u;
      const bool keep = ((uint32_t)lane < ncur) && (need_ab || is_obj) &&
                        !splat_misses_rect(a.x, a.y, b.x, b.y, b.z, a.w, is_obj ? ox0 : ax0, is_obj ? ox1 : ax1,
                                           is_obj ? oy0 : ay0, is_obj ? oy1 : ay1);
      const uint64_t mask = __ballot(keep);
      const int cnt = (int)__popcll(mask);
      if (__ballot(keep && is_obj) != 0ull) L.fork();
      if (keep) {
        const int slot = (int)__popcll(mask & lt);
        store_pair_half(my, slot, a.x, a.y, splat_q(b.x, b.y, b.z), a.w, make_float4(b.w, c.x, c.y, a.z), pos);
        store_slot_class(my, slot, is_obj ? 1u : 0u);
      }
      if (lane < ((4 - (cnt & 3)) & 3)) {   // neutral pads up to a multiple of 4 (opacity 0)
        const SplatQ zq = {0.f, 0.f, 0.f};
        store_pair_half(my, cnt + lane, 0.f, 0.f, zq, 0.f, make_float4(0.f, 0.f, 0.f, 0.f), 0u);
        store_slot_class(my, cnt + lane, 0u);
      }
      __builtin_amdgcn_wave_barrier();
      if (L.forked) {
        for (int j0 = 0; j0 < cnt; j0 += 4) blend_quad_layers(L.a, L.b, L.o, my, j0, pxf, (float)py, lane);
      } else {   // one state: the plain quarter wave's quad
        for (int j0 = 0; j0 < cnt; j0 += 4) blend_quad<false, 0, false>(L.a, my, j0, pxf, (float)py);
      }
      __builtin_amdgcn_wave_barrier();
    }
    a = a_n; b = b_n; c = c_n; pos = pos_n; idc = id_n; ncur = nn;
    if (ncur == 0 && ls.exhausted()) break;
  }
  L.write(out, px, py, W, H);
  out.arrive(x0, y0, wave_pixels_inside(px < W && py < H));
}

// ------------------------------------------------------------------------------------------
// Long tiles WITH object entries (>= LAYERS_PC_MIN entries): ONE WALK PER LAYER, on the plain frame's own paths.
//
// A layer's chain depends on nothing but the list, so a long object tile is not walked with three states at all: the
// composition, the background layer and the object layer each get a walk of their own that takes the entries of its
// class (ClassFilter: all / non-objects / objects) and writes its own planes (LayerRoleOut) -- wave pairs from
// RENDER_PC_MIN entries, four quarter waves below.  Each walk is literally the plain frame's (same functions):
// bit-identical to the op's call on that subset by construction, and as fast as the plain frame's chain -- the
// launch's critical path.  History (DESIGN_EXPERIMENTS.md R6.1): three states on one consumer wave made the class-0
// chain of a ten-actor frame 0.41 ms (plain: 0.19); one producer feeding three consumer waves 0.26 ms (a consumer
// per layer is as cheap as the plain one, but the shared producer serves the slowest and culls for the union);
// independent walks: the background layer's own chain, 0.20 ms.  The object layer's walk scans the whole list
// (FILL drops what is not an object: 1 / 256 of a step per entry) and blends the few object entries.
// ------------------------------------------------------------------------------------------
template <int ROLE, class LOut>
struct LayerRoleOut {   // ROLE 0 composition (epilogue included), 1 background layer, 2 object layer
  const LOut& o;
  template <bool AUX>
  __device__ __forceinline__ void pixel(const int px, const int py, const float T, const v2f CrCg, const v2f CbD,
                                        const uint32_t) const {
    if (ROLE == 0) o.pixel_a(px, py, T, CrCg, CbD);
    else if (ROLE == 1) o.pixel_b(px, py, T, CrCg, CbD);
    else o.pixel_o(px, py, T, CrCg, CbD);
  }
  __device__ __forceinline__ void arrive(const int x0, const int y0, const uint32_t npix) const {
    if (ROLE == 0) o.arrive(x0, y0, npix);
  }
};
__device__ __forceinline__ ClassFilter layer_filter(const int role) {
  ClassFilter cf;
  cf.wmask = role == 0 ? 0u : LAYER_BIT;
  cf.wval = role == 2 ? LAYER_BIT : 0u;
  cf.id_and = LAYER_ID_MASK;
  return cf;
}

// waves per SIMD the register allocator must fit.  4 (128 VGPRs, no spills in the 4-pixel light path,
// 112 KB of LDS per CU so that the other stream's sort workgroups can co-reside) measured slightly
// ahead of 5 (96 VGPRs, 136 B of scratch per lane): 0.327 vs 0.333 ms, 1640 vs 1600 frames/s.
// A frame with semantic planes (NSEM > 0) carries 16 accumulator registers per lane (MFMA) and 4 KB of
// staged rows per wave: 3 waves per SIMD / 3 workgroups per CU.
constexpr int RENDER_MIN_WAVES = 4;

template <bool LAYERS, bool EPI> struct FrameOut { typedef PlainOutT<EPI> type; };
template <bool EPI> struct FrameOut<true, EPI> { typedef LayersOutT<EPI> type; };
template <bool EPI>
__device__ __forceinline__ PlainOutT<EPI> make_out(const PlainOutT<EPI>& p, const LayerOut&, PlainOutT<EPI>*) { return p; }
template <bool EPI>
__device__ __forceinline__ LayersOutT<EPI> make_out(const PlainOutT<EPI>& p, const LayerOut& lo, LayersOutT<EPI>*) {
  return LayersOutT<EPI>{p.bg, p.out_color, p.out_depth, p.out_alpha, p.W, p.H, lo, p.epi};
}

// Measured and removed: XCD-contiguous work assignment (0.248 vs 0.232 ms:
// neighbouring tiles are similarly long, contiguous eighths unbalance the XCDs), occupancy capped
// with unused LDS, one splat per iteration in the light path, persistent waves.
// LAYERS: a layered frame (above) -- the same launch structure, one state where the tile holds no object entry
#ifndef GRPG_LAYERS_MIN_WAVES   // experiment builds: waves per SIMD the layered kernel is allocated for
#define GRPG_LAYERS_MIN_WAVES 4
#endif
// EPI: the frame epilogue (FrameEpi above) behind the blend, evaluation frames only
template <bool WRITE_AUX, int GPI_L, bool TRACE = false, int NSEM = 0, bool LAYERS = false, bool EPI = false>
__global__ void __launch_bounds__(256, NSEM > 0 ? 3 : (LAYERS ? GRPG_LAYERS_MIN_WAVES : RENDER_MIN_WAVES))
render_forward_kernel(const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                      const RecView rec, const int W, const int H, const int gx,
                      const uint32_t T, const uint32_t* __restrict__ work,
                      const float* __restrict__ bg,
                      float* __restrict__ out_color, float* __restrict__ out_depth,
                      float* __restrict__ out_alpha, uint32_t* __restrict__ n_contrib,
                      const uint32_t pc_slots, const CkptArgs ck, const PCErr pc_err,
                      const SemSrc sem, float* __restrict__ out_semantic,
                      uint32_t* __restrict__ trace = nullptr, const int ablate = 0,
                      const LayerOut lo = LayerOut{nullptr, nullptr, nullptr, nullptr, nullptr},
                      const TileObjBits ob = TileObjBits{nullptr, 0}, const FrameEpi epi = FrameEpi{}) {
  static_assert(!(LAYERS || EPI) || (!WRITE_AUX && NSEM == 0 && !TRACE), "layered frames / frame epilogue: evaluation only, no semantic planes");
  typedef typename FrameOut<LAYERS, EPI>::type Out;
  const Out out = make_out(PlainOutT<EPI>{bg, out_color, out_depth, out_alpha, n_contrib, W, H, epi}, lo, (Out*)nullptr);
  __shared__ float4 s_rec[RW_WAVES][WAVE * REC_F4];
  __shared__ uint32_t s_qid[RW_WAVES][QCAP];
  __shared__ uint32_t s_qpos[RW_WAVES][QCAP];
  __shared__ PCCtrl s_ctl[2];
  // staged semantic rows of the batch being blended: per heavy wave, or per batch buffer of the pairs
  __shared__ float s_sem[NSEM > 0 ? RW_WAVES : 1][NSEM > 0 ? WAVE * SEM_ROW : 4];
#ifdef GRPG_RENDER_LDS_PAD   // experiment build: unused LDS that caps the workgroups per CU
  __shared__ uint32_t s_pad[GRPG_RENDER_LDS_PAD / 4];
  if (W < 0) s_pad[threadIdx.x] = (uint32_t)H;   // never true: keeps the array allocated
  if (W < -1) out_color[0] = (float)s_pad[0];
#endif
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  WaveTrace tr = {0, 0, 0, 0, 0, {0, 0, 0, 0}, {0, 0, 0, 0}};
  const uint64_t t_start = TRACE ? wall_clock64() : 0;
  uint32_t tr_tile = 0xFFFFFFFFu, tr_len = 0;
  const uint32_t n0 = work[0], n1 = work[1], n2 = work[2], nlight = work[3];
  const uint32_t* lists = work + NUM_CLASSES;
  if (WRITE_AUX && ck.bin_hdr != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {
    ck.bin_hdr->ckpt_off256 = ck.off256;   // tells the backward where this frame's checkpoints are
    ck.bin_hdr->ckpt_slots = ck.slots;
  }
  // Class 0 holds the few longest tiles (>= RENDER_PC_MIN entries); each is rendered by two
  // workgroups (half tiles) with a producer and a consumer wave per quarter.  The first pc_slots
  // workgroups are reserved for them (upper bound of 2 n0 computed on the host from num_rendered).
  // A layered frame puts SIX workgroups per class-0 tile at the front of the grid -- as many as the device finds
  // (6 n0; the surplus of the host's upper bound idles at the END of the grid, not in front of the other classes):
  //   tile without object entries: two half-tile wave pairs (workgroups 0, 1), as in a plain frame;
  //   with object entries, >= RENDER_PC_MIN: per layer two half-tile wave pairs (workgroup = 2 layer + half);
  //   with object entries, shorter: per layer four quarter waves (workgroups 0 .. 2).
  const uint32_t pc_n = LAYERS ? min(6u * n0, pc_slots) : pc_slots;
  // a drained frame (FrameEpi::host8): the first drain_wgs workgroups carry finished units to the host
  uint32_t bx = blockIdx.x;
  if constexpr (EPI) {
    if (epi.host8 != nullptr) {
      const uint32_t ndw = (uint32_t)epi.drain_wgs;
      if (bx < ndw) {
        drain_units(epi, W, H, bx * RW_WAVES + (uint32_t)wave, ndw * RW_WAVES, lane, pc_err);
        return;
      }
      bx -= ndw;
    }
  }
#ifdef GRPG_RENDER_ONLY_CLASS0   // experiment build: the class-0 tiles alone (their chain, nothing beside it)
  if (bx >= pc_n) return;
#endif
#ifdef GRPG_RENDER_NO_CLASS0     // experiment build: everything BUT the class-0 tiles (the throughput part)
  if (bx < pc_n) return;
#endif
  if (bx < pc_n) {
    const uint32_t pc_ti = LAYERS ? bx / 6u : bx >> 1;
    uint32_t pc_sub = LAYERS ? bx - 6u * pc_ti : bx & 1u;
    if (pc_ti >= n0) return;
    const uint32_t tile = lists[pc_ti];
    const int ty = (int)(tile / (uint32_t)gx), tx = (int)(tile - (uint32_t)ty * (uint32_t)gx);
    const uint2 range = ranges[tile];
    const uint32_t rb = __builtin_amdgcn_readfirstlane(range.x);
    const uint32_t re = __builtin_amdgcn_readfirstlane(range.y);
    int role = -1;   // -1: one state for all planes (no object entry in the list)
    if constexpr (LAYERS) {
      if (ob.tile(tile)) {   // (workgroup-uniform)
        if (re - rb < RENDER_PC_MIN) {   // three workgroups, one per layer: four quarter waves each
          if (pc_sub >= 3u) return;
          const ClassFilter cf = layer_filter((int)pc_sub);
          const CkptWriter nock = ckpt_writer(ck, tile, wave, rb, re);
          const SemSrc nosem = {nullptr, 0, nullptr};
#define LAYER_HEAVY(R)                                                                                                \
          blend_heavy<false, false, 0>(s_rec[wave], s_qid[wave], s_qpos[wave], lane, wave, rb, re, tx * TILE,         \
                                       ty * TILE + wave * 4, W, H, point_list, rec, LayerRoleOut<R, Out>{out}, nullptr, \
                                       nock, nosem, nullptr, nullptr, cf)
          if (pc_sub == 0u) LAYER_HEAVY(0); else if (pc_sub == 1u) LAYER_HEAVY(1); else LAYER_HEAVY(2);
#undef LAYER_HEAVY
          return;
        }
        role = (int)(pc_sub >> 1);
        pc_sub &= 1u;
      } else if (pc_sub >= 2u) {
        return;
      }
    }
    const int slot = wave & 1;                                   // quarter slot inside this workgroup
    const int q = (int)pc_sub * 2 + slot;                        // quarter of the tile
    const int x0 = tx * TILE, y0 = ty * TILE + q * 4;
    if (wave < 2 && lane == 0) {   // the consumer initialises its quarter's control block
      s_ctl[slot].flag[0] = 0u; s_ctl[slot].flag[1] = 0u; s_ctl[slot].stop = 0u; s_ctl[slot].pad = 0u;
      s_ctl[slot].box[0] = (float)x0; s_ctl[slot].box[1] = (float)(x0 + 15);
      s_ctl[slot].box[2] = (float)y0; s_ctl[slot].box[3] = (float)(y0 + 3);
    }
    __syncthreads();   // the only workgroup barrier: all 4 waves of the workgroup take this branch
    WaveTrace* const trp = TRACE ? &tr : nullptr;
    if constexpr (LAYERS) {
      if (role >= 0) {   // a layer's own wave pairs on a tile with object entries
        const SemSrc nosem = {nullptr, 0, nullptr};
        if (wave < 2) {
          const CkptWriter nock = ckpt_writer(ck, tile, q, rb, re);
#define LAYER_CONSUMER(R)                                                                                          \
          pc_consumer<false, 0>(s_rec[slot], s_rec[slot + 2], &s_ctl[slot], lane, x0, y0, W, H,                     \
                                LayerRoleOut<R, Out>{out}, nock, re - rb, pc_err)
          if (role == 0) LAYER_CONSUMER(0); else if (role == 1) LAYER_CONSUMER(1); else LAYER_CONSUMER(2);
#undef LAYER_CONSUMER
        } else {
          pc_producer<false>(s_rec[slot], s_rec[slot + 2], s_qid[wave], s_qpos[wave], &s_ctl[slot], lane, q, rb, re,
                             point_list, rec, pc_err, nullptr, nosem, nullptr, nullptr, layer_filter(role));
        }
        return;
      }
    }
    if (wave < 2) {
      pc_consumer<WRITE_AUX, NSEM>(s_rec[slot], s_rec[slot + 2], &s_ctl[slot], lane, x0, y0, W, H, out,
                  ckpt_writer(ck, tile, q, rb, re), re - rb, pc_err, sem,
                  out_semantic, trp, s_sem[NSEM > 0 ? slot : 0], s_sem[NSEM > 0 ? slot + 2 : 0],
                  reinterpret_cast<float*>(s_qid[wave]), reinterpret_cast<float*>(s_qpos[wave]));
      if (WRITE_AUX && q == 0) ckpt_publish_items(ck, lane, tile, re - rb);
    } else
      pc_producer<(NSEM > 0)>(s_rec[slot], s_rec[slot + 2], s_qid[wave], s_qpos[wave], &s_ctl[slot], lane, q,
                  rb, re, point_list, rec, pc_err, trp, sem, s_sem[NSEM > 0 ? slot : 0], s_sem[NSEM > 0 ? slot + 2 : 0]);
    if (TRACE && lane == 0) {   // class-0 roles: waves 0, 1 consumers, 2, 3 producers
      // [0] tile | 0x40000000  [1] list length  [2] batches  [3] survivors  [4] waits
      // [5] wave life (wall clock ticks)  [6] start  [7] wave | waiting cycles / 256 << 4
      uint32_t* o = trace + ((size_t)blockIdx.x * RW_WAVES + wave) * 8;
      o[0] = tile | 0x40000000u; o[1] = re - rb; o[2] = tr.batches; o[3] = tr.survivors; o[4] = tr.blends;
      o[5] = (uint32_t)(wall_clock64() - t_start); o[6] = (uint32_t)(t_start & 0xFFFFFFFFu);
      o[7] = (uint32_t)wave | ((tr.t_stage >> 8) << 4);
    }
    return;
  }
  // Dispatch order (longest processing time first): class 0, class 1, the LIGHT tiles, class 2.
  // A light tile is a whole tile on one wave (measured 40-70 us for 100-250 entries), longer than
  // a quarter wave of a class-2 tile (20-30 us), so it must not come last.
  const uint32_t nlwg = (nlight + RW_WAVES - 1) / RW_WAVES;
  const uint32_t b0 = bx - pc_n;
  const bool is_heavy = b0 < n1 || b0 >= n1 + nlwg;
  const uint32_t b = b0 < n1 ? b0 : (is_heavy ? b0 - nlwg : b0 - n1);   // index in its kind
  if (is_heavy) {
    if (b >= n1 + n2) return;
    // heavy tile: four independent 16x4 sub-tiles, 1 pixel per lane, 4 splats per iteration
    const uint32_t tile = b < n1 ? lists[T + b] : lists[2 * T + (b - n1)];
    const int ty = (int)(tile / (uint32_t)gx), tx = (int)(tile - (uint32_t)ty * (uint32_t)gx);
    const uint2 range = ranges[tile];
    const uint32_t rb = __builtin_amdgcn_readfirstlane(range.x);
    const uint32_t re = __builtin_amdgcn_readfirstlane(range.y);
    tr_tile = tile; tr_len = re - rb;
    if constexpr (LAYERS) {
      if (ob.tile(tile)) {   // object entries in the list: three states (forked late)
        blend_heavy_layers(s_rec[wave], s_qid[wave], s_qpos[wave], lane, wave, rb, re, tx * TILE, ty * TILE + wave * 4,
                           W, H, point_list, rec, out);
        return;
      }
    }
    blend_heavy<TRACE, WRITE_AUX, NSEM>(s_rec[wave], s_qid[wave], s_qpos[wave], lane, wave, rb, re, tx * TILE,
                       ty * TILE + wave * 4, W, H, point_list, rec, out, &tr, ckpt_writer(ck, tile, wave, rb, re),
                       sem, out_semantic, s_sem[NSEM > 0 ? wave : 0]);
    if (WRITE_AUX && wave == 0) ckpt_publish_items(ck, lane, tile, re - rb);
  } else {
    if (b >= nlwg) return;
    const uint32_t li = b * RW_WAVES + (uint32_t)wave;
    if (li >= nlight) return;   // whole wave exits together; no workgroup barriers are used
    const uint32_t tile = lists[3 * (size_t)T + li];
    const int ty = (int)(tile / (uint32_t)gx), tx = (int)(tile - (uint32_t)ty * (uint32_t)gx);
    const uint2 range = ranges[tile];
    const uint32_t rb = __builtin_amdgcn_readfirstlane(range.x);
    const uint32_t re = __builtin_amdgcn_readfirstlane(range.y);
    tr_tile = tile | 0x80000000u; tr_len = re - rb;
    // (a layered frame: light tiles hold no object entry -- the tile scan sends the others down the heavy path)
    blend_rect<4, GPI_L, WRITE_AUX, TRACE>(s_rec[wave], lane, rb, re, tx * TILE, ty * TILE, W, H,
                                           point_list, rec, out, &tr, ablate);
    if (NSEM > 0) {
      // a semantic frame sends every non-empty tile down the heavy path (tile_classes): what arrives
      // here holds no splat at all -- its semantic planes are zero (they get no background)
      const int px = tx * TILE + (lane & 15), py0 = ty * TILE + (lane >> 4) * 4;
      const size_t HW = (size_t)H * W;
      for (int k = 0; k < 4; k++)
        if (px < W && py0 + k < H)
          for (int c = 0; c < sem.S && c < NSEM; c++) out_semantic[(size_t)c * HW + (size_t)(py0 + k) * W + px] = 0.f;
    }
  }
  if (TRACE && lane == 0) {   // per-wave trace record (experiment build -DGRPG_TRACE)
    const uint64_t t_end = wall_clock64();
    uint32_t* o = trace + ((size_t)blockIdx.x * RW_WAVES + wave) * 8;
    o[0] = tr_tile; o[1] = tr_len; o[2] = tr.batches; o[3] = tr.survivors; o[4] = tr.blends;
    o[5] = (uint32_t)(t_end - t_start); o[6] = (uint32_t)(t_start & 0xFFFFFFFFu);
    o[7] = (uint32_t)wave | ((tr.t_stage >> 8) << 4);   // stage cycles / 256 in the upper bits
    o[2] = tr.batches | 0u; o[4] = tr.blends;
    trace[(size_t)gridDim.x * RW_WAVES * 8 + ((size_t)blockIdx.x * RW_WAVES + wave)] = tr.t_loop;
    uint32_t* o2 = trace + (size_t)gridDim.x * RW_WAVES * 9 + ((size_t)blockIdx.x * RW_WAVES + wave) * 8;
    for (int i = 0; i < 4; i++) { o2[i] = tr.it[i]; o2[4 + i] = tr.cyc[i]; }
  }
}

static FrameEpi make_frame_epi(const FrameEpilogue* e, const int W = 0, const bool may_drain = false) {
  FrameEpi d{};
  if (e == nullptr) return d;
  d.sky.cube = e->sky_cube; d.sky.res = e->sky_res; d.sky.fill = e->sky_fill; d.sky.clamp_out = e->clamp;
  d.sky.m_dev = e->ray_m_dev; d.sky.mask = nullptr; d.sky.jitter = nullptr;
  for (int i = 0; i < 9; i++) d.sky.m[i] = e->ray_m_dev ? 0.f : e->ray_m[i];
  d.clamp = e->clamp; d.rgb8 = e->rgb8; d.bias = e->truncate ? 0.0f : 0.5f; d.planes = e->planes;
  d.host = e->rgb8_host;
  if (may_drain && e->rgb8_host && e->drain_stage != nullptr && e->drain_cnt != nullptr && e->drain_wgs > 0 &&
      (W & 15) == 0 && ((uintptr_t)e->rgb8 & 15) == 0 && ((uintptr_t)e->drain_stage & 15) == 0) {
    d.host8 = e->rgb8; d.rgb8 = e->drain_stage; d.host = 2;
    d.unit_cnt = e->drain_cnt; d.drain_wgs = e->drain_wgs; d.units_x = (W + 63) / 64;
  }
  return d;
}

// a frame without Gaussians through the same epilogue: blended colour 0 (NOT the background: the reference's
// pre-zeroed planes stay zero when nothing is launched, rasterize_points.cu:85-86,123), T = 1
__global__ void __launch_bounds__(256)
frame_epilogue_empty_kernel(const int W, const int H, const FrameEpi epi, float* __restrict__ out_color) {
  const int px = blockIdx.x * 64 + (threadIdx.x & 63), py = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (px >= W || py >= H) return;
  float rgb[3] = {0.f, 0.f, 0.f};
  frame_finish<true>(epi, px, py, W, H, 1.0f, rgb);
  if (epi.planes && out_color != nullptr) {
    const size_t HW = (size_t)H * W, pix = (size_t)py * W + px;
    out_color[pix] = rgb[0]; out_color[HW + pix] = rgb[1]; out_color[2 * HW + pix] = rgb[2];
  }
}
void launch_frame_epilogue_empty(hipStream_t s, int W, int H, const FrameEpilogue& epi, float* out_color) {
  const dim3 grid((W + 63) / 64, (H + 3) / 4);
  frame_epilogue_empty_kernel<<<grid, 256, 0, s>>>(W, H, make_frame_epi(&epi), out_color);
}

__global__ void __launch_bounds__(256)
fill_layer_planes_kernel(const size_t N, const float* __restrict__ layer_background, float* __restrict__ color_bg,
                         float* __restrict__ alpha_bg, float* __restrict__ color_obj, float* __restrict__ alpha_obj) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (size_t)gridDim.x * 256) {
    for (int c = 0; c < 3; c++) { color_bg[c * N + i] = layer_background[c]; color_obj[c * N + i] = layer_background[c]; }
    alpha_bg[i] = 0.f; alpha_obj[i] = 0.f;
  }
}
void launch_fill_layer_planes(hipStream_t s, size_t N, const float* layer_background, float* color_bg, float* alpha_bg,
                              float* color_obj, float* alpha_obj) {
  if (N == 0) return;
  const size_t nb = (N + 255) / 256;
  fill_layer_planes_kernel<<<(unsigned)(nb < 2048 ? nb : 2048), 256, 0, s>>>(N, layer_background, color_bg, alpha_bg,
                                                                               color_obj, alpha_obj);
}

uint32_t render_pc_slots(uint32_t R);

void launch_render_layers(hipStream_t s, const uint2* ranges, uint32_t* point_list, const RecView rec, int W, int H,
                          int gx, int gy, const float* bg, float* out_color, float* out_depth, float* out_alpha,
                          uint32_t* work, const TileClasses cls, uint32_t cap, bool classified,
                          const float* layer_background, float* out_color_bg,
                          float* out_alpha_bg, float* out_color_obj, float* out_alpha_obj, const PCErr pc_err,
                          const FrameEpilogue* epi) {
  const int ntiles = gx * gy;
  if (ntiles <= 0) return;
  // (the class of every list entry is already there: bit 27, from the record -- hier_binning.hip / binning.hip)
  if (!classified)
    classify_tiles_kernel<<<(ntiles + 255) / 256, 256, 0, s>>>((uint32_t)ntiles, ranges, cls, work);
  const LayerOut lo = {layer_background, out_color_bg, out_alpha_bg, out_color_obj, out_alpha_obj};
  // six workgroups per class-0 tile; in a layered frame a tile with object entries is class 0 from
  // cls.c0_obj_min entries (common.h tile_class_of): at most cap / c0_obj_min tiles.  An upper bound of the grid
  // only: the kernel maps by the device's count and the surplus exits at the end of the grid.
  const uint32_t pc_slots = 6u * (uint32_t)((size_t)cap / (cls.c0_obj_min < cls.c0_min ? cls.c0_obj_min : cls.c0_min) + 1);
  const CkptArgs ck = CkptArgs{nullptr, nullptr, nullptr, nullptr, 0u, 0u};
  const SemSrc sem = {nullptr, 0, nullptr};
  if (epi) {
    const FrameEpi fe = make_frame_epi(epi, W, true);
    render_forward_kernel<false, 2, false, 0, true, true><<<ntiles + pc_slots + fe.drain_wgs, 256, 0, s>>>(
        ranges, point_list, rec, W, H, gx, (uint32_t)ntiles, work, bg, out_color, out_depth, out_alpha, nullptr,
        pc_slots, ck, pc_err, sem, nullptr, nullptr, 0, lo, cls.obj, fe);
  } else
    render_forward_kernel<false, 2, false, 0, true><<<ntiles + pc_slots, 256, 0, s>>>(
        ranges, point_list, rec, W, H, gx, (uint32_t)ntiles, work, bg, out_color, out_depth, out_alpha, nullptr,
        pc_slots, ck, pc_err, sem, nullptr, nullptr, 0, lo, cls.obj);
}

// N-channel "semantic" planes (forward.cu:442-444): same traversal and the same accept/reject
// arithmetic (blend_math.h), NCH channels per launch accumulated in registers instead of the
// reference's per-contribution global read-modify-write.  Non-default path (use_semantic=False
// in every shipped config), kept simple: one wave per tile, 4 pixels per lane.
template <int NCH>
__global__ void __launch_bounds__(256)
render_semantic_kernel(const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                       const RecView rec, const float* __restrict__ semantics,
                       const int S, const int c0, const int W, const int H, const int gx,
                       const int ntiles, float* __restrict__ out_semantic) {
  __shared__ float4 s_rec[RW_WAVES][WAVE * 2];
  __shared__ float s_sem[RW_WAVES][WAVE * NCH];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tile = blockIdx.x * RW_WAVES + wave;
  if (tile >= ntiles) return;
  const int ty = tile / gx, tx = tile - ty * gx;
  const int px = tx * TILE + (lane & 15);
  const int py0 = ty * TILE + (lane >> 4) * 4;
  const float pxf = (float)px;
  const uint2 range = ranges[tile];
  float T[4], acc[4][NCH];
  bool done[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    T[k] = 1.0f;
    done[k] = !(px < W && (py0 + k) < H);
#pragma unroll
    for (int c = 0; c < NCH; c++) acc[k][c] = 0.f;
  }
  float4* my = s_rec[wave];
  float* mysem = s_sem[wave];
  for (uint32_t base = range.x; base < range.y; base += WAVE) {
    const bool alldone = done[0] && done[1] && done[2] && done[3];
    if (__ballot(!alldone) == 0ull) break;
    const uint32_t n = min((uint32_t)WAVE, range.y - base);
    if ((uint32_t)lane < n) {
      const uint32_t id = point_list[base + lane] & ID_MASK;
      float4 ra, rb, rc;
      rec.load(id, ra, rb, rc);
      my[lane * 2 + 0] = ra;
      my[lane * 2 + 1] = rb;
#pragma unroll
      for (int c = 0; c < NCH; c++)
        mysem[lane * NCH + c] = (c0 + c < S) ? semantics[(size_t)id * S + c0 + c] : 0.f;
    }
    __builtin_amdgcn_wave_barrier();
    for (uint32_t j = 0; j < n; j++) {
      const float4 a = my[j * 2 + 0];
      const float4 b = my[j * 2 + 1];
      float sv[NCH];
#pragma unroll
      for (int c = 0; c < NCH; c++) sv[c] = mysem[j * NCH + c];
      const SplatTerms st = splat_terms(a.x - pxf, b.x, b.y, b.z);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const float dy = a.y - (float)(py0 + k);
        float G, alpha;
        bool valid = pair_alpha(pair_power(st, dy), a.w, G, alpha) && !done[k];
        const float test_T = T[k] * (1.0f - alpha);
        const bool term = valid && (test_T < 0.0001f);
        done[k] = done[k] || term;
        valid = valid && !term;
        const float w = valid ? alpha * T[k] : 0.0f;
#pragma unroll
        for (int c = 0; c < NCH; c++) acc[k][c] = fmaf(sv[c], w, acc[k][c]);
        T[k] = valid ? test_T : T[k];
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  const size_t HW = (size_t)H * W;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int py = py0 + k;
    if (px < W && py < H) {
      const size_t pix = (size_t)py * W + px;
#pragma unroll
      for (int c = 0; c < NCH; c++)
        if (c0 + c < S) out_semantic[(size_t)(c0 + c) * HW + pix] = acc[k][c];
    }
  }
}

// workgroups reserved for half tiles of class 0 (producer / consumer wave pairs): at most
// R / RENDER_PC_MIN tiles can be that long, two workgroups each.  The forward and the backward of one
// frame agree on it (both derive it from num_rendered).
uint32_t render_pc_slots(uint32_t R) { return 2u * (uint32_t)((size_t)R / RENDER_PC_MIN + 1); }

void launch_render_forward(hipStream_t s, const uint2* ranges, const uint32_t* point_list,
                           const RecView rec, int W, int H, int gx, int gy, const float* bg,
                           float* out_color, float* out_depth, float* out_alpha,
                           uint32_t* n_contrib, uint32_t* work /* [4 + 4T] scratch */,
                           const TileClasses cls, uint32_t R, bool aux, bool classified,
                           const CkptArgs* ckp, const PCErr pc_err, const float* semantics, int S,
                           float* out_semantic, const FrameEpilogue* epi) {
  const int ntiles = gx * gy;
  if (ntiles <= 0) return;
  const CkptArgs ck = (aux && ckp) ? *ckp : CkptArgs{nullptr, nullptr, nullptr, nullptr, 0u, 0u};
  // work[0..3] were zeroed by frame_init_kernel (same stream, earlier in the frame)
  const uint32_t pc_slots = render_pc_slots(R);
  // classified: the hierarchical binning's tile scan has already built the work lists
  if (!classified)
    classify_tiles_kernel<<<(ntiles + 255) / 256, 256, 0, s>>>((uint32_t)ntiles, ranges, cls, work);
  const SemSrc sem = {semantics, S, nullptr};
  // nheavy + ceil(nlight/4) <= ntiles: launch the upper bound, surplus workgroups exit at once
  // aux == false (no backward will follow): n_contrib is neither tracked nor written
#define RF_ARGS ranges, point_list, rec, W, H, gx, (uint32_t)ntiles, work, bg, out_color, out_depth, \
                out_alpha, n_contrib, pc_slots, ck, pc_err, sem, out_semantic
#ifdef GRPG_TRACE   // experiment build: per-wave cycle counts and survivor statistics to a file
  static const char* trace_path = getenv("GRPG_RENDER_TRACE");
  if (trace_path && S == 0) {
    uint32_t* d_trace = nullptr;
    const size_t words = (size_t)(ntiles + pc_slots) * RW_WAVES * 17;
    if (hipMalloc((void**)&d_trace, words * 4) == hipSuccess) {
      (void)hipMemsetAsync(d_trace, 0xFF, words * 4, s);
      render_forward_kernel<true, 1, true><<<ntiles + pc_slots, 256, 0, s>>>(
          RF_ARGS, d_trace, getenv("GRPG_RENDER_ABLATE") ? atoi(getenv("GRPG_RENDER_ABLATE")) : 0);
      std::vector<uint32_t> h(words);
      (void)hipMemcpyAsync(h.data(), d_trace, words * 4, hipMemcpyDeviceToHost, s);
      (void)hipStreamSynchronize(s);
      if (FILE* f = fopen(trace_path, "wb")) { fwrite(h.data(), 4, words, f); fclose(f); }
      (void)hipFree(d_trace);
      return;
    }
  }
#endif
  // the light (4 pixels per lane) path evaluates two splats per inner iteration (one: measured
  // slower); the heavy path always evaluates quads
  if (epi != nullptr && !aux && S == 0) {   // evaluation frame with the callers' epilogue fused in (grpg_forward_frame)
    const FrameEpi fe = make_frame_epi(epi, W, true);
    render_forward_kernel<false, 2, false, 0, false, true><<<ntiles + pc_slots + fe.drain_wgs, 256, 0, s>>>(
        RF_ARGS, nullptr, 0, LayerOut{nullptr, nullptr, nullptr, nullptr, nullptr}, TileObjBits{nullptr, 0}, fe);
  } else if (S > 0) {
    // semantic planes ride in the heavy path: up to RENDER_NSEM channels in this launch, the rest
    // (S > RENDER_NSEM) in the stand-alone kernel below
    if (aux) render_forward_kernel<true, 2, false, RENDER_NSEM><<<ntiles + pc_slots, 256, 0, s>>>(RF_ARGS);
    else render_forward_kernel<false, 2, false, RENDER_NSEM><<<ntiles + pc_slots, 256, 0, s>>>(RF_ARGS);
  } else {
    if (aux) render_forward_kernel<true, 2><<<ntiles + pc_slots, 256, 0, s>>>(RF_ARGS);
    else render_forward_kernel<false, 2><<<ntiles + pc_slots, 256, 0, s>>>(RF_ARGS);
  }
#undef RF_ARGS
}

// channels [c_begin, S) of the semantic planes by the stand-alone kernel (the first RENDER_NSEM ride
// in the main render launch)
void launch_render_semantic(hipStream_t s, const uint2* ranges, const uint32_t* point_list,
                            const RecView rec, const float* semantics, int S, int c_begin, int W, int H,
                            int gx, int gy, float* out_semantic) {
  const int ntiles = gx * gy;
  if (ntiles <= 0 || S <= c_begin) return;
  const int blocks = (ntiles + RW_WAVES - 1) / RW_WAVES;
  constexpr int NCH = 4;
  for (int c0 = c_begin; c0 < S; c0 += NCH)
    render_semantic_kernel<NCH><<<blocks, 256, 0, s>>>(ranges, point_list, rec, semantics, S, c0,
                                                       W, H, gx, ntiles, out_semantic);
}

}  // namespace grpg
