// Backward of the per-tile blend for gfx950.
//
// Replaces renderCUDA<3,20> backward (cuda_rasterizer/backward.cu:415-641): back-to-front
// re-traversal of each tile's list starting from n_contrib, T recovered by division (:547),
// gradients w.r.t. colour / depth / semantic / accumulated alpha / background (:556-614) pushed to
// (mean2D, conic, opacity, colour, depth, semantic) of every contributing Gaussian (:618-638).
//
// CDNA4 mapping (same decomposition as the forward, render_fwd.hip, whose tile work lists in the
// image blob are reused):
//   * Tiles with fewer than 2048 entries: two half-tile waves at 2 pixels per lane.  Longer ones:
//     four 16x4 quarter-tile waves at 1 pixel per lane (their chains are the critical path).
//   * LONG tiles (>= CK_LONG_MIN entries) are not one serial chain: the training forward left blend
//     checkpoints behind (common.h CK_*: per quarter the state (T, C, D) at batch ends >= CK_SEG
//     positions apart), and the first workgroups of the launch walk (tile, segment) items that start
//     from them: T at the segment's end from the checkpoint, the accumulators from
//     (C_final - C_k) / T_k.  render_backward_kernel 744 -> 505 us at config 5.
//   * The list is walked back to front, only up to the deepest contributor of the wave's pixels,
//     through the forward's FILL / POP / process pipeline: FILL scans 128 entries per step and
//     keeps those whose sub-tile mask (set by emit) concerns the wave in an LDS ring; POP starts
//     the record gather of up to 64 of them; the previous batch is meanwhile culled against the
//     bounding box of the pixels that are ACTIVE at that depth (last contributor behind the
//     batch), compacted deepest-first into LDS and processed.  No record is loaded for an entry
//     the mask rules out.
//   * The reference issues 11+S float atomics per (pixel, Gaussian) pair.  Here the 11 partial
//     gradients of a splat are summed over the lane's pixels in registers and reduced across the
//     wave together: two halving exchanges (v_permlane32_swap / v_permlane16_swap: two
//     quantities per swap + add) leave three quantities per 16-lane row, four DPP adds finish the
//     row sums -- 30 VALU instructions for all of them -- and lane 16 r + k of row r then owns
//     quantity 3 r + k, so ONE global_atomic_add_f32 with 11 active lanes (5 cache lines) updates
//     the Gaussian: 64-256x fewer atomic operations than the reference.  Summation order differs
//     from the reference's (unspecified) atomic order, so gradients agree to rounding, not
//     bitwise -- exactly as two runs of the reference differ from each other.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "blend_math.h"
#include "common.h"

namespace grpg {

constexpr int RB_WAVES = 4;
constexpr int BQCAP = 256;     // ring capacity in entries (>= 64 + 128), power of two
constexpr int BFILL_Q = 2;     // list entries per lane per FILL step

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_step(float v) {
  const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false);
  return v + __int_as_float(t);
}

// Sum over the 64 lanes; the total is valid in lane 63.
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
  v = dpp_step<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
  v = dpp_step<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
  v = dpp_step<0x141, 0xF>(v);  // row_half_mirror
  v = dpp_step<0x140, 0xF>(v);  // row_mirror      -> every lane holds its row's (16-lane) sum
  v = dpp_step<0x142, 0xA>(v);  // row_bcast:15 into rows 1,3
  v = dpp_step<0x143, 0xC>(v);  // row_bcast:31 into rows 2,3 -> lane 63 = total
  return v;
}

// Row sum (16 lanes) left in EVERY lane of the row: 4 DPP adds.
__device__ __forceinline__ float row_sum_all(float v) {
  v = dpp_step<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
  v = dpp_step<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
  v = dpp_step<0x141, 0xF>(v);  // row_half_mirror
  v = dpp_step<0x140, 0xF>(v);  // row_mirror
  return v;
}

// Halving exchanges of gfx950: a and b are two different quantities to be summed over the wave.
// After the call `a` holds, in lanes 0-31, (a of lane L) + (a of lane L+32), and in lanes 32-63 the
// same for b: two quantities cost one swap + one add instead of two full-width butterfly steps.
__device__ __forceinline__ float halve32(const float a, const float b) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// Same across the two 16-lane rows of each half: even rows end up with a, odd rows with b.
__device__ __forceinline__ float halve16(const float a, const float b) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// Wave totals of 12 quantities in 30 VALU instructions (66+ with one butterfly each).
// Row r (lanes 16r .. 16r+15) returns in out[k] the total of v[k + 3 r]:
//   step 1 pairs (v[k], v[k+6]) across the wave halves, step 2 pairs (., .+3) across rows,
//   then a 16-lane row sum.
__device__ __forceinline__ void wave_sum_12(const float (&v)[12], float (&out)[3]) {
  float h[6];
#pragma unroll
  for (int k = 0; k < 6; k++) h[k] = halve32(v[k], v[k + 6]);
#pragma unroll
  for (int k = 0; k < 3; k++) out[k] = row_sum_all(halve16(h[k], h[k + 3]));
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d, 64));
  return v;
}

template <int CTRL>
__device__ __forceinline__ float dpp_take(const float v) {   // v of the lane CTRL selects (all lanes valid)
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
// lanes whose bank (group of 4 lanes within a 16-lane row) is in BANKS take `src`, the others keep `old`
template <int BANKS>
__device__ __forceinline__ float bank_merge(const float old, const float src) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src),
                                                    0xE4 /* quad_perm [0,1,2,3] */, 0xF, BANKS, false));
}

// Wave totals of 12 quantities by a FOLDING tree: every level halves the number of lanes a
// quantity's partial sums occupy AND the number of registers, so the deeper levels are shared:
//   64 -> 32 lanes, 12 -> 6 registers   v_permlane32_swap + add                       (12 instructions)
//   32 -> 16 lanes,  6 -> 3 registers   v_permlane16_swap + add                       ( 6)
//   quad level,      3 -> 1 register    two select-pairs + quad_perm adds: lane j of every quad ends
//                                       up with its quad's sum of quantity j           ( 7)
//   4 quads -> 1                        row_ror:8 and row_ror:4 adds                  ( 2)
// 27 VALU instructions (three full row sums before: 37).  Row r (lanes 16 r ...) holds the totals of
// v[3r], v[3r+1], v[3r+2] in lanes 0, 1, 2 of each of its quads: adjacent lanes, so the 11-lane
// atomic that follows is one request per row (totals parked in lanes 0 / 4 / 8 cost 3.5x the
// atomic time: every quad of lanes became a request of its own).
__device__ __forceinline__ float wave_fold_12(const float (&v)[12], const bool lane_odd, const bool lane_hi2) {
  float h[6], g[3];
#pragma unroll
  for (int k = 0; k < 6; k++) h[k] = halve32(v[k], v[k + 6]);
#pragma unroll
  for (int k = 0; k < 3; k++) g[k] = halve16(h[k], h[k + 3]);
  // lanes 0,2 of a quad: g0 pair sums; lanes 1,3: g1 pair sums
  const float keep01 = lane_odd ? g[1] : g[0], give01 = lane_odd ? g[0] : g[1];
  const float e01 = keep01 + dpp_take<0xB1>(give01);          // quad_perm [1,0,3,2]
  const float e2 = g[2] + dpp_take<0xB1>(g[2]);               // g2 pair sums in every lane
  // lanes 0,1: quad totals of g0, g1; lanes 2,3: quad total of g2
  const float keep = lane_hi2 ? e2 : e01, give = lane_hi2 ? e01 : e2;
  float t = keep + dpp_take<0x4E>(give);                       // quad_perm [2,3,0,1]
  t += dpp_take<0x128>(t);                                     // row_ror:8  (quads q and q^2)
  t += dpp_take<0x124>(t);                                     // row_ror:4  (all four quads)
  return t;
}

// Round 5 tried the matrix pipe for these totals (one v_mfma_f32_16x16x4_f32 per quantity with a one-hot B
// column, 3 adds and two row exchanges behind them: 11 adjacent lanes hold the totals) -- parity-green and
// 40 % SLOWER (backward 0.91 vs 0.66 ms): eleven dependent 8-pass MFMAs per (wave, splat) are 352 cycles of
// the one matrix pipe four waves share, more than the trip's vector work.  DESIGN_EXPERIMENTS.md.
constexpr int BREC = 4;   // float4 per compacted survivor in LDS: record (3) + pre-scaled conic

// One wave, PX pixels per lane: rows y0 + (lane>>4)*PX + k.  bits_mask selects the sub-tile bits
// of a point-list entry that concern this wave (one bit for a quarter, all four for a whole tile).
// SEG (quarter waves of a long tile only): the wave walks list positions [seg_lo, seg_hi) and starts
// from the forward's checkpoint at seg_hi (ck_end; NULL for the list's last segment) -- see below.
template <int PX, int SMAX, bool SEG = false>
__device__ __forceinline__ void backward_rect(
    float4* __restrict__ my, uint32_t* __restrict__ qid, uint32_t* __restrict__ qpos,
    const int lane, const uint32_t r_begin, const uint32_t r_end,
    const int x0, const int y0, const uint32_t bits_mask, const int W, const int H, const int S,
    const uint32_t* __restrict__ point_list, const RecView rec,
    const float* __restrict__ semantics, const float* __restrict__ bg,
    const float* __restrict__ alphas, const uint32_t* __restrict__ n_contrib,
    const float* __restrict__ dL_dpix, const float* __restrict__ dL_dpix_depth,
    const float* __restrict__ dL_dalphas, const float* __restrict__ dL_dpix_semantic,
    float* __restrict__ grad_rec, float* __restrict__ dL_dsemantic, const int ablate_in,
    unsigned long long* __restrict__ stats_in, const uint32_t seg_lo = 0u, const uint32_t seg_hi = 0u,
    const float* __restrict__ ck_end = nullptr, const float* __restrict__ ck_final = nullptr) {
#ifdef GRPG_TRACE   // experiment build: ablation switches and loop counters are live
  const int ablate = ablate_in;
  unsigned long long* const stats = stats_in;
#else               // production: constants, so that the counters and the switches' masks cost nothing in the trip loop
  constexpr int ablate = 0;
  constexpr unsigned long long* stats = nullptr;
#endif
  constexpr int SM = SMAX > 0 ? SMAX : 1;
  const int px = x0 + (lane & 15);
  const int py0 = y0 + (lane >> 4) * PX;
  const float pxf = (float)px;
  float rx0 = (float)x0, rx1 = (float)(x0 + 15);   // cull rectangle, see the active-pixel box below
  float ry0 = (float)y0, ry1 = (float)(y0 + 4 * PX - 1);
  uint64_t prev_act[PX];
#pragma unroll
  for (int k = 0; k < PX; k++) prev_act[k] = ~0ull;
  const size_t HW = (size_t)H * W;
  const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];

  // Per-pixel state.  The reference carries (last_alpha, last_color) per channel and forms
  //   accum_rec = last_alpha * last_color + (1 - last_alpha) * accum_rec        (backward.cu:556-561)
  // at the NEXT contributor; the same value is obtained by blending the current contributor into
  // the accumulator right after it has been used, acc += alpha (c - acc).  The accumulators are
  // only ever used inside ONE sum, dL/dalpha = sum_ch (c_ch - acc_ch) dL_ch (colour, depth,
  // accumulated alpha with c = 1, semantics: backward.cu:565,584,594,600), and both that sum and the
  // update are linear in them -- so the pixel carries the single scalar
  //   A = sum_ch acc_ch dL_ch,        d = (sum_ch c_ch dL_ch) - A,        A += alpha d
  // instead of 5 + S accumulators (round 6: 9 of the 56 vector instructions per (pixel, splat) and 8
  // registers gone; the difference is formed once instead of per channel, which moves the result by
  // rounding only).  Plain (unpacked) fp32 throughout: on gfx950 a v_pk_fma_f32 occupies the VALU
  // 1.8x as long as a v_fma_f32 (tools/ubench/valu_rate.hip), so packing buys nothing and costs moves.
  // The accumulated-alpha plane needs no state at all: its channel has c = 1, so (1 - acc) is the
  // transmittance of everything BEHIND the splat, T (1 - acc) = T_final / (1 - alpha) -- the very factor
  // of the background term (backward.cu:600,611-614), which takes dL_dalpha in with it (nTfBg below).
  // (Inside A it would be the one badly conditioned channel: acc -> 1 on opaque pixels.)
  float T[PX], nTfBg[PX], A[PX];
  float dLc[PX][4];                      // channels: r, g, b, depth
  float dLs[PX][SM];
  float pyf[PX];
  uint32_t lastc[PX];
  uint32_t maxlast = 0;
#pragma unroll
  for (int k = 0; k < PX; k++) {
    const int py = py0 + k;
    const bool inside = px < W && py < H;
    const size_t pix = inside ? (size_t)py * W + px : 0;
    const float T_final = inside ? 1.0f - alphas[pix] : 0.f;
    T[k] = T_final;
    lastc[k] = inside ? n_contrib[pix] : 0u;
    dLc[k][0] = inside ? dL_dpix[pix] : 0.f;
    dLc[k][1] = inside ? dL_dpix[HW + pix] : 0.f;
    dLc[k][2] = inside ? dL_dpix[2 * HW + pix] : 0.f;
    dLc[k][3] = inside ? dL_dpix_depth[pix] : 0.f;
    const float dLa = inside ? dL_dalphas[pix] : 0.f;
    // background term of backward.cu:611-614, (-T_final / (1 - alpha)) * (bg . dL_dpixel), and the
    // accumulated alpha's T (1 - accum_alpha_rec) dL_dalpha = (T_final / (1 - alpha)) dL_dalpha (:600)
    nTfBg[k] = T_final * (dLa - (bg0 * dLc[k][0] + bg1 * dLc[k][1] + bg2 * dLc[k][2]));
    pyf[k] = (float)py;
    A[k] = 0.f;
#pragma unroll
    for (int c = 0; c < SM; c++)
      dLs[k][c] = (SMAX > 0 && c < S && inside) ? dL_dpix_semantic[(size_t)c * HW + pix] : 0.f;
    maxlast = max(maxlast, lastc[k]);
  }
  if (SEG && ck_end != nullptr) {
    // Start in the MIDDLE of the list, from the forward's state right behind position seg_hi:
    // T there, and what the back-to-front walk would have accumulated by then -- the composite of
    // everything behind, as seen from this depth: (C_final - C_k) / T_k per channel, folded into A.
    // A pixel that terminated in front of seg_hi has T_k == T_final and C_k == C_final (its state
    // stopped changing): zeros, and pos < lastc rejects every entry anyway.
    const float Tk = ck_end[lane];
    const float inv = 1.0f / fmaxf(Tk, 1e-30f);
    T[0] = Tk;
    float a0 = 0.f;
#pragma unroll
    for (int c = 0; c < 4; c++)
      a0 = fmaf((ck_final[64 * (c + 1) + lane] - ck_end[64 * (c + 1) + lane]) * inv, dLc[0][c], a0);
    A[0] = a0;
  }
  maxlast = wave_max_u32(maxlast);   // nothing behind the tile's deepest contributor matters
  const float nddelx = -(float)(0.5 * W), nddely = -(float)(0.5 * H);   // -ddelx_dx, -ddely_dy (backward.cu:501-502)

  // Gradient scatter: wave_fold_12 leaves the total of quantity 3 r + {0, 1, 2} in lanes 0 / 1 / 2 of
  // row r (lanes 16 r ...):
  //   row 0: dL_dmean2D x, y, |.|   row 1: dL_dconic xx, xy, yy   row 2: dL_dcolor r, g, b
  //   row 3: dL_dopacity, dL_ddepth, -
  // i.e. float 3 r + k of the Gaussian's 64-byte gradient record (common.h GRAD_*): ONE
  // global_atomic_add_f32 with 11 active lanes and ONE memory line per (wave, Gaussian) --
  // the reference issues 11 atomics per PIXEL and Gaussian; five separate arrays cost five lines per
  // atomic, and on this part every line is a transaction of its own at the memory side (the XCDs'
  // L2s are not coherent): 0.28 of 0.96 ms before the record.  The conic moments are accumulated
  // without their -1/2 (backward.cu:634-636); the issuing lanes apply it to the wave total.
  const int row = lane >> 4, sub = lane & 15;
  const bool lane_odd = (lane & 1) != 0, lane_hi2 = (lane & 2) != 0;
  const bool sc_lane = sub < 3 && (3 * row + sub) < 11;
  float* const sc_ptr = grad_rec + (3 * row + sub);
  const float sc_scale = row == 1 ? -0.5f : 1.0f;

  // Back-to-front traversal of list positions [0, count), decoupled like the forward's heavy path
  // (render_fwd.hip): FILL scans 128 entries per step (prefetched one window ahead) and appends
  // the ones whose sub-tile mask concerns this wave to an LDS ring, deepest first; POP takes up to
  // 64 of them and starts the gather of their records; the PREVIOUS batch is culled, compacted and
  // processed while that gather is in flight.  A dead entry costs 1/128 of a FILL step and no
  // record is loaded for it; in a horizon tile only one entry in six concerns a given quarter.
  const uint64_t lt = lanemask_lt();
  const uint32_t lo = SEG ? seg_lo : 0u;   // list positions [lo, count) are walked
  const uint32_t count = max(min(SEG ? seg_hi : r_end - r_begin, maxlast), lo);
  uint32_t in_hi = count;           // list positions [lo, in_hi) are still unread (wave-uniform)
  uint32_t head = 0, rcount = 0;    // ring state                                (wave-uniform)
  uint32_t win[BFILL_Q];
#pragma unroll
  for (int q = 0; q < BFILL_Q; q++) {
    const uint32_t off = (uint32_t)(q * WAVE + lane);
    win[q] = off < in_hi - lo ? point_list[r_begin + in_hi - 1 - off] : 0u;
  }
  const uint64_t ablate_m = (ablate & 4) ? 0ull : ~0ull;
  uint32_t st_fill = 0, st_batches = 0, st_iters = 0, st_used = 0, st_rows = 0;   // GRPG_BWD_STATS
  unsigned long long st_small = 0ull, st_lanes_sum = 0ull;
  const unsigned long long st_t0 = stats ? __builtin_readcyclecounter() : 0ull;
  float4 la = make_float4(0, 0, 0, 0), lb = la, lc = la;   // current batch (records arrived)
  uint32_t lpos = 0, lid = 0, ncur = 0;
  for (;;) {
    // ---- FILL ----
    while (rcount < (uint32_t)WAVE && in_hi > lo) {
      uint32_t v[BFILL_Q];
#pragma unroll
      for (int q = 0; q < BFILL_Q; q++) v[q] = win[q];
      const uint32_t nxt = in_hi - lo > (uint32_t)(BFILL_Q * WAVE) ? in_hi - BFILL_Q * WAVE : lo;
#pragma unroll
      for (int q = 0; q < BFILL_Q; q++) {
        const uint32_t off = (uint32_t)(q * WAVE + lane);
        win[q] = off < nxt - lo ? point_list[r_begin + nxt - 1 - off] : 0u;
      }
#pragma unroll
      for (int q = 0; q < BFILL_Q; q++) {
        const uint32_t off = (uint32_t)(q * WAVE + lane);
        const bool keep = (off < in_hi - lo) && (v[q] & bits_mask);
        const uint64_t m = __ballot(keep);
        if (keep) {
          const uint32_t slot = (head + rcount + (uint32_t)__popcll(m & lt)) & (BQCAP - 1);
          qid[slot] = v[q] & ID_MASK;
          qpos[slot] = in_hi - 1 - off;   // 0-based list position == the reference's `contributor`
        }
        rcount += (uint32_t)__popcll(m);
      }
      in_hi = nxt;
      st_fill++;
    }
    // the ring entries written by FILL are read by OTHER lanes in POP: keep the compiler from
    // reordering the LDS accesses across this point (costs no instruction)
    __builtin_amdgcn_wave_barrier();
    // ---- POP: next batch of up to 64 entries (deepest first), start its record gather ----
    const uint32_t nn = min(rcount, (uint32_t)WAVE);
    float4 na = make_float4(0, 0, 0, 0), nb = na, nc = na;
    uint32_t npos = 0, nid = 0;
    if ((uint32_t)lane < nn) {
      const uint32_t slot = (head + lane) & (BQCAP - 1);
      nid = qid[slot];
      npos = qpos[slot];
      rec.load(nid, na, nb, nc);
    }
    head = (head + nn) & (BQCAP - 1);
    rcount -= nn;
    // ---- the previous batch: cull, compact (deepest first), process ----
    int cnt = 0;
    if (ncur > 0) {
      // Only pixels whose last contributor lies behind the batch's shallowest entry can use any of
      // its splats: cull against the bounding box of THOSE pixels (it grows as the walk approaches
      // the front; in a horizon tile only the never-saturating sky pixels are active for most of
      // the list).  Exact: a splat is dropped only if no active pixel can accept it.
      const uint32_t pos_min = (uint32_t)__builtin_amdgcn_readlane((int)lpos, (int)ncur - 1);
      bool changed = false;
#pragma unroll
      for (int k = 0; k < PX; k++) {
        const uint64_t act = __ballot(lastc[k] > pos_min);
        changed = changed || (act != prev_act[k]);
        prev_act[k] = act;
      }
      if (changed) {
        // from the bits of the lane masks, on the scalar unit (render_fwd.hip mask_box: the butterfly over
        // four floats per lane this replaces cost 24 ds_bpermute round trips + 24 min / max per update, and
        // the active set changes at almost every batch); rows y0 + PX (lane >> 4) + k
        int c0 = 15, c1 = 0, r0 = 4 * PX, r1 = -1;
#pragma unroll
        for (int k = 0; k < PX; k++) {
          const uint64_t m = prev_act[k];
          if (m != 0ull) {
            const int ra = (int)(__builtin_ctzll(m) >> 4), rb = (int)((63 - __builtin_clzll(m)) >> 4);
            uint32_t cols = (uint32_t)m | (uint32_t)(m >> 32);
            cols = (cols | (cols >> 16)) & 0xFFFFu;
            c0 = min(c0, (int)__builtin_ctz(cols)); c1 = max(c1, 31 - (int)__builtin_clz(cols));
            r0 = min(r0, PX * ra + k); r1 = max(r1, PX * rb + k);
          }
        }
        rx0 = (float)(x0 + c0); rx1 = (float)(x0 + c1); ry0 = (float)(y0 + r0); ry1 = (float)(y0 + r1);
        // no active pixel at all (cannot happen while the walk starts at the deepest contributor, kept for the
        // day it does not): the inverted box above does not make splat_misses_rect reject -- a far-away one does
        if (r1 < 0) { rx0 = 3e38f; rx1 = -3e38f; ry0 = 3e38f; ry1 = -3e38f; }
      }
      const bool keep = ((uint32_t)lane < ncur) &&
                        !splat_misses_rect(la.x, la.y, lb.x, lb.y, lb.z, la.w, rx0, rx1, ry0, ry1);
      const uint64_t mask = __ballot(keep);
      cnt = (int)__popcll(mask);
      if (keep) {
        const int slot = (int)__popcll(mask & lt);
        const SplatQ sq = splat_q(lb.x, lb.y, lb.z);   // once per survivor, not once per (wave, survivor)
        // the conic enters the trip loop only through dL_dmean2D = -ddel (conic t): its four products
        // with -ddelx_dx / -ddely_dy are formed here, once per survivor, not per (pixel, survivor)
        my[slot * BREC + 0] = la;
        my[slot * BREC + 1] = make_float4(nddelx * lb.x, nddelx * lb.y, nddely * lb.z, lb.w);
        my[slot * BREC + 2] = make_float4(lc.x, lc.y, __uint_as_float(lpos), __uint_as_float(lid));
        my[slot * BREC + 3] = make_float4(sq.A, sq.B, sq.C, nddely * lb.y);
      }
    }
    __builtin_amdgcn_wave_barrier();
    st_batches += ncur > 0 ? 1u : 0u;
    st_iters += (uint32_t)cnt;
    const int cnt_u = __builtin_amdgcn_readfirstlane(cnt);   // loop control on the scalar unit
    for (int j = 0; j < cnt_u; j++) {
      const float4 a = my[j * BREC + 0];   // px, py, depth, opacity
      const float4 b = my[j * BREC + 1];   // -ddelx conic.x, -ddelx conic.y, -ddely conic.z, R
      const float4 c = my[j * BREC + 2];   // G, B, list position, Gaussian id
      const float4 q = my[j * BREC + 3];   // pre-scaled conic A, B, C (blend_math.h), -ddely conic.y
      const uint32_t pos = __float_as_uint(c.z);   // 0-based == the reference's `contributor`
      const uint32_t gid = __float_as_uint(c.w);
      const float dx = a.x - pxf;
      SplatTerms st;                       // == splat_terms(dx, conic): same operations, same bits
      st.hA = (q.x * dx) * dx;
      st.bdx = q.y * dx;
      st.hc = q.z;
      const float col[4] = {b.w, c.x, c.y, a.z};
      // per-lane partial sums of the 11 gradient quantities of this splat over the lane's pixels
      float g_mx = 0.f, g_my = 0.f, g_mabs = 0.f, g_cxx = 0.f, g_cxy = 0.f, g_cyy = 0.f, g_op = 0.f;
      float g_c[4] = {0.f, 0.f, 0.f, 0.f};
      float g_s[SM];
#pragma unroll
      for (int cc = 0; cc < SM; cc++) g_s[cc] = 0.f;
      bool any = false;
      uint64_t st_lanes = 0ull;   // GRPG_BWD_STATS: lanes that take the splat for some pixel
#pragma unroll
      for (int k = 0; k < PX; k++) {
        const float dy = a.y - pyf[k];
        float G, alpha;   // identical arithmetic to the forward (blend_math.h)
        const float power2 = pair_power(st, dy);
        G = __builtin_amdgcn_exp2f(power2);
        alpha = fminf(ALPHA_MAX, a.w * G);
        // lane masks in SGPRs: the compares ARE ballots, the boolean algebra runs on the scalar unit
        const uint64_t valid_m = __builtin_amdgcn_ballot_w64(!(power2 > 0.0f)) &
                                 __builtin_amdgcn_ballot_w64(!(alpha < ALPHA_MIN)) &
                                 __builtin_amdgcn_ballot_w64(pos < lastc[k]) & ablate_m;
        if (valid_m == 0ull) continue;   // wave-uniform: no pixel of this row takes the splat
        const bool valid = __builtin_amdgcn_inverse_ballot_w64(valid_m);
        any = true;
        st_rows++;
        st_lanes |= valid_m;
        // Branch-free below: a lane that rejects the splat runs the same arithmetic with
        // alpha = G = 0, which leaves its T and accumulators unchanged (T / (1 - 0), acc + 0 * d)
        // and contributes exact zeros to every sum.
        alpha = valid ? alpha : 0.f;
        G = valid ? G : 0.f;
        // one reciprocal serves both divisions of backward.cu:547,596 (1 ulp: the recovered T is
        // the inverse of a rounded product chain anyway)
        const float inv_1ma = __builtin_amdgcn_rcpf(1.f - alpha);
        T[k] = T[k] * inv_1ma;
        const float dch = alpha * T[k];
        // dL/dalpha through what lies behind this splat: sum_c (c - acc_c) dL_c = (sum_c c dL_c) - A
        // (backward.cu:556-596: colour, semantics, depth)
        float cD = 0.f;
#pragma unroll
        for (int ch = 0; ch < 4; ch++) {
          cD = fmaf(col[ch], dLc[k][ch], cD);
          g_c[ch] = fmaf(dch, dLc[k][ch], g_c[ch]);
        }
        if (SMAX > 0) {
#pragma unroll
          for (int cc = 0; cc < SM; cc++) {
            if (cc < S) {
              cD = fmaf(semantics[(size_t)gid * S + cc], dLs[k][cc], cD);
              g_s[cc] = fmaf(dch, dLs[k][cc], g_s[cc]);
            }
          }
        }
        const float d = cD - A[k];
        A[k] = fmaf(alpha, d, A[k]);                    // every accumulator: alpha c + (1 - alpha) acc
        const float dL_dopa = fmaf(inv_1ma, nTfBg[k], d * T[k]);
        const float gd = G * dL_dopa;
        g_op += gd;
        // t = dL_dG G (dx, dy), dL_dG = opacity dL_dopa:  dL_dmean2D = -ddel (conic t),  dL_dconic = -1/2 t (dx, dy)^T
        const float w = a.w * gd;
        const float tdx = w * dx, tdy = w * dy;
        const float mx = fmaf(b.y, tdy, b.x * tdx);
        const float my_ = fmaf(q.w, tdx, b.z * tdy);
        g_mx += mx;
        g_my += my_;
        g_mabs += fabsf(mx);
        g_mabs += fabsf(my_);
        g_cxx = fmaf(tdx, dx, g_cxx);
        g_cxy = fmaf(tdx, dy, g_cxy);
        g_cyy = fmaf(tdy, dy, g_cyy);
      }
      if (!any) continue;         // wave-uniform: nobody in the tile used this splat
      st_used++;
      if (stats != nullptr) {
        const int nl = (int)__popcll(st_lanes);
        st_lanes_sum += (unsigned long long)nl;
        st_small += nl <= 1 ? 1ull : (nl <= 4 ? (1ull << 20) : (nl <= 8 ? (1ull << 40) : 0ull));
      }
      if (ablate & 2) continue;   // experiment switch (GRPG_BWD_ABLATE): no reduction, no atomics
      {
        const float qv[12] = {g_mx, g_my, g_mabs, g_cxx, g_cxy, g_cyy, g_c[0], g_c[1], g_c[2], g_op,
                              g_c[3], 0.f};
        const float tot = wave_fold_12(qv, lane_odd, lane_hi2);
        if (sc_lane && !(ablate & 1)) atomicAdd(sc_ptr + gid * (uint32_t)GRAD_STRIDE, tot * sc_scale);
      }
      if (SMAX > 0) {
#pragma unroll
        for (int cc = 0; cc < SM; cc++) g_s[cc] = wave_sum_to_lane63(g_s[cc]);
        if (lane == 63) {
#pragma unroll
          for (int cc = 0; cc < SM; cc++)
            if (cc < S) atomicAdd(&dL_dsemantic[(size_t)gid * S + cc], g_s[cc]);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    la = na; lb = nb; lc = nc; lpos = npos; lid = nid; ncur = nn;
    if (ncur == 0 && in_hi == lo) break;   // ring empty (rcount == 0 here) and list exhausted
  }
  if (stats != nullptr && lane == 0) {   // experiment counters (GRPG_BWD_STATS=1), off in production
    // one record of 8 words per wave, no atomics (same-address atomics would dominate the launch)
    unsigned long long* r = stats + 8ull * ((unsigned long long)blockIdx.x * RB_WAVES + (threadIdx.x >> 6));
    // wave kind: 1 quarter wave, 3 half-tile wave; + 4: a (tile, segment) item; above: the tile's list length
    r[0] = (1ull + (PX == 1 ? 0ull : 2ull)) | (SEG ? 4ull : 0ull) | ((unsigned long long)(r_end - r_begin) << 8);
    // list entries in reach of the wave (up to the deepest contributor) | the wave's share of the list
    r[1] = (unsigned long long)(count - lo) | ((unsigned long long)(SEG ? seg_hi - seg_lo : r_end - r_begin) << 32);
    r[2] = ((unsigned long long)st_fill << 32) | st_batches;
    r[3] = st_iters;                          // survivors of the rectangle cull = loop trips
    r[4] = st_used;                           // ... of which some pixel used (reduction + atomic)
    r[5] = (unsigned long long)st_rows | (st_lanes_sum << 32);   // gradient blocks executed | accepting lanes, summed over the used trips
    r[6] = __builtin_readcyclecounter() - st_t0;
    r[7] = st_small;                          // used trips with 1 / 2-4 / 5-8 accepting lanes (20 bits each)
  }
}

// Work lists: written by the forward's classify_tiles_kernel (render_fwd.hip) into the image blob:
// counts[4] (three heavy classes, light), then four lists of T tile ids.
constexpr int NUM_CLASSES_B = 4;

// MINW: waves per SIMD the register allocator must fit.  Since the scalar accumulator (round 6) the
// S = 0 variant takes 96 VGPRs (five waves per SIMD, no scratch; rounds 3-5: 128 with 24 B of scratch per
// lane), the S <= 4 variant 114 (four waves; before: 139, three), the S <= 32 variant all 256 + 31 AGPRs
template <int SMAX, int LIGHT_SPLIT, int MINW = 1>
__global__ void __launch_bounds__(256, MINW)
render_backward_kernel(const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                       const RecView rec, const float* __restrict__ semantics,
                       const int S, const int W, const int H, const int gx, const uint32_t T,
                       const uint32_t* __restrict__ work, const float* __restrict__ bg,
                       const float* __restrict__ alphas, const uint32_t* __restrict__ n_contrib,
                       const float* __restrict__ dL_dpix, const float* __restrict__ dL_dpix_depth,
                       const float* __restrict__ dL_dalphas,
                       const float* __restrict__ dL_dpix_semantic, float* __restrict__ grad_rec,
                       float* __restrict__ dL_dsemantic, const int ablate, const int wide_classes,
                       unsigned long long* __restrict__ stats,
                       const BlobHeader* __restrict__ bin_hdr, const uint32_t* __restrict__ ck_count,
                       const uint32_t* __restrict__ bwd_ctl, const uint32_t items_cap) {
  __shared__ float4 s_rec[RB_WAVES][WAVE * BREC];
  __shared__ uint32_t s_qid[RB_WAVES][BQCAP];
  __shared__ uint32_t s_qpos[RB_WAVES][BQCAP];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // Long tiles (forward checkpoints present): the first items_cap workgroups take (tile, segment)
  // items, four quarter waves each; the tiles themselves are skipped further down.
  const bool ck_on = SMAX == 0 && items_cap != 0u && bin_hdr->ckpt_off256 != 0u;
  if (SMAX == 0 && items_cap != 0u && blockIdx.x < items_cap) {
    if (!ck_on || blockIdx.x >= min(bwd_ctl[0], bin_hdr->ckpt_slots)) return;
    const float* recs = (const float*)((const char*)bin_hdr + (size_t)bin_hdr->ckpt_off256 * 256);
    const uint2 item = ((const uint2*)((const char*)recs + ckpt_items_offset(bin_hdr->ckpt_slots)))[blockIdx.x];
    const uint32_t tile = item.x, k = item.y;
    const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)ck_count[(size_t)tile * 4 + wave]);
    if (k >= n) return;   // this quarter cut its list into fewer pieces
    const int ty = (int)(tile / (uint32_t)gx), tx = (int)(tile - (uint32_t)ty * (uint32_t)gx);
    const uint2 range = ranges[tile];
    const uint32_t rb = __builtin_amdgcn_readfirstlane(range.x);
    const uint32_t re = __builtin_amdgcn_readfirstlane(range.y);
    const float* r0 = recs + ((size_t)ckpt_tile_base(rb) * 4 + (size_t)wave) * CK_REC_FLOATS;
    const float* rk = r0 + (size_t)k * 4 * CK_REC_FLOATS;
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(rk[320]));
    const uint32_t lo = k ? (uint32_t)__builtin_amdgcn_readfirstlane(
                                (int)__float_as_uint((rk - 4 * CK_REC_FLOATS)[320])) : 0u;
    backward_rect<1, SMAX, true>(s_rec[wave], s_qid[wave], s_qpos[wave], lane, rb, re, tx * TILE,
                                 ty * TILE + wave * 4, 1u << (SUBTILE_SHIFT + wave), W, H, S, point_list,
                                 rec, semantics, bg, alphas, n_contrib, dL_dpix, dL_dpix_depth,
                                 dL_dalphas, dL_dpix_semantic, grad_rec, dL_dsemantic, ablate, stats, lo,
                                 hi, k + 1u == n ? nullptr : rk, r0 + (size_t)(n - 1u) * 4 * CK_REC_FLOATS);
    return;
  }
  // wide_classes = 1: the shortest heavy class of the forward's classification is walked like the
  // light tiles here (two half-tile waves, 2 pixels per lane) -- one reduction + atomic per
  // (half tile, splat) instead of one per quarter (see launch_render_backward)
  const uint32_t n0 = work[0], n1 = work[1];
  const uint32_t n2 = wide_classes >= 1 ? 0u : work[2];
  const uint32_t nmid = wide_classes >= 1 ? work[2] : 0u;
  const uint32_t nlight = work[3] + nmid;
  const uint32_t nheavy = n0 + n1 + n2;
  const uint32_t b = blockIdx.x - (SMAX == 0 ? items_cap : 0u);
  const uint32_t* lists = work + NUM_CLASSES_B;
  uint32_t tile;
  uint32_t half = 0;   // light tiles: which 16x8 half of the tile this wave owns
  if (b < nheavy) {
    tile = b < n0 ? lists[b] : (b < n0 + n1 ? lists[T + (b - n0)] : lists[2 * T + (b - n0 - n1)]);
  } else {
    // LIGHT tiles: LIGHT_SPLIT waves per tile, 4 / LIGHT_SPLIT pixels per lane.
    const uint32_t li = (b - nheavy) * RB_WAVES + (uint32_t)wave;
    if (li >= LIGHT_SPLIT * nlight) return;   // whole wave exits together; no workgroup barriers are used
    const uint32_t lt_ = li / LIGHT_SPLIT;   // mid tiles first (longer lists start earlier)
    tile = lt_ < nmid ? lists[2 * (size_t)T + lt_] : lists[3 * (size_t)T + (lt_ - nmid)];
    half = li % LIGHT_SPLIT;
  }
  const int ty = (int)(tile / (uint32_t)gx), tx = (int)(tile - (uint32_t)ty * (uint32_t)gx);
  const uint2 range = ranges[tile];
  const uint32_t rb = __builtin_amdgcn_readfirstlane(range.x);
 const uint32_t re = __builtin_amdgcn_readfirstlane(range.y);
  if (ck_on && re - rb >= CK_LONG_MIN) return;   // walked as (tile, segment) items above
#define RB_CALL(PXV, YOFF, BITS)                                                                  \
  backward_rect<PXV, SMAX>(s_rec[wave], s_qid[wave], s_qpos[wave], lane, rb, re, tx * TILE, ty * TILE + (YOFF), (BITS), W, H, \
                           S, point_list, rec, semantics, bg, alphas, n_contrib, dL_dpix,          \
                           dL_dpix_depth, dL_dalphas, dL_dpix_semantic, grad_rec, dL_dsemantic,     \
                           ablate, stats)
  if (b < nheavy)
    RB_CALL(1, wave * 4, 1u << (SUBTILE_SHIFT + wave));
  else if (LIGHT_SPLIT == 2)
    RB_CALL(2, (int)half * 8, (half ? 0xCu : 0x3u) << SUBTILE_SHIFT);
  else
    RB_CALL(4, 0, 0xFu << SUBTILE_SHIFT);
#undef RB_CALL
}

void launch_render_backward(hipStream_t s, const uint2* ranges, const uint32_t* point_list,
                            const RecView rec, const float* semantics, int S, int W, int H, int gx,
                            int gy, const float* bg, const float* alphas,
                            const uint32_t* n_contrib, const uint32_t* work, const float* dL_dpix,
                            const float* dL_dpix_depth, const float* dL_dalphas,
                            const float* dL_dpix_semantic, float* grad_rec, float* dL_dsemantic,
                            const BlobHeader* bin_hdr, const uint32_t* ck_count, const uint32_t* bwd_ctl,
                            uint32_t R) {
  const int ntiles = gx * gy;
  if (ntiles <= 0) return;
  uint32_t items_cap = 0;
  // the forward left (tile, segment) items behind (S = 0 training forwards only); their number is on
  // the device, bounded here by the sum of ckpt_tile_cap over lists of >= CK_LONG_MIN entries.
  // Without them (semantic frames) every list is walked as one chain.
  if (S <= 0 && bin_hdr && ck_count && bwd_ctl) items_cap = ckpt_slots(R);
  // The forward's shortest heavy class (lists of 256 .. 2047 entries) is walked by two half-tile
  // waves at 2 pixels per lane like the light tiles -- one reduction + atomic per (half tile, splat)
  // instead of one per quarter.  The kernel is VALU-bound and a third of its cycles are that
  // reduction: 515 -> 489 us at config 5, 645 -> 573 us on the 2 M-Gaussian scene (quarter waves for
  // every heavy tile, or half tiles for lists of 2048 .. 4095 entries as well: slower).
  const int wide = 1;
  int ablate = 0;
  unsigned long long* stats = nullptr;
#ifdef GRPG_TRACE   // experiment build: ablation switch and per-launch loop counters (synchronises)
  // GRPG_BWD_ABLATE: 1 = no atomics, 2 = no reduction either, 4 = traversal + alpha only
  static const int ablate_env = [] { const char* e = getenv("GRPG_BWD_ABLATE"); return e ? atoi(e) : 0; }();
  ablate = ablate_env;
  static const int want_stats = [] { const char* e = getenv("GRPG_BWD_STATS"); return e ? atoi(e) : 0; }();
  static unsigned long long* stats_dev = nullptr;
  static size_t stats_cap = 0;
  const int grid_max = ntiles + ntiles / 2 + 1 + (int)ckpt_slots(R);
  const size_t stats_words = 8ull * (size_t)grid_max * RB_WAVES;
  if (want_stats) {
    if (stats_words > stats_cap) {
      if (stats_dev) (void)hipFree(stats_dev);
      stats_dev = nullptr;
      if (hipMalloc((void**)&stats_dev, stats_words * sizeof(unsigned long long)) != hipSuccess) return;
      stats_cap = stats_words;
    }
    (void)hipMemsetAsync(stats_dev, 0, stats_words * sizeof(unsigned long long), s);
    stats = stats_dev;
  }
#endif
#define RB_ARGS                                                                                  \
  ranges, point_list, rec, semantics, S, W, H, gx, (uint32_t)ntiles, work, bg, alphas, n_contrib, \
      dL_dpix, dL_dpix_depth, dL_dalphas, dL_dpix_semantic, grad_rec, dL_dsemantic, ablate, wide, stats, \
      bin_hdr, ck_count, bwd_ctl, items_cap
  // nheavy + ceil(2 nlight / 4) workgroups at most = ntiles + ntiles / 2 + 1; surplus ones exit at once.
  // Light tiles: two waves per tile at 2 pixels per lane: the kernel then fits 128 VGPRs = 4 waves per
  // SIMD (one wave at 4 pixels per lane / 160 VGPRs and an uncapped allocation measured within 1 %;
  // 96 VGPRs with 36 spilled: 8 % slower).
  const int grid = ntiles + ntiles / 2 + 1 + (int)items_cap;
#ifndef GRPG_BWD_LIGHT_SPLIT   // experiment switch: 1 = one wave per light / mid tile at 4 pixels per lane
#define GRPG_BWD_LIGHT_SPLIT 2
#endif
  if (S <= 0) {
    render_backward_kernel<0, GRPG_BWD_LIGHT_SPLIT, 4><<<grid, 256, 0, s>>>(RB_ARGS);
  } else if (S <= 4) {
    render_backward_kernel<4, 2><<<grid, 256, 0, s>>>(RB_ARGS);
  } else {
    render_backward_kernel<32, 1><<<ntiles, 256, 0, s>>>(RB_ARGS);   // S <= 32 (reference: 20)
  }
#undef RB_ARGS
#ifdef GRPG_TRACE
  if (stats) {
    std::vector<unsigned long long> h(stats_words);
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(h.data(), stats_dev, stats_words * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    // Per tile class (by list length; how the launch walks it): tile instances, (wave, splat) reductions, and how
    // many lanes accept per reduction -- VERDICT r5 item 4: what has to be sized is the NUMBER of reductions.
    struct Cls { unsigned long long waves, inst4, reach, trips, used, rows, lanes, s1, s4, s8, cyc; };
    Cls c[4] = {};
    static const char* const cname[4] = {"<256 (2 half-tile waves)", "256-2047 (2 half-tile waves)",
                                         "2048-4095 (4 quarter waves)", ">=4096 (4 quarter waves per segment)"};
    unsigned long long cyc_max = 0, trips_of_longest = 0, reach_of_longest = 0;
    for (size_t w = 0; w < stats_words / 8; w++) {
      const unsigned long long* r = &h[8 * w];
      if (!r[0]) continue;
      const unsigned kind = (unsigned)(r[0] & 3ull);
      const unsigned long long len = r[0] >> 8;
      Cls& k = c[len < 256 ? 0 : (len < 2048 ? 1 : (len < 4096 ? 2 : 3))];
      k.waves++;
      // the wave's share of the list in quarter-instances: a quarter wave covers 1/4 of its entries' pixels, a
      // half-tile wave 2/4
      k.inst4 += (r[1] >> 32) * (kind == 1 ? 1ull : 2ull);
      k.reach += r[1] & 0xFFFFFFFFull; k.trips += r[3]; k.used += r[4];
      k.rows += r[5] & 0xFFFFFFFFull; k.lanes += r[5] >> 32; k.cyc += r[6];
      k.s1 += r[7] & 0xFFFFFull; k.s4 += (r[7] >> 20) & 0xFFFFFull; k.s8 += (r[7] >> 40) & 0xFFFFFull;
      if (r[6] > cyc_max) { cyc_max = r[6]; trips_of_longest = r[3]; reach_of_longest = r[1] & 0xFFFFFFFFull; }
    }
    unsigned long long tu = 0, ti = 0;
    for (int i = 0; i < 4; i++) {
      const double inst = (double)c[i].inst4 / 4.0;   // tile instances of the class
      tu += c[i].used; ti += c[i].inst4;
      fprintf(stderr, "[bwd stats] class %-38s waves %8llu  tile instances %10.0f  cull survivors (trips) %10llu  "
                      "reductions %10llu = %.3f per instance  lanes per reduction %.2f  (1 lane: %.3f, 2-4: %.3f, 5-8: %.3f)  "
                      "wave cycles %llu\n", cname[i], c[i].waves, inst, c[i].trips, c[i].used,
              inst > 0 ? (double)c[i].used / inst : 0.0, c[i].used ? (double)c[i].lanes / (double)c[i].used : 0.0,
              c[i].used ? (double)c[i].s1 / (double)c[i].used : 0.0, c[i].used ? (double)c[i].s4 / (double)c[i].used : 0.0,
              c[i].used ? (double)c[i].s8 / (double)c[i].used : 0.0, c[i].cyc);
    }
    fprintf(stderr, "[bwd stats] all classes: reductions %llu, tile instances %.0f, %.3f per instance; longest wave %llu cycles "
                    "(%llu trips, %llu entries in reach)\n", tu, (double)ti / 4.0, ti ? 4.0 * (double)tu / (double)ti : 0.0,
            cyc_max, trips_of_longest, reach_of_longest);
  }
#endif
}

}  // namespace grpg
