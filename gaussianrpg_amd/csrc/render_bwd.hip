// Backward of the per-tile blend for gfx950.
//
// Replaces renderCUDA<3,20> backward (cuda_rasterizer/backward.cu:415-641): back-to-front
// re-traversal of each tile's list starting from n_contrib, T recovered by division (:547),
// gradients w.r.t. colour / depth / semantic / accumulated alpha / background (:556-614) pushed to
// (mean2D, conic, opacity, colour, depth, semantic) of every contributing Gaussian (:618-638).
//
// CDNA4 mapping (same decomposition as the forward, render_fwd.hip, whose tile work lists in the
// image blob are reused):
//   * LIGHT tiles: one wave per tile, 4 pixels per lane.  HEAVY tiles: four 16x4 quarter-tile
//     waves at 1 pixel per lane (a heavy tile no longer serialises behind one wave).
//   * The list is walked back to front, only up to the deepest contributor of the wave's pixels,
//     through the forward's FILL / POP / process pipeline: FILL scans 256 entries per step and
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
#include <cstdlib>

#include "blend_math.h"
#include "common.h"

namespace grpg {

constexpr int RB_WAVES = 4;
constexpr int BQCAP = 512;     // ring capacity in entries (>= 64 + 256), power of two
constexpr int BFILL_Q = 4;     // list entries per lane per FILL step

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

// One wave, PX pixels per lane: rows y0 + (lane>>4)*PX + k.  bits_mask selects the sub-tile bits
// of a point-list entry that concern this wave (one bit for a quarter, all four for a whole tile).
template <int PX, int SMAX>
__device__ __forceinline__ void backward_rect(
    float4* __restrict__ my, uint32_t* __restrict__ qid, uint32_t* __restrict__ qpos,
    const int lane, const uint32_t r_begin, const uint32_t r_end,
    const int x0, const int y0, const uint32_t bits_mask, const int W, const int H, const int S,
    const uint32_t* __restrict__ point_list, const RecView rec,
    const float* __restrict__ semantics, const float* __restrict__ bg,
    const float* __restrict__ alphas, const uint32_t* __restrict__ n_contrib,
    const float* __restrict__ dL_dpix, const float* __restrict__ dL_dpix_depth,
    const float* __restrict__ dL_dalphas, const float* __restrict__ dL_dpix_semantic,
    float* __restrict__ dL_dmean2D, float* __restrict__ dL_dconic, float* __restrict__ dL_dopacity,
    float* __restrict__ dL_dcolor, float* __restrict__ dL_ddepth,
    float* __restrict__ dL_dsemantic, const int ablate) {
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

  // Per-pixel state.  The reference carries (last_alpha, last_color) and forms
  //   accum_rec = last_alpha * last_color + (1 - last_alpha) * accum_rec        (backward.cu:556-561)
  // at the NEXT contributor; the same value is obtained by blending the current contributor into
  // the accumulator right after it has been used (acc = alpha c + (1 - alpha) acc), which needs no
  // copy of the previous splat's colour per pixel.  Channels are kept in pairs (r,g) / (b,depth)
  // so that the updates are v_pk_* instructions.
  float T[PX], T_final[PX], dLa[PX], bgdot[PX], acc_a[PX];
  v2f dL_rg[PX], dL_bd[PX], acc_rg[PX], acc_bd[PX];
  float dLs[PX][SM], acc_s[PX][SM];
  uint32_t lastc[PX];
  uint32_t maxlast = 0;
#pragma unroll
  for (int k = 0; k < PX; k++) {
    const int py = py0 + k;
    const bool inside = px < W && py < H;
    const size_t pix = inside ? (size_t)py * W + px : 0;
    T_final[k] = inside ? 1.0f - alphas[pix] : 0.f;
    T[k] = T_final[k];
    lastc[k] = inside ? n_contrib[pix] : 0u;
    const float dLr = inside ? dL_dpix[pix] : 0.f;
    const float dLg = inside ? dL_dpix[HW + pix] : 0.f;
    const float dLb = inside ? dL_dpix[2 * HW + pix] : 0.f;
    const float dLd = inside ? dL_dpix_depth[pix] : 0.f;
    dL_rg[k] = (v2f){dLr, dLg};
    dL_bd[k] = (v2f){dLb, dLd};
    dLa[k] = inside ? dL_dalphas[pix] : 0.f;
    bgdot[k] = bg0 * dLr + bg1 * dLg + bg2 * dLb;
    acc_rg[k] = (v2f){0.f, 0.f};
    acc_bd[k] = (v2f){0.f, 0.f};
    acc_a[k] = 0.f;
#pragma unroll
    for (int c = 0; c < SM; c++) {
      dLs[k][c] = (SMAX > 0 && c < S && inside) ? dL_dpix_semantic[(size_t)c * HW + pix] : 0.f;
      acc_s[k][c] = 0.f;
    }
    maxlast = max(maxlast, lastc[k]);
  }
  maxlast = wave_max_u32(maxlast);   // nothing behind the tile's deepest contributor matters
  const float ddelx_dx = (float)(0.5 * W), ddely_dy = (float)(0.5 * H);   // backward.cu:501-502

  // Gradient scatter: after wave_sum_12 row r of the wave holds the totals of quantities 3r..3r+2
  //   row 0: dL_dmean2D x, y, |.|   row 1: dL_dconic xx, xy, yy   row 2: dL_dcolor r, g, b
  //   row 3: dL_dopacity, dL_ddepth, -
  // Lane 16 r + k takes quantity 3 r + k, so ONE global_atomic_add_f32 with 11 active lanes covers a
  // Gaussian: 5 cache lines per (wave, Gaussian) instead of 11 single-lane atomics.
  const int row = lane >> 4, sub = lane & 15;
  float* sc_ptr = nullptr;
  uint32_t sc_stride = 0;
  if (sub < 3) {
    if (row == 0) { sc_ptr = dL_dmean2D + sub; sc_stride = 3; }
    else if (row == 1) { sc_ptr = dL_dconic + (sub == 2 ? 3 : sub); sc_stride = 4; }
    else if (row == 2) { sc_ptr = dL_dcolor + sub; sc_stride = 3; }
    else if (sub < 2) { sc_ptr = sub == 0 ? dL_dopacity : dL_ddepth; sc_stride = 1; }
  }

  // Back-to-front traversal of list positions [0, count), decoupled like the forward's heavy path
  // (render_fwd.hip): FILL scans 256 entries per step (prefetched one window ahead) and appends
  // the ones whose sub-tile mask concerns this wave to an LDS ring, deepest first; POP takes up to
  // 64 of them and starts the gather of their records; the PREVIOUS batch is culled, compacted and
  // processed while that gather is in flight.  A dead entry costs 1/256 of a FILL step and no
  // record is loaded for it; in a horizon tile only one entry in six concerns a given quarter.
  const uint64_t lt = lanemask_lt();
  const uint32_t count = min(r_end - r_begin, maxlast);
  uint32_t in_hi = count;           // list positions [0, in_hi) are still unread (wave-uniform)
  uint32_t head = 0, rcount = 0;    // ring state                                (wave-uniform)
  uint32_t win[BFILL_Q];
#pragma unroll
  for (int q = 0; q < BFILL_Q; q++) {
    const uint32_t off = (uint32_t)(q * WAVE + lane);
    win[q] = off < in_hi ? point_list[r_begin + in_hi - 1 - off] : 0u;
  }
  float4 la = make_float4(0, 0, 0, 0), lb = la, lc = la;   // current batch (records arrived)
  uint32_t lpos = 0, lid = 0, ncur = 0;
  for (;;) {
    // ---- FILL ----
    while (rcount < (uint32_t)WAVE && in_hi > 0) {
      uint32_t v[BFILL_Q];
#pragma unroll
      for (int q = 0; q < BFILL_Q; q++) v[q] = win[q];
      const uint32_t nxt = in_hi > (uint32_t)(BFILL_Q * WAVE) ? in_hi - BFILL_Q * WAVE : 0u;
#pragma unroll
      for (int q = 0; q < BFILL_Q; q++) {
        const uint32_t off = (uint32_t)(q * WAVE + lane);
        win[q] = off < nxt ? point_list[r_begin + nxt - 1 - off] : 0u;
      }
#pragma unroll
      for (int q = 0; q < BFILL_Q; q++) {
        const uint32_t off = (uint32_t)(q * WAVE + lane);
        const bool keep = (off < in_hi) && (v[q] & bits_mask);
        const uint64_t m = __ballot(keep);
        if (keep) {
          const uint32_t slot = (head + rcount + (uint32_t)__popcll(m & lt)) & (BQCAP - 1);
          qid[slot] = v[q] & ID_MASK;
          qpos[slot] = in_hi - 1 - off;   // 0-based list position == the reference's `contributor`
        }
        rcount += (uint32_t)__popcll(m);
      }
      in_hi = nxt;
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
        float bx0 = 3e38f, bx1 = -3e38f, by0 = 3e38f, by1 = -3e38f;
#pragma unroll
        for (int k = 0; k < PX; k++) {
          if (lastc[k] > pos_min) {
            bx0 = pxf; bx1 = pxf;
            by0 = fminf(by0, (float)(py0 + k));
            by1 = fmaxf(by1, (float)(py0 + k));
          }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
          bx0 = fminf(bx0, __shfl_xor(bx0, d, 64));
          bx1 = fmaxf(bx1, __shfl_xor(bx1, d, 64));
          by0 = fminf(by0, __shfl_xor(by0, d, 64));
          by1 = fmaxf(by1, __shfl_xor(by1, d, 64));
        }
        rx0 = bx0; rx1 = bx1; ry0 = by0; ry1 = by1;
      }
      const bool keep = ((uint32_t)lane < ncur) &&
                        !splat_misses_rect(la.x, la.y, lb.x, lb.y, lb.z, la.w, rx0, rx1, ry0, ry1);
      const uint64_t mask = __ballot(keep);
      cnt = (int)__popcll(mask);
      if (keep) {
        const int slot = (int)__popcll(mask & lt);
        my[slot * REC_F4 + 0] = la;
        my[slot * REC_F4 + 1] = lb;
        my[slot * REC_F4 + 2] = make_float4(lc.x, lc.y, __uint_as_float(lpos), __uint_as_float(lid));
      }
    }
    __builtin_amdgcn_wave_barrier();
    for (int j = 0; j < cnt; j++) {
      const float4 a = my[j * REC_F4 + 0];   // px, py, depth, opacity
      const float4 b = my[j * REC_F4 + 1];   // conic.x, conic.y, conic.z, R
      const float4 c = my[j * REC_F4 + 2];   // G, B, list position, Gaussian id
      const uint32_t pos = __float_as_uint(c.z);   // 0-based == the reference's `contributor`
      const uint32_t gid = __float_as_uint(c.w);
      const float dx = a.x - pxf;
      const SplatTerms st = splat_terms(dx, b.x, b.y, b.z);
      // packed accumulators: (mean2D x, y), (conic xx, yy), (colour r, g), (colour b, depth)
      v2f g_mxy = {0.f, 0.f}, g_cxw = {0.f, 0.f}, g_rg = {0.f, 0.f}, g_bd = {0.f, 0.f};
      float g_mabs = 0.f, g_cy = 0.f, g_op = 0.f;
      float g_s[SM];
#pragma unroll
      for (int cc = 0; cc < SM; cc++) g_s[cc] = 0.f;
      const v2f col_rg = {b.w, c.x}, col_bd = {c.y, a.z};
      const v2f con_xz = {b.x, b.z}, ncy2 = {-b.y, -b.y};
      const v2f ddel = {ddelx_dx, ddely_dy};
      bool any = false;
#pragma unroll
      for (int k = 0; k < PX; k++) {
        const float dy = a.y - (float)(py0 + k);
        float G, alpha;   // identical arithmetic to the forward (blend_math.h)
        const bool valid = pair_alpha(pair_power(st, dy), a.w, G, alpha) && (pos < lastc[k]) &&
                           !(ablate & 4);
        if (valid) {
          any = true;
          // one reciprocal serves both divisions of backward.cu:547,596 (1 ulp: the recovered T is
          // the inverse of a rounded product chain anyway)
          const float om = 1.f - alpha;
          const float inv_1ma = __builtin_amdgcn_rcpf(om);
          T[k] = T[k] * inv_1ma;
          const float dch = alpha * T[k];
          const v2f dch2 = {dch, dch}, al2 = {alpha, alpha}, om2 = {om, om};
          // dL/dalpha through the colours behind this splat (acc = what lies behind, blended)
          const v2f d1 = (col_rg - acc_rg[k]) * dL_rg[k];
          const v2f d2 = (col_bd - acc_bd[k]) * dL_bd[k];
          const v2f ds = d1 + d2;
          float dL_dopa = ds.x + ds.y;
          g_rg = __builtin_elementwise_fma(dch2, dL_rg[k], g_rg);
          g_bd = __builtin_elementwise_fma(dch2, dL_bd[k], g_bd);
          acc_rg[k] = __builtin_elementwise_fma(al2, col_rg, om2 * acc_rg[k]);
          acc_bd[k] = __builtin_elementwise_fma(al2, col_bd, om2 * acc_bd[k]);
          if (SMAX > 0) {
#pragma unroll
            for (int cc = 0; cc < SM; cc++) {
              if (cc < S) {
                const float sv = semantics[(size_t)gid * S + cc];
                dL_dopa += (sv - acc_s[k][cc]) * dLs[k][cc];
                g_s[cc] += dch * dLs[k][cc];
                acc_s[k][cc] = alpha * sv + om * acc_s[k][cc];
              }
            }
          }
          dL_dopa += (1.f - acc_a[k]) * dLa[k];
          acc_a[k] = alpha + om * acc_a[k];
          dL_dopa *= T[k];
          dL_dopa += (-T_final[k] * inv_1ma) * bgdot[k];
          const float dL_dG = a.w * dL_dopa;
          const v2f d2v = {dx, dy};
          const v2f gd = (v2f){G, G} * d2v;                                   // (G dx, G dy)
          // dG/ddelx = -gdx cx - gdy cy ;  dG/ddely = -gdy cz - gdx cy
          const v2f dG = __builtin_elementwise_fma((v2f){gd.y, gd.x}, ncy2, -(gd * con_xz));
          const v2f mxy = ((v2f){dL_dG, dL_dG} * dG) * ddel;
          g_mxy += mxy;
          g_mabs += fabsf(mxy.x) + fabsf(mxy.y);
          const float h = -0.5f * dL_dG;
          g_cxw = __builtin_elementwise_fma((v2f){h, h}, gd * d2v, g_cxw);    // (gdx dx, gdy dy)
          g_cy = fmaf(h * gd.x, dy, g_cy);
          g_op = fmaf(G, dL_dopa, g_op);
        }
      }
      if (__ballot(any) == 0ull) continue;   // wave-uniform: nobody in the tile used this splat
      if (ablate & 2) continue;   // experiment switch (GRPG_BWD_ABLATE): no reduction, no atomics
      {
        const float q[12] = {g_mxy.x, g_mxy.y, g_mabs, g_cxw.x, g_cy, g_cxw.y, g_rg.x, g_rg.y, g_bd.x, g_op,
                             g_bd.y, 0.f};
        float tot[3];
        wave_sum_12(q, tot);
        if (sc_ptr && !(ablate & 1))
          atomicAdd(sc_ptr + (size_t)gid * sc_stride, sub == 0 ? tot[0] : (sub == 1 ? tot[1] : tot[2]));
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
    if (ncur == 0 && in_hi == 0) break;   // ring empty (rcount == 0 here) and list exhausted
  }
}

// Work lists: written by the forward's classify_tiles_kernel (render_fwd.hip) into the image blob:
// counts[4] (three heavy classes, light), then four lists of T tile ids.
constexpr int NUM_CLASSES_B = 4;

// MINW: waves per SIMD the register allocator must fit (4 -> 128 VGPRs, 24 B of scratch per lane in
// the S = 0 / two-pixel-light variant; 1 -> whatever it takes: 138 VGPRs, 3 waves per SIMD)
template <int SMAX, int LIGHT_SPLIT, int MINW = 1>
__global__ void __launch_bounds__(256, MINW)
render_backward_kernel(const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                       const RecView rec, const float* __restrict__ semantics,
                       const int S, const int W, const int H, const int gx, const uint32_t T,
                       const uint32_t* __restrict__ work, const float* __restrict__ bg,
                       const float* __restrict__ alphas, const uint32_t* __restrict__ n_contrib,
                       const float* __restrict__ dL_dpix, const float* __restrict__ dL_dpix_depth,
                       const float* __restrict__ dL_dalphas,
                       const float* __restrict__ dL_dpix_semantic, float* __restrict__ dL_dmean2D,
                       float* __restrict__ dL_dconic, float* __restrict__ dL_dopacity,
                       float* __restrict__ dL_dcolor, float* __restrict__ dL_ddepth,
                       float* __restrict__ dL_dsemantic, const int ablate, const int wide_classes) {
  __shared__ float4 s_rec[RB_WAVES][WAVE * REC_F4];
  __shared__ uint32_t s_qid[RB_WAVES][BQCAP];
  __shared__ uint32_t s_qpos[RB_WAVES][BQCAP];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // wide_classes = k: the k shortest heavy classes of the forward's classification are walked like
  // light tiles here (one wave per tile, 4 pixels per lane) -- one reduction + atomic per
  // (tile, splat) instead of four
  const uint32_t n0 = work[0], n1 = work[1];
  const uint32_t n2 = wide_classes >= 1 ? 0u : work[2];
  const uint32_t nmid = wide_classes >= 1 ? work[2] : 0u;
  const uint32_t nlight = work[3] + nmid;
  const uint32_t nheavy = n0 + n1 + n2;
  const uint32_t b = blockIdx.x;
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
#define RB_CALL(PXV, YOFF, BITS)                                                                  \
  backward_rect<PXV, SMAX>(s_rec[wave], s_qid[wave], s_qpos[wave], lane, rb, re, tx * TILE, ty * TILE + (YOFF), (BITS), W, H, \
                           S, point_list, rec, semantics, bg, alphas, n_contrib, dL_dpix,          \
                           dL_dpix_depth, dL_dalphas, dL_dpix_semantic, dL_dmean2D, dL_dconic,     \
                           dL_dopacity, dL_dcolor, dL_ddepth, dL_dsemantic, ablate)
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
                            const float* dL_dpix_semantic, float* dL_dmean2D, float* dL_dconic,
                            float* dL_dopacity, float* dL_dcolor, float* dL_ddepth,
                            float* dL_dsemantic) {
  const int ntiles = gx * gy;
  if (ntiles <= 0) return;
  // experiment switch: 1 = no atomics, 2 = no reduction either, 4 = traversal + alpha only
  static const int ablate = [] { const char* e = getenv("GRPG_BWD_ABLATE"); return e ? atoi(e) : 0; }();
  static const int wide = [] { const char* e = getenv("GRPG_BWD_WIDE"); return e ? atoi(e) : 0; }();
#define RB_ARGS                                                                                  \
  ranges, point_list, rec, semantics, S, W, H, gx, (uint32_t)ntiles, work, bg, alphas, n_contrib, \
      dL_dpix, dL_dpix_depth, dL_dalphas, dL_dpix_semantic, dL_dmean2D, dL_dconic, dL_dopacity,  \
      dL_dcolor, dL_ddepth, dL_dsemantic, ablate, wide
  // nheavy + ceil(LIGHT_SPLIT nlight / 4) <= ntiles for LIGHT_SPLIT <= 2 ... not for nheavy ~ ntiles/2:
  // launch nheavy_max + light workgroups = ntiles + ntiles/2 + 1, surplus workgroups exit at once.
  // Light tiles: one wave per tile, 4 pixels per lane (default).  GRPG_BWD_LIGHT=2 selects two
  // waves per light tile at 2 pixels per lane (138 VGPRs / 3 waves per SIMD instead of 178 / 2, or
  // 128 / 4 with GRPG_BWD_WAVES=4): measured SLOWER (1.25 / 1.36 vs 1.14 ms at config 5) -- the
  // second wave repeats the tile's list traversal, which outweighs the occupancy.
  static const int light4 = [] { const char* e = getenv("GRPG_BWD_LIGHT"); return !(e && atoi(e) == 2); }();
  const int grid = light4 ? ntiles : ntiles + ntiles / 2 + 1;
  if (S <= 0) {
    static const int minw = [] { const char* e = getenv("GRPG_BWD_WAVES"); return e ? atoi(e) : 1; }();
    if (light4) render_backward_kernel<0, 1><<<grid, 256, 0, s>>>(RB_ARGS);
    else if (minw >= 4) render_backward_kernel<0, 2, 4><<<grid, 256, 0, s>>>(RB_ARGS);
    else render_backward_kernel<0, 2><<<grid, 256, 0, s>>>(RB_ARGS);
  } else if (S <= 4) {
    if (light4) render_backward_kernel<4, 1><<<grid, 256, 0, s>>>(RB_ARGS);
    else render_backward_kernel<4, 2><<<grid, 256, 0, s>>>(RB_ARGS);
  } else {
    render_backward_kernel<32, 1><<<ntiles, 256, 0, s>>>(RB_ARGS);   // S <= 32 (reference: 20)
  }
#undef RB_ARGS
}

}  // namespace grpg
