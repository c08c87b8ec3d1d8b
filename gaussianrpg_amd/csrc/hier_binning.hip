// Hierarchical tile binning for gfx950: an alternative to "emit R (tile, value) pairs + stable
// two-pass partition" (binning.hip / sort.hip) that never materialises 8-byte pairs for the R
// instances.  Same result, bit for bit: per tile the (depth_bits, id)-ordered list of the reference's
// duplicateWithKeys + SortPairs + identifyTileRanges (cuda_rasterizer/rasterizer_impl.cu:70-138,
// 306-321), `num_rendered` = sum of the tile rectangles.
//
//   coarse emit   one entry per (Gaussian, SUPER-TILE) -- a super-tile is 8x8 tiles -- in depth
//                 order: Rc ~ 0.2 R entries (emit_kernel<true> in binning.hip, slot-parallel)
//   coarse sort   stable partition of the Rc entries by super-tile id (<= 8 bits at 1920x1280:
//                 ONE radix pass whose histogram the coarse emit leaves behind)
//   segments      every super-tile's run is cut into segments of 1024 entries (numbered by the
//                 count kernel's workgroups themselves from the partition's digit totals; a
//                 separate setup launch only beyond 256 super-tiles)
//   count         per segment: how many of its Gaussians touch each of the super-tile's 64 tiles
//   prefix, scan  per tile the exclusive prefix over its super-tile's segments; exclusive scan
//                 over all tiles = the tile ranges, total = num_rendered; the same launch sorts
//                 the tiles into the render's work lists
//   fill          per segment and tile: the touching Gaussians, in order, go to
//                 point_list[tile_start + segment prefix ...] with their quarter-reach mask (a
//                 wave per tile COLUMN: the splat's extent over the column strip gives the masks
//                 of the column's 8 tiles at once)
//
// Order argument: the coarse sort is stable from depth order, segments are consecutive pieces of a
// super-tile's run, and inside a segment the entries of a tile are written in segment order: the
// list of a tile is in (depth_bits, id) order like the sorted 64-bit keys of the reference.
// A coarse key carries, above the super-tile id, the 8-bit column and row masks of the tiles of that
// super-tile the Gaussian's rectangle covers: count and fill never look at the rectangle again.
// HBM traffic: 8 B x Rc x 4 (coarse emit / partition / count, fill) + one record line per coarse
// entry in the fill (quarter masks) + 4 B x R (the point list).
#include "blend_math.h"
#include "common.h"

#ifndef GRPG_FILL_ABLATE   // experiment builds only
#define GRPG_FILL_ABLATE 0
#endif

namespace grpg {

#ifndef GRPG_HB_SEG   // experiment builds: 512
#define GRPG_HB_SEG 1024
#endif
constexpr int HB_SEG = GRPG_HB_SEG;  // coarse entries per segment
constexpr int HB_CNT_THREADS = 256;
constexpr int HB_FILL_THREADS = 512;   // 8 waves: one per tile column of the super-tile

struct SegDesc { uint32_t st, begin, end, pad; };

// ---- segments: cranges[st] (run of super-tile st in the sorted coarse list) -> descriptors.
// Stand-alone launch for grids of more than 256 super-tiles (hb_count_kernel<true> does this
// itself otherwise). ----
__global__ void __launch_bounds__(1024)
hb_seg_setup_kernel(const uint2* __restrict__ cranges, const uint32_t NS, SegDesc* __restrict__ seg,
                    uint2* __restrict__ st_seg /* [NS] (first segment, count) */,
                    uint32_t* __restrict__ nseg_total, const uint32_t max_seg) {
  __shared__ uint32_t s_w[16];
  __shared__ uint32_t s_carry;
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (uint32_t base = 0; base < NS; base += 1024) {
    const uint32_t st = base + tid;
    uint2 r = make_uint2(0u, 0u);
    if (st < NS) r = cranges[st];
    const uint32_t ns = (r.y - r.x + HB_SEG - 1) / HB_SEG;
    uint32_t inc = ns;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t t = __shfl_up(inc, d, 64);
      if (lane >= (uint32_t)d) inc += t;
    }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    uint32_t before = s_carry;
    for (uint32_t w = 0; w < wave; w++) before += s_w[w];
    if (st < NS) st_seg[st] = make_uint2(before + inc - ns, ns);
    __syncthreads();
    if (tid == 1023) s_carry = before + inc;
    __syncthreads();
  }
  const uint32_t total = min(s_carry, max_seg);
  if (tid == 0) *nseg_total = total;
  __threadfence_block();
  __syncthreads();
  // descriptors, one thread per segment: the super-tile is the LAST one whose first segment is
  // <= the segment's index (super-tiles without segments share their first index with a successor)
  for (uint32_t q = tid; q < total; q += 1024) {
    uint32_t lo = 0, hi = NS - 1;
    while (lo < hi) {
      const uint32_t mid = lo + (hi - lo + 1) / 2;
      if (st_seg[mid].x <= q) lo = mid; else hi = mid - 1;
    }
    const uint2 r = cranges[lo];
    const uint32_t k = q - st_seg[lo].x;
    SegDesc d;
    d.st = lo; d.begin = r.x + k * HB_SEG; d.end = min(r.y, r.x + (k + 1) * HB_SEG); d.pad = 0u;
    seg[q] = d;
  }
}

// ---- count: table[seg][64] = Gaussians of the segment touching each tile of its super-tile ----
// SETUP (<= 256 super-tiles, one-pass coarse partition): there is no segment-setup launch; every
// workgroup derives the super-tile runs and the segment numbering itself from the partition's digit
// totals (1 KB, one scan in wave 0), finds its own segment and leaves the descriptor behind for the
// fill; workgroup 0 also publishes the runs, the per-super-tile segment ranges and the count.
template <bool SETUP>
__global__ void __launch_bounds__(HB_CNT_THREADS)
hb_count_kernel(SegDesc* __restrict__ seg, uint32_t* __restrict__ nseg_total,
                const uint32_t* __restrict__ ckey, uint32_t* __restrict__ table,
                const uint32_t* __restrict__ run_totals, const uint32_t NS, uint2* __restrict__ cranges,
                uint2* __restrict__ st_seg, const uint32_t max_seg) {
  __shared__ uint16_t s_m[HB_SEG];
  __shared__ uint32_t s_run[256 + 1], s_first[256 + 1];
  const uint32_t sid = blockIdx.x;
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  SegDesc d;
  if (SETUP) {
    if (wave == 0) {
      uint32_t len[4], ns[4], slen = 0, sns = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t st = lane * 4 + k;
        len[k] = st < NS ? run_totals[st] : 0u;
        ns[k] = (len[k] + HB_SEG - 1) / HB_SEG;
        slen += len[k]; sns += ns[k];
      }
      uint32_t ilen = slen, ins = sns;
#pragma unroll
      for (int dd = 1; dd < 64; dd <<= 1) {
        const uint32_t a = __shfl_up(ilen, dd, 64), b = __shfl_up(ins, dd, 64);
        if (lane >= (uint32_t)dd) { ilen += a; ins += b; }
      }
      uint32_t rs = ilen - slen, fs = ins - sns;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        s_run[lane * 4 + k] = rs; s_first[lane * 4 + k] = fs;
        rs += len[k]; fs += ns[k];
      }
      if (lane == 63) { s_run[256] = rs; s_first[256] = fs; }
    }
    __syncthreads();
    const uint32_t total = min(s_first[256], max_seg);
    if (sid == 0) {
      if (tid == 0) *nseg_total = total;
      if (tid < NS) {
        const uint32_t b = s_run[tid], e = s_run[tid + 1];
        cranges[tid] = e > b ? make_uint2(b, e) : make_uint2(0u, 0u);
        st_seg[tid] = make_uint2(s_first[tid], s_first[tid + 1] - s_first[tid]);
      }
    }
    if (sid >= total) return;
    // the LAST super-tile whose first segment is <= sid (empty ones share theirs with a successor)
    uint32_t lo = 0, hi = NS - 1;
    while (lo < hi) {
      const uint32_t mid = lo + (hi - lo + 1) / 2;
      if (s_first[mid] <= sid) lo = mid; else hi = mid - 1;
    }
    const uint32_t k = sid - s_first[lo];
    d.st = lo; d.begin = s_run[lo] + k * HB_SEG; d.end = min(s_run[lo + 1], d.begin + HB_SEG); d.pad = 0u;
    if (tid == 0) seg[sid] = d;
  } else {
    if (sid >= *nseg_total) return;
    d = seg[sid];
  }
  const uint32_t n = d.end - d.begin;
#pragma unroll
  for (int i = 0; i < HB_SEG / HB_CNT_THREADS; i++) {
    const uint32_t e = i * HB_CNT_THREADS + tid;
    s_m[e] = e < n ? (uint16_t)(ckey[d.begin + e] >> 16) : (uint16_t)0;
  }
  __syncthreads();
  // wave w counts tile rows 2w and 2w+1: every lane walks 16 entries and keeps the 8 column
  // counters of a row as bytes of two words (a lane adds at most 16: no overflow)
#pragma unroll
  for (int rr = 0; rr < 2; rr++) {
    const int row = (int)wave * 2 + rr;
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int r = 0; r < HB_SEG / WAVE; r++) {
      const uint32_t m = s_m[r * WAVE + lane];
      const uint32_t cm = ((m >> (8 + row)) & 1u) ? (m & 0xFFu) : 0u;
      lo += ((cm & 0xFu) * 0x00204081u) & 0x01010101u;          // bits 0-3 -> bytes 0-3
      hi += (((cm >> 4) & 0xFu) * 0x00204081u) & 0x01010101u;   // bits 4-7 -> bytes 0-3
    }
#pragma unroll
    for (int tx = 0; tx < STILE; tx++) {
      uint32_t c = ((tx < 4 ? lo : hi) >> (8 * (tx & 3))) & 0xFFu;
#pragma unroll
      for (int s = 32; s >= 1; s >>= 1) c += (uint32_t)__shfl_xor((int)c, s, 64);
      if ((int)lane == tx) table[(size_t)sid * STILE_TILES + row * STILE + tx] = c;
    }
  }
}

// ---- per tile: exclusive prefix over the segments of its super-tile (in place) + tile total ----
// One workgroup per super-tile, lane = tile; the 16 waves each take a contiguous share of the
// segments (sum it, exchange, then write the prefixes): a super-tile with hundreds of segments
// costs a few dependent batches, not hundreds of dependent loads.
__global__ void __launch_bounds__(1024)
hb_tile_prefix_kernel(const uint2* __restrict__ st_seg, uint32_t* __restrict__ table, const int sgx,
                      const int gx, const int gy, uint32_t* __restrict__ tile_tot) {
  __shared__ uint32_t s_part[16][STILE_TILES];
  const uint32_t st = blockIdx.x, l = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint2 sg = st_seg[st];
  const uint32_t per = (sg.y + 15u) / 16u;
  const uint32_t q0 = min(sg.y, wave * per), q1 = min(sg.y, q0 + per);
  uint32_t sum = 0;
  for (uint32_t q = q0; q < q1; q++) sum += table[(size_t)(sg.x + q) * STILE_TILES + l];
  s_part[wave][l] = sum;
  __syncthreads();
  uint32_t run = 0, tot = 0;
#pragma unroll
  for (uint32_t w = 0; w < 16; w++) {
    const uint32_t v = s_part[w][l];
    if (w < wave) run += v;
    tot += v;
  }
  for (uint32_t qb = q0; qb < q1; qb += 8) {
    uint32_t v[8];
#pragma unroll
    for (uint32_t j = 0; j < 8; j++)
      v[j] = qb + j < q1 ? table[(size_t)(sg.x + qb + j) * STILE_TILES + l] : 0u;
#pragma unroll
    for (uint32_t j = 0; j < 8; j++) {
      if (qb + j < q1) table[(size_t)(sg.x + qb + j) * STILE_TILES + l] = run;
      run += v[j];
    }
  }
  if (wave == 0) {
    const int sy = (int)(st / (uint32_t)sgx), sx = (int)(st - (uint32_t)sy * (uint32_t)sgx);
    const int tx = sx * STILE + (int)(l & 7u), ty = sy * STILE + (int)(l >> 3);
    if (tx < gx && ty < gy) tile_tot[(size_t)ty * gx + tx] = tot;
  }
}

// ---- exclusive scan over all tiles -> ranges (untouched tiles stay (0,0) like the reference's
// memset + identifyTileRanges), monotone tile_start[] for the fill, num_rendered; and, since every
// tile's length passes through here, the render's work lists (render_fwd.hip: classify_tiles_kernel
// does the same from the ranges; slots come from LDS counters here, no global atomics) ----
__global__ void __launch_bounds__(1024)
hb_tile_scan_kernel(const uint32_t T, const uint32_t* __restrict__ tile_tot,
                    uint32_t* __restrict__ tile_start, uint2* __restrict__ ranges,
                    uint32_t* __restrict__ R_out, uint32_t* __restrict__ host_word,
                    const uint32_t* __restrict__ Rc_dev, BlobHeader* __restrict__ bin_header,
                    const uint32_t R_cap, const uint32_t coarse_cap, uint32_t* __restrict__ work,
                    const TileClasses tc) {
  __shared__ uint32_t s_w[16];
  __shared__ uint32_t s_carry;
  __shared__ uint32_t s_cls[4];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_carry = 0;
  if (tid < 4) s_cls[tid] = 0;
  __syncthreads();
  constexpr int PER = 4;   // one 16-byte load / store per lane: a single CU must not waste lines
  for (uint32_t base = 0; base < T; base += 1024 * PER) {
    uint32_t v[PER], s = 0;
    const uint32_t t0 = base + tid * PER;
    const bool full = t0 + PER <= T;
    if (full) {
      const uint4 q = *reinterpret_cast<const uint4*>(tile_tot + t0);
      v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
#pragma unroll
      for (int k = 0; k < PER; k++) v[k] = t0 + k < T ? tile_tot[t0 + k] : 0u;
    }
#pragma unroll
    for (int k = 0; k < PER; k++) s += v[k];
    uint32_t inc = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t u = __shfl_up(inc, d, 64);
      if (lane >= (uint32_t)d) inc += u;
    }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    uint32_t ex = s_carry + inc - s;
    for (uint32_t w = 0; w < wave; w++) ex += s_w[w];
    if (full) {   // starts and ranges of the lane's four tiles: 16 + 32 bytes, stored whole
      uint32_t st[PER + 1];
      st[0] = ex;
#pragma unroll
      for (int k = 0; k < PER; k++) st[k + 1] = st[k] + v[k];
      *reinterpret_cast<uint4*>(tile_start + t0) = make_uint4(st[0], st[1], st[2], st[3]);
      uint32_t r[2 * PER];
#pragma unroll
      for (int k = 0; k < PER; k++) {
        // clamped to the point list's capacity: on an overflow the frame's tail is redone, but
        // the render enqueued behind this launch must stay inside the list
        r[2 * k] = v[k] ? min(st[k], R_cap) : 0u;
        r[2 * k + 1] = v[k] ? min(st[k + 1], R_cap) : 0u;
      }
      uint4* rp = reinterpret_cast<uint4*>(ranges + t0);
      rp[0] = make_uint4(r[0], r[1], r[2], r[3]);
      rp[1] = make_uint4(r[4], r[5], r[6], r[7]);
    }
    // work-list class of each tile from its (clamped) list length; slots: per class one LDS counter
    // bump per wave (lane c for class c), positions inside the wave from a packed lane scan
    int cls[PER];
    uint32_t n01 = 0, n23 = 0;   // per-lane class counts, 16-bit fields: (class 0, 1), (class 2, 3)
#pragma unroll
    for (int k = 0; k < PER; k++) {
      const uint32_t t = t0 + k;
      if (!full && t < T) {
        tile_start[t] = ex;
        ranges[t] = v[k] ? make_uint2(min(ex, R_cap), min(ex + v[k], R_cap)) : make_uint2(0u, 0u);
      }
      const uint32_t len = min(ex + v[k], R_cap) - min(ex, R_cap);
      cls[k] = t >= T ? -1 : tile_class_of(t, len, tc);
      n01 += cls[k] == 0 ? 1u : (cls[k] == 1 ? 0x10000u : 0u);
      n23 += cls[k] == 2 ? 1u : (cls[k] == 3 ? 0x10000u : 0u);
      ex += v[k];
    }
    uint32_t i01 = n01, i23 = n23;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t a = __shfl_up(i01, d, 64), b = __shfl_up(i23, d, 64);
      if (lane >= (uint32_t)d) { i01 += a; i23 += b; }
    }
    const uint32_t tot01 = __shfl(i01, 63, 64), tot23 = __shfl(i23, 63, 64);
    uint32_t slot = 0;
    if (lane < 4) {
      const uint32_t mine = lane == 0 ? tot01 & 0xFFFFu : (lane == 1 ? tot01 >> 16 : (lane == 2 ? tot23 & 0xFFFFu : tot23 >> 16));
      if (mine) slot = atomicAdd(&s_cls[lane], mine);
    }
    uint32_t pos[4];
    pos[0] = __shfl(slot, 0, 64) + ((i01 - n01) & 0xFFFFu);
    pos[1] = __shfl(slot, 1, 64) + ((i01 - n01) >> 16);
    pos[2] = __shfl(slot, 2, 64) + ((i23 - n23) & 0xFFFFu);
    pos[3] = __shfl(slot, 3, 64) + ((i23 - n23) >> 16);
#pragma unroll
    for (int k = 0; k < PER; k++) {
#pragma unroll
      for (int c = 0; c < 4; c++)
        if (cls[k] == c) { work[4 + (size_t)c * T + pos[c]] = t0 + k; pos[c]++; }
    }
    __syncthreads();
    if (tid == 1023) s_carry = ex;
    __syncthreads();
  }
  if (tid < 4) work[tid] = s_cls[tid];
  if (tid == 0) {
    *R_out = s_carry;
    if (bin_header) { bin_header->hier = 1u; bin_header->Rc = coarse_cap; }
    if (host_word) { host_word[0] = s_carry; host_word[1] = Rc_dev ? *Rc_dev : 0u; }
  }
}

// ---- fill ----
// Ids, masks and the splat-only part of the quarter-reach test of a segment go to LDS (one record
// line per entry).  Wave w owns tile COLUMN w of the super-tile: it compacts the entries whose
// rectangle reaches the column (in order), then walks that list 64 entries at a time: the y-extent
// of the splat over the column strip is evaluated once, and for each of the column's 8 tiles the
// lanes whose entry covers the tile derive the quarter mask from it and store to consecutive
// positions of the tile's list (ballot prefix).
// Workgroups are persistent and walk the segments with a stride: the descriptor and the (key, id)
// pairs of the NEXT segment are fetched while the current one is processed, so a segment exposes
// one memory round trip (the record gather) instead of three.
#ifdef GRPG_FILL_TRACE   // experiment build (tools/fill_trace.py): per workgroup start / end, segments, instances
__device__ unsigned long long g_fill_trace[1024][4];
__device__ unsigned long long g_fill_phase[1024][8];   // wave 0's accumulated ticks per phase
#endif

__global__ void __launch_bounds__(HB_FILL_THREADS, HB_SEG <= 512 ? 8 : 1)
hb_fill_kernel(const SegDesc* __restrict__ seg, const uint32_t* __restrict__ nseg_total,
               const uint32_t* __restrict__ ckey, const uint32_t* __restrict__ cval, const RecView rec,
               const int sgx, const int gx, const int gy, const uint32_t* __restrict__ table,
               const uint32_t* __restrict__ tile_start, const uint32_t R_cap,
               uint32_t* __restrict__ point_list) {
  __shared__ uint32_t s_gid[HB_SEG];
  __shared__ uint16_t s_m[HB_SEG];
  __shared__ uint8_t s_fl[HB_SEG];       // bit 0: sane, bit 1: transparent
  __shared__ float2 s_xy[HB_SEG];        // px, py
  __shared__ float2 s_bi[HB_SEG];        // conic b, 1 / conic c
  __shared__ float2 s_dc[HB_SEG];        // det, c t
  __shared__ float s_xt[HB_SEG];         // xtop
  __shared__ uint16_t s_col[STILE][HB_SEG];   // per wave: entries reaching its column
  constexpr int ITEMS = HB_SEG / HB_FILL_THREADS;   // 2
  const uint32_t nseg = *nseg_total;
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint64_t lt = (1ull << lane) - 1ull;
  const int col = (int)wave;
  uint16_t* coll = s_col[wave];
  uint32_t sid = blockIdx.x;
#ifdef GRPG_FILL_TRACE
  const unsigned long long tr_t0 = wall_clock64();
  unsigned long long tr_segs = 0, tr_inst = 0;
  if (tid == 0 && blockIdx.x < 1024) {
    for (int k = 0; k < 8; k++) g_fill_phase[blockIdx.x][k] = 0ull;
    g_fill_trace[blockIdx.x][0] = tr_t0; g_fill_trace[blockIdx.x][1] = tr_t0; g_fill_trace[blockIdx.x][2] = 0; g_fill_trace[blockIdx.x][3] = 0;
  }
#endif
  if (sid >= nseg) return;
  SegDesc d = seg[sid];
  uint32_t gid[ITEMS], key[ITEMS];
#pragma unroll
  for (int i = 0; i < ITEMS; i++) {
    const uint32_t e = i * HB_FILL_THREADS + tid;
    const bool on = e < d.end - d.begin;
    gid[i] = on ? cval[d.begin + e] : 0u;
    key[i] = on ? ckey[d.begin + e] : 0u;
  }
  while (true) {
    const uint32_t n = d.end - d.begin;
    const int sy = (int)(d.st / (uint32_t)sgx), sx = (int)(d.st - (uint32_t)sy * (uint32_t)sgx);
    const int tx = sx * STILE + col;
#ifdef GRPG_FILL_TRACE
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tp = wall_clock64();
#define FILL_PH(K) do { const unsigned long long t_ = wall_clock64(); ph[K] += t_ - tp; tp = t_; } while (0)
#else
#define FILL_PH(K) do {} while (0)
#endif
    float4 g0[ITEMS], g1[ITEMS];
#pragma unroll
#if GRPG_FILL_ABLATE & 2   // experiment build (masks of the wrong splats: images wrong, nothing out of bounds): no gather
    for (int i = 0; i < ITEMS; i++) { g0[i] = rec.geo0(gid[i] & 1u); g1[i] = rec.geo1(gid[i] & 1u); }
#else
    for (int i = 0; i < ITEMS; i++) { g0[i] = rec.geo0(gid[i]); g1[i] = rec.geo1(gid[i]); }
#endif
    // running output position of each of the column's tiles (independent of the gather)
    uint32_t out[STILE];
    {
      const int txc = min(tx, gx - 1);
#pragma unroll
      for (int r = 0; r < STILE; r++) {
        const int ty = min(sy * STILE + r, gy - 1);
        out[r] = tile_start[(uint32_t)ty * (uint32_t)gx + (uint32_t)txc] +
                 table[(size_t)sid * STILE_TILES + r * STILE + col];
      }
    }
    const uint32_t sid_next = sid + gridDim.x;
    const bool more = sid_next < nseg;
    SegDesc dn = d;
    if (more) dn = seg[sid_next];
    FILL_PH(0);   // records, output positions, next descriptor: issued AND (the adds need them) arrived
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
      const uint32_t e = i * HB_FILL_THREADS + tid;
      const QuarterPre q = quarter_pre(g1[i].x, g1[i].y, g1[i].z, g0[i].z);
      // (layered frame: the class bit of the record becomes bit 27 of the list entries; 0 in every other frame)
      s_gid[e] = gid[i] | ((__float_as_uint(g0[i].w) & REC_CLASS_BIT) ? POINT_CLASS_BIT : 0u);
      s_m[e] = (uint16_t)(key[i] >> 16);      // entries beyond n carry mask 0
      s_fl[e] = (uint8_t)((q.sane ? 1u : 0u) | (q.transparent ? 2u : 0u));
      s_xy[e] = make_float2(g0[i].x, g0[i].y);
      s_bi[e] = make_float2(q.b, q.ic);
      s_dc[e] = make_float2(q.det, q.ct);
      s_xt[e] = q.xtop;
    }
    FILL_PH(2);   // quarter_pre + staging
    __syncthreads();
    FILL_PH(3);   // barrier
    // the next segment's pairs travel while this one is processed
    if (more) {
#pragma unroll
      for (int i = 0; i < ITEMS; i++) {
        const uint32_t e = i * HB_FILL_THREADS + tid;
        const bool on = e < dn.end - dn.begin;
        gid[i] = on ? cval[dn.begin + e] : 0u;
        key[i] = on ? ckey[dn.begin + e] : 0u;
      }
    }
    if (tx < gx) {
      const uint32_t nround = (n + WAVE - 1) / WAVE;
      uint32_t ncol = 0;
      for (uint32_t r = 0; r < nround; r++) {
        const uint32_t e = r * WAVE + lane;
        const bool hit = (s_m[e] >> col) & 1u;
        const uint64_t m = __ballot(hit);
        if (hit) coll[ncol + (uint32_t)__popcll(m & lt)] = (uint16_t)e;
        ncol += (uint32_t)__popcll(m);
      }
      __builtin_amdgcn_wave_barrier();
      FILL_PH(4);   // column compaction
      const float tile_x = (float)(tx * TILE), super_y = (float)(sy * STILE * TILE);
      for (uint32_t i0 = 0; i0 < ncol; i0 += WAVE) {
        const uint32_t i = i0 + lane;
        const bool valid = i < ncol;
        const uint32_t e = valid ? coll[i] : 0u;
        const uint32_t rm = valid ? (uint32_t)s_m[e] >> 8 : 0u;
        const uint32_t g = s_gid[e];
        const uint32_t fl = s_fl[e];
        const float2 xy = s_xy[e], bi = s_bi[e], dc = s_dc[e];
        QuarterPre q;
        q.b = bi.x; q.ic = bi.y; q.det = dc.x; q.ct = dc.y; q.xtop = s_xt[e];
        q.sane = fl & 1u; q.transparent = fl & 2u;
        float ylo, yup;
        bool empty;
        strip_extent(q, xy.x, tile_x, ylo, yup, empty);
        // quarter masks of the column's 8 tiles, 4 bits each
        const uint32_t qm = quarter_column_mask(q.sane, q.transparent, ylo, yup, empty, xy.y, super_y);
#pragma unroll
        for (int r = 0; r < STILE; r++) {
          const bool hit = (rm >> r) & 1u;
          const uint64_t m = __ballot(hit);
          if (m == 0ull) continue;       // wave-uniform (also skips rows beyond the grid: mask 0)
          if (hit) {
            const uint32_t bits = (qm >> (4 * r)) & 15u;
            const uint32_t pos = out[r] + (uint32_t)__popcll(m & lt);
            if (pos < R_cap) point_list[pos] = g | (bits << SUBTILE_SHIFT);
          }
          out[r] += (uint32_t)__popcll(m);
#ifdef GRPG_FILL_TRACE
          if (wave == 0) tr_inst += (unsigned long long)__popcll(m);
#endif
        }
      }
    }
    FILL_PH(5);   // column walk: extents, masks, stores
#ifdef GRPG_FILL_TRACE
    tr_segs++;
    if (tid == 0 && blockIdx.x < 1024) {
      for (int k = 0; k < 8; k++) g_fill_phase[blockIdx.x][k] += ph[k];
      g_fill_trace[blockIdx.x][1] = wall_clock64(); g_fill_trace[blockIdx.x][2] = tr_segs; g_fill_trace[blockIdx.x][3] += tr_inst;
      tr_inst = 0;
    }
#endif
    if (!more) break;
    sid = sid_next;
    d = dn;
    __syncthreads();   // LDS is refilled
  }
}

void launch_hier_count(hipStream_t s, uint2* cranges, const uint32_t* run_totals, uint32_t NS,
                       char* seg_desc, uint2* st_seg,
                       uint32_t* nseg_total, uint32_t max_seg, const uint32_t* ckey_sorted,
                       int gx, int gy, uint32_t* seg_table, uint32_t* tile_tot,
                       uint32_t* tile_start, uint2* ranges, uint32_t* R_out, uint32_t* host_word,
                       const uint32_t* Rc_dev, BlobHeader* bin_header, uint32_t R_cap,
                       uint32_t coarse_cap, uint32_t* work, TileClasses cls) {
  const int sgx = (gx + STILE - 1) / STILE;
  const uint32_t T = (uint32_t)gx * (uint32_t)gy;
  SegDesc* seg = (SegDesc*)seg_desc;
  if (run_totals != nullptr && NS <= 256u) {
    hb_count_kernel<true><<<max_seg, HB_CNT_THREADS, 0, s>>>(seg, nseg_total, ckey_sorted, seg_table,
                                                             run_totals, NS, cranges, st_seg, max_seg);
  } else {
    hb_seg_setup_kernel<<<1, 1024, 0, s>>>(cranges, NS, seg, st_seg, nseg_total, max_seg);
    hb_count_kernel<false><<<max_seg, HB_CNT_THREADS, 0, s>>>(seg, nseg_total, ckey_sorted, seg_table,
                                                              nullptr, NS, cranges, st_seg, max_seg);
  }
  hb_tile_prefix_kernel<<<NS, 1024, 0, s>>>(st_seg, seg_table, sgx, gx, gy, tile_tot);
  hb_tile_scan_kernel<<<1, 1024, 0, s>>>(T, tile_tot, tile_start, ranges, R_out, host_word, Rc_dev,
                                         bin_header, R_cap, coarse_cap, work, cls);
}

void launch_hier_fill(hipStream_t s, const char* seg_desc, const uint32_t* nseg_total, uint32_t max_seg,
                      const uint32_t* ckey_sorted, const uint32_t* cval_sorted, const RecView rec,
                      int gx, int gy, const uint32_t* seg_table, const uint32_t* tile_start,
                      uint32_t R_cap, uint32_t* point_list) {
  const int sgx = (gx + STILE - 1) / STILE;
  // persistent workgroups: 3 fit a CU (LDS), 256 CUs
  const uint32_t wgs = HB_SEG <= 512 ? 1024u : 768u;   // (512-entry segments: 26 KB of LDS, four workgroups per CU)
  const uint32_t grid = max_seg < wgs ? max_seg : wgs;
  hb_fill_kernel<<<grid, HB_FILL_THREADS, 0, s>>>((const SegDesc*)seg_desc, nseg_total, ckey_sorted,
                                                     cval_sorted, rec, sgx, gx, gy, seg_table, tile_start,
                                                     R_cap, point_list);
}

uint32_t hier_max_segments(uint32_t Rcap, uint32_t NS) { return Rcap / HB_SEG + NS + 1; }

#ifdef GRPG_FILL_TRACE
}  // namespace grpg
extern "C" __attribute__((visibility("default"))) int grpg_debug_fill_trace(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(grpg::g_fill_trace), sizeof(grpg::g_fill_trace));
}
extern "C" __attribute__((visibility("default"))) int grpg_debug_fill_phase(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(grpg::g_fill_phase), sizeof(grpg::g_fill_phase));
}
namespace grpg {
#endif

}  // namespace grpg
