// Tile binning for gfx950: instance emission, tile ranges, and the debug/parity decoder.
//
// Replaces duplicateWithKeys and identifyTileRanges (cuda_rasterizer/rasterizer_impl.cu:70-138).
//
// Order argument (why the result is bit-identical to the reference's stable 64-bit sort):
// the reference emits instances in ascending Gaussian id, keys them (tile << 32 | depth_bits)
// and sorts stably, so the final order is (tile, depth_bits, id).  Here the Gaussians are first
// sorted stably by depth_bits from id order -> (depth_bits, id); instances are emitted in that
// order and then stably partitioned by tile -> (tile, depth_bits, id).  A Gaussian touches a
// tile at most once, so there are no further ties.
#include "blend_math.h"
#include "common.h"

namespace grpg {

// One thread per OUTPUT instance (not per Gaussian): a workgroup owns 2048 consecutive slots of the
// instance list and maps every slot back to its (depth-sorted) Gaussian.  The offsets scan left the
// index window [lo, hi] of Gaussians that can own those slots; every window Gaussian marks the slot
// where its run starts in an LDS array, and an inclusive max-scan over the 2048 slots (4 consecutive
// slots per thread, wave scan, 8-wave spine) yields the owner of every slot -- about 10
// instructions per slot instead of an 11-step binary search each.
// Work per thread is independent of the footprint distribution -- in the reference one thread loops
// over every tile of its Gaussian (rasterizer_impl.cu:98-108), thousands of serial iterations for a
// near-camera splat, and after the depth sort the largest footprints sit next to each other -- and
// each thread stores its 4 keys / 4 values with one 16-byte store.
// A workgroup's 2048 slots are exactly one chunk of the tile partition's radix passes (sort.hip),
// so it also leaves that chunk's pass-0 digit histogram behind: one launch and one read of the
// 4 R key bytes less.
// The instance count R lives in device memory (the host learns it asynchronously, api.hip); the
// grid is sized for the blob's capacity and surplus workgroups exit.
constexpr int EMIT_WIN = EMIT_PER_BLOCK + 64;   // window offsets staged in LDS (else global search)
constexpr int EMIT_WAVES = EMIT_THREADS / WAVE;

__device__ __forceinline__ uint32_t last_leq(const uint32_t* __restrict__ a, uint32_t lo,
                                             uint32_t hi, const uint32_t v) {
  // largest i in [lo, hi] with a[i] <= v; requires a[lo] <= v
  while (lo < hi) {
    const uint32_t mid = lo + (hi - lo + 1) / 2;
    if (a[mid] <= v) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// One instance: slot k of Gaussian g's tile rectangle (row-major, rasterizer_impl.cu:98-108)
// -> tile id and value (id | quarter-reach mask << 28).  r0 / r1 = first 32 bytes of g's record.
__device__ __forceinline__ void emit_instance(const uint32_t g, const uint32_t k, const float4 r0,
                                              const float4 r1, const int gx, const int gy,
                                              uint32_t& key, uint32_t& val) {
  const int radius = (int)(__float_as_uint(r0.w) & ~REC_CLASS_BIT);
  const uint32_t cls = (__float_as_uint(r0.w) & REC_CLASS_BIT) ? POINT_CLASS_BIT : 0u;   // layered frames only
  int minx, miny, maxx, maxy;
  get_rect(r0.x, r0.y, radius, gx, gy, minx, miny, maxx, maxy);
  const uint32_t w = (uint32_t)(maxx - minx);
  // k / w for k < gx*gy <= 2^24 (exact in fp32): float estimate, one correction step each way
  uint32_t row = (uint32_t)((float)k * __builtin_amdgcn_rcpf((float)w));
  row -= (row * w > k) ? 1u : 0u;
  row += ((row + 1u) * w <= k) ? 1u : 0u;
  const uint32_t col = k - row * w;
  const int tx = minx + (int)col, ty = miny + (int)row;
  key = (uint32_t)(ty * gx + tx);
  // Sub-tile mask: which 16x4 quarters of the tile can this splat reach at all?  Evaluated once
  // here (the record is in registers anyway) and carried through the sort in the spare top bits
  // of the value, so render knows before loading a record whether it is needed.
  const uint32_t bits = quarter_reach_mask(r0.x, r0.y, r1.x, r1.y, r1.z, r0.z, (float)(tx * TILE),
                                           (float)(ty * TILE));
  val = g | cls | (bits << SUBTILE_SHIFT);
}

// Coarse instance (hierarchical binning, hier_binning.hip): slot k of Gaussian g's SUPER-TILE
// rectangle (row-major) -> key = super-tile id | local column mask << 16 | local row mask << 24
// (the tiles of that super-tile the rectangle covers; the coarse partition sorts on the id bits
// only), value = g.  rc = the packed tile rectangle.
__device__ __forceinline__ void emit_coarse_instance(const uint32_t g, const uint32_t k, const uint2 rc,
                                                     const int sgx, uint32_t& key, uint32_t& val) {
  const uint32_t rx = rc.x, ry = rc.y;
  const int minx = (int)(rx & 0xFFFFu), maxx = (int)(rx >> 16);
  const int miny = (int)(ry & 0xFFFFu), maxy = (int)(ry >> 16);
  const uint32_t sx0 = (uint32_t)minx / STILE, sx1 = ((uint32_t)maxx + STILE - 1) / STILE;
  const uint32_t sy0 = (uint32_t)miny / STILE;
  const uint32_t w = sx1 - sx0;
  uint32_t row = (uint32_t)((float)k * __builtin_amdgcn_rcpf((float)w));
  row -= (row * w > k) ? 1u : 0u;
  row += ((row + 1u) * w <= k) ? 1u : 0u;
  const uint32_t col = k - row * w;
  const int sx = (int)(sx0 + col), sy = (int)(sy0 + row);
  const int lx0 = max(minx - sx * STILE, 0), lx1 = min(maxx - sx * STILE, STILE);   // half-open
  const int ly0 = max(miny - sy * STILE, 0), ly1 = min(maxy - sy * STILE, STILE);
  const uint32_t cm = ((1u << lx1) - 1u) & ~((1u << lx0) - 1u);
  const uint32_t rm = ((1u << ly1) - 1u) & ~((1u << ly0) - 1u);
  key = ((uint32_t)sy * (uint32_t)sgx + (uint32_t)sx) | (cm << 16) | (rm << 24);
  val = g;
}

// COARSE: the slots are (Gaussian, super-tile) pairs -- `offsets` is the scan over the super-tile
// counts, gx = super-tiles per row -- and instead of the records the packed tile rectangles are
// read, which the scan left behind in depth order (no dependent gather).
template <bool COARSE>
__global__ void __launch_bounds__(EMIT_THREADS)
emit_kernel(const uint32_t* __restrict__ V_dev, const uint32_t* __restrict__ R_dev,
            const uint32_t R_cap, const uint32_t* __restrict__ sorted_gid,
            const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ emit_win,
            const uint32_t emit_win_cap, const RecView rec,
            const uint2* __restrict__ rect_sorted /* COARSE only */, const int gx, const int gy,
            uint32_t* __restrict__ tile_keys, uint32_t* __restrict__ vals,
            uint32_t* __restrict__ hist_table /* [digit][nchunks], may be NULL */,
            const uint32_t hist_mask, const uint32_t nchunks,
            uint2* __restrict__ zero_ranges /* COARSE: super-tile runs cleared by workgroup 0 */,
            const uint32_t nzero) {
  __shared__ uint32_t s_win[2];
  __shared__ uint32_t s_off[EMIT_WIN];
  __shared__ uint32_t s_own[EMIT_PER_BLOCK];
  __shared__ uint32_t s_wave[EMIT_WAVES];
  __shared__ uint32_t s_hist[RS_MAX_RADIX];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t R = min(*R_dev, R_cap);
  const uint32_t o0 = blockIdx.x * EMIT_PER_BLOCK;
  if (COARSE && blockIdx.x == 0)
    for (uint32_t i = tid; i < nzero; i += EMIT_THREADS) zero_ranges[i] = make_uint2(0u, 0u);
  if (o0 >= R) return;   // whole workgroup
  const uint32_t o1 = min(R, o0 + EMIT_PER_BLOCK);
  const uint32_t nG = *V_dev;   // Gaussians in the depth-sorted arrays
  // window bounds: owner of the block's first slot, and (an upper bound of) the owner of its last
  // slot = owner of the next block's first slot / of the list's last slot -- both left behind by
  // the offsets scan (sort.hip:scan_down_kernel); binary search only beyond the table's capacity
  if (tid < 2) {
    const uint32_t b = blockIdx.x + tid;
    s_win[tid] = b <= emit_win_cap ? emit_win[b]
                                   : last_leq(offsets, 0, nG - 1, tid == 0 ? o0 : o1 - 1);
  }
#pragma unroll
  for (int r = 0; r < EMIT_PER_BLOCK / EMIT_THREADS; r++) s_own[r * EMIT_THREADS + tid] = 0u;
  if (tid < RS_MAX_RADIX) s_hist[tid] = 0u;
  __syncthreads();
  const uint32_t lo = s_win[0], hi = s_win[1];
  // the window normally holds <= 2049 Gaussians (every visible Gaussian owns >= 1 slot)
  const uint32_t nwin = hi - lo + 1;
  if (nwin <= (uint32_t)EMIT_WIN) {
    // a Gaussian with zero instances shares its offset with its successor, so the LAST index with
    // offsets[i] <= s owns slot s: mark run starts with the largest window index, then max-scan
    for (uint32_t j = tid; j < nwin; j += EMIT_THREADS) {
      const uint32_t o = offsets[lo + j];
      s_off[j] = o;
      if (j > 0 && o < o1) atomicMax(&s_own[o - o0], j);   // j > 0  =>  o > o0
    }
    __syncthreads();
    uint32_t own[4];
    {
      const uint4 v = reinterpret_cast<const uint4*>(s_own)[tid];
      own[0] = v.x; own[1] = max(own[0], v.y); own[2] = max(own[1], v.z); own[3] = max(own[2], v.w);
    }
    uint32_t inc = own[3];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t t = __shfl_up(inc, d, 64);
      if (lane >= (uint32_t)d) inc = max(inc, t);
    }
    if (lane == 63) s_wave[wave] = inc;
    const uint32_t excl_w = __shfl_up(inc, 1, 64);
    __syncthreads();
    uint32_t before = lane > 0 ? excl_w : 0u;   // max over the preceding threads of this wave ...
#pragma unroll
    for (int w2 = 0; w2 < EMIT_WAVES - 1; w2++)
      if ((uint32_t)w2 < wave) before = max(before, s_wave[w2]);   // ... and of the preceding waves
    uint32_t key[4], val[4];
    const uint32_t s0 = o0 + 4 * tid;
    // the dependent loads (owner id -> its record) of the thread's 4 slots are issued together:
    // two memory round trips per thread instead of eight
    uint32_t gid4[4], k4[4], j4[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const uint32_t j = max(own[r], before);
      const bool on = s0 + r < o1;
      j4[r] = j;
      gid4[r] = on ? sorted_gid[lo + j] : 0u;
      k4[r] = s0 + r - s_off[j];
    }
    float4 q0[4], q1[4];
    uint2 rc4[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      if (COARSE) { rc4[r] = (s0 + r < o1) ? rect_sorted[lo + j4[r]] : make_uint2(0u, 0u); }
      else { q0[r] = rec.geo0(gid4[r]); q1[r] = rec.geo1(gid4[r]); }
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
      key[r] = 0u; val[r] = 0u;
      if (s0 + r < o1) {
        if (COARSE) emit_coarse_instance(gid4[r], k4[r], rc4[r], gx, key[r], val[r]);
        else emit_instance(gid4[r], k4[r], q0[r], q1[r], gx, gy, key[r], val[r]);
        if (hist_table) atomicAdd(&s_hist[key[r] & hist_mask], 1u);
      }
    }
    if (s0 + 3 < o1) {   // o0 is a multiple of 2048 and the arrays are 256-byte aligned
      reinterpret_cast<uint4*>(tile_keys)[s0 >> 2] = make_uint4(key[0], key[1], key[2], key[3]);
      reinterpret_cast<uint4*>(vals)[s0 >> 2] = make_uint4(val[0], val[1], val[2], val[3]);
    } else {
#pragma unroll
      for (int r = 0; r < 4; r++)
        if (s0 + r < o1) { tile_keys[s0 + r] = key[r]; vals[s0 + r] = val[r]; }
    }
  } else {   // oversized window (runs of zero-instance Gaussians): per-slot search in global memory
    for (uint32_t s = o0 + tid; s < o1; s += EMIT_THREADS) {
      const uint32_t i = last_leq(offsets, lo, hi, s);
      const uint32_t g = sorted_gid[i];
      uint32_t key, val;
      if (COARSE) emit_coarse_instance(g, s - offsets[i], rect_sorted[i], gx, key, val);
      else emit_instance(g, s - offsets[i], rec.geo0(g), rec.geo1(g), gx, gy, key, val);
      tile_keys[s] = key;
      vals[s] = val;
      if (hist_table) atomicAdd(&s_hist[key & hist_mask], 1u);
    }
  }
  if (hist_table) {   // pass-0 digit histogram of this chunk of the tile partition
    __syncthreads();
    if (tid <= hist_mask) hist_table[(size_t)tid * nchunks + blockIdx.x] = s_hist[tid];
  }
}

// identifyTileRanges, rasterizer_impl.cu:116-138 (ranges pre-zeroed by the caller, :313).
// Four consecutive keys per thread (one 16-byte load + the two neighbours).
__global__ void __launch_bounds__(256)
tile_ranges_kernel(const uint32_t* __restrict__ R_dev, const uint32_t R_cap,
                   const uint32_t* __restrict__ tile_keys, uint2* __restrict__ ranges,
                   const uint32_t key_mask /* coarse keys carry payload above the id bits */) {
  const uint32_t R = min(*R_dev, R_cap);
  const uint32_t i0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (i0 >= R) return;
  uint32_t k[6];   // k[0] = predecessor, k[1..4] = own keys, k[5] = successor
  if (i0 + 4 <= R) {
    const uint4 v = *reinterpret_cast<const uint4*>(tile_keys + i0);
    k[1] = v.x & key_mask; k[2] = v.y & key_mask; k[3] = v.z & key_mask; k[4] = v.w & key_mask;
  } else {
#pragma unroll
    for (int j = 0; j < 4; j++) k[1 + j] = i0 + j < R ? tile_keys[i0 + j] & key_mask : 0xFFFFFFFFu;
  }
  k[0] = i0 > 0 ? tile_keys[i0 - 1] & key_mask : 0xFFFFFFFFu;
  k[5] = i0 + 4 < R ? tile_keys[i0 + 4] & key_mask : 0xFFFFFFFFu;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const uint32_t i = i0 + j;
    if (i >= R) break;
    const uint32_t t = k[1 + j];
    if (i == 0 || k[j] != t) ranges[t].x = i;
    if (i == R - 1 || k[2 + j] != t) ranges[t].y = i + 1;
  }
}

void launch_emit(hipStream_t s, const uint32_t* V_dev, const uint32_t* R_dev, uint32_t R_cap,
                 const uint32_t* sorted_gid, const uint32_t* offsets, const uint32_t* emit_win,
                 uint32_t emit_win_cap, const RecView rec, int gx, int gy, uint32_t* tile_keys,
                 uint32_t* vals, uint32_t* hist_table, uint32_t hist_mask, uint32_t nchunks) {
  if (R_cap == 0) return;
  emit_kernel<false><<<(R_cap + EMIT_PER_BLOCK - 1) / EMIT_PER_BLOCK, EMIT_THREADS, 0, s>>>(
      V_dev, R_dev, R_cap, sorted_gid, offsets, emit_win, emit_win_cap, rec, nullptr, gx, gy, tile_keys,
      vals, hist_table, hist_mask, nchunks, nullptr, 0u);
}

void launch_emit_coarse(hipStream_t s, const uint32_t* V_dev, const uint32_t* Rc_dev, uint32_t cap,
                        const uint32_t* sorted_gid, const uint2* rect_sorted, const uint32_t* offsets,
                        const uint32_t* emit_win, uint32_t emit_win_cap, int sgx, int sgy, uint32_t* st_keys,
                        uint32_t* vals, uint32_t* hist_table, uint32_t hist_mask, uint32_t nchunks,
                        uint2* cranges, uint32_t NS) {
  const uint32_t grid = cap ? (cap + EMIT_PER_BLOCK - 1) / EMIT_PER_BLOCK : 1u;   // block 0 clears the runs
  emit_kernel<true><<<grid, EMIT_THREADS, 0, s>>>(
      V_dev, Rc_dev, cap, sorted_gid, offsets, emit_win, emit_win_cap, RecView{nullptr}, rect_sorted, sgx,
      sgy, st_keys, vals, hist_table, hist_mask, nchunks, cranges, NS);
}

// ------------------------------------------------------------------------------------------
// Round 5: the coarse emit with the offsets scan's down-sweep inside it (one launch and one pass over
// the counts fewer: scan_down wrote the V offsets and the window table only for this kernel to read
// them back).  Preconditions (api.hip): the super-tile counts and rectangles arrive in depth order from
// the fat sort's last pass, which drops culled Gaussians -- so EVERY one of the V entries owns >= 1
// slot -- and the scan's reduce launch has left the sums of 2048-Gaussian blocks (<= FUSED_MAX_BLOCKS).
// A workgroup owns EMIT_PER_BLOCK output slots [o0, o1):
//   1. it scans the block sums itself (the spine: a few KB, L2-resident) -> total (= the coarse count,
//      published by workgroup 0) and the block B0 that holds the owner of slot o0;
//   2. it loads the counts of the 4096 Gaussians from B0's first one on and scans them: the owners of
//      its slots lie among them (the owner of o0 is within B0's 2048, and 2048 slots have <= 2048
//      owners behind it);
//   3. then exactly emit_kernel<true>'s window logic on those offsets.
// ------------------------------------------------------------------------------------------
constexpr int FUSED_MAX_BLOCKS = 4 * EMIT_THREADS;   // 2048 blocks = 4 M Gaussians (the fat sort's limit)
// the fused emit needs the fat sort's depth-ordered payload: whatever the fat sort accepts must fit the spine
static_assert((size_t)DS_MAX_CHUNKS * DS_CHUNK <= (size_t)FUSED_MAX_BLOCKS * SC_CHUNK,
              "emit_coarse_fused_kernel: the spine must hold the block sums of the largest fat-sorted frame");
constexpr int FUSED_WIN = 2 * SC_CHUNK;              // Gaussians whose offsets a workgroup computes
static_assert(FUSED_WIN == 8 * EMIT_THREADS, "8 counts per thread");
static_assert(SC_CHUNK == EMIT_PER_BLOCK, "a slot block's owners must fit two Gaussian blocks");

// inclusive scan of one value per thread over the 512-thread workgroup; s_wave: EMIT_WAVES words
__device__ __forceinline__ uint32_t emit_block_inclusive(uint32_t v, uint32_t* s_wave, uint32_t* total) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t t = __shfl_up(inc, d, 64);
    if (lane >= (uint32_t)d) inc += t;
  }
  __syncthreads();   // s_wave is reused across calls
  if (lane == 63) s_wave[wave] = inc;
  __syncthreads();
  uint32_t base = 0, sum = 0;
#pragma unroll
  for (int w = 0; w < EMIT_WAVES; w++) {
    const uint32_t c = s_wave[w];
    if ((uint32_t)w < wave) base += c;
    sum += c;
  }
  *total = sum;
  return base + inc;
}

__global__ void __launch_bounds__(EMIT_THREADS)
emit_coarse_fused_kernel(const uint32_t* __restrict__ V_dev, uint32_t* __restrict__ Rc_out, const uint32_t R_cap,
                         const uint32_t* __restrict__ sorted_gid, const uint32_t* __restrict__ counts,
                         const uint32_t* __restrict__ block_sums, const uint32_t nblocks,
                         const uint2* __restrict__ rect_sorted, const int sgx,
                         uint32_t* __restrict__ st_keys, uint32_t* __restrict__ vals,
                         uint32_t* __restrict__ hist_table /* [digit][nchunks], may be NULL */,
                         const uint32_t hist_mask, const uint32_t nchunks,
                         uint2* __restrict__ zero_ranges, const uint32_t nzero) {
  __shared__ uint32_t s_spine[FUSED_MAX_BLOCKS];   // exclusive prefix of the block sums
  __shared__ uint32_t s_off[FUSED_WIN];            // offsets of Gaussians G0 .. G0 + 4095
  __shared__ uint32_t s_own[EMIT_PER_BLOCK];
  __shared__ uint32_t s_wave[EMIT_WAVES];
  __shared__ uint32_t s_hist[RS_MAX_RADIX];
  __shared__ uint32_t s_win[3];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t o0 = blockIdx.x * EMIT_PER_BLOCK;
  if (blockIdx.x == 0)
    for (uint32_t i = tid; i < nzero; i += EMIT_THREADS) zero_ranges[i] = make_uint2(0u, 0u);
  // ---- 1. the spine ----
  uint32_t total;
  {
    uint32_t v[4], sum = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t b = 4 * tid + k;
      v[k] = b < nblocks ? block_sums[b] : 0u;
      sum += v[k];
    }
    uint32_t ex = emit_block_inclusive(sum, s_wave, &total) - sum;
#pragma unroll
    for (int k = 0; k < 4; k++) { s_spine[4 * tid + k] = ex; ex += v[k]; }
  }
  if (blockIdx.x == 0 && tid == 0) *Rc_out = total;
  const uint32_t R = min(total, R_cap);
  if (o0 >= R) return;   // whole workgroup
  const uint32_t o1 = min(R, o0 + EMIT_PER_BLOCK);
  const uint32_t nG = *V_dev;
#pragma unroll
  for (int r = 0; r < EMIT_PER_BLOCK / EMIT_THREADS; r++) s_own[r * EMIT_THREADS + tid] = 0u;
  if (tid < RS_MAX_RADIX) s_hist[tid] = 0u;
  __syncthreads();
  // the last block whose prefix is <= o0 (blocks without Gaussians sit at prefix == total > o0)
  uint32_t B0;
  {
    uint32_t lo = 0, hi = nblocks - 1;
    while (lo < hi) {
      const uint32_t mid = lo + (hi - lo + 1) / 2;
      if (s_spine[mid] <= o0) lo = mid; else hi = mid - 1;
    }
    B0 = lo;
  }
  // ---- 2. offsets of the 4096 Gaussians from block B0 on ----
  const uint32_t G0 = B0 * SC_CHUNK;
  {
    const uint32_t i0 = G0 + 8 * tid;
    uint32_t c[8], sum = 0;
    if (i0 + 8 <= nG) {   // G0 and 8 tid are multiples of 8, the array is 256-byte aligned
      const uint4 a = reinterpret_cast<const uint4*>(counts + i0)[0], b = reinterpret_cast<const uint4*>(counts + i0)[1];
      c[0] = a.x; c[1] = a.y; c[2] = a.z; c[3] = a.w; c[4] = b.x; c[5] = b.y; c[6] = b.z; c[7] = b.w;
    } else {
#pragma unroll
      for (int k = 0; k < 8; k++) c[k] = i0 + k < nG ? counts[i0 + k] : 0u;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) sum += c[k];
    uint32_t dummy;
    uint32_t ex = emit_block_inclusive(sum, s_wave, &dummy) - sum + s_spine[B0];
    if (tid < 2) s_win[tid] = 0xFFFFFFFFu;   // "owner not seen" (a zero count must not go unnoticed)
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; k++) {
      s_off[8 * tid + k] = ex;
      // window bounds: the owners of the first and the last slot (every Gaussian owns >= 1 slot: unique)
      if (c[k] > 0u) {
        if (ex <= o0 && o0 < ex + c[k]) s_win[0] = 8 * tid + k;
        if (ex <= o1 - 1u && o1 - 1u < ex + c[k]) s_win[1] = 8 * tid + k;
      }
      ex += c[k];
    }
  }
  __syncthreads();
  // The owners of the first and the last slot were seen by the threads that hold them -- as long as every one of
  // the V entries owns a slot (the fat sort drops culled Gaussians, a visible one covers >= 1 super-tile).  Should a
  // zero count ever arrive, a slot's owner may go unseen (its offset interval is found by nobody) or lie beyond the
  // first 2049 entries: fall back to emit_kernel<true>'s search -- the LAST entry whose offset is <= the slot (zero-
  // count entries share their successor's offset and sort in front of it; step 3's atomicMax resolves such ties the
  // same way) -- over the 4096 offsets this workgroup holds; a window that would reach past them is clamped (only a
  // run of > 2048 slot-less entries could do that, and those entries own no output).
  if (tid < 2 && s_win[tid] == 0xFFFFFFFFu) {
    const uint32_t slot = tid == 0 ? o0 : o1 - 1u;
    uint32_t a = 0, b = FUSED_WIN - 1;
    while (a < b) {
      const uint32_t mid = a + (b - a + 1) / 2;
      if (s_off[mid] <= slot) a = mid; else b = mid - 1;
    }
    s_win[tid] = a;
  }
  __syncthreads();
  const uint32_t lo = s_win[0], hi = max(s_win[1], lo);   // local indices (Gaussian G0 + index)
  const uint32_t nwin = hi - lo + 1;                      // <= 2049 (<= 4096 after a fallback)
  const uint32_t* __restrict__ woff = s_off + lo;
  // ---- 3. as emit_kernel<true>: run starts marked with the largest window index, then a max-scan ----
  for (uint32_t j = tid; j < nwin; j += EMIT_THREADS) {
    const uint32_t o = woff[j];
    if (j > 0 && o < o1) atomicMax(&s_own[o - o0], j);   // j > 0  =>  o > o0
  }
  __syncthreads();
  uint32_t own[4];
  {
    const uint4 v = reinterpret_cast<const uint4*>(s_own)[tid];
    own[0] = v.x; own[1] = max(own[0], v.y); own[2] = max(own[1], v.z); own[3] = max(own[2], v.w);
  }
  uint32_t inc = own[3];
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t t = __shfl_up(inc, d, 64);
    if (lane >= (uint32_t)d) inc = max(inc, t);
  }
  __syncthreads();   // s_wave: the scans above are done with it
  if (lane == 63) s_wave[wave] = inc;
  const uint32_t excl_w = __shfl_up(inc, 1, 64);
  __syncthreads();
  uint32_t before = lane > 0 ? excl_w : 0u;
#pragma unroll
  for (int w2 = 0; w2 < EMIT_WAVES - 1; w2++)
    if ((uint32_t)w2 < wave) before = max(before, s_wave[w2]);
  const uint32_t s0 = o0 + 4 * tid;
  uint32_t key[4], val[4], gid4[4], k4[4], j4[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const uint32_t j = max(own[r], before);
    const bool on = s0 + r < o1;
    j4[r] = j;
    gid4[r] = on ? sorted_gid[G0 + lo + j] : 0u;
    k4[r] = s0 + r - woff[j];
  }
  uint2 rc4[4];
#pragma unroll
  for (int r = 0; r < 4; r++) rc4[r] = (s0 + r < o1) ? rect_sorted[G0 + lo + j4[r]] : make_uint2(0u, 0u);
#pragma unroll
  for (int r = 0; r < 4; r++) {
    key[r] = 0u; val[r] = 0u;
    if (s0 + r < o1) {
      emit_coarse_instance(gid4[r], k4[r], rc4[r], sgx, key[r], val[r]);
      if (hist_table) atomicAdd(&s_hist[key[r] & hist_mask], 1u);
    }
  }
  if (s0 + 3 < o1) {   // o0 is a multiple of 2048 and the arrays are 256-byte aligned
    reinterpret_cast<uint4*>(st_keys)[s0 >> 2] = make_uint4(key[0], key[1], key[2], key[3]);
    reinterpret_cast<uint4*>(vals)[s0 >> 2] = make_uint4(val[0], val[1], val[2], val[3]);
  } else {
#pragma unroll
    for (int r = 0; r < 4; r++)
      if (s0 + r < o1) { st_keys[s0 + r] = key[r]; vals[s0 + r] = val[r]; }
  }
  if (hist_table) {   // pass-0 digit histogram of this chunk of the tile partition
    __syncthreads();
    if (tid <= hist_mask) hist_table[(size_t)tid * nchunks + blockIdx.x] = s_hist[tid];
  }
}

bool emit_coarse_fused_ok(uint32_t nblocks_scan) { return nblocks_scan >= 1u && nblocks_scan <= (uint32_t)FUSED_MAX_BLOCKS; }

void launch_emit_coarse_fused(hipStream_t s, const uint32_t* V_dev, uint32_t* Rc_out, uint32_t cap,
                              const uint32_t* sorted_gid, const uint32_t* counts_sorted,
                              const uint32_t* block_sums, uint32_t nblocks, const uint2* rect_sorted, int sgx,
                              uint32_t* st_keys, uint32_t* vals, uint32_t* hist_table, uint32_t hist_mask,
                              uint32_t nchunks, uint2* cranges, uint32_t NS) {
  const uint32_t grid = cap ? (cap + EMIT_PER_BLOCK - 1) / EMIT_PER_BLOCK : 1u;   // block 0 clears the runs
  emit_coarse_fused_kernel<<<grid, EMIT_THREADS, 0, s>>>(V_dev, Rc_out, cap, sorted_gid, counts_sorted, block_sums,
                                                         nblocks, rect_sorted, sgx, st_keys, vals, hist_table,
                                                         hist_mask, nchunks, cranges, NS);
}

void launch_tile_ranges(hipStream_t s, const uint32_t* R_dev, uint32_t R_cap,
                        const uint32_t* tile_keys, uint2* ranges, uint32_t T, uint32_t key_mask) {
  // ranges were zeroed by frame_init_kernel (same stream, earlier in the frame)
  if (R_cap == 0) return;
  tile_ranges_kernel<<<(R_cap + 1023) / 1024, 256, 0, s>>>(R_dev, R_cap, tile_keys, ranges, key_mask);
}

// ---------------------------------- debug / parity decoder --------------------------------
__global__ void __launch_bounds__(256)
debug_geom_kernel(const int P, const RecView rec, const uint32_t* __restrict__ tiles,
                  float* means2D, float* depths, float* conic_opacity, float* rgb,
                  uint32_t* tiles_out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  const bool vis = tiles[i] > 0;
  float4 a = make_float4(0, 0, 0, 0), b = a, c = a;
  if (vis) rec.load((size_t)i, a, b, c);
  if (means2D) { means2D[2 * i] = a.x; means2D[2 * i + 1] = a.y; }
  if (depths) depths[i] = a.z;
  if (conic_opacity) {
    conic_opacity[4 * i] = b.x; conic_opacity[4 * i + 1] = b.y; conic_opacity[4 * i + 2] = b.z;
    conic_opacity[4 * i + 3] = a.w;
  }
  if (rgb) { rgb[3 * i] = b.w; rgb[3 * i + 1] = c.x; rgb[3 * i + 2] = c.y; }
  if (tiles_out) tiles_out[i] = tiles[i];
}

__global__ void __launch_bounds__(256)
debug_keys_kernel(const uint32_t R, const uint32_t* __restrict__ tile_keys,
                  const uint32_t* __restrict__ tile_start, const uint32_t T,
                  const uint32_t* __restrict__ point_list, const RecView rec,
                  uint64_t* keys_sorted, uint32_t* point_list_out) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= R) return;
  const uint32_t g = point_list[i] & ID_MASK;
  // hierarchical blob: the tile of instance i is the LAST tile whose start is <= i (empty tiles
  // share their start with their successor)
  const uint32_t tile = tile_start ? last_leq(tile_start, 0u, T - 1u, i) : tile_keys[i];
  if (keys_sorted)
    keys_sorted[i] = ((uint64_t)tile << 32) | (uint64_t)__float_as_uint(rec.geo1(g).w);
  if (point_list_out) point_list_out[i] = g;
}

void launch_debug_export(hipStream_t s, int P, uint32_t R, int W, int H, int gx, int gy,
                         const RecView rec, const uint32_t* tiles, const uint32_t* tile_keys,
                         const uint32_t* tile_start, const uint32_t* point_list, const uint2* ranges,
                         const uint32_t* n_contrib_in, uint64_t* keys_sorted,
                         uint32_t* point_list_out, uint32_t* ranges_out, uint32_t* n_contrib_out,
                         float* means2D, float* depths, float* conic_opacity, float* rgb,
                         uint32_t* tiles_out) {
  if (P > 0 && (means2D || depths || conic_opacity || rgb || tiles_out))
    debug_geom_kernel<<<(P + 255) / 256, 256, 0, s>>>(P, rec, tiles, means2D, depths,
                                                      conic_opacity, rgb, tiles_out);
  if (R > 0 && (keys_sorted || point_list_out))
    debug_keys_kernel<<<(R + 255) / 256, 256, 0, s>>>(R, tile_keys, tile_start, (uint32_t)(gx * gy),
                                                      point_list, rec, keys_sorted, point_list_out);
  if (ranges_out)
    (void)hipMemcpyAsync(ranges_out, ranges, (size_t)gx * gy * sizeof(uint2), hipMemcpyDeviceToDevice, s);
  if (n_contrib_out)
    (void)hipMemcpyAsync(n_contrib_out, n_contrib_in, (size_t)W * H * 4, hipMemcpyDeviceToDevice, s);
}

// ------------------------------------------------------------------------------------------
// One launch at the head of every frame: blob headers (instead of three pageable H2D copies), the
// tile ranges (identifyTileRanges leaves empty tiles untouched, rasterizer_impl.cu:313), the render
// work-list counters, and the words of the geometry blob that this frame accumulates into with
// atomics (depth-sort count tables, scan block sums).  `bin` may be NULL when the binning blob
// does not exist yet; write_bin_header_kernel then follows once it does.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
frame_init_kernel(BlobHeader* geom, BlobHeader* bin, BlobHeader* img, const uint32_t P,
                  const uint32_t V_init, const uint32_t Rcap, const uint32_t W, const uint32_t H,
                  const uint32_t S, uint2* __restrict__ ranges, const uint32_t T,
                  uint32_t* __restrict__ work, uint4* __restrict__ zero16, const size_t nzero16,
                  const uint32_t geom_has_grad, uint32_t* __restrict__ zero_words, const uint32_t n_zero_words) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = t; i < nzero16; i += stride) zero16[i] = make_uint4(0u, 0u, 0u, 0u);
  for (size_t i = t; i < n_zero_words; i += stride) zero_words[i] = 0u;   // layered frame: object-tile bit rows
  if (ranges)
    for (size_t i = t; i < T; i += stride) ranges[i] = make_uint2(0u, 0u);
  if (work && t < 4) { work[t] = 0u; work[4 + 4 * (size_t)T + t] = 0u; }   // list counters, bwd_ctl
  if (t < 3) {
    BlobHeader* h = t == 0 ? geom : (t == 1 ? bin : img);
    if (h) {
      h->magic = t == 0 ? GEOM_MAGIC : (t == 1 ? BIN_MAGIC : IMG_MAGIC);
      h->P = P; h->R = 0u; h->W = W; h->H = H; h->S = S; h->V = V_init; h->Rcap = Rcap;
      h->Rc = 0u; h->hier = 0u; h->ckpt_off256 = 0u; h->ckpt_slots = 0u;
      h->key_base = 0u; h->key_far = 0u; h->pc_timeout = 0u;
      h->has_grad_rec = t == 0 ? geom_has_grad : 0u;
    }
  }
}

void launch_frame_init(hipStream_t s, char* geom, char* bin, char* img, uint32_t P, uint32_t V_init,
                       uint32_t Rcap, uint32_t W, uint32_t H, uint32_t S, uint2* ranges, uint32_t T,
                       uint32_t* work, char* zero_begin, size_t zero_bytes, bool geom_has_grad,
                       uint32_t* zero_words, uint32_t n_zero_words) {
  const size_t nzero16 = zero_bytes / 16;   // the region is 256-byte aligned at both ends
  size_t items = nzero16 > (size_t)T ? nzero16 : (size_t)T;
  uint32_t blocks = (uint32_t)((items + 255) / 256);
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  frame_init_kernel<<<blocks, 256, 0, s>>>((BlobHeader*)geom, (BlobHeader*)bin, (BlobHeader*)img, P,
                                            V_init, Rcap, W, H, S, ranges, T, work,
                                            (uint4*)zero_begin, nzero16, geom_has_grad ? 1u : 0u,
                                            zero_words, zero_words ? n_zero_words : 0u);
}

// Binning-blob header alone (the blob was sized after num_rendered became known), or a re-run of
// the tail of the frame after a capacity overflow: ranges / work counters are cleared again.
__global__ void __launch_bounds__(256)
bin_header_kernel(BlobHeader* bin, const uint32_t P, const uint32_t Rcap, const uint32_t W,
                  const uint32_t H, const uint32_t S, uint2* __restrict__ ranges, const uint32_t T,
                  uint32_t* __restrict__ work) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (ranges)
    for (uint32_t i = t; i < T; i += gridDim.x * 256) ranges[i] = make_uint2(0u, 0u);
  if (work && t < 4) { work[t] = 0u; work[4 + 4 * (size_t)T + t] = 0u; }   // list counters, bwd_ctl
  if (t == 0 && bin) {
    bin->magic = BIN_MAGIC; bin->P = P; bin->R = 0u; bin->W = W; bin->H = H; bin->S = S;
    bin->V = 0u; bin->Rcap = Rcap; bin->Rc = 0u; bin->hier = 0u;
    bin->ckpt_off256 = 0u; bin->ckpt_slots = 0u;
  }
}

void launch_bin_header(hipStream_t s, char* bin, uint32_t P, uint32_t Rcap, uint32_t W, uint32_t H,
                       uint32_t S, uint2* ranges, uint32_t T, uint32_t* work) {
  const uint32_t blocks = ranges ? (T + 255) / 256 : 1;
  bin_header_kernel<<<blocks ? blocks : 1, 256, 0, s>>>((BlobHeader*)bin, P, Rcap, W, H, S, ranges,
                                                        T, work);
}

// ------------------------------------------------------------------------------------------
// Frame delivery helper for trajectory mode: float [n] -> uint8 [n], clamp(x,0,1)*255 + 0.5.
// Fuses the eval-mode clamp of render_kernel (street_gaussian_renderer.py:236-237) with the
// x255 / uint8 conversion of the image writers (one launch instead of four torch kernels).
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
pack_u8_kernel(const float4* __restrict__ src, uchar4* __restrict__ dst, const size_t n4,
               const float* __restrict__ src_tail, unsigned char* __restrict__ dst_tail,
               const size_t ntail) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) {
    const float4 v = src[i];
    uchar4 o;
    o.x = (unsigned char)(fminf(fmaxf(v.x, 0.f), 1.f) * 255.0f + 0.5f);
    o.y = (unsigned char)(fminf(fmaxf(v.y, 0.f), 1.f) * 255.0f + 0.5f);
    o.z = (unsigned char)(fminf(fmaxf(v.z, 0.f), 1.f) * 255.0f + 0.5f);
    o.w = (unsigned char)(fminf(fmaxf(v.w, 0.f), 1.f) * 255.0f + 0.5f);
    dst[i] = o;
  }
  if (i < ntail) dst_tail[i] = (unsigned char)(fminf(fmaxf(src_tail[i], 0.f), 1.f) * 255.0f + 0.5f);
}

void launch_pack_u8(hipStream_t s, const float* src, unsigned char* dst, size_t n) {
  if (n == 0) return;
  const size_t n4 = n / 4, ntail = n - n4 * 4;
  const size_t blocks = (n4 + 255) / 256 > 0 ? (n4 + 255) / 256 : 1;
  pack_u8_kernel<<<(unsigned)blocks, 256, 0, s>>>((const float4*)src, (uchar4*)dst, n4,
                                                  src + n4 * 4, dst + n4 * 4, ntail);
}

// planar float [3,H,W] -> interleaved rgb8 [H,W,3] (a ROS sensor_msgs/Image "rgb8" payload).
// One thread per 4 pixels: three 16-byte reads, one 12-byte write (three dwords).
__global__ void __launch_bounds__(256)
pack_hwc_kernel(const float* __restrict__ src, uint32_t* __restrict__ dst, const size_t npix,
                const int truncate) {
  const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;   // group of 4 pixels
  const size_t p0 = g * 4;
  if (p0 >= npix) return;
  const float bias = truncate ? 0.0f : 0.5f;
  unsigned char b[12];
  if (p0 + 4 <= npix && (npix & 3) == 0) {
    const float4 r = *reinterpret_cast<const float4*>(src + p0);
    const float4 gg = *reinterpret_cast<const float4*>(src + npix + p0);
    const float4 bl = *reinterpret_cast<const float4*>(src + 2 * npix + p0);
    const float rv[4] = {r.x, r.y, r.z, r.w}, gv[4] = {gg.x, gg.y, gg.z, gg.w},
                bv[4] = {bl.x, bl.y, bl.z, bl.w};
#pragma unroll
    for (int k = 0; k < 4; k++) {
      b[3 * k + 0] = colour_u8(rv[k], bias);
      b[3 * k + 1] = colour_u8(gv[k], bias);
      b[3 * k + 2] = colour_u8(bv[k], bias);
    }
    uint32_t w[3];
#pragma unroll
    for (int k = 0; k < 3; k++)
      w[k] = (uint32_t)b[4 * k] | ((uint32_t)b[4 * k + 1] << 8) | ((uint32_t)b[4 * k + 2] << 16) |
             ((uint32_t)b[4 * k + 3] << 24);
    dst[3 * g + 0] = w[0]; dst[3 * g + 1] = w[1]; dst[3 * g + 2] = w[2];
  } else {   // ragged pixel count: byte stores
    unsigned char* d8 = reinterpret_cast<unsigned char*>(dst);
    for (size_t p = p0; p < npix && p < p0 + 4; p++) {
#pragma unroll
      for (int c = 0; c < 3; c++)
        d8[3 * p + c] = colour_u8(src[c * npix + p], bias);
    }
  }
}

void launch_pack_hwc(hipStream_t s, const float* src, unsigned char* dst, size_t npix,
                     int truncate) {
  if (npix == 0) return;
  const size_t groups = (npix + 3) / 4;
  pack_hwc_kernel<<<(unsigned)((groups + 255) / 256), 256, 0, s>>>(src, (uint32_t*)dst, npix,
                                                                 truncate);
}

}  // namespace grpg
