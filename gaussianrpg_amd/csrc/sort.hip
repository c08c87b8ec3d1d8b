// On-device stable LSD radix sort of (u32 key, u32 value) pairs + exclusive scan, for gfx950.
//
// Replaces cub::DeviceRadixSort::SortPairs / cub::DeviceScan::InclusiveSum as used by the
// reference (cuda_rasterizer/rasterizer_impl.cu:280, :306-311).  The reference sorts R 64-bit
// (tile|depth) keys over 32+ceil(log2 T) bits; we reach the IDENTICAL final order with far less
// HBM traffic by (1) sorting the P Gaussians once by their 32-bit depth key and (2) stably
// partitioning the R instances by their <=16-bit tile id -- both use this sort.
//
// Two pass shapes:
//  * classic (tile partition of the R instances; depth sort of very large P), per pass of <= 8 bits:
//      radix_hist       : one workgroup per 2048-key chunk, LDS histogram -> table[digit][chunk]
//                         (pass 0 of the tile partition gets it from emit_kernel instead)
//      radix_digit_scan : one workgroup per digit, exclusive scan across chunks (in place) + totals
//      radix_scatter    : wave64 match-any ranking (ballot per key bit, popcount prefix), per-wave
//                         digit counters in LDS, LDS reorder of the chunk, coalesced run writes.
//  * fat (depth sort of the P Gaussians, P <= 4 M): 1024-thread workgroups own 8192 keys, the count
//    table is [chunk][digit] and so small (245 rows at P = 2 M) that every workgroup sweeps it
//    itself for its digit bases: a pass is ONE scatter launch plus one small histogram launch for
//    the next pass (pass 0's table comes from preprocess_kernel); pass 0 drops the culled
//    Gaussians, so the later passes move V instead of P pairs; and THREE 9-bit passes over
//    key - key_base settle the order whenever the visible depths span less than 2^27 key values
//    (a fourth pass over the remaining 5 bits runs only when the device finds they do not).
// Element counts that only the device knows (V visible Gaussians, R instances) are read from
// device memory; grids are sized by the host-side upper bound and surplus workgroups exit.
// Stability: a chunk is split wave-major, each wave walks its keys in rounds of 64 consecutive
// keys, ranks are (digit, wave, round, lane)-ordered.
#include "common.h"

namespace grpg {

// Exclusive scan of one value per thread over a 256-thread workgroup (4 waves).
// Returns the exclusive prefix; *total gets the workgroup sum.  `s_wave` is 4 words of LDS.
__device__ __forceinline__ uint32_t block_exclusive_scan_256(uint32_t v, uint32_t* s_wave,
                                                             uint32_t* total) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t t = __shfl_up(inc, d, 64);
    if (lane >= (uint32_t)d) inc += t;
  }
  __syncthreads();  // protect s_wave reuse across calls
  if (lane == 63) s_wave[wave] = inc;
  __syncthreads();
  uint32_t base = 0, sum = 0;
#pragma unroll
  for (int w = 0; w < 4; w++) {
    const uint32_t c = s_wave[w];
    if ((uint32_t)w < wave) base += c;
    sum += c;
  }
  *total = sum;
  return base + inc - v;
}

__global__ void __launch_bounds__(RS_THREADS)
radix_hist_kernel(const uint32_t* __restrict__ keys, const uint32_t n_cap,
                  const uint32_t* __restrict__ n_dev, const int shift,
                  const uint32_t mask, uint32_t* __restrict__ table, const uint32_t nchunks) {
  __shared__ uint32_t h[RS_MAX_RADIX];
  const uint32_t n = n_dev ? min(*n_dev, n_cap) : n_cap;
  const uint32_t base = blockIdx.x * RS_CHUNK;
  if (base >= n) return;   // the digit scan never reads columns beyond ceil(n / RS_CHUNK)
  h[threadIdx.x] = 0;
  __syncthreads();
  uint32_t kk[RS_ITEMS];   // all loads first: one memory round trip per workgroup
#pragma unroll
  for (int k = 0; k < RS_ITEMS; k++) {
    const uint32_t idx = base + k * RS_THREADS + threadIdx.x;
    kk[k] = idx < n ? keys[idx] : 0u;
  }
#pragma unroll
  for (int k = 0; k < RS_ITEMS; k++) {
    const uint32_t idx = base + k * RS_THREADS + threadIdx.x;
    if (idx < n) atomicAdd(&h[(kk[k] >> shift) & mask], 1u);
  }
  __syncthreads();
  if (threadIdx.x <= mask) table[(size_t)threadIdx.x * nchunks + blockIdx.x] = h[threadIdx.x];
}

// One workgroup per digit: in-place exclusive scan of table[digit][0..nchunks) and totals[digit].
__global__ void __launch_bounds__(256)
radix_digit_scan_kernel(uint32_t* __restrict__ table, const uint32_t nchunks_cap,
                        const uint32_t n_cap, const uint32_t* __restrict__ n_dev,
                        uint32_t* __restrict__ totals) {
  __shared__ uint32_t s_wave[4];
  const uint32_t n = n_dev ? min(*n_dev, n_cap) : n_cap;
  const uint32_t nchunks = (n + RS_CHUNK - 1) / RS_CHUNK;   // <= nchunks_cap (the row stride)
  uint32_t* row = table + (size_t)blockIdx.x * nchunks_cap;
  uint32_t carry = 0;
  for (uint32_t base = 0; base < nchunks; base += 1024) {
    uint32_t v[4], s = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t i = base + threadIdx.x * 4 + k;
      v[k] = i < nchunks ? row[i] : 0;
      s += v[k];
    }
    uint32_t tot;
    uint32_t ex = block_exclusive_scan_256(s, s_wave, &tot) + carry;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t i = base + threadIdx.x * 4 + k;
      if (i < nchunks) row[i] = ex;
      ex += v[k];
    }
    carry += tot;
  }
  if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

// CBITS > 0: digit width known at compile time (the match-any loop unrolls into straight-line
// code: one bit test + ballot, one sign-extended bit, two xor, two and per key bit, no branch);
// CBITS == 0: width taken from `bits_rt`.
// ITEMS keys per thread: a workgroup owns ITEMS / 8 consecutive 2048-key chunks of the count
// table (ITEMS = 16: twice as long output runs per digit; the table keeps its 2048-key columns,
// the workgroup takes the column of its first chunk as base and ranks all its keys itself).
template <int CBITS, int ITEMS>
__global__ void __launch_bounds__(RS_THREADS)
radix_scatter_kernel(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                     uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                     const uint32_t n_cap, const uint32_t* __restrict__ n_dev, const int shift,
                     const int bits_rt,
                     const uint32_t* __restrict__ table, const uint32_t* __restrict__ totals,
                     const uint32_t nchunks, const uint32_t* __restrict__ gather_src,
                     uint32_t* __restrict__ gather_dst) {
  const uint32_t n = n_dev ? min(*n_dev, n_cap) : n_cap;
  constexpr int CHUNK = RS_THREADS * ITEMS;
  constexpr int COLS = ITEMS / RS_ITEMS;   // table columns (2048-key chunks) per workgroup
  if (blockIdx.x * CHUNK >= n) return;
  // [wave][digit]: in the ranking loop the lanes of a wave index by digit -> distinct LDS banks
  // ([digit][wave] put every access of a wave on 16 of the 64 banks)
  __shared__ uint32_t s_cnt[4 * RS_MAX_RADIX];
  __shared__ uint32_t s_gbase[RS_MAX_RADIX];    // global position of local slot 0 of each digit
  __shared__ uint32_t s_wave[4];
  __shared__ uint32_t s_keys[CHUNK];
  __shared__ uint32_t s_vals[CHUNK];

  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bits = CBITS > 0 ? CBITS : bits_rt;
  const uint32_t mask = (1u << bits) - 1u;
  const uint32_t chunk = blockIdx.x;
  const uint32_t chunk_base = chunk * CHUNK;
  const uint32_t chunk_n = min((uint32_t)CHUNK, n - chunk_base);

#pragma unroll
  for (int k = 0; k < 4; k++) s_cnt[k * RS_MAX_RADIX + tid] = 0;
  __syncthreads();

  uint32_t key[ITEMS], val[ITEMS], rnk[ITEMS];
  const uint64_t lt = (1ull << lane) - 1ull;
  // all of the wave's loads first (one memory round trip per workgroup instead of RS_ITEMS)
#pragma unroll
  for (int i = 0; i < ITEMS; i++) {
    const uint32_t local = wave * (ITEMS * 64) + i * 64 + lane;
    const uint32_t idx = chunk_base + local;
    const bool valid = local < chunk_n;
    key[i] = valid ? keys_in[idx] : 0xFFFFFFFFu;
    val[i] = valid ? (vals_in ? vals_in[idx] : idx) : 0u;
  }
#pragma unroll
  for (int i = 0; i < ITEMS; i++) {
    const uint32_t local = wave * (ITEMS * 64) + i * 64 + lane;
    const bool valid = local < chunk_n;
    const uint32_t d = (key[i] >> shift) & mask;
    // match-any over the wave: lanes holding the same digit
    uint64_t peers = __builtin_amdgcn_ballot_w64(valid);
#pragma unroll
    for (int b = 0; b < (CBITS > 0 ? CBITS : RS_MAX_BITS); b++) {
      if (CBITS == 0 && b >= bits) break;
      const uint64_t bal = __builtin_amdgcn_ballot_w64(((d >> b) & 1u) != 0u);
      // lanes whose bit b equals mine:  ~(bal ^ mine),  mine = all-ones if my bit is set
      const uint32_t mine = (uint32_t)(((int32_t)(d << (31 - b))) >> 31);
      peers &= ~(bal ^ (((uint64_t)mine << 32) | mine));
    }
    const uint32_t before = (uint32_t)__popcll(peers & lt);
    const uint32_t prev = valid ? s_cnt[wave * RS_MAX_RADIX + d] : 0u;
    rnk[i] = prev + before;
    if (valid && before == 0) s_cnt[wave * RS_MAX_RADIX + d] = prev + (uint32_t)__popcll(peers);
  }
  __syncthreads();

  // Thread d owns digit d: turn per-(digit,wave) counts into local start slots, and compute the
  // global base of the digit for this chunk.
  {
    const uint32_t c0 = s_cnt[tid], c1 = s_cnt[RS_MAX_RADIX + tid], c2 = s_cnt[2 * RS_MAX_RADIX + tid],
                   c3 = s_cnt[3 * RS_MAX_RADIX + tid];
    const uint32_t dsum = c0 + c1 + c2 + c3;
    uint32_t tot;
    const uint32_t dstart = block_exclusive_scan_256(dsum, s_wave, &tot);
    const uint32_t gtot = tid <= mask ? totals[tid] : 0u;
    uint32_t tot2;
    const uint32_t gstart = block_exclusive_scan_256(gtot, s_wave, &tot2);
    s_cnt[tid] = dstart;
    s_cnt[RS_MAX_RADIX + tid] = dstart + c0;
    s_cnt[2 * RS_MAX_RADIX + tid] = dstart + c0 + c1;
    s_cnt[3 * RS_MAX_RADIX + tid] = dstart + c0 + c1 + c2;
    const uint32_t tb = tid <= mask ? table[(size_t)tid * nchunks + chunk * COLS] : 0u;
    s_gbase[tid] = gstart + tb - dstart;  // wraps mod 2^32; only used as base + slot
  }
  __syncthreads();

#pragma unroll
  for (int i = 0; i < ITEMS; i++) {
    const uint32_t local = wave * (ITEMS * 64) + i * 64 + lane;
    if (local < chunk_n) {
      const uint32_t d = (key[i] >> shift) & mask;
      const uint32_t slot = s_cnt[wave * RS_MAX_RADIX + d] + rnk[i];
      s_keys[slot] = key[i];
      s_vals[slot] = val[i];
    }
  }
  __syncthreads();

  // last pass of the depth sort: also emit the per-Gaussian tile count in sorted order, so the
  // offsets scan streams a contiguous array instead of gathering tiles[gid[i]].  The gather loads
  // are issued for all of the thread's items before the first store (one round trip, not 8).
  uint32_t gsrc[ITEMS];
  if (gather_src) {
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
      const uint32_t j = i * RS_THREADS + tid;
      gsrc[i] = j < chunk_n ? gather_src[s_vals[j]] : 0u;
    }
  }
#pragma unroll
  for (int i = 0; i < ITEMS; i++) {
    const uint32_t j = i * RS_THREADS + tid;
    if (j < chunk_n) {
      const uint32_t k = s_keys[j];
      const uint32_t d = (k >> shift) & mask;
      const uint32_t g = s_gbase[d] + j;
      keys_out[g] = k;
      vals_out[g] = s_vals[j];
      if (gather_src) gather_dst[g] = gsrc[i];
    }
  }
}

// ------------------------------------------------------------------------------------------
// Fat depth sort (P <= DS_MAX_CHUNKS * 8192): one scatter launch per 9-bit pass, THREE passes.
//
// The reference sorts all 32 depth bits (rasterizer_impl.cu:303-311).  Visible depths are floats
// > 0.2 whose bit patterns span far fewer than 2^32 values: with key_base = the smallest visible
// key rounded down to a multiple of 2^18, every visible key satisfies key - key_base < 2^27 as long
// as the farthest visible Gaussian is less than ~65 000 times as far away as the nearest one.  The
// order of the keys is the order of (key - key_base), and subtracting a multiple of 2^18 leaves the
// low 18 bits alone: passes 0 and 1 take bits [0,9) and [9,18) of the RAW key (so preprocess can
// histogram pass 0 before anybody knows the minimum), pass 2 takes bits [18,27) of key - key_base.
// When the range test fails (key_far != 0, decided on the device from preprocess' per-workgroup
// min / max), pass 2 writes into the spare buffers and pass 3 sorts the remaining bits [27,32) of
// key - key_base; otherwise pass 2 writes the final arrays and the two launches of pass 3 exit at
// once.  Either way the result is the exact stable (depth_bits, id) order, in the same arrays.
//
// The count table of a pass is [chunk][digit] (u32).  A workgroup sweeps the rows of ALL chunks
// itself (245 rows x 2 KB at P = 2 M, L2-resident): per digit the sum over the earlier chunks (its
// base inside the digit's region) and the grand total (exclusive scan over digits = start of the
// region).  Pass 0 reads the P depth keys written by preprocess, whose workgroups also left
// table 0 behind; culled Gaussians (CULLED_KEY) are dropped right there, so from pass 1 on the
// arrays hold only the V visible ones.  V = sum of table 0, published by chunk 0 of pass 0.
// The tables of the later passes come from depth_hist_kernel, launched after the previous scatter;
// they are written for the ceil(V / 8192) chunks that hold data and swept for exactly those.
// Measured dead ends: accumulating the offsets scan's per-2048 block sums in the last pass with one
// global atomic per wave (23 k agent-scope atomics on ~45 cache lines: 175 us; the separate reduce
// launch costs 5), gathering the tile counts into sorted order in the last pass (one
// 1024-thread workgroup per CU keeps too few random loads in flight: +38 us), the next pass's
// counts by one global atomic per key (0.096 -> 0.42 ms), 11-bit digits (a [2048][chunks] table
// costs more than the pass it saves), 512-thread workgroups.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v, const uint32_t lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t t = __shfl_up(v, d, 64);
    if (lane >= (uint32_t)d) v += t;
  }
  return v;
}

// digit of pass PASS; `base` = key_base (used from pass 2 on)
template <int PASS>
__device__ __forceinline__ uint32_t ds_digit(const uint32_t key, const uint32_t base) {
  if (PASS == 0) return key & (DS_RADIX - 1);
  if (PASS == 1) return (key >> DS_BITS) & (DS_RADIX - 1);
  if (PASS == 2) return ((key - base) >> (2 * DS_BITS)) & (DS_RADIX - 1);
  return (key - base) >> (3 * DS_BITS);   // 5 bits
}

// range[0] = key_base, range[1] = key_far (geometry header, written by pass 0's publishing workgroup)
template <int PASS>
__global__ void __launch_bounds__(DS_THREADS)
depth_hist_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ n_dev,
                  const uint32_t* __restrict__ range,
                  uint32_t* __restrict__ table /* [chunk][DS_RADIX] of this pass */) {
  __shared__ uint32_t h[DS_RADIX];
  if (PASS == 3 && range[1] == 0u) return;   // three passes settle the order
  const uint32_t n = *n_dev;
  const uint32_t base = blockIdx.x * DS_CHUNK;
  if (base >= n) return;   // rows beyond the data are never swept
  const uint32_t kb = PASS >= 2 ? range[0] : 0u;
  if (threadIdx.x < DS_RADIX) h[threadIdx.x] = 0;
  __syncthreads();
  uint32_t kk[DS_ITEMS];
#pragma unroll
  for (int k = 0; k < DS_ITEMS; k++) {
    const uint32_t idx = base + k * DS_THREADS + threadIdx.x;
    kk[k] = idx < n ? keys[idx] : 0u;
  }
#pragma unroll
  for (int k = 0; k < DS_ITEMS; k++) {
    const uint32_t idx = base + k * DS_THREADS + threadIdx.x;
    if (idx < n) atomicAdd(&h[ds_digit<PASS>(kk[k], kb)], 1u);
  }
  __syncthreads();
  if (threadIdx.x < DS_RADIX) table[(size_t)blockIdx.x * DS_RADIX + threadIdx.x] = h[threadIdx.x];
}

// Payload of the hierarchical binning (RECT): a Gaussian's tile rectangle travels with its (key, id)
// pair, packed into one word (8 bits per bound: grids of <= 255 x 255 tiles).  Pass 0 reads the
// rectangles in id order -- fully coalesced, its ids ARE its positions -- and the last pass leaves
// them behind in DEPTH order (rect_sorted) together with the super-tile counts the coarse scan
// sums: the scan's random gather by sorted id (26 us: 8 useful bytes per line) disappears for
// 4 bytes per pair and pass.
struct RectPayload {
  const uint2* rects_by_id;   // pass 0 input
  const uint32_t* aux_in;     // later passes' input (packed)
  uint32_t* aux_out;          // output of a pass that is not the last (packed)
  uint2* rect_sorted;         // the last pass' output
  uint32_t* counts_sorted;    // the last pass' output: super-tiles per Gaussian, depth order
};
__device__ __forceinline__ uint32_t rect_pack(const uint2 r) {
  return (r.x & 0xFFu) | ((r.x >> 16) << 8) | ((r.y & 0xFFu) << 16) | ((r.y >> 16) << 24);
}
__device__ __forceinline__ uint2 rect_unpack(const uint32_t p) {
  return make_uint2((p & 0xFFu) | (((p >> 8) & 0xFFu) << 16), ((p >> 16) & 0xFFu) | ((p >> 24) << 16));
}

// The frame's count publish rides in pass 0 (its last, usually partial, chunk's workgroup): the sum
// of preprocess' per-workgroup (instances, coarse pairs) goes to the pinned host words and the
// geometry header, the min / max of the visible depth keys become key_base / key_far; the host's
// event is recorded behind this launch.
struct CountPublish {
  const uint4* pre_counts;   // per preprocess workgroup: instances, coarse pairs, min key, max key
  uint32_t nblocks;
  uint32_t* host_word;       // pinned, device-mapped: [0] num_rendered, [1] coarse pairs, [2] key_far (may be NULL)
  uint32_t* header_words;    // geometry header: R_pre, Rc_pre, key_base, key_far
};

// Sum / min / max of preprocess' per-workgroup words -> header (+ pinned host words).  Called by
// all threads of ONE workgroup of NT threads.
template <int NT>
__device__ __forceinline__ void publish_frame_counts(const CountPublish pub) {
  constexpr int NW = NT / 64;
  __shared__ unsigned long long s_pr[NW], s_pc[NW];
  __shared__ uint32_t s_mn[NW], s_mx[NW];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned long long r = 0ull, c = 0ull;
  uint32_t mn = 0xFFFFFFFFu, mx = 0u;
  for (uint32_t i = tid; i < pub.nblocks; i += NT) {
    const uint4 v = pub.pre_counts[i];
    r += v.x; c += v.y;
    mn = min(mn, v.z); mx = max(mx, v.w);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    r += __shfl_xor(r, o);
    c += __shfl_xor(c, o);
    mn = min(mn, (uint32_t)__shfl_xor((int)mn, o));
    mx = max(mx, (uint32_t)__shfl_xor((int)mx, o));
  }
  if (lane == 0) { s_pr[wave] = r; s_pc[wave] = c; s_mn[wave] = mn; s_mx[wave] = mx; }
  __syncthreads();
  if (tid == 0) {
    r = 0ull; c = 0ull; mn = 0xFFFFFFFFu; mx = 0u;
    for (int w = 0; w < NW; w++) {
      r += s_pr[w]; c += s_pc[w];
      mn = min(mn, s_mn[w]); mx = max(mx, s_mx[w]);
    }
    const uint32_t r32 = r > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)r;   // saturating
    const uint32_t c32 = c > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)c;
    // nothing visible: mn > mx, any base will do
    const uint32_t kbase = mn <= mx ? (mn & ~((1u << (2 * DS_BITS)) - 1u)) : 0u;
    const uint32_t kfar = (mn <= mx && ((mx - kbase) >> (3 * DS_BITS)) != 0u) ? 1u : 0u;
    if (pub.header_words) {
      pub.header_words[0] = r32; pub.header_words[1] = c32;
      pub.header_words[2] = kbase; pub.header_words[3] = kfar;
    }
    if (pub.host_word) {
      pub.host_word[0] = r32; pub.host_word[1] = c32; pub.host_word[2] = kfar;
      __threadfence_system();
    }
  }
}

__global__ void __launch_bounds__(1024)
publish_counts_kernel(const CountPublish pub) { publish_frame_counts<1024>(pub); }

void launch_publish_counts(hipStream_t s, const uint4* pre_counts, uint32_t nblocks,
                           uint32_t* host_word, uint32_t* header_words) {
  const CountPublish pub = {pre_counts, nblocks, host_word, header_words};
  publish_counts_kernel<<<1, 1024, 0, s>>>(pub);
}

// Where a pass writes.  Pass 2 chooses on the device: `near` when three passes settle the order
// (it is the last pass then), `far` when pass 3 has to follow.
struct PassOut {
  uint32_t* keys;
  uint32_t* vals;
};

#ifdef GRPG_DS_TRACE   // experiment build (tools/gpu_ds_trace.sh): phase timestamps of a few workgroups
__device__ unsigned long long g_ds_trace[4][4][10];   // [pass][probe workgroup][phase]
#define DS_TRACE(PH)                                                                             \
  do {                                                                                           \
    if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == 50 || blockIdx.x == 120 || blockIdx.x == 170)) \
      g_ds_trace[PASS][blockIdx.x == 0 ? 0 : (blockIdx.x == 50 ? 1 : (blockIdx.x == 120 ? 2 : 3))][PH] = wall_clock64(); \
  } while (0)
#else
#define DS_TRACE(PH) do {} while (0)
#endif

// GRPG_SORT_LEAN (experiment build): 96 instead of 126 registers and 101 instead of 133 KB of LDS per
// workgroup, so that one workgroup of another stream's render / preprocess fits beside a sort
// workgroup on its CU.  The staging area becomes dynamic LDS (the compiler honours a register cap only
// when it cannot see that the LDS allows one workgroup per CU anyway), the per-wave digit counters
// move into the payload plane, which is dead until the staging, and the table sweep keeps 4 instead
// of 8 rows in flight per wave.
#ifdef GRPG_SORT_LEAN
#define DS_KERNEL_ATTR __attribute__((amdgpu_waves_per_eu(5, 5)))
constexpr bool DS_LEAN = true;
#else
#define DS_KERNEL_ATTR
constexpr bool DS_LEAN = false;
#endif
constexpr size_t DS_DYN_LDS = DS_LEAN ? 3 * DS_CHUNK * sizeof(uint32_t) : 0;

template <int PASS, bool RECT>
DS_KERNEL_ATTR __global__ void __launch_bounds__(DS_THREADS)
depth_scatter_kernel(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                     const PassOut out_near, const PassOut out_far,
                     const uint32_t P, const uint32_t* __restrict__ table /* [nchunks][DS_RADIX] */,
                     const uint32_t nchunks, uint32_t* __restrict__ V_out,
                     const uint32_t* __restrict__ range /* key_base, key_far */, const int allow_far,
                     const RectPayload rp, const CountPublish pub) {
  // keys | values | payload staged for the coalesced run writes; the per-wave partial column sums
  // of the table sweep live in the first two thirds until the keys are staged
#ifdef GRPG_SORT_LEAN
  extern __shared__ uint32_t s_dyn[];
  uint32_t* const s_buf = s_dyn;
  uint32_t* const s_cnt = s_dyn + 2 * DS_CHUNK;     // RECT: the payload plane, staged into only after the last s_cnt read
#else
  __shared__ uint32_t s_buf[(RECT ? 3 : 2) * DS_CHUNK];
  __shared__ uint32_t s_cnt[DS_WAVES * DS_RADIX];   // [wave][digit]: bank-conflict-free ranking
#endif
  __shared__ uint32_t s_gbase[DS_RADIX];
  __shared__ uint32_t s_w[2 * (DS_RADIX / 64)];
  uint32_t* const s_keys = s_buf;
  uint32_t* const s_vals = s_buf + DS_CHUNK;
  uint32_t* const s_aux = s_buf + (RECT ? 2 : 0) * DS_CHUNK;
  uint32_t* const s_pex = s_buf;                          // [DS_WAVES][DS_RADIX]
  uint32_t* const s_ptot = s_buf + DS_WAVES * DS_RADIX;   // [DS_WAVES][DS_RADIX]
  static_assert(2 * DS_WAVES * DS_RADIX <= 2 * DS_CHUNK, "sweep partials must fit in the staging area");
  constexpr int DW = DS_RADIX / 64;   // waves whose threads own a digit

  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t chunk = blockIdx.x;
  DS_TRACE(0);
  // allow_far == 0: the host enqueued no fourth pass behind pass 2 (the previous frames of this shape
  // did not need one); should THIS frame need it, its order is wrong and the host, which sees key_far
  // next to num_rendered, renders the frame again with the fourth pass (api.hip)
  const uint32_t far = (PASS >= 2 && allow_far) ? range[1] : 0u;
  if (PASS == 3 && far == 0u) return;   // pass 2 was the last one
  const uint32_t kb = PASS >= 2 ? range[0] : 0u;
  const bool last = PASS == 3 || (PASS == 2 && far == 0u);
  uint32_t* const keys_out = (PASS == 2 && far != 0u) ? out_far.keys : out_near.keys;
  uint32_t* const vals_out = (PASS == 2 && far != 0u) ? out_far.vals : out_near.vals;
  if (PASS == 0 && pub.pre_counts != nullptr && chunk == nchunks - 1u) publish_frame_counts<DS_THREADS>(pub);

  // ---- the chunk's own loads go out first: they overlap the table sweep below.  Pass 0 moves
  // P keys; later passes move the V visible ones (published by chunk 0 of pass 0). ----
  const uint32_t n_in = PASS == 0 ? P : *V_out;
  const uint32_t nrows = PASS == 0 ? nchunks : min(nchunks, (n_in + DS_CHUNK - 1) / DS_CHUNK);
  const uint32_t chunk_base = chunk * DS_CHUNK;
  const uint32_t chunk_n = chunk_base < n_in ? min((uint32_t)DS_CHUNK, n_in - chunk_base) : 0u;
  if (PASS > 0 && chunk_n == 0u) return;   // a chunk beyond the data (pass 0: chunk 0 still owes V)
  uint32_t key[DS_ITEMS], val[DS_ITEMS], rnk[DS_ITEMS], aux[RECT ? DS_ITEMS : 1];
  uint2 rraw[(RECT && PASS == 0) ? DS_ITEMS : 1];   // pass 0: packed only when staged (nothing waits for the loads here)
#pragma unroll
  for (int i = 0; i < DS_ITEMS; i++) {
    const uint32_t local = wave * (DS_ITEMS * 64) + i * 64 + lane;
    const uint32_t idx = chunk_base + local;
    const bool inb = local < chunk_n;
    key[i] = inb ? keys_in[idx] : CULLED_KEY;
    val[i] = PASS == 0 ? idx : (inb ? vals_in[idx] : 0u);
    if (RECT) {   // a culled Gaussian's rectangle was never written: loaded, never used
      if (PASS == 0) rraw[i] = inb ? rp.rects_by_id[idx] : make_uint2(0u, 0u);
      else aux[i] = inb ? rp.aux_in[idx] : 0u;
    }
  }

  DS_TRACE(1);
  // ---- sweep the count table: per digit, sum over earlier chunks and over all chunks ----
  // A wave reads whole 2 KB rows (lane l: digits 8l..8l+7, two 16-byte loads); wave w owns rows
  // w, w+16, ...; 8 rows (16 loads) are in flight per wave: 245 rows cost two round trips.
  // (Measured: group rows -- the chunk rows of 16 chunks summed, so that a workgroup sweeps 31
  // instead of 245 rows -- save 3.5 us per pass, but producing them costs far more: +21 us in
  // preprocess for the second atomic per digit, +60 us per histogram launch for the release / acquire
  // fences of a last-arriver reduction (an agent-scope fence writes back and invalidates the XCD's L2).)
  {
    const uint4* t4 = reinterpret_cast<const uint4*>(table);
    uint4 exa = make_uint4(0u, 0u, 0u, 0u), exb = exa, tota = exa, totb = exa;
    constexpr int SW = DS_LEAN ? 4 : 8;
    for (uint32_t r0 = wave; r0 < nrows; r0 += DS_WAVES * SW) {
      uint4 va[SW], vb[SW];
#pragma unroll
      for (int k = 0; k < SW; k++) {
        const uint32_t r = r0 + (uint32_t)k * DS_WAVES;
        const bool in = r < nrows;
        va[k] = in ? t4[(size_t)r * (DS_RADIX / 4) + 2 * lane] : make_uint4(0u, 0u, 0u, 0u);
        vb[k] = in ? t4[(size_t)r * (DS_RADIX / 4) + 2 * lane + 1] : make_uint4(0u, 0u, 0u, 0u);
      }
#pragma unroll
      for (int k = 0; k < SW; k++) {
        const uint32_t r = r0 + (uint32_t)k * DS_WAVES;
        tota.x += va[k].x; tota.y += va[k].y; tota.z += va[k].z; tota.w += va[k].w;
        totb.x += vb[k].x; totb.y += vb[k].y; totb.z += vb[k].z; totb.w += vb[k].w;
        if (r < chunk) {
          exa.x += va[k].x; exa.y += va[k].y; exa.z += va[k].z; exa.w += va[k].w;
          exb.x += vb[k].x; exb.y += vb[k].y; exb.z += vb[k].z; exb.w += vb[k].w;
        }
      }
    }
    reinterpret_cast<uint4*>(s_pex + wave * DS_RADIX)[2 * lane] = exa;
    reinterpret_cast<uint4*>(s_pex + wave * DS_RADIX)[2 * lane + 1] = exb;
    reinterpret_cast<uint4*>(s_ptot + wave * DS_RADIX)[2 * lane] = tota;
    reinterpret_cast<uint4*>(s_ptot + wave * DS_RADIX)[2 * lane + 1] = totb;
  }
  DS_TRACE(2);
#pragma unroll
  for (int k = 0; k < (DS_RADIX * DS_WAVES) / DS_THREADS; k++) s_cnt[k * DS_THREADS + tid] = 0;
  __syncthreads();
  DS_TRACE(3);
  uint32_t dex = 0, dtot = 0, inc = 0;
  if (tid < DS_RADIX) {
#pragma unroll
    for (int w = 0; w < DS_WAVES; w++) { dex += s_pex[w * DS_RADIX + tid]; dtot += s_ptot[w * DS_RADIX + tid]; }
    inc = wave_inclusive_sum(dtot, lane);
    if (lane == 63) s_w[wave] = inc;
  }
  __syncthreads();
  uint32_t n_all = 0;   // elements this pass moves (= V)
#pragma unroll
  for (int w = 0; w < DW; w++) n_all += s_w[w];
  if (tid < DS_RADIX) {
    uint32_t before = 0;
#pragma unroll
    for (int w = 0; w < DW - 1; w++) before += (uint32_t)w < wave ? s_w[w] : 0u;
    s_gbase[tid] = before + inc - dtot + dex;   // first output position of (digit, this chunk)
  }
  if (PASS == 0 && chunk == 0 && tid == 0) *V_out = n_all;
  if (chunk_n == 0u) return;   // whole workgroup (pass 0: a chunk beyond the data)

  DS_TRACE(4);
  // ---- rank (wave w owns the 512 consecutive keys [512 w, 512 w + 512)) ----
  const uint64_t lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int i = 0; i < DS_ITEMS; i++) {
    const uint32_t local = wave * (DS_ITEMS * 64) + i * 64 + lane;
    // pass 0 drops culled Gaussians; later passes hold visible ones only (a visible depth key is
    // the bit pattern of a float > 0.2, never CULLED_KEY)
    const bool valid = local < chunk_n && (PASS > 0 || key[i] != CULLED_KEY);
    const uint32_t d = ds_digit<PASS>(key[i], kb) & (DS_RADIX - 1);
    uint64_t peers = __builtin_amdgcn_ballot_w64(valid);
#pragma unroll
    for (int b = 0; b < (PASS == 3 ? 32 - 3 * DS_BITS : DS_BITS); b++) {
      const uint64_t bal = __builtin_amdgcn_ballot_w64(((d >> b) & 1u) != 0u);
      const uint32_t mine = (uint32_t)(((int32_t)(d << (31 - b))) >> 31);
      peers &= ~(bal ^ (((uint64_t)mine << 32) | mine));
    }
    const uint32_t before = (uint32_t)__popcll(peers & lt);
    const uint32_t prev = valid ? s_cnt[wave * DS_RADIX + d] : 0u;
    rnk[i] = valid ? prev + before : 0xFFFFFFFFu;
    if (valid && before == 0) s_cnt[wave * DS_RADIX + d] = prev + (uint32_t)__popcll(peers);
  }
  DS_TRACE(5);
  __syncthreads();
  DS_TRACE(6);

  // ---- thread d < DS_RADIX owns digit d: local start slots of its 16 (digit, wave) runs ----
  uint32_t c[DS_WAVES], dsum = 0, inc2 = 0;
  if (tid < DS_RADIX) {
#pragma unroll
    for (int w = 0; w < DS_WAVES; w++) { c[w] = s_cnt[w * DS_RADIX + tid]; dsum += c[w]; }
    inc2 = wave_inclusive_sum(dsum, lane);
    if (lane == 63) s_w[DW + wave] = inc2;
  }
  __syncthreads();
  uint32_t nvalid = 0;
#pragma unroll
  for (int w = 0; w < DW; w++) nvalid += s_w[DW + w];
  if (tid < DS_RADIX) {
    uint32_t before = 0;
#pragma unroll
    for (int w = 0; w < DW - 1; w++) before += (uint32_t)w < wave ? s_w[DW + w] : 0u;
    const uint32_t dstart = before + inc2 - dsum;
    uint32_t run = dstart;
#pragma unroll
    for (int w = 0; w < DS_WAVES; w++) { s_cnt[w * DS_RADIX + tid] = run; run += c[w]; }
    s_gbase[tid] -= dstart;   // wraps mod 2^32; only used as base + local slot
  }
  __syncthreads();
  if (DS_LEAN) {   // the counters share LDS with the payload plane: every slot is read before anything is staged
#pragma unroll
    for (int i = 0; i < DS_ITEMS; i++) {
      if (rnk[i] != 0xFFFFFFFFu)
        rnk[i] += s_cnt[wave * DS_RADIX + (ds_digit<PASS>(key[i], kb) & (DS_RADIX - 1))];
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < DS_ITEMS; i++) {
    if (rnk[i] != 0xFFFFFFFFu) {
      const uint32_t d = ds_digit<PASS>(key[i], kb) & (DS_RADIX - 1);
      const uint32_t slot = DS_LEAN ? rnk[i] : s_cnt[wave * DS_RADIX + d] + rnk[i];
      s_keys[slot] = key[i];
      s_vals[slot] = val[i];
      if (RECT) s_aux[slot] = PASS == 0 ? rect_pack(rraw[i]) : aux[i];
    }
  }
  __syncthreads();
  DS_TRACE(7);

  // ---- coalesced run writes ----
#pragma unroll
  for (int i = 0; i < DS_ITEMS; i++) {
    const uint32_t j = i * DS_THREADS + tid;
    if (j < nvalid) {
      const uint32_t k = s_keys[j];
      const uint32_t g = s_gbase[ds_digit<PASS>(k, kb) & (DS_RADIX - 1)] + j;
      keys_out[g] = k;
      vals_out[g] = s_vals[j];
      if (RECT) {
        if (PASS < 2 || !last) {
          rp.aux_out[g] = s_aux[j];
        } else {
          const uint2 r = rect_unpack(s_aux[j]);
          rp.rect_sorted[g] = r;
          rp.counts_sorted[g] = rect_super_tiles(r.x, r.y);
        }
      }
    }
  }
  DS_TRACE(8);
}

// Depth sort of the P (key, id) pairs; ids are implicit in pass 0.  Result: (key_a, val_a) hold
// the V visible pairs in (depth_bits, id) order, *V_out = V (the per-Gaussian tile counts are
// brought into sorted order by the offsets scan's reduce launch -- or, with the rectangle payload
// below, arrive in sorted order with the last pass).  Table 0 of ds_table must hold preprocess'
// pass-0 counts; pre_counts = preprocess'
// per-workgroup (instances, coarse pairs, min key, max key); `range` = the geometry header's
// (key_base, key_far) words, written by pass 0 when publish_here, by launch_publish_counts before
// this call otherwise.  key_a (the raw keys by id) is consumed; (key_b, val_b) and (key_c, val_c) are
// scratch.  5 launches; with_pass3: two more, which exit at once unless the frame's depth range
// needs the fourth pass.  !with_pass3 and key_far != 0 (the host learns it with num_rendered): the
// order is wrong, the caller renders the frame again with_pass3.
// rects_by_id != NULL (hierarchical binning, grid <= 255 x 255 tiles): the tile rectangles ride along
// (RectPayload above); aux_a / aux_b / aux_c are scratch arrays of P words, rect_sorted [P] and
// counts_sorted [P] receive the rectangles and super-tile counts in depth order.  aux_a may alias
// rect_sorted (it is dead before the last pass writes that array); aux_b and aux_c must not.
void depth_sort_fat(hipStream_t s, uint32_t P, uint32_t* key_a, uint32_t* val_a, uint32_t* key_b,
                    uint32_t* val_b, uint32_t* key_c, uint32_t* val_c, uint32_t* ds_table,
                    uint32_t nchunks, uint32_t* V_out, const uint32_t* range, bool with_pass3, const uint2* rects_by_id, uint32_t* aux_a, uint32_t* aux_b,
                    uint32_t* aux_c, uint2* rect_sorted, uint32_t* counts_sorted,
                    const uint4* pre_counts, uint32_t pre_nblocks, bool publish_here,
                    uint32_t* count_host_word, uint32_t* count_header_words, hipEvent_t count_event) {
  if (P == 0) return;
#ifdef GRPG_SORT_LEAN
  static const bool lds_ok = [] {   // more than 64 KB of dynamic LDS has to be asked for, once per kernel
    auto ask = [](const void* f) {
      return hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DS_DYN_LDS) == hipSuccess;
    };
    return ask((const void*)depth_scatter_kernel<0, true>) & ask((const void*)depth_scatter_kernel<1, true>) &
           ask((const void*)depth_scatter_kernel<2, true>) & ask((const void*)depth_scatter_kernel<3, true>) &
           ask((const void*)depth_scatter_kernel<0, false>) & ask((const void*)depth_scatter_kernel<1, false>) &
           ask((const void*)depth_scatter_kernel<2, false>) & ask((const void*)depth_scatter_kernel<3, false>);
  }();
  (void)lds_ok;
#endif
  const size_t tsz = (size_t)nchunks * DS_RADIX;
  const bool rect = rects_by_id != nullptr;
  const int allow_far = with_pass3 ? 1 : 0;
#define DS_SCATTER(PASS, KI, VI, KN, VN, KF, VF, AI, AO)                                         \
  do {                                                                                           \
    const RectPayload rp = {rects_by_id, AI, AO, rect_sorted, counts_sorted};                    \
    const CountPublish pub = {(PASS == 0 && publish_here) ? pre_counts : nullptr, pre_nblocks,   \
                              count_host_word, count_header_words};                              \
    const PassOut on = {KN, VN}, of = {KF, VF};                                                  \
    if (rect)                                                                                    \
      depth_scatter_kernel<PASS, true><<<nchunks, DS_THREADS, DS_DYN_LDS, s>>>(                  \
          KI, VI, on, of, P, ds_table + PASS * tsz, nchunks, V_out, range, allow_far, rp, pub);  \
    else                                                                                         \
      depth_scatter_kernel<PASS, false><<<nchunks, DS_THREADS, DS_DYN_LDS, s>>>(                 \
          KI, VI, on, of, P, ds_table + PASS * tsz, nchunks, V_out, range, allow_far, rp, pub);  \
  } while (0)
#define DS_HIST(PASS, K)                                                                         \
  depth_hist_kernel<PASS><<<nchunks, DS_THREADS, 0, s>>>(K, V_out, range, ds_table + PASS * tsz)
  DS_SCATTER(0, key_a, nullptr, key_b, val_b, nullptr, nullptr, nullptr, aux_a);
  if (publish_here && count_event != nullptr) (void)hipEventRecord(count_event, s);
  DS_HIST(1, key_b);
  DS_SCATTER(1, key_b, val_b, key_c, val_c, nullptr, nullptr, aux_a, aux_b);
  DS_HIST(2, key_c);
  // three passes settle it: c -> a (final).  Otherwise c -> b, and pass 3: b -> a.
  DS_SCATTER(2, key_c, val_c, key_a, val_a, key_b, val_b, aux_b, aux_c);
  if (with_pass3) {   // both launches exit at once when key_far == 0
    DS_HIST(3, key_b);
    DS_SCATTER(3, key_b, val_b, key_a, val_a, nullptr, nullptr, aux_c, nullptr);
  }
#undef DS_HIST
#undef DS_SCATTER
}

#ifdef GRPG_DS_TRACE
}  // namespace grpg
extern "C" __attribute__((visibility("default"))) int grpg_debug_ds_trace(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(grpg::g_ds_trace), sizeof(grpg::g_ds_trace));
}
namespace grpg {
#endif

int radix_pass_bits(int begin_bit, int end_bit, int p);

// Classic passes.  n = host-side upper bound (sizes the grids), n_dev = device-side element count
// (may be NULL: n is exact).  have_hist0: the caller already filled table[digit][chunk] for the
// first pass (emit_kernel does, for the tile partition).
bool radix_sort_pairs(hipStream_t s, uint32_t n, const uint32_t* n_dev, uint32_t* key_a,
                      uint32_t* val_a, uint32_t* key_b, uint32_t* val_b, bool vals_iota,
                      int begin_bit, int end_bit, uint32_t* table, uint32_t* totals,
                      uint32_t nchunks, bool have_hist0, const uint32_t* gather_src,
                      uint32_t* gather_dst) {
  bool in_b = false;
  if (n == 0) return in_b;
  const int nbits = end_bit - begin_bit;
  if (nbits <= 0) return in_b;
  const int passes = (nbits + RS_MAX_BITS - 1) / RS_MAX_BITS;
  int shift = begin_bit;
  for (int p = 0; p < passes; p++) {
    const int bits = radix_pass_bits(begin_bit, end_bit, p);
    const uint32_t mask = (1u << bits) - 1u;
    uint32_t* kin = in_b ? key_b : key_a;
    uint32_t* vin = in_b ? val_b : val_a;
    uint32_t* kout = in_b ? key_a : key_b;
    uint32_t* vout = in_b ? val_a : val_b;
    if (!(p == 0 && have_hist0))
      radix_hist_kernel<<<nchunks, RS_THREADS, 0, s>>>(kin, n, n_dev, shift, mask, table, nchunks);
    radix_digit_scan_kernel<<<1u << bits, 256, 0, s>>>(table, nchunks, n, n_dev, totals);
#define RS_SCATTER(CB, IT)                                                                    \
  radix_scatter_kernel<CB, IT><<<(nchunks + (IT / RS_ITEMS) - 1) / (IT / RS_ITEMS), RS_THREADS, 0, s>>>( \
      kin, (p == 0 && vals_iota) ? nullptr : vin, kout, vout, n, n_dev, shift, bits, table,   \
      totals, nchunks, p == passes - 1 ? gather_src : nullptr, gather_dst)
    if (bits == 8) RS_SCATTER(8, 8);        // depth sort of very large P
    else if (bits == 7) RS_SCATTER(7, 8);   // tile partition of a 1920x1280 frame (14 bits)
    else RS_SCATTER(0, 8);
#undef RS_SCATTER
    in_b = !in_b;
    shift += bits;
  }
  return in_b;
}

int radix_sort_num_passes(int begin_bit, int end_bit) {
  const int nbits = end_bit - begin_bit;
  return nbits <= 0 ? 0 : (nbits + RS_MAX_BITS - 1) / RS_MAX_BITS;
}
// Digit width of pass p: the bits are spread evenly over the passes (14 bits -> 7 + 7; measured
// against 6 + 8, 8 + 6 and 4096-key workgroups: within noise or worse).
int radix_pass_bits(int begin_bit, int end_bit, int p) {
  const int passes = radix_sort_num_passes(begin_bit, end_bit);
  const int nbits = end_bit - begin_bit;
  int shift = 0;
  int bits = 0;
  for (int q = 0; q <= p; q++) {
    bits = (nbits - shift + (passes - q) - 1) / (passes - q);
    shift += bits;
  }
  return bits;
}
int radix_sort_first_pass_bits(int begin_bit, int end_bit) {
  return radix_sort_num_passes(begin_bit, end_bit) == 0 ? 0 : radix_pass_bits(begin_bit, end_bit, 0);
}

// ------------------------------------------------------------------------------------------
// Exclusive scan of the per-Gaussian instance counts in depth-sorted order (written contiguously
// by the last depth-sort pass) -> offsets[i]; replaces cub::DeviceScan::InclusiveSum (rasterizer_impl.cu:280).
// Two launches: per-workgroup reduce, per-workgroup down-sweep (which recomputes the spine).
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(SC_THREADS)
scan_reduce_kernel(const uint32_t n_cap, const uint32_t* __restrict__ n_dev,
                   const uint32_t* __restrict__ tiles, uint32_t* __restrict__ block_sums,
                   const uint32_t* __restrict__ gid, const uint32_t* __restrict__ tiles_by_id,
                   uint32_t* __restrict__ tiles_out, const uint2* __restrict__ rects_by_id,
                   uint2* __restrict__ rect_out) {
  __shared__ uint32_t s_wave[4];
  const uint32_t n = n_dev ? min(*n_dev, n_cap) : n_cap;
  const uint32_t base = blockIdx.x * SC_CHUNK;
  if (base >= n) {   // the down-sweep sums all nblocks entries
    if (threadIdx.x == 0) block_sums[blockIdx.x] = 0u;
    return;
  }
  uint32_t v[SC_ITEMS], s = 0;   // all loads in flight together
  if (gid != nullptr) {
    // fat depth sort: the tile counts are still in id order.  Gather them here (this launch has
    // several times the memory parallelism of the sort's 1024-thread workgroups) and leave them
    // behind in sorted order for the down-sweep.
    uint32_t g[SC_ITEMS];
#pragma unroll
    for (int k = 0; k < SC_ITEMS; k++) {
      const uint32_t i = base + k * SC_THREADS + threadIdx.x;
      g[k] = i < n ? gid[i] : 0u;
    }
    if (rects_by_id != nullptr) {
      // hierarchical binning: the counts are super-tile counts, derived from the gathered tile
      // rectangles, which stay behind in sorted order for the coarse emit
      uint2 rc[SC_ITEMS];
#pragma unroll
      for (int k = 0; k < SC_ITEMS; k++) {
        const uint32_t i = base + k * SC_THREADS + threadIdx.x;
        rc[k] = i < n ? rects_by_id[g[k]] : make_uint2(0u, 0u);
      }
#pragma unroll
      for (int k = 0; k < SC_ITEMS; k++) {
        const uint32_t i = base + k * SC_THREADS + threadIdx.x;
        v[k] = rect_super_tiles(rc[k].x, rc[k].y);
        if (i < n) { tiles_out[i] = v[k]; rect_out[i] = rc[k]; }
      }
    } else {
#pragma unroll
      for (int k = 0; k < SC_ITEMS; k++) {
        const uint32_t i = base + k * SC_THREADS + threadIdx.x;
        v[k] = i < n ? tiles_by_id[g[k]] : 0u;
      }
#pragma unroll
      for (int k = 0; k < SC_ITEMS; k++) {
        const uint32_t i = base + k * SC_THREADS + threadIdx.x;
        if (i < n) tiles_out[i] = v[k];
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < SC_ITEMS; k++) {
      const uint32_t i = base + k * SC_THREADS + threadIdx.x;
      v[k] = i < n ? tiles[i] : 0u;
    }
  }
#pragma unroll
  for (int k = 0; k < SC_ITEMS; k++) s += v[k];
  uint32_t tot;
  block_exclusive_scan_256(s, s_wave, &tot);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(SC_THREADS)
scan_down_kernel(const uint32_t n_cap, const uint32_t* __restrict__ n_dev,
                 const uint32_t* __restrict__ tiles,
                 const uint32_t* __restrict__ block_sums, const uint32_t nblocks,
                 uint32_t* __restrict__ offsets, uint32_t* __restrict__ total_out,
                 uint32_t* __restrict__ total_host, uint32_t* __restrict__ emit_win,
                 const uint32_t emit_win_cap) {
  __shared__ uint32_t s_wave[4];
  __shared__ uint32_t s_spine[8];
  const uint32_t n = n_dev ? min(*n_dev, n_cap) : n_cap;
  // the spine (prefix of the workgroup sums before this workgroup, and the grand total) is
  // recomputed by every workgroup from the <= a few thousand L2-resident sums: cheaper than a
  // third launch with a single workgroup.  Sums of workgroups beyond n are zero.
  {
    uint32_t pre = 0, tot = 0;
    for (uint32_t i = threadIdx.x; i < nblocks; i += SC_THREADS) {
      const uint32_t v = block_sums[i];
      tot += v;
      pre += i < blockIdx.x ? v : 0u;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      pre += (uint32_t)__shfl_xor((int)pre, d, 64);
      tot += (uint32_t)__shfl_xor((int)tot, d, 64);
    }
    if ((threadIdx.x & 63) == 0) { s_spine[(threadIdx.x >> 6) * 2] = pre; s_spine[(threadIdx.x >> 6) * 2 + 1] = tot; }
    __syncthreads();
  }
  const uint32_t block_prefix = s_spine[0] + s_spine[2] + s_spine[4] + s_spine[6];
  const uint32_t total = s_spine[1] + s_spine[3] + s_spine[5] + s_spine[7];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    *total_out = total;
    // num_rendered for the host: a plain store into pinned, device-mapped host memory; the host
    // reads it after the event recorded behind this kernel has fired (api.hip)
    if (total_host) { total_host[0] = total; total_host[1] = n; }
  }
  if (blockIdx.x * SC_CHUNK >= n) return;
  // thread t owns SC_ITEMS consecutive elements so that the scan order is the array order
  const uint32_t base = blockIdx.x * SC_CHUNK + threadIdx.x * SC_ITEMS;
  uint32_t v[SC_ITEMS], s = 0;
#pragma unroll
  for (int k = 0; k < SC_ITEMS; k++) {
    const uint32_t i = base + k;
    v[k] = i < n ? tiles[i] : 0;
    s += v[k];
  }
  uint32_t tot;
  uint32_t ex = block_exclusive_scan_256(s, s_wave, &tot) + block_prefix;
#pragma unroll
  for (int k = 0; k < SC_ITEMS; k++) {
    const uint32_t i = base + k;
    if (i < n) offsets[i] = ex;
    // Element i owns instance slots [ex, ex + v): tell emit (binning.hip) which element owns the
    // first slot of each of its 2048-slot blocks, and which element owns the very last slot.
    if (v[k] > 0) {
      const uint32_t end = ex + v[k];
      for (uint32_t b = (ex + EMIT_PER_BLOCK - 1) / EMIT_PER_BLOCK; b * EMIT_PER_BLOCK < end; b++)
        if (b <= emit_win_cap) emit_win[b] = i;
      if (end == total) {
        const uint32_t nb = (end + EMIT_PER_BLOCK - 1) / EMIT_PER_BLOCK;
        if (nb <= emit_win_cap && nb * EMIT_PER_BLOCK >= end) emit_win[nb] = i;
      }
    }
    ex += v[k];
  }
}

// the reduce launch alone: sums of SC_CHUNK-element blocks of `counts` (zeros beyond *n_dev) -- what the fused
// coarse emit (binning.hip emit_coarse_fused_kernel) scans itself
void launch_offsets_reduce(hipStream_t s, uint32_t n, const uint32_t* n_dev, const uint32_t* counts,
                           uint32_t* block_sums, uint32_t nblocks) {
  if (n == 0) return;
  scan_reduce_kernel<<<nblocks, SC_THREADS, 0, s>>>(n, n_dev, counts, block_sums, nullptr, nullptr, nullptr,
                                                     nullptr, nullptr);
}

void launch_offsets_scan(hipStream_t s, uint32_t n, const uint32_t* n_dev,
                         uint32_t* tiles_sorted, const uint32_t* gather_gid,
                         const uint32_t* tiles_by_id, uint32_t* offsets, uint32_t* block_sums,
                         uint32_t nblocks, uint32_t* total_out, uint32_t* total_host,
                         uint32_t* emit_win, uint32_t emit_win_cap, const uint2* rects_by_id,
                         uint2* rect_sorted) {
  if (n == 0) return;
  scan_reduce_kernel<<<nblocks, SC_THREADS, 0, s>>>(n, n_dev, tiles_sorted, block_sums, gather_gid,
                                                     tiles_by_id, tiles_sorted, rects_by_id, rect_sorted);
  scan_down_kernel<<<nblocks, SC_THREADS, 0, s>>>(n, n_dev, tiles_sorted, block_sums, nblocks, offsets,
                                                   total_out, total_host, emit_win, emit_win_cap);
}

}  // namespace grpg
