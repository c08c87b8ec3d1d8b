// On-device stable LSD radix sort of (u32 key, u32 value) pairs + exclusive scan, for gfx950.
//
// Replaces cub::DeviceRadixSort::SortPairs / cub::DeviceScan::InclusiveSum as used by the
// reference (cuda_rasterizer/rasterizer_impl.cu:280, :306-311).  The reference sorts R 64-bit
// (tile|depth) keys over 32+ceil(log2 T) bits; we reach the IDENTICAL final order with far less
// HBM traffic by (1) sorting the P Gaussians once by their 32-bit depth key and (2) stably
// partitioning the R instances by their <=16-bit tile id -- both use this sort.
//
// Per pass (<= 8 bits):
//   radix_hist    : one workgroup per 4096-key chunk, LDS histogram  -> table[digit][chunk]
//   radix_digit_scan : one workgroup per digit, exclusive scan across chunks (in place) + totals
//   radix_scatter : wave64 match-any ranking (ballot per key bit, popcount prefix), per-wave
//                   digit counters in LDS, LDS reorder of the chunk, then coalesced run writes.
// Stability: a chunk is split wave-major (wave w owns keys [1024w, 1024w+1024)), each wave walks
// its keys in 16 rounds of 64 consecutive keys, ranks are (digit, wave, round, lane)-ordered.
#include <cstdlib>

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
radix_hist_kernel(const uint32_t* __restrict__ keys, const uint32_t n, const int shift,
                  const uint32_t mask, uint32_t* __restrict__ table, const uint32_t nchunks) {
  __shared__ uint32_t h[RS_MAX_RADIX];
  h[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t base = blockIdx.x * RS_CHUNK;
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
radix_digit_scan_kernel(uint32_t* __restrict__ table, const uint32_t nchunks,
                        uint32_t* __restrict__ totals) {
  __shared__ uint32_t s_wave[4];
  uint32_t* row = table + (size_t)blockIdx.x * nchunks;
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
template <int CBITS>
__global__ void __launch_bounds__(RS_THREADS)
radix_scatter_kernel(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                     uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                     const uint32_t n, const int shift, const int bits_rt,
                     const uint32_t* __restrict__ table, const uint32_t* __restrict__ totals,
                     const uint32_t nchunks, const uint32_t* __restrict__ gather_src,
                     uint32_t* __restrict__ gather_dst) {
  __shared__ uint32_t s_cnt[RS_MAX_RADIX * 4];  // [digit][wave]
  __shared__ uint32_t s_gbase[RS_MAX_RADIX];    // global position of local slot 0 of each digit
  __shared__ uint32_t s_wave[4];
  __shared__ uint32_t s_keys[RS_CHUNK];
  __shared__ uint32_t s_vals[RS_CHUNK];

  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bits = CBITS > 0 ? CBITS : bits_rt;
  const uint32_t mask = (1u << bits) - 1u;
  const uint32_t chunk = blockIdx.x;
  const uint32_t chunk_base = chunk * RS_CHUNK;
  const uint32_t chunk_n = min((uint32_t)RS_CHUNK, n - chunk_base);

#pragma unroll
  for (int k = 0; k < 4; k++) s_cnt[tid * 4 + k] = 0;
  __syncthreads();

  uint32_t key[RS_ITEMS], val[RS_ITEMS], rnk[RS_ITEMS];
  const uint64_t lt = (1ull << lane) - 1ull;
  // all of the wave's loads first (one memory round trip per workgroup instead of RS_ITEMS)
#pragma unroll
  for (int i = 0; i < RS_ITEMS; i++) {
    const uint32_t local = wave * (RS_ITEMS * 64) + i * 64 + lane;
    const uint32_t idx = chunk_base + local;
    const bool valid = local < chunk_n;
    key[i] = valid ? keys_in[idx] : 0xFFFFFFFFu;
    val[i] = valid ? (vals_in ? vals_in[idx] : idx) : 0u;
  }
#pragma unroll
  for (int i = 0; i < RS_ITEMS; i++) {
    const uint32_t local = wave * (RS_ITEMS * 64) + i * 64 + lane;
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
    const uint32_t prev = valid ? s_cnt[d * 4 + wave] : 0u;
    rnk[i] = prev + before;
    if (valid && before == 0) s_cnt[d * 4 + wave] = prev + (uint32_t)__popcll(peers);
  }
  __syncthreads();

  // Thread d owns digit d: turn per-(digit,wave) counts into local start slots, and compute the
  // global base of the digit for this chunk.
  {
    const uint32_t c0 = s_cnt[tid * 4 + 0], c1 = s_cnt[tid * 4 + 1], c2 = s_cnt[tid * 4 + 2],
                   c3 = s_cnt[tid * 4 + 3];
    const uint32_t dsum = c0 + c1 + c2 + c3;
    uint32_t tot;
    const uint32_t dstart = block_exclusive_scan_256(dsum, s_wave, &tot);
    const uint32_t gtot = tid <= mask ? totals[tid] : 0u;
    uint32_t tot2;
    const uint32_t gstart = block_exclusive_scan_256(gtot, s_wave, &tot2);
    s_cnt[tid * 4 + 0] = dstart;
    s_cnt[tid * 4 + 1] = dstart + c0;
    s_cnt[tid * 4 + 2] = dstart + c0 + c1;
    s_cnt[tid * 4 + 3] = dstart + c0 + c1 + c2;
    const uint32_t tb = tid <= mask ? table[(size_t)tid * nchunks + chunk] : 0u;
    s_gbase[tid] = gstart + tb - dstart;  // wraps mod 2^32; only used as base + slot
  }
  __syncthreads();

#pragma unroll
  for (int i = 0; i < RS_ITEMS; i++) {
    const uint32_t local = wave * (RS_ITEMS * 64) + i * 64 + lane;
    if (local < chunk_n) {
      const uint32_t d = (key[i] >> shift) & mask;
      const uint32_t slot = s_cnt[d * 4 + wave] + rnk[i];
      s_keys[slot] = key[i];
      s_vals[slot] = val[i];
    }
  }
  __syncthreads();

  // last pass of the depth sort: also emit the per-Gaussian tile count in sorted order, so the
  // offsets scan streams a contiguous array instead of gathering tiles[gid[i]].  The gather loads
  // are issued for all of the thread's items before the first store (one round trip, not 8).
  uint32_t gsrc[RS_ITEMS];
  if (gather_src) {
#pragma unroll
    for (int i = 0; i < RS_ITEMS; i++) {
      const uint32_t j = i * RS_THREADS + tid;
      gsrc[i] = j < chunk_n ? gather_src[s_vals[j]] : 0u;
    }
  }
#pragma unroll
  for (int i = 0; i < RS_ITEMS; i++) {
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
// Single-pass-per-digit variant ("onesweep"): the per-chunk histogram pass and the table scan are
// replaced by a decoupled look-back between workgroups, so a pass reads the keys ONCE.
//   radix_global_hist : one launch per sort, digit histograms of ALL passes (LDS atomics, one
//                       global atomic per non-empty (workgroup, pass, digit)).
//   radix_onesweep    : per pass.  A workgroup ranks its chunk exactly like radix_scatter, then
//                       thread d publishes the chunk's count of digit d and walks back over the
//                       predecessors' status words until it meets an inclusive prefix.
// Inter-workgroup protocol (MI355X: per-XCD L2s are not coherent, a CU's L1 is never refreshed):
// a status word is ONE naturally aligned 8-byte granule {count:32 | tag:30 | state:2} written with
// a single relaxed agent-scope atomic store (sc1, write-through) and polled with relaxed
// agent-scope atomic loads (sc1, L1-bypassing) -- the data travels inside the granule, so no
// fence and no separate flag are needed (cdna_hip_programming.md G16 "R2").  The tag is the pass
// number, so words of an earlier pass read as "empty" and the array is zeroed once per sort.
// Forward progress: a workgroup only ever waits for LOWER-numbered workgroups of the same launch,
// and workgroups are dispatched in index order, so whoever it waits for is resident or finished.
// ------------------------------------------------------------------------------------------
constexpr int GH_ITEMS = 32;                       // keys per thread in radix_global_hist
constexpr int GH_CHUNK = 256 * GH_ITEMS;
constexpr int OS_MAX_PASSES = 4;

__global__ void __launch_bounds__(256)
radix_global_hist_kernel(const uint32_t* __restrict__ keys, const uint32_t n, const int begin_bit,
                         const int passes, const int bits0, const int bits_rest,
                         uint32_t* __restrict__ totals /* [passes][256] */) {
  __shared__ uint32_t h[OS_MAX_PASSES][RS_MAX_RADIX];
#pragma unroll
  for (int p = 0; p < OS_MAX_PASSES; p++) h[p][threadIdx.x] = 0;
  __syncthreads();
  const uint32_t base = blockIdx.x * GH_CHUNK;
  for (int k = 0; k < GH_ITEMS; k++) {
    const uint32_t idx = base + k * 256 + threadIdx.x;
    if (idx < n) {
      uint32_t key = keys[idx] >> begin_bit;
      for (int p = 0; p < passes; p++) {
        const int b = p == 0 ? bits0 : bits_rest;
        atomicAdd(&h[p][key & ((1u << b) - 1u)], 1u);
        key >>= b;
      }
    }
  }
  __syncthreads();
  for (int p = 0; p < passes; p++) {
    const uint32_t c = h[p][threadIdx.x];
    if (c) atomicAdd(&totals[p * RS_MAX_RADIX + threadIdx.x], c);
  }
}

constexpr uint32_t OS_AGG = 1u, OS_PFX = 2u;

__global__ void __launch_bounds__(RS_THREADS)
radix_onesweep_kernel(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                      uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                      const uint32_t n, const int shift, const int bits,
                      const uint32_t* __restrict__ totals /* this pass: [256] */,
                      unsigned long long* __restrict__ status /* [nchunks][256] */,
                      const uint32_t tag, const uint32_t* __restrict__ gather_src,
                      uint32_t* __restrict__ gather_dst) {
  __shared__ uint32_t s_cnt[RS_MAX_RADIX * 4];  // [digit][wave]
  __shared__ uint32_t s_gbase[RS_MAX_RADIX];
  __shared__ uint32_t s_wave[4];
  __shared__ uint32_t s_keys[RS_CHUNK];
  __shared__ uint32_t s_vals[RS_CHUNK];

  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t mask = (1u << bits) - 1u;
  const uint32_t chunk = blockIdx.x;
  const uint32_t chunk_base = chunk * RS_CHUNK;
  const uint32_t chunk_n = min((uint32_t)RS_CHUNK, n - chunk_base);

#pragma unroll
  for (int k = 0; k < 4; k++) s_cnt[tid * 4 + k] = 0;
  __syncthreads();

  uint32_t key[RS_ITEMS], val[RS_ITEMS], rnk[RS_ITEMS];
  const uint64_t lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int i = 0; i < RS_ITEMS; i++) {
    const uint32_t local = wave * (RS_ITEMS * 64) + i * 64 + lane;
    const uint32_t idx = chunk_base + local;
    const bool valid = local < chunk_n;
    key[i] = valid ? keys_in[idx] : 0xFFFFFFFFu;
    val[i] = valid ? (vals_in ? vals_in[idx] : idx) : 0u;
    const uint32_t d = (key[i] >> shift) & mask;
    uint64_t peers = __ballot(valid);
    for (int b = 0; b < bits; b++) {
      const bool bit = (d >> b) & 1u;
      const uint64_t bal = __ballot(bit);
      peers &= bit ? bal : ~bal;
    }
    const uint32_t before = (uint32_t)__popcll(peers & lt);
    const uint32_t prev = valid ? s_cnt[d * 4 + wave] : 0u;
    rnk[i] = prev + before;
    if (valid && before == 0) s_cnt[d * 4 + wave] = prev + (uint32_t)__popcll(peers);
  }
  __syncthreads();

  {
    const uint32_t c0 = s_cnt[tid * 4 + 0], c1 = s_cnt[tid * 4 + 1], c2 = s_cnt[tid * 4 + 2],
                   c3 = s_cnt[tid * 4 + 3];
    const uint32_t dsum = c0 + c1 + c2 + c3;
    // ---- decoupled look-back: exclusive count of digit `tid` over all earlier chunks ----
    uint32_t excl = 0;
    if (tid <= mask) {
      unsigned long long* mine = status + (size_t)chunk * RS_MAX_RADIX + tid;
      const unsigned long long tagw = (unsigned long long)(tag << 2) << 32;
      if (chunk == 0) {
        __hip_atomic_store(mine, tagw | ((unsigned long long)OS_PFX << 32) | dsum, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      } else {
        __hip_atomic_store(mine, tagw | ((unsigned long long)OS_AGG << 32) | dsum, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long* p = mine - RS_MAX_RADIX;
        for (;;) {
          const unsigned long long w = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const uint32_t hi = (uint32_t)(w >> 32);
          if ((hi >> 2) == tag && (hi & 3u) != 0u) {
            excl += (uint32_t)w;
            if ((hi & 3u) == OS_PFX) break;
            p -= RS_MAX_RADIX;              // aggregate only: keep walking back (chunk 0 is a prefix)
          } else {
            __builtin_amdgcn_s_sleep(1);    // predecessor has not published yet
          }
        }
        __hip_atomic_store(mine, tagw | ((unsigned long long)OS_PFX << 32) | (excl + dsum),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    uint32_t tot;
    const uint32_t dstart = block_exclusive_scan_256(dsum, s_wave, &tot);
    const uint32_t gtot = tid <= mask ? totals[tid] : 0u;
    uint32_t tot2;
    const uint32_t gstart = block_exclusive_scan_256(gtot, s_wave, &tot2);
    s_cnt[tid * 4 + 0] = dstart;
    s_cnt[tid * 4 + 1] = dstart + c0;
    s_cnt[tid * 4 + 2] = dstart + c0 + c1;
    s_cnt[tid * 4 + 3] = dstart + c0 + c1 + c2;
    s_gbase[tid] = gstart + excl - dstart;  // wraps mod 2^32; only used as base + slot
  }
  __syncthreads();

#pragma unroll
  for (int i = 0; i < RS_ITEMS; i++) {
    const uint32_t local = wave * (RS_ITEMS * 64) + i * 64 + lane;
    if (local < chunk_n) {
      const uint32_t d = (key[i] >> shift) & mask;
      const uint32_t slot = s_cnt[d * 4 + wave] + rnk[i];
      s_keys[slot] = key[i];
      s_vals[slot] = val[i];
    }
  }
  __syncthreads();

  for (uint32_t j = tid; j < chunk_n; j += RS_THREADS) {
    const uint32_t k = s_keys[j];
    const uint32_t d = (k >> shift) & mask;
    const uint32_t g = s_gbase[d] + j;
    keys_out[g] = k;
    const uint32_t v = s_vals[j];
    vals_out[g] = v;
    if (gather_src) gather_dst[g] = gather_src[v];
  }
}

bool radix_sort_pairs(hipStream_t s, uint32_t n, uint32_t* key_a, uint32_t* val_a, uint32_t* key_b,
                      uint32_t* val_b, bool vals_iota, int begin_bit, int end_bit, uint32_t* table,
                      uint32_t* totals, uint32_t nchunks, const uint32_t* gather_src,
                      uint32_t* gather_dst) {
  bool in_b = false;
  if (n == 0) return in_b;
  const int nbits = end_bit - begin_bit;
  if (nbits <= 0) return in_b;
  const int passes = (nbits + RS_MAX_BITS - 1) / RS_MAX_BITS;
  int shift = begin_bit;
  // Default: the three-kernel reduce-then-scan pass.  The look-back variant is correct (parity
  // tests pass with GRPG_SORT=onesweep) but measured SLOWER on MI355X (depth sort 0.164 vs 0.129 ms,
  // tile partition 0.204 vs 0.140 ms): a status word crosses XCDs at ~1 us per hop, and with
  // ~1800 resident workgroups publishing "aggregate" at once the walk-back is long.
  static const bool onesweep = [] { const char* e = getenv("GRPG_SORT"); return e && e[0] == 'o'; }();
  if (onesweep && passes <= OS_MAX_PASSES) {
    // onesweep: `table` holds the status words (8 B x 256 x nchunks), `totals` the per-pass digit
    // histograms ([4][256]); both are zeroed once per sort.
    int bits_p[OS_MAX_PASSES];
    {
      int sh = begin_bit;
      for (int p = 0; p < passes; p++) { bits_p[p] = (end_bit - sh + (passes - p) - 1) / (passes - p); sh += bits_p[p]; }
    }
    // the histogram kernel assumes pass 0 may be wider than the (equal) remaining passes
    bool uniform_rest = true;
    for (int p = 2; p < passes; p++) uniform_rest = uniform_rest && bits_p[p] == bits_p[1];
    if (uniform_rest) {
      (void)hipMemsetAsync(totals, 0, OS_MAX_PASSES * RS_MAX_RADIX * sizeof(uint32_t), s);
      (void)hipMemsetAsync(table, 0, (size_t)nchunks * RS_MAX_RADIX * 8, s);
      radix_global_hist_kernel<<<(n + GH_CHUNK - 1) / GH_CHUNK, 256, 0, s>>>(
          key_a, n, begin_bit, passes, bits_p[0], passes > 1 ? bits_p[1] : bits_p[0], totals);
      for (int p = 0; p < passes; p++) {
        uint32_t* kin = in_b ? key_b : key_a;
        uint32_t* vin = in_b ? val_b : val_a;
        uint32_t* kout = in_b ? key_a : key_b;
        uint32_t* vout = in_b ? val_a : val_b;
        radix_onesweep_kernel<<<nchunks, RS_THREADS, 0, s>>>(
            kin, (p == 0 && vals_iota) ? nullptr : vin, kout, vout, n, shift, bits_p[p],
            totals + p * RS_MAX_RADIX, (unsigned long long*)table, (uint32_t)(p + 1),
            p == passes - 1 ? gather_src : nullptr, gather_dst);
        in_b = !in_b;
        shift += bits_p[p];
      }
      return in_b;
    }
  }
  for (int p = 0; p < passes; p++) {
    // spread the bits evenly over the passes (e.g. 14 bits -> 7 + 7)
    const int bits = (end_bit - shift + (passes - p) - 1) / (passes - p);
    const uint32_t mask = (1u << bits) - 1u;
    uint32_t* kin = in_b ? key_b : key_a;
    uint32_t* vin = in_b ? val_b : val_a;
    uint32_t* kout = in_b ? key_a : key_b;
    uint32_t* vout = in_b ? val_a : val_b;
    radix_hist_kernel<<<nchunks, RS_THREADS, 0, s>>>(kin, n, shift, mask, table, nchunks);
    radix_digit_scan_kernel<<<1u << bits, 256, 0, s>>>(table, nchunks, totals);
#define RS_SCATTER(CB)                                                                        \
  radix_scatter_kernel<CB><<<nchunks, RS_THREADS, 0, s>>>(                                    \
      kin, (p == 0 && vals_iota) ? nullptr : vin, kout, vout, n, shift, bits, table, totals,  \
      nchunks, p == passes - 1 ? gather_src : nullptr, gather_dst)
    if (bits == 8) RS_SCATTER(8);        // depth sort
    else if (bits == 7) RS_SCATTER(7);   // tile partition of a 1920x1280 frame (14 bits)
    else RS_SCATTER(0);
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

// ------------------------------------------------------------------------------------------
// Exclusive scan of the per-Gaussian instance counts in depth-sorted order (written contiguously
// by the last depth-sort pass) -> offsets[i]; replaces cub::DeviceScan::InclusiveSum (rasterizer_impl.cu:280).
// Two launches: per-workgroup reduce, per-workgroup downsweep (which recomputes the spine).
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(SC_THREADS)
scan_reduce_kernel(const uint32_t n, const uint32_t* __restrict__ tiles,
                   uint32_t* __restrict__ block_sums) {
  __shared__ uint32_t s_wave[4];
  const uint32_t base = blockIdx.x * SC_CHUNK;
  uint32_t v[SC_ITEMS], s = 0;   // all loads in flight together
#pragma unroll
  for (int k = 0; k < SC_ITEMS; k++) {
    const uint32_t i = base + k * SC_THREADS + threadIdx.x;
    v[k] = i < n ? tiles[i] : 0u;
  }
#pragma unroll
  for (int k = 0; k < SC_ITEMS; k++) s += v[k];
  uint32_t tot;
  block_exclusive_scan_256(s, s_wave, &tot);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(SC_THREADS)
scan_down_kernel(const uint32_t n, const uint32_t* __restrict__ tiles,
                 const uint32_t* __restrict__ block_sums, const uint32_t nblocks,
                 uint32_t* __restrict__ offsets, uint32_t* __restrict__ total_out,
                 uint32_t* __restrict__ emit_win, const uint32_t emit_win_cap) {
  __shared__ uint32_t s_wave[4];
  __shared__ uint32_t s_spine[8];
  // the spine (prefix of the workgroup sums before this workgroup, and the grand total) is
  // recomputed by every workgroup from the <= a few thousand L2-resident sums: cheaper than a
  // third launch with a single workgroup
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
  if (blockIdx.x == 0 && threadIdx.x == 0) *total_out = total;
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
    // first slot of each of its 1024-slot blocks, and which element owns the very last slot.
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

void launch_offsets_scan(hipStream_t s, uint32_t n, const uint32_t* tiles_sorted,
                         uint32_t* offsets, uint32_t* block_sums, uint32_t nblocks,
                         uint32_t* total_out, uint32_t* emit_win, uint32_t emit_win_cap) {
  if (n == 0) return;
  scan_reduce_kernel<<<nblocks, SC_THREADS, 0, s>>>(n, tiles_sorted, block_sums);
  scan_down_kernel<<<nblocks, SC_THREADS, 0, s>>>(n, tiles_sorted, block_sums, nblocks, offsets,
                                                   total_out, emit_win, emit_win_cap);
}

}  // namespace grpg
