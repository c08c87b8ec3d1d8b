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

// One thread per OUTPUT instance (not per Gaussian): slot s of the instance list is mapped back to
// its (depth-sorted) Gaussian by a binary search over the exclusive offsets, restricted to the
// index window [lo, hi] that the workgroup's 1024 consecutive slots can touch (two searches per
// workgroup).  Work per thread is therefore independent of the footprint distribution -- in the
// reference one thread loops over every tile of its Gaussian (rasterizer_impl.cu:98-108), which
// is thousands of serial iterations for a near-camera splat, and after the depth sort the largest
// footprints sit next to each other -- and the stores are perfectly coalesced.
constexpr int EMIT_PER_BLOCK = 1024;

__device__ __forceinline__ uint32_t last_leq(const uint32_t* __restrict__ a, uint32_t lo,
                                             uint32_t hi, const uint32_t v) {
  // largest i in [lo, hi] with a[i] <= v; requires a[lo] <= v
  while (lo < hi) {
    const uint32_t mid = lo + (hi - lo + 1) / 2;
    if (a[mid] <= v) lo = mid; else hi = mid - 1;
  }
  return lo;
}

__global__ void __launch_bounds__(256)
emit_kernel(const uint32_t P, const uint32_t R, const uint32_t* __restrict__ sorted_gid,
            const uint32_t* __restrict__ offsets, const float4* __restrict__ rec, const int gx,
            const int gy, uint32_t* __restrict__ tile_keys, uint32_t* __restrict__ vals) {
  __shared__ uint32_t s_win[2];
  __shared__ uint32_t s_off[EMIT_PER_BLOCK + 64];
  const uint32_t o0 = blockIdx.x * EMIT_PER_BLOCK;
  const uint32_t o1 = min(R, o0 + EMIT_PER_BLOCK);
  if (threadIdx.x < 2)
    s_win[threadIdx.x] = last_leq(offsets, 0, P - 1, threadIdx.x == 0 ? o0 : o1 - 1);
  __syncthreads();
  const uint32_t lo = s_win[0], hi = s_win[1];
  // the window normally holds <= 1025 Gaussians (every visible Gaussian owns >= 1 slot; culled
  // ones sort to the very end): stage its offsets in LDS so the per-slot search never leaves the CU
  const uint32_t nwin = hi - lo + 1;
  const bool in_lds = nwin <= (uint32_t)(EMIT_PER_BLOCK + 64);
  if (in_lds)
    for (uint32_t j = threadIdx.x; j < nwin; j += 256) s_off[j] = offsets[lo + j];
  __syncthreads();
#pragma unroll
  for (int r = 0; r < EMIT_PER_BLOCK / 256; r++) {
    const uint32_t s = o0 + r * 256 + threadIdx.x;
    if (s >= o1) break;
    // a Gaussian with zero instances shares its offset with its successor, so the LAST index
    // with offsets[i] <= s is always the owner of slot s
    const uint32_t i = in_lds ? lo + last_leq(s_off, 0, nwin - 1, s) : last_leq(offsets, lo, hi, s);
    const uint32_t g = sorted_gid[i];
    const uint32_t k = s - (in_lds ? s_off[i - lo] : offsets[i]);
    const float4 r0 = rec[3 * (size_t)g];
    const int radius = __float_as_int(rec[3 * (size_t)g + 2].w);
    int minx, miny, maxx, maxy;
    get_rect(r0.x, r0.y, radius, gx, gy, minx, miny, maxx, maxy);
    const uint32_t w = (uint32_t)(maxx - minx);
    const uint32_t row = k / w, col = k - row * w;
    const int tx = minx + (int)col, ty = miny + (int)row;
    tile_keys[s] = (uint32_t)(ty * gx + tx);
    // Sub-tile mask: can this splat reach the w-th 16x4 quarter of the tile at all?  Evaluated
    // once here (the record is in registers anyway) and carried through the sort in the spare
    // top bits of the value, so render knows before loading a record whether it is needed.
    const float4 r1 = rec[3 * (size_t)g + 1];
    uint32_t bits = 0;
#pragma unroll
    for (int q = 0; q < 4; q++)
      if (!splat_misses_rect(r0.x, r0.y, r1.x, r1.y, r1.z, r0.w, (float)(tx * TILE),
                             (float)(tx * TILE + 15), (float)(ty * TILE + 4 * q),
                             (float)(ty * TILE + 4 * q + 3)))
        bits |= 1u << q;
    vals[s] = g | (bits << SUBTILE_SHIFT);
  }
}

// identifyTileRanges, rasterizer_impl.cu:116-138 (ranges pre-zeroed by the caller, :313).
__global__ void __launch_bounds__(256)
tile_ranges_kernel(const uint32_t R, const uint32_t* __restrict__ tile_keys,
                   uint2* __restrict__ ranges) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= R) return;
  const uint32_t t = tile_keys[i];
  if (i == 0 || tile_keys[i - 1] != t) ranges[t].x = i;
  if (i == R - 1 || tile_keys[i + 1] != t) ranges[t].y = i + 1;
}

void launch_emit(hipStream_t s, uint32_t P, uint32_t R, const uint32_t* sorted_gid,
                 const uint32_t* offsets, const float4* rec, int gx, int gy, uint32_t* tile_keys,
                 uint32_t* vals) {
  if (P == 0 || R == 0) return;
  emit_kernel<<<(R + EMIT_PER_BLOCK - 1) / EMIT_PER_BLOCK, 256, 0, s>>>(
      P, R, sorted_gid, offsets, rec, gx, gy, tile_keys, vals);
}

void launch_tile_ranges(hipStream_t s, uint32_t R, const uint32_t* tile_keys, uint2* ranges,
                        uint32_t T) {
  (void)hipMemsetAsync(ranges, 0, (size_t)T * sizeof(uint2), s);
  if (R == 0) return;
  tile_ranges_kernel<<<(R + 255) / 256, 256, 0, s>>>(R, tile_keys, ranges);
}

// ---------------------------------- debug / parity decoder --------------------------------
__global__ void __launch_bounds__(256)
debug_geom_kernel(const int P, const float4* __restrict__ rec, const uint32_t* __restrict__ tiles,
                  float* means2D, float* depths, float* conic_opacity, float* rgb,
                  uint32_t* tiles_out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  const bool vis = tiles[i] > 0;
  float4 a = make_float4(0, 0, 0, 0), b = a, c = a;
  if (vis) { a = rec[3 * (size_t)i]; b = rec[3 * (size_t)i + 1]; c = rec[3 * (size_t)i + 2]; }
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
                  const uint32_t* __restrict__ point_list, const float4* __restrict__ rec,
                  uint64_t* keys_sorted, uint32_t* point_list_out) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= R) return;
  const uint32_t g = point_list[i] & ID_MASK;
  if (keys_sorted)
    keys_sorted[i] = ((uint64_t)tile_keys[i] << 32) | (uint64_t)__float_as_uint(rec[3 * (size_t)g].z);
  if (point_list_out) point_list_out[i] = g;
}

void launch_debug_export(hipStream_t s, int P, uint32_t R, int W, int H, int gx, int gy,
                         const float4* rec, const uint32_t* tiles, const uint32_t* tile_keys,
                         const uint32_t* point_list, const uint2* ranges,
                         const uint32_t* n_contrib_in, uint64_t* keys_sorted,
                         uint32_t* point_list_out, uint32_t* ranges_out, uint32_t* n_contrib_out,
                         float* means2D, float* depths, float* conic_opacity, float* rgb,
                         uint32_t* tiles_out) {
  if (P > 0 && (means2D || depths || conic_opacity || rgb || tiles_out))
    debug_geom_kernel<<<(P + 255) / 256, 256, 0, s>>>(P, rec, tiles, means2D, depths,
                                                      conic_opacity, rgb, tiles_out);
  if (R > 0 && (keys_sorted || point_list_out))
    debug_keys_kernel<<<(R + 255) / 256, 256, 0, s>>>(R, tile_keys, point_list, rec, keys_sorted,
                                                      point_list_out);
  if (ranges_out)
    (void)hipMemcpyAsync(ranges_out, ranges, (size_t)gx * gy * sizeof(uint2), hipMemcpyDeviceToDevice, s);
  if (n_contrib_out)
    (void)hipMemcpyAsync(n_contrib_out, n_contrib_in, (size_t)W * H * 4, hipMemcpyDeviceToDevice, s);
}

// ------------------------------------------------------------------------------------------
// Blob headers are written by one tiny launch instead of three pageable H2D copies (each of
// which costs a staging copy + a ~5 us copy kernel on the stream).
// ------------------------------------------------------------------------------------------
__global__ void write_headers_kernel(BlobHeader* geom, BlobHeader* bin, BlobHeader* img,
                                     const uint32_t P, const uint32_t R, const uint32_t W,
                                     const uint32_t H, const uint32_t S) {
  const uint32_t t = threadIdx.x;
  if (t < 3) {
    BlobHeader* h = t == 0 ? geom : (t == 1 ? bin : img);
    if (h) {
      h->magic = t == 0 ? GEOM_MAGIC : (t == 1 ? BIN_MAGIC : IMG_MAGIC);
      h->P = P; h->R = R; h->W = W; h->H = H; h->S = S;
    }
  }
}

void launch_write_headers(hipStream_t s, char* geom, char* bin, char* img, uint32_t P, uint32_t R,
                          uint32_t W, uint32_t H, uint32_t S) {
  write_headers_kernel<<<1, 64, 0, s>>>((BlobHeader*)geom, (BlobHeader*)bin, (BlobHeader*)img, P,
                                        R, W, H, S);
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

}  // namespace grpg
