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
#include "common.h"

namespace grpg {

// One thread per depth-sorted Gaussian.  Small footprints are written by their own lane; a
// footprint of more than EMIT_SERIAL_MAX tiles (a near-camera splat can cover all 9600 tiles)
// is written cooperatively by the whole wave with coalesced stores, so no lane ever runs the
// reference's thousands-of-iterations serial loop (rasterizer_impl.cu:98-108).
constexpr uint32_t EMIT_SERIAL_MAX = 24;

__global__ void __launch_bounds__(256)
emit_kernel(const uint32_t P, const uint32_t* __restrict__ sorted_gid,
            const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ tiles,
            const float4* __restrict__ rec, const int gx, const int gy,
            uint32_t* __restrict__ tile_keys, uint32_t* __restrict__ vals) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63;
  uint32_t g = 0, cnt = 0, off = 0;
  int minx = 0, miny = 0, w = 1;
  if (i < P) {
    g = sorted_gid[i];
    cnt = tiles[g];
    if (cnt) {
      off = offsets[i];
      const float4 r0 = rec[3 * (size_t)g];
      const int radius = __float_as_int(rec[3 * (size_t)g + 2].w);
      int maxx, maxy;
      get_rect(r0.x, r0.y, radius, gx, gy, minx, miny, maxx, maxy);
      w = maxx - minx;
    }
  }
  if (cnt > 0 && cnt <= EMIT_SERIAL_MAX) {
    int tx = minx, ty = miny;
    for (uint32_t k = 0; k < cnt; k++) {
      tile_keys[off + k] = (uint32_t)(ty * gx + tx);
      vals[off + k] = g;
      if (++tx == minx + w) { tx = minx; ty++; }
    }
  }
  uint64_t big = __ballot(cnt > EMIT_SERIAL_MAX);
  while (big) {
    const int src = __ffsll((unsigned long long)big) - 1;
    big &= big - 1;
    const uint32_t c = __shfl(cnt, src, 64), o = __shfl(off, src, 64), gg = __shfl(g, src, 64);
    const int mx = __shfl(minx, src, 64), my = __shfl(miny, src, 64), ww = __shfl(w, src, 64);
    for (uint32_t k = lane; k < c; k += 64) {
      const uint32_t row = k / (uint32_t)ww, col = k - row * (uint32_t)ww;
      tile_keys[o + k] = (uint32_t)((my + (int)row) * gx + mx + (int)col);
      vals[o + k] = gg;
    }
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

void launch_emit(hipStream_t s, uint32_t P, const uint32_t* sorted_gid, const uint32_t* offsets,
                 const uint32_t* tiles, const float4* rec, int gx, int gy, uint32_t* tile_keys,
                 uint32_t* vals) {
  if (P == 0) return;
  emit_kernel<<<(P + 255) / 256, 256, 0, s>>>(P, sorted_gid, offsets, tiles, rec, gx, gy,
                                              tile_keys, vals);
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
  const uint32_t g = point_list[i];
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

}  // namespace grpg
