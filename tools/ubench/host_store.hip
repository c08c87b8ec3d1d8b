// Can the render's epilogue store the rgb8 frame STRAIGHT into pinned host memory (the simulator reads its frame on
// the host: simulator.py:313-328), instead of HBM + a 7.4 MB copy behind the launch?  The answer depends on how the
// link takes a kernel's small scattered writes, which nothing in the guides states: this measures it.
//
// A 1920x1280 rgb8 frame [H,W,3] written by 16x16-pixel tiles (one 256-thread workgroup per tile, a 48-byte row
// segment per tile row), each pattern into device memory and into hipHostMalloc memory:
//   bytes3     three 1-byte stores per pixel (what frame_finish does today)
//   dwords     lanes 0-2 of every four pack the quad's 12 bytes into three dwords (48 contiguous bytes per row)
//   rows64     the same bytes, but a workgroup owns a 64-pixel x 4-row strip: 192 contiguous bytes = whole 64-byte lines
//   memcpy     hipMemcpyAsync device -> host of the finished frame (today's path)
//   dwords_wt  the dwords pattern with write-through stores (system scope: sc0 sc1), so that the lines leave the L2
//              while the launch runs instead of at its end-of-kernel write-back
// spread = K: every workgroup idles K / 4.7 us first, to mimic a ~K us render whose tiles finish over time.
//
// build: hipcc --offload-arch=gfx950 -O3 -o host_store host_store.hip ; run: ./host_store
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int W = 1920, H = 1280, TX = W / 16, TY = H / 16;

__device__ __forceinline__ void idle(const int spread_us, const int slot, const int slots) {
  if (spread_us <= 0) return;
  // 100 MHz constant clock: 100 ticks per us; a workgroup's wake-up time is its slot's share of the spread
  const uint64_t t0 = wall_clock64();
  // every workgroup "works" for the same time: with ~2048 of the 9600 resident at once (8 x 256 threads per CU) a
  // wait of spread / 4.7 per workgroup makes the launch last ~spread and spreads the stores evenly over it
  (void)slot; (void)slots;
  const uint64_t wait = (uint64_t)spread_us * 100ull * 10ull / 47ull;
  while (wall_clock64() - t0 < wait) __builtin_amdgcn_s_sleep(8);
}

__global__ void __launch_bounds__(256) bytes3(unsigned char* __restrict__ dst, int spread_us) {
  idle(spread_us, blockIdx.x, gridDim.x);
  const int tx = blockIdx.x % TX, ty = blockIdx.x / TX;
  const int px = tx * 16 + (threadIdx.x & 15), py = ty * 16 + (threadIdx.x >> 4);
  unsigned char* d = dst + 3 * ((size_t)py * W + px);
  d[0] = (unsigned char)px; d[1] = (unsigned char)py; d[2] = (unsigned char)(px + py);
}

__global__ void __launch_bounds__(256) dwords(unsigned char* __restrict__ dst, int spread_us) {
  idle(spread_us, blockIdx.x, gridDim.x);
  const int tx = blockIdx.x % TX, ty = blockIdx.x / TX;
  const int px = tx * 16 + (threadIdx.x & 15), py = ty * 16 + (threadIdx.x >> 4);
  const uint32_t pk = (uint32_t)(px & 255) | ((uint32_t)(py & 255) << 8) | ((uint32_t)((px + py) & 255) << 16);
  const uint32_t nx = __shfl_down(pk, 1, 4);     // the next pixel of the quad
  const int j = threadIdx.x & 3;
  const uint32_t v = (pk >> (8 * j)) | (nx << (24 - 8 * j));
  if (j < 3) *(uint32_t*)(dst + 3 * ((size_t)py * W + (px & ~3)) + 4 * j) = v;
}

__global__ void __launch_bounds__(256) dwords_wt(unsigned char* __restrict__ dst, int spread_us) {
  idle(spread_us, blockIdx.x, gridDim.x);
  const int tx = blockIdx.x % TX, ty = blockIdx.x / TX;
  const int px = tx * 16 + (threadIdx.x & 15), py = ty * 16 + (threadIdx.x >> 4);
  const uint32_t pk = (uint32_t)(px & 255) | ((uint32_t)(py & 255) << 8) | ((uint32_t)((px + py) & 255) << 16);
  const uint32_t nx = __shfl_down(pk, 1, 4);
  const int j = threadIdx.x & 3;
  const uint32_t v = (pk >> (8 * j)) | (nx << (24 - 8 * j));
  if (j < 3) __hip_atomic_store((uint32_t*)(dst + 3 * ((size_t)py * W + (px & ~3)) + 4 * j), v, __ATOMIC_RELAXED,
                                __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void __launch_bounds__(256) bytes3_wt(unsigned char* __restrict__ dst, int spread_us) {
  idle(spread_us, blockIdx.x, gridDim.x);
  const int tx = blockIdx.x % TX, ty = blockIdx.x / TX;
  const int px = tx * 16 + (threadIdx.x & 15), py = ty * 16 + (threadIdx.x >> 4);
  unsigned char* d = dst + 3 * ((size_t)py * W + px);
  __hip_atomic_store(d, (unsigned char)px, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(d + 1, (unsigned char)py, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(d + 2, (unsigned char)(px + py), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void __launch_bounds__(256) rows64_wt(unsigned char* __restrict__ dst, int spread_us) {
  idle(spread_us, blockIdx.x, gridDim.x);
  const int sx = blockIdx.x % (W / 64), sy = blockIdx.x / (W / 64);
  const int px = sx * 64 + (threadIdx.x & 63), py = sy * 4 + (threadIdx.x >> 6);
  const uint32_t pk = (uint32_t)(px & 255) | ((uint32_t)(py & 255) << 8) | ((uint32_t)((px + py) & 255) << 16);
  const uint32_t nx = __shfl_down(pk, 1, 4);
  const int j = threadIdx.x & 3;
  const uint32_t v = (pk >> (8 * j)) | (nx << (24 - 8 * j));
  if (j < 3) __hip_atomic_store((uint32_t*)(dst + 3 * ((size_t)py * W + (px & ~3)) + 4 * j), v, __ATOMIC_RELAXED,
                                __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void __launch_bounds__(256) dwords_nt(unsigned char* __restrict__ dst, int spread_us) {
  idle(spread_us, blockIdx.x, gridDim.x);
  const int tx = blockIdx.x % TX, ty = blockIdx.x / TX;
  const int px = tx * 16 + (threadIdx.x & 15), py = ty * 16 + (threadIdx.x >> 4);
  const uint32_t pk = (uint32_t)(px & 255) | ((uint32_t)(py & 255) << 8) | ((uint32_t)((px + py) & 255) << 16);
  const uint32_t nx = __shfl_down(pk, 1, 4);
  const int j = threadIdx.x & 3;
  const uint32_t v = (pk >> (8 * j)) | (nx << (24 - 8 * j));
  if (j < 3) __builtin_nontemporal_store(v, (uint32_t*)(dst + 3 * ((size_t)py * W + (px & ~3)) + 4 * j));
}

// strip: 64 pixels x 4 rows per 256-thread workgroup; thread t: row t >> 6, pixel t & 63 -> 48 dwords per row
__global__ void __launch_bounds__(256) rows64(unsigned char* __restrict__ dst, int spread_us) {
  idle(spread_us, blockIdx.x, gridDim.x);
  const int sx = blockIdx.x % (W / 64), sy = blockIdx.x / (W / 64);
  const int px = sx * 64 + (threadIdx.x & 63), py = sy * 4 + (threadIdx.x >> 6);
  const uint32_t pk = (uint32_t)(px & 255) | ((uint32_t)(py & 255) << 8) | ((uint32_t)((px + py) & 255) << 16);
  const uint32_t nx = __shfl_down(pk, 1, 4);
  const int j = threadIdx.x & 3;
  const uint32_t v = (pk >> (8 * j)) | (nx << (24 - 8 * j));
  if (j < 3) *(uint32_t*)(dst + 3 * ((size_t)py * W + (px & ~3)) + 4 * j) = v;
}

template <class F>
static float time_ms(F launch, int reps) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  launch(); CHECK(hipDeviceSynchronize());
  float best = 1e9f, sum = 0.f;
  for (int r = 0; r < reps; r++) {
    CHECK(hipEventRecord(e0, 0));
    launch();
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best; sum += ms;
  }
  (void)best;
  return sum / reps;
}

int main() {
  const size_t bytes = (size_t)W * H * 3;
  unsigned char *dev, *host, *ref;
  CHECK(hipMalloc(&dev, bytes));
  CHECK(hipHostMalloc(&host, bytes, hipHostMallocDefault));
  ref = (unsigned char*)malloc(bytes);
  const int tiles = TX * TY, strips = (W / 64) * (H / 4);
  // reference bytes
  hipLaunchKernelGGL(bytes3, dim3(tiles), dim3(256), 0, 0, dev, 0);
  CHECK(hipMemcpy(ref, dev, bytes, hipMemcpyDeviceToHost));
  printf("{\"frame_bytes\": %zu", bytes);
  const int reps = 30;
  for (int spread : {0, 200, 400}) {
    for (int target = 0; target < 2; target++) {
      unsigned char* dst = target ? host : dev;
      const char* tn = target ? "host" : "device";
      float a = time_ms([&] { hipLaunchKernelGGL(bytes3, dim3(tiles), dim3(256), 0, 0, dst, spread); }, target && spread == 0 ? 5 : reps);
      CHECK(hipMemset(dev, 0, bytes)); memset(host, 0, bytes);
      float b = time_ms([&] { hipLaunchKernelGGL(dwords, dim3(tiles), dim3(256), 0, 0, dst, spread); }, reps);
      CHECK(hipDeviceSynchronize());
      if (target) { if (memcmp(host, ref, bytes) != 0) { fprintf(stderr, "dwords(host) bytes differ\n"); return 1; } }
      float c = time_ms([&] { hipLaunchKernelGGL(rows64, dim3(strips), dim3(256), 0, 0, dst, spread); }, reps);
      CHECK(hipDeviceSynchronize());
      if (target) { if (memcmp(host, ref, bytes) != 0) { fprintf(stderr, "rows64(host) bytes differ\n"); return 1; } }
      memset(host, 0, bytes);
      float d = time_ms([&] { hipLaunchKernelGGL(dwords_wt, dim3(tiles), dim3(256), 0, 0, dst, spread); }, reps);
      CHECK(hipDeviceSynchronize());
      if (target) { if (memcmp(host, ref, bytes) != 0) { fprintf(stderr, "dwords_wt(host) bytes differ\n"); return 1; } }
      float e = time_ms([&] { hipLaunchKernelGGL(dwords_nt, dim3(tiles), dim3(256), 0, 0, dst, spread); }, reps);
      CHECK(hipDeviceSynchronize());
      float f = time_ms([&] { hipLaunchKernelGGL(bytes3_wt, dim3(tiles), dim3(256), 0, 0, dst, spread); }, target && spread == 0 ? 5 : reps);
      float g = time_ms([&] { hipLaunchKernelGGL(rows64_wt, dim3(strips), dim3(256), 0, 0, dst, spread); }, reps);
      printf(", \"%s_spread%d\": {\"bytes3_ms\": %.4f, \"dwords_ms\": %.4f, \"rows64_ms\": %.4f, \"dwords_wt_ms\": %.4f, \"dwords_nt_ms\": %.4f, \"bytes3_wt_ms\": %.4f, \"rows64_wt_ms\": %.4f}", tn, spread, a, b, c, d, e, f, g);
    }
  }
  float m = time_ms([&] { CHECK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, 0)); }, reps);
  float km = time_ms([&] { hipLaunchKernelGGL(dwords, dim3(tiles), dim3(256), 0, 0, dev, 200);
                            CHECK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, 0)); }, reps);
  printf(", \"memcpy_d2h_ms\": %.4f, \"dwords_device_spread200_then_memcpy_ms\": %.4f}\n", m, km);
  return 0;
}
