// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for THIS build's access patterns.
// MI355X_MICROARCH.md calibrates FETCH_SIZE only for wide coalesced streaming reads (it reports 1/2 of the bytes);
// the binning fill (hier_binning.hip hb_fill_kernel) and the render (render_fwd.hip RecView::load) GATHER 64-byte
// records by index, and tools/summarize_profile.py doubled their FETCH_SIZE like a stream's.  Every kernel here
// moves a KNOWN number of bytes with one of those patterns; run under
//     rocprofv3 --kernel-trace --pmc FETCH_SIZE   (and, separately, --pmc WRITE_SIZE)
// and divide: tools/calibrate_fetch.py prints  counter bytes / true bytes  per pattern.
//
//   stream16      every lane one 16-byte load, consecutive lanes consecutive addresses           (the guide's case)
//   gather64_32   every lane 2 x 16 bytes (float4 0, 1) of a RANDOM 64-byte record               (hb_fill: geo0, geo1)
//   gather64_48   every lane 3 x 16 bytes (float4 0, 1, 2) of a RANDOM 64-byte record            (render: RecView::load)
//   gather4       every lane one 4-byte load at a random word                                    (point list by index)
//   write4_stream every lane one 4-byte store, consecutive                                       (point list store)
//   write16_stream every lane one 16-byte store, consecutive
//   write64_rec   every lane one 64-byte record as 4 x 16-byte stores, consecutive records       (preprocess records)
// The working set (N records = 256 MiB + an index array) exceeds the 256 MiB Infinity Cache together with the
// indices; every record is touched exactly once (the index array is a permutation), so true bytes = N x touched.
//
// build: hipcc --offload-arch=gfx950 -O3 -o gather_fetch gather_fetch.hip ; run: ./gather_fetch [log2 N = 22]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) stream16(const float4* __restrict__ src, size_t n16, float* __restrict__ sink) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n16) return;
  const float4 v = src[i];
  if (v.x + v.y + v.z + v.w == 12345.678f) sink[0] = v.x;
}
template <int NF4>
__global__ void __launch_bounds__(256) gather64(const float4* __restrict__ rec, const uint32_t* __restrict__ idx, size_t n,
                                                float* __restrict__ sink) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const size_t id = idx[i];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NF4; k++) { const float4 v = rec[4 * id + k]; s += v.x + v.y + v.z + v.w; }
  if (s == 12345.678f) sink[0] = s;
}
__global__ void __launch_bounds__(256) gather4(const float* __restrict__ words, const uint32_t* __restrict__ idx, size_t n,
                                               float* __restrict__ sink) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float v = words[(size_t)idx[i] * 16];   // one word of a random 64-byte line
  if (v == 12345.678f) sink[0] = v;
}
__global__ void __launch_bounds__(256) write4_stream(float* __restrict__ dst, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = (float)i;
}
__global__ void __launch_bounds__(256) write16_stream(float4* __restrict__ dst, size_t n16) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n16) dst[i] = make_float4((float)i, 1.f, 2.f, 3.f);
}
__global__ void __launch_bounds__(256) write64_rec(float4* __restrict__ dst, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
#pragma unroll
  for (int k = 0; k < 4; k++) dst[4 * i + k] = make_float4((float)i, (float)k, 2.f, 3.f);
}

int main(int argc, char** argv) {
  const int lg = argc > 1 ? atoi(argv[1]) : 22;
  const size_t N = (size_t)1 << lg;              // records of 64 bytes
  float4* rec; uint32_t* idx; float* sink; float4* out;
  CHECK(hipMalloc(&rec, N * 64));
  CHECK(hipMalloc(&out, N * 64));
  CHECK(hipMalloc(&idx, N * 4));
  CHECK(hipMalloc(&sink, 256));
  CHECK(hipMemset(rec, 0, N * 64));
  // a permutation of [0, N): multiplicative hash with an odd multiplier (bijective mod 2^lg), so neighbouring lanes
  // hit unrelated lines -- like sorted-by-depth ids gathered in tile order
  std::vector<uint32_t> h(N);
  for (size_t i = 0; i < N; i++) h[i] = (uint32_t)((i * 2654435761ull + 12345ull) & (N - 1));
  CHECK(hipMemcpy(idx, h.data(), N * 4, hipMemcpyHostToDevice));
  const unsigned gN = (unsigned)((N + 255) / 256), g4N = (unsigned)((4 * N + 255) / 256);
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(stream16, dim3(g4N), dim3(256), 0, 0, rec, 4 * N, sink);
    hipLaunchKernelGGL(gather64<2>, dim3(gN), dim3(256), 0, 0, rec, idx, N, sink);
    hipLaunchKernelGGL(gather64<3>, dim3(gN), dim3(256), 0, 0, rec, idx, N, sink);
    hipLaunchKernelGGL(gather4, dim3(gN), dim3(256), 0, 0, (const float*)rec, idx, N, sink);
    hipLaunchKernelGGL(write4_stream, dim3(g4N), dim3(256), 0, 0, (float*)out, 4 * N);
    hipLaunchKernelGGL(write16_stream, dim3(g4N), dim3(256), 0, 0, out, 4 * N);
    hipLaunchKernelGGL(write64_rec, dim3(gN), dim3(256), 0, 0, out, N);
  }
  CHECK(hipDeviceSynchronize());
  // true bytes per launch, one line per kernel (tools/calibrate_fetch.py joins this with the counter CSV)
  printf("{\"N\": %zu, \"true_bytes\": {\"stream16\": {\"read\": %zu, \"write\": 0}, "
         "\"gather64<2>\": {\"read\": %zu, \"read_sectors64\": %zu, \"write\": 0}, "
         "\"gather64<3>\": {\"read\": %zu, \"read_sectors64\": %zu, \"write\": 0}, "
         "\"gather4\": {\"read\": %zu, \"read_sectors64\": %zu, \"write\": 0}, "
         "\"write4_stream\": {\"read\": 0, \"write\": %zu}, \"write16_stream\": {\"read\": 0, \"write\": %zu}, "
         "\"write64_rec\": {\"read\": 0, \"write\": %zu}}}\n",
         N, N * 64, N * 32 + N * 4, N * 64 + N * 4, N * 48 + N * 4, N * 64 + N * 4, N * 4 + N * 4, N * 64 + N * 4,
         N * 16, N * 64, N * 64);
  return 0;
}
