// Microbenchmark: instruction issue cost for ONE wave alone on a SIMD (the situation of the
// longest tile's wave in the tail of render_forward_kernel).  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_IT 4096
template <int MODE>
__global__ void k(float* out, unsigned long long* cyc, float a, float b) {
  float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < N_IT; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
      if (MODE == 0) { x0 = fmaf(x0, a, b); }                                   // dependent fma chain
      if (MODE == 1) { x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b); }  // 4 independent
      if (MODE == 2) { x0 = (x0 > b) ? x1 : x2; x1 = fmaf(x1, a, b); x2 += a; }  // cmp -> cndmask (VCC)
      if (MODE == 3) { bool p = x0 > b, q = x1 > a; x2 = (p && q) ? x2 + a : x2 * a; x0 += a; x1 -= a; }  // cmp,cmp -> s_and -> cndmask
      if (MODE == 4) { x0 = __expf(x0) * a; }                                   // dependent exp
      if (MODE == 5) { x0 = __expf(x0) * a; x1 = __expf(x1) * a; x2 = __expf(x2) * a; x3 = __expf(x3) * a; }
      if (MODE == 6) { unsigned long long m = __ballot(x0 > b); x0 += (float)__popcll(m) * a; }  // ballot round trip
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = x0 + x1 + x2 + x3;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE> void run(const char* name, int per_iter) {
  float* out; unsigned long long* cyc; hipMalloc(&out, 256); hipMalloc(&cyc, 8);
  k<MODE><<<1, 64>>>(out, cyc, 0.999f, 0.001f); hipDeviceSynchronize();
  k<MODE><<<1, 64>>>(out, cyc, 0.999f, 0.001f); hipDeviceSynchronize();
  unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-40s %8.2f cycles per group (%d VALU-ish instr/group)\n", name, (double)h / (N_IT * 16.0), per_iter);
}
int main() {
  run<0>("dependent fma", 1); run<1>("4 independent fma", 4); run<2>("cmp->cndmask + 2", 4);
  run<3>("2cmp->s_and->cndmask + 3", 7); run<4>("dependent exp+mul", 3); run<5>("4 indep exp+mul", 12);
  run<6>("ballot->popc->cvt->fma", 5);
  return 0;
}
