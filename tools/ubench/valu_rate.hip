// Microbenchmark: VALU issue rate of a gfx950 SIMD -- how many cycles does one wave64 VALU
// instruction occupy the SIMD's vector pipe?  ("Next round" item 2: DESIGN.md's
// roofline statements assumed 4; MI355X_MICROARCH.md says SIMD-32, 2.)
//
// Dependency-free streams (16 independent accumulators, round-robin) of
//   v_fma_f32      1 FMA per lane
//   v_pk_fma_f32   2 FMAs per lane (the instruction the render / backward quad loops are made of)
//   v_exp_f32      transcendental (quarter rate on earlier CDNA parts)
//   v_cndmask_b32 with an SGPR-pair mask, v_cmp -> SGPR pair + v_cndmask (the predicate algebra)
// at 1, 2, 4 and 8 waves per SIMD on every CU.  Reported per configuration:
//   clock           = shader clock while the waves ran: s_memtime ticks per s_memrealtime (100 MHz) tick
//   cyc/instr/SIMD  = launch wall time (hipEvents) x measured clock x SIMDs / wave-instructions issued
//                     (and, as a cross-check, from the waves' own s_memtime deltas)
//   Ginstr/s        = all wave-instructions of the launch / wall time
// Build + run:  hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

constexpr int UNROLL = 64;   // instructions per loop trip (4 x 16 accumulators)

template <int MODE>
__global__ void __launch_bounds__(256) rate_kernel(float* out, unsigned long long* cyc, int iters,
                                                    float b, float c) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,
        a6 = a0 + 6, a7 = a0 + 7, a8 = a0 + 8, a9 = a0 + 9, a10 = a0 + 10, a11 = a0 + 11,
        a12 = a0 + 12, a13 = a0 + 13, a14 = a0 + 14, a15 = a0 + 15;
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a8, a9}, p5 = {a10, a11},
     p6 = {a12, a13}, p7 = {a14, a15}, p8 = {a1, a0}, p9 = {a3, a2}, p10 = {a5, a4}, p11 = {a7, a6},
     p12 = {a9, a8}, p13 = {a11, a10}, p14 = {a13, a12}, p15 = {a15, a14};
  const f2 pb = {b, b}, pc = {c, c};
  const unsigned long long lanemask = __ballot(a0 > b * 20.f);
  unsigned long long m2 = 0;
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();   // constant 100 MHz
  const unsigned long long t0 = __builtin_readcyclecounter();        // s_memtime: shader clock
  for (int i = 0; i < iters; i++) {
    if (MODE == 0) {
#define I(r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(b), "v"(c));
      REP4(I(a0) I(a1) I(a2) I(a3) I(a4) I(a5) I(a6) I(a7) I(a8) I(a9) I(a10) I(a11) I(a12) I(a13) I(a14) I(a15))
#undef I
    } else if (MODE == 1) {
#define I(r) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(pb), "v"(pc));
      REP4(I(p0) I(p1) I(p2) I(p3) I(p4) I(p5) I(p6) I(p7) I(p8) I(p9) I(p10) I(p11) I(p12) I(p13) I(p14) I(p15))
#undef I
    } else if (MODE == 2) {
#define I(r) asm volatile("v_exp_f32 %0, %0" : "+v"(r));
      REP4(I(a0) I(a1) I(a2) I(a3) I(a4) I(a5) I(a6) I(a7) I(a8) I(a9) I(a10) I(a11) I(a12) I(a13) I(a14) I(a15))
#undef I
    } else if (MODE == 3) {   // v_cndmask_b32 with the lane mask in an SGPR pair (VOP3): what a
                              // predicate handed back from the scalar unit costs on the VALU
#define I(r) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(r) : "v"(b), "s"(lanemask));
      REP4(I(a0) I(a1) I(a2) I(a3) I(a4) I(a5) I(a6) I(a7) I(a8) I(a9) I(a10) I(a11) I(a12) I(a13) I(a14) I(a15))
#undef I
    } else if (MODE == 6) {   // v_cmp_gt_f32 writing an SGPR pair + v_cndmask reading it (1 : 1)
#define I(r) asm volatile("v_cmp_gt_f32_e64 %1, %0, %2\n\tv_cndmask_b32_e64 %0, %0, %2, %3" : "+v"(r), "=&s"(m2) : "v"(b), "s"(lanemask));
      REP4(I(a0) I(a1) I(a2) I(a3) I(a4) I(a5) I(a6) I(a7)) REP4(I(a8) I(a9) I(a10) I(a11) I(a12) I(a13) I(a14) I(a15))
#undef I
    } else if (MODE == 4) {   // v_pk_mul_f32 (the other packed op of the quad loops)
#define I(r) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(r) : "v"(pb));
      REP4(I(p0) I(p1) I(p2) I(p3) I(p4) I(p5) I(p6) I(p7) I(p8) I(p9) I(p10) I(p11) I(p12) I(p13) I(p14) I(p15))
#undef I
    } else if (MODE == 7) {   // cross-lane exchanges of the backward's reduction tree
#define I(r, q) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(r), "+v"(q));
      REP4(I(a0, a1) I(a2, a3) I(a4, a5) I(a6, a7) I(a8, a9) I(a10, a11) I(a12, a13) I(a14, a15)
           I(a0, a2) I(a1, a3) I(a4, a6) I(a5, a7) I(a8, a10) I(a9, a11) I(a12, a14) I(a13, a15))
#undef I
    } else if (MODE == 8) {
#define I(r, q) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(r), "+v"(q));
      REP4(I(a0, a1) I(a2, a3) I(a4, a5) I(a6, a7) I(a8, a9) I(a10, a11) I(a12, a13) I(a14, a15)
           I(a0, a2) I(a1, a3) I(a4, a6) I(a5, a7) I(a8, a10) I(a9, a11) I(a12, a14) I(a13, a15))
#undef I
    } else if (MODE == 9) {   // DPP add, quad_perm (source = another accumulator: no RAW hazard nops)
#define I(r, q) asm volatile("v_add_f32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(r) : "v"(q));
      REP4(I(a0, a8) I(a1, a9) I(a2, a10) I(a3, a11) I(a4, a12) I(a5, a13) I(a6, a14) I(a7, a15)
           I(a8, a0) I(a9, a1) I(a10, a2) I(a11, a3) I(a12, a4) I(a13, a5) I(a14, a6) I(a15, a7))
#undef I
    } else if (MODE == 10) {  // DPP add, row_ror:8
#define I(r, q) asm volatile("v_add_f32_dpp %0, %1, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(r) : "v"(q));
      REP4(I(a0, a8) I(a1, a9) I(a2, a10) I(a3, a11) I(a4, a12) I(a5, a13) I(a6, a14) I(a7, a15)
           I(a8, a0) I(a9, a1) I(a10, a2) I(a11, a3) I(a12, a4) I(a13, a5) I(a14, a6) I(a15, a7))
#undef I
    } else if (MODE == 11) {  // VOP2 multiply
#define I(r) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(r) : "v"(b));
      REP4(I(a0) I(a1) I(a2) I(a3) I(a4) I(a5) I(a6) I(a7) I(a8) I(a9) I(a10) I(a11) I(a12) I(a13) I(a14) I(a15))
#undef I
    } else if (MODE == 12) {  // VOP2 subtract
#define I(r) asm volatile("v_sub_f32_e32 %0, %1, %0" : "+v"(r) : "v"(b));
      REP4(I(a0) I(a1) I(a2) I(a3) I(a4) I(a5) I(a6) I(a7) I(a8) I(a9) I(a10) I(a11) I(a12) I(a13) I(a14) I(a15))
#undef I
    } else if (MODE == 13) {
#define I(r) asm volatile("v_rcp_f32_e32 %0, %0" : "+v"(r));
      REP4(I(a0) I(a1) I(a2) I(a3) I(a4) I(a5) I(a6) I(a7) I(a8) I(a9) I(a10) I(a11) I(a12) I(a13) I(a14) I(a15))
#undef I
    } else if (MODE == 14) {  // compare into an SGPR pair (no consumer)
#define I(r) asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(m2) : "v"(r), "v"(b));
      REP4(I(a0) I(a1) I(a2) I(a3) I(a4) I(a5) I(a6) I(a7) I(a8) I(a9) I(a10) I(a11) I(a12) I(a13) I(a14) I(a15))
#undef I
    } else if (MODE == 15) {  // VOP2 fused multiply-accumulate
#define I(r) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(r) : "v"(b), "v"(c));
      REP4(I(a0) I(a1) I(a2) I(a3) I(a4) I(a5) I(a6) I(a7) I(a8) I(a9) I(a10) I(a11) I(a12) I(a13) I(a14) I(a15))
#undef I
    } else if (MODE == 16) {
#define I(r) asm volatile("v_min_f32_e32 %0, %1, %0" : "+v"(r) : "v"(b));
      REP4(I(a0) I(a1) I(a2) I(a3) I(a4) I(a5) I(a6) I(a7) I(a8) I(a9) I(a10) I(a11) I(a12) I(a13) I(a14) I(a15))
#undef I
    } else if (MODE == 17) {  // integer add (loop / address arithmetic)
#define I(r) asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(r) : "v"(b));
      REP4(I(a0) I(a1) I(a2) I(a3) I(a4) I(a5) I(a6) I(a7) I(a8) I(a9) I(a10) I(a11) I(a12) I(a13) I(a14) I(a15))
#undef I
    } else if (MODE == 18) {  // VOP3 multiply with a negated source (what the gradient code emits)
#define I(r) asm volatile("v_mul_f32_e64 %0, %0, -%1" : "+v"(r) : "v"(b));
      REP4(I(a0) I(a1) I(a2) I(a3) I(a4) I(a5) I(a6) I(a7) I(a8) I(a9) I(a10) I(a11) I(a12) I(a13) I(a14) I(a15))
#undef I
    } else if (MODE == 19) {  // v_mov_b32
#define I(r, q) asm volatile("v_mov_b32_e32 %0, %1" : "=v"(r) : "v"(q));
      REP4(I(a0, a8) I(a1, a9) I(a2, a10) I(a3, a11) I(a4, a12) I(a5, a13) I(a6, a14) I(a7, a15)
           I(a8, a0) I(a9, a1) I(a10, a2) I(a11, a3) I(a12, a4) I(a13, a5) I(a14, a6) I(a15, a7))
#undef I
    } else if (MODE == 20) {  // signed integer minimum (orders non-negative floats like v_min_f32)
#define I(r) asm volatile("v_min_i32_e32 %0, %1, %0" : "+v"(r) : "v"(b));
      REP4(I(a0) I(a1) I(a2) I(a3) I(a4) I(a5) I(a6) I(a7) I(a8) I(a9) I(a10) I(a11) I(a12) I(a13) I(a14) I(a15))
#undef I
    } else if (MODE == 21) {
#define I(r) asm volatile("v_max_f32_e32 %0, %1, %0" : "+v"(r) : "v"(b));
      REP4(I(a0) I(a1) I(a2) I(a3) I(a4) I(a5) I(a6) I(a7) I(a8) I(a9) I(a10) I(a11) I(a12) I(a13) I(a14) I(a15))
#undef I
    } else if (MODE == 22) {
#define I(r) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(r) : "v"(b), "v"(c));
      REP4(I(a0) I(a1) I(a2) I(a3) I(a4) I(a5) I(a6) I(a7) I(a8) I(a9) I(a10) I(a11) I(a12) I(a13) I(a14) I(a15))
#undef I
    } else if (MODE == 23) {  // compare into VCC (VOPC encoding), no consumer
#define I(r) asm volatile("v_cmp_gt_f32_e32 vcc, %0, %1" : : "v"(r), "v"(b) : "vcc");
      REP4(I(a0) I(a1) I(a2) I(a3) I(a4) I(a5) I(a6) I(a7) I(a8) I(a9) I(a10) I(a11) I(a12) I(a13) I(a14) I(a15))
#undef I
    } else if (MODE == 24) {  // integer compare into an SGPR pair
#define I(r) asm volatile("v_cmp_lt_u32_e64 %0, %1, %2" : "=s"(m2) : "v"(r), "v"(b));
      REP4(I(a0) I(a1) I(a2) I(a3) I(a4) I(a5) I(a6) I(a7) I(a8) I(a9) I(a10) I(a11) I(a12) I(a13) I(a14) I(a15))
#undef I
    } else if (MODE == 25) {  // v_cndmask_b32 on VCC (VOP2 encoding)
#define I(r) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(r) : "v"(b) : "vcc");
      REP4(I(a0) I(a1) I(a2) I(a3) I(a4) I(a5) I(a6) I(a7) I(a8) I(a9) I(a10) I(a11) I(a12) I(a13) I(a14) I(a15))
#undef I
    } else if (MODE == 26) {  // v_pk_add_f32
#define I(r) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(r) : "v"(pb));
      REP4(I(p0) I(p1) I(p2) I(p3) I(p4) I(p5) I(p6) I(p7) I(p8) I(p9) I(p10) I(p11) I(p12) I(p13) I(p14) I(p15))
#undef I
    } else if (MODE == 27) {  // v_fma_f32 with three DIFFERENT rotating sources (register-bank pressure)
#define I(r, x, y) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r) : "v"(x), "v"(y));
      REP4(I(a0, a5, a10) I(a1, a6, a11) I(a2, a7, a12) I(a3, a8, a13) I(a4, a9, a14) I(a5, a10, a15) I(a6, a11, a0) I(a7, a12, a1)
           I(a8, a13, a2) I(a9, a14, a3) I(a10, a15, a4) I(a11, a0, a5) I(a12, a1, a6) I(a13, a2, a7) I(a14, a3, a8) I(a15, a4, a9))
#undef I
    } else if (MODE == 5) {   // alternating v_fma_f32 / v_exp_f32 (3 : 1, the quad loop's mix)
#define I4(r, s, t, u)                                                      \
  asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(b), "v"(c));      \
  asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s) : "v"(b), "v"(c));      \
  asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(t) : "v"(b), "v"(c));      \
  asm volatile("v_exp_f32 %0, %0" : "+v"(u));
      REP4(I4(a0, a1, a2, a3) I4(a4, a5, a6, a7) I4(a8, a9, a10, a11) I4(a12, a13, a14, a15))
#undef I4
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + a8 + a9 + a10 + a11 + a12 + a13 + a14 + a15;
  s += p0.x + p1.x + p2.x + p3.x + p4.x + p5.x + p6.x + p7.x + p8.y + p9.y + p10.y + p11.y + p12.y +
       p13.y + p14.y + p15.y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)(m2 & 1ull);
  if ((threadIdx.x & 63) == 0) {
    const size_t w = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    cyc[2 * w] = t1 - t0;
    cyc[2 * w + 1] = r1 - r0;
  }
}

template <int MODE>
void run(const char* name, int flop_per_lane_instr, int num_cu, double clock_ghz, FILE* js, bool first) {
  const int iters = 4096;
  for (int wps : {1, 2, 4, 8}) {
    // 256-thread workgroups = 4 waves, one per SIMD (the dispatcher spreads a workgroup's waves over
    // the CU's SIMDs); wps workgroups per CU
    const int blocks = num_cu * wps;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipMalloc(&cyc, (size_t)blocks * 4 * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    rate_kernel<MODE><<<blocks, 256>>>(out, cyc, 64, 0.999f, 0.001f);   // warm-up
    hipDeviceSynchronize();
    hipEventRecord(e0);
    rate_kernel<MODE><<<blocks, 256>>>(out, cyc, iters, 0.999f, 0.001f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)blocks * 4 * 2);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, rate_kernel<MODE>, 256, 0);
    const size_t nw = h.size() / 2;
    double sum = 0, mx = 0, rsum = 0;
    for (size_t w = 0; w < nw; w++) {
      sum += (double)h[2 * w]; rsum += (double)h[2 * w + 1];
      if ((double)h[2 * w] > mx) mx = (double)h[2 * w];
    }
    // shader clock while the waves ran: s_memtime ticks per s_memrealtime tick (100 MHz)
    const double ghz_meas = sum / rsum * 0.1;
    const double mean_cyc = sum / nw;
    const double instr_per_wave = (double)iters * UNROLL;
    const double total_instr = instr_per_wave * nw;
    const double ginstr_s = total_instr / (ms * 1e-3) * 1e-9;
    const double tflops = ginstr_s * 64.0 * flop_per_lane_instr * 1e-3;
    // cycles (of the MEASURED shader clock) one SIMD spends per wave-instruction, from the launch's
    // wall time: SIMD-seconds per instruction x clock
    const double cyc_wall = (ms * 1e-3) * ghz_meas * 1e9 * (4.0 * num_cu) / total_instr;
    // the same from a wave's own cycle count (valid when all wps waves of a SIMD are co-resident)
    const double cyc_wave = mean_cyc / instr_per_wave / (wps < occ ? wps : occ);
    printf("%-26s waves/SIMD %d (resident <= %d): %5.2f cyc/instr/SIMD from wall, %5.2f from the waves' s_memtime; clock %.2f GHz; %7.1f Ginstr/s %6.1f TFLOP/s\n",
           name, wps, occ, cyc_wall, cyc_wave, ghz_meas, ginstr_s, tflops);
    if (js) fprintf(js, "%s{\"stream\": \"%s\", \"waves_per_simd\": %d, \"resident_limit\": %d, \"cycles_per_instr_per_simd\": %.4f, "
                        "\"cycles_per_instr_per_simd_from_wave_counters\": %.4f, \"shader_clock_ghz\": %.3f, \"ginstr_per_s\": %.2f, \"tflops\": %.2f}",
                    (first && wps == 1) ? "" : ",\n  ", name, wps, occ, cyc_wall, cyc_wave, ghz_meas, ginstr_s, tflops);
    (void)mx; (void)clock_ghz;
    hipFree(out); hipFree(cyc); hipEventDestroy(e0); hipEventDestroy(e1);
  }
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { fprintf(stderr, "no HIP device\n"); return 1; }
  const int num_cu = prop.multiProcessorCount;
  const double clock_ghz = prop.clockRate * 1e-6;   // kHz -> GHz
  printf("device %s  CUs %d  clock %.3f GHz  (4 SIMDs per CU)\n", prop.gcnArchName, num_cu, clock_ghz);
  FILE* js = argc > 1 ? fopen(argv[1], "w") : nullptr;
  if (js) fprintf(js, "{\"device\": \"%s\", \"cus\": %d, \"clock_ghz\": %.3f, \"results\": [\n  ", prop.gcnArchName, num_cu, clock_ghz);
  run<0>("v_fma_f32", 2, num_cu, clock_ghz, js, true);
  run<1>("v_pk_fma_f32", 4, num_cu, clock_ghz, js, false);
  run<4>("v_pk_mul_f32", 2, num_cu, clock_ghz, js, false);
  run<2>("v_exp_f32", 1, num_cu, clock_ghz, js, false);
  run<3>("v_cndmask_b32 (sgpr mask)", 0, num_cu, clock_ghz, js, false);
  run<6>("v_cmp + v_cndmask", 0, num_cu, clock_ghz, js, false);
  run<5>("3 v_fma + 1 v_exp", 2, num_cu, clock_ghz, js, false);
  run<15>("v_fmac_f32 (VOP2)", 2, num_cu, clock_ghz, js, false);
  run<11>("v_mul_f32 (VOP2)", 1, num_cu, clock_ghz, js, false);
  run<18>("v_mul_f32 (VOP3, neg)", 1, num_cu, clock_ghz, js, false);
  run<12>("v_sub_f32 (VOP2)", 1, num_cu, clock_ghz, js, false);
  run<16>("v_min_f32", 0, num_cu, clock_ghz, js, false);
  run<17>("v_add_u32", 0, num_cu, clock_ghz, js, false);
  run<19>("v_mov_b32", 0, num_cu, clock_ghz, js, false);
  run<13>("v_rcp_f32", 1, num_cu, clock_ghz, js, false);
  run<14>("v_cmp_gt_f32 -> sgpr", 0, num_cu, clock_ghz, js, false);
  run<20>("v_min_i32", 0, num_cu, clock_ghz, js, false);
  run<21>("v_max_f32", 0, num_cu, clock_ghz, js, false);
  run<22>("v_med3_f32", 0, num_cu, clock_ghz, js, false);
  run<23>("v_cmp_gt_f32 -> vcc (e32)", 0, num_cu, clock_ghz, js, false);
  run<24>("v_cmp_lt_u32 -> sgpr", 0, num_cu, clock_ghz, js, false);
  run<26>("v_pk_add_f32", 2, num_cu, clock_ghz, js, false);
  run<27>("v_fma_f32 (3 rotating sources)", 2, num_cu, clock_ghz, js, false);
  run<7>("v_permlane32_swap", 0, num_cu, clock_ghz, js, false);
  run<8>("v_permlane16_swap", 0, num_cu, clock_ghz, js, false);
  run<9>("v_add_f32_dpp quad_perm", 1, num_cu, clock_ghz, js, false);
  run<10>("v_add_f32_dpp row_ror", 1, num_cu, clock_ghz, js, false);
  if (js) { fprintf(js, "\n]}\n"); fclose(js); }
  return 0;
}
