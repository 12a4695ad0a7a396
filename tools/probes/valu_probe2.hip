// issue-rate probe 2 for gfx950 (diagnostics only): packed-f32 forms by operand count, the DPP forms of the lane-pair
// step, dependent chains with and without the s_nop the compiler puts between packed producers and their consumers, at
// 1 / 2 / 4 / 8 waves per SIMD.  Prints shader cycles per instruction per SIMD (s_memtime ticks of wave 0 of workgroup 0
// are not used: the figure is wall time x the clock the same run measures with a plain v_fma_f32 loop = 4 cycles assumed
// NOT; instead every mode is reported relative to wall time and the launch's own cycle counter).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define ITERS 1024

template <int MODE>
__global__ void __launch_bounds__(64) k(float* out, long long* cyc, float a, float b) {
  const long long w0 = wall_clock64();
  f32x2 y[8];
  float x[8];
  for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 1e-3f + i; y[i] = f32x2{x[i], x[i] + 0.5f}; }
  f32x2 av = {a, a + 1e-3f}, bv = {b, b - 1e-3f};
  const float sg = (threadIdx.x & 1) ? -1.f : 1.f;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
      if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[i]) : "v"(av), "v"(bv));
      if (MODE == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(y[i]) : "v"(av));
      if (MODE == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(y[i]) : "v"(av));
      if (MODE == 4) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(y[i]) : "v"(av));
      if (MODE == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "+v"(y[i]) : "v"(av), "v"(bv));
      if (MODE == 6) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(x[i]) : "v"(x[(i + 1) & 7]));
      if (MODE == 7) asm volatile("v_fmac_f32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x[i]) : "v"(sg));
      if (MODE == 8) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
      // one dependent chain of packed instructions (y[0] only), 8 per iteration
      if (MODE == 9) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[0]) : "v"(av), "v"(bv));
      if (MODE == 10) asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n s_nop 0" : "+v"(y[0]) : "v"(av), "v"(bv));
      // independent packed instructions with an s_nop 0 behind each
      if (MODE == 11) asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n s_nop 0" : "+v"(y[i]) : "v"(av), "v"(bv));
      // two chains interleaved
      if (MODE == 12) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[i & 1]) : "v"(av), "v"(bv));
      if (MODE == 13) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x[i]), "+v"(x[(i + 1) & 7]));
      if (MODE == 14) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[0,1]" : "+v"(y[i]) : "v"(av));
      if (MODE == 15) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[0]) : "v"(a), "v"(b));
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float t = 0.f;
  for (int i = 0; i < 8; ++i) t += x[i] + y[i].x + y[i].y;
  out[blockIdx.x * 64 + threadIdx.x] = t;
  if (threadIdx.x == 0) {                             // per wave: cycles, wall ticks (100 MHz), start and end wall stamps
    const long long w1 = wall_clock64();
    cyc[4 * blockIdx.x + 0] = t1 - t0;
    cyc[4 * blockIdx.x + 1] = w1 - w0;
    cyc[4 * blockIdx.x + 2] = w0;
    cyc[4 * blockIdx.x + 3] = w1;
  }
}

template <int MODE>
static void run(const char* name, float* out, long long* cyc) {
  printf("%-42s", name);
  static long long h[4 * 256 * 4 * 8];
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int wlist[] = {1, 2, 3, 4, 8};
  for (int wi = 0; wi < 5; ++wi) {
    const int wps = wlist[wi];
    const int wgs = 256 * 4 * wps;                      // one-wave workgroups: wps waves on every SIMD
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      k<MODE><<<wgs, 64>>>(out, cyc, 0.999f, 0.001f);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    hipMemcpy(h, cyc, sizeof(long long) * 4 * wgs, hipMemcpyDeviceToHost);
    double sum_c = 0, sum_w = 0; long long first = h[2], last = h[3], last_start = h[2];
    for (int i = 0; i < wgs; ++i) {
      sum_c += h[4 * i]; sum_w += h[4 * i + 1];
      if (h[4 * i + 2] < first) first = h[4 * i + 2];
      if (h[4 * i + 2] > last_start) last_start = h[4 * i + 2];
      if (h[4 * i + 3] > last) last = h[4 * i + 3];
    }
    const double clk_ghz = sum_c / sum_w * 0.1;          // shader cycles per 100 MHz tick
    const double per_wave = sum_c / wgs / (ITERS * 8.0);  // cycles between two instructions of one wave (mean over waves)
    const double span_us = (last - first) * 0.01;        // first wave start to last wave end
    const double per_simd = span_us * 1e-6 * clk_ghz * 1e9 / (ITERS * 8.0) / wps;   // cycles per instruction per SIMD (throughput)
    printf(" | %dw: wave %5.2f simd %5.2f clk %.2f start-spread %.1fus", wps, per_wave, per_simd, clk_ghz, (last_start - first) * 0.01);
    (void)ms;
  }
  printf("\n");
}

int main() {
  float* out; hipMalloc(&out, 256 * 4 * 8 * 64 * 4);
  long long* cyc; hipMalloc(&cyc, 8 * 4 * 256 * 4 * 8);
  run<0>("v_fma_f32 (3 vgpr)", out, cyc);
  run<8>("v_add_f32", out, cyc);
  run<1>("v_pk_fma_f32 (3 vgpr pairs)", out, cyc);
  run<4>("v_pk_fma_f32 (src1 == src2)", out, cyc);
  run<5>("v_pk_fma_f32 op_sel/neg (cmul 2nd half)", out, cyc);
  run<2>("v_pk_mul_f32", out, cyc);
  run<14>("v_pk_mul_f32 op_sel_hi (cmul 1st half)", out, cyc);
  run<3>("v_pk_add_f32", out, cyc);
  run<6>("v_mov_b32_dpp quad_perm", out, cyc);
  run<7>("v_fmac_f32_dpp quad_perm", out, cyc);
  run<13>("v_permlane32_swap_b32", out, cyc);
  run<15>("v_fma_f32, one dependent chain", out, cyc);
  run<9>("v_pk_fma_f32, one dependent chain", out, cyc);
  run<10>("v_pk_fma_f32 chain + s_nop 0 each", out, cyc);
  run<12>("v_pk_fma_f32, two chains interleaved", out, cyc);
  run<11>("v_pk_fma_f32 independent + s_nop 0 each", out, cyc);
  return 0;
}
