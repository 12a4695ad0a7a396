// issue-rate probe for gfx950: v_fma_f32 vs v_pk_fma_f32 vs v_sin_f32 and their mixes (diagnostics only)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define ITERS 512
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, float a, float b) {
  float x[8];
  f32x2 y[8];
  for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 1e-3f + i; y[i] = f32x2{x[i], x[i] + 0.5f}; }
  const f32x2 av = {a, a}, bv = {b, b};
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#define FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b))
#define PK(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[i]) : "v"(av), "v"(bv))
#define SIN(i) asm volatile("v_sin_f32 %0, %0" : "+v"(x[i]))
      if (MODE == 0) FMA(i);
      if (MODE == 1) PK(i);
      if (MODE == 2) SIN(i);
      if (MODE == 3) { FMA(i); if (i < 2) SIN(i + 4); }
      if (MODE == 4) { PK(i); if (i < 2) SIN(i); }
      if (MODE == 5) { FMA(i); if (i < 4) PK(i); }
    }
  }
  float t = 0.f;
  for (int i = 0; i < 8; ++i) t += x[i] + y[i].x + y[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = t;
}
int main() {
  const int wgs = 2048;                       // 8 waves per SIMD
  float* out; hipMalloc(&out, wgs * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[] = {"8 v_fma_f32", "8 v_pk_fma_f32", "8 v_sin_f32", "8 fma + 2 sin", "8 pk_fma + 2 sin", "8 fma + 4 pk_fma"};
  for (int m = 0; m < 6; ++m) {
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      switch (m) {
        case 0: k<0><<<wgs, 256>>>(out, 0.999f, 0.001f); break;
        case 1: k<1><<<wgs, 256>>>(out, 0.999f, 0.001f); break;
        case 2: k<2><<<wgs, 256>>>(out, 0.999f, 0.001f); break;
        case 3: k<3><<<wgs, 256>>>(out, 0.999f, 0.001f); break;
        case 4: k<4><<<wgs, 256>>>(out, 0.999f, 0.001f); break;
        case 5: k<5><<<wgs, 256>>>(out, 0.999f, 0.001f); break;
      }
      hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    // cycles per loop iteration per SIMD-resident wave: ms * clk / (ITERS * waves_per_simd)
    double cyc = ms * 1e-3 * 2.4e9 / (ITERS * 8.0);
    printf("%-20s %.3f ms  -> %.1f cycles per iteration per wave (at 2.4 GHz, 8 waves/SIMD)\n", names[m], ms, cyc);
  }
  return 0;
}
