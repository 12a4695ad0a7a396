// sustained rate of the float32 MFMA instructions on gfx950 (diagnostics only): what ceiling do k_ir_gemm
// (v_mfma_f32_32x32x2_f32) and k_fir_mfma (v_mfma_f32_16x16x4_f32) run against in practice?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define ITERS 2048
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, float a, float b) {
  const float av = a + threadIdx.x * 1e-6f, bv = b;
  float t = 0.f;
  if (MODE == 0) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) t += acc[i][j];
  } else {
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) t += acc[i][j];
  }
  out[blockIdx.x * 256 + threadIdx.x] = t;
}
int main() {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wps = 1; wps <= 2; ++wps) {
    const int wgs = 256 * wps;                        // 4 waves per workgroup: wps waves per SIMD
    for (int m = 0; m < 2; ++m) {
      float ms = 0, best = 1e9;
      for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        if (m == 0) k<0><<<wgs, 256>>>(out, 0.5f, 0.25f); else k<1><<<wgs, 256>>>(out, 0.5f, 0.25f);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      const double flops = (double)wgs * 4 * ITERS * 4 * (m == 0 ? 32.0 * 32 * 2 * 2 : 16.0 * 16 * 4 * 2);
      printf("%s, %d wave(s)/SIMD: %.3f ms -> %.1f TFLOP/s (%.1f cycles per MFMA at 2.4 GHz)\n",
             m == 0 ? "v_mfma_f32_32x32x2_f32" : "v_mfma_f32_16x16x4_f32", wps, best, flops / (best * 1e-3) / 1e12,
             best * 1e-3 * 2.4e9 / (ITERS * 4.0 * wps));
    }
  }
  return 0;
}
