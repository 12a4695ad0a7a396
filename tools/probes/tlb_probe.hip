// Address-translation probe: N independent 4-byte loads, one per `stride` bytes of a large buffer, from a cold start
// (the buffer is written once, then an unrelated 1 GiB is swept to push its translations out).  Time per load against
// the stride tells how large the mapped fragments are and what a translation miss costs on this box.
// hipcc --offload-arch=gfx950 -O2 tools/probes/tlb_probe.hip -o tools/ab/tlb_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_touch(const float* __restrict__ p, size_t stride_floats, size_t n, float* sink) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const float v = p[i * stride_floats]; if (v == 12345.678f) sink[0] = v; }
}
__global__ void k_fill(float* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.0f;
}
// every workgroup reads its own contiguous 64 KiB (as a filter workgroup's first loads do): wall time of the launch
__global__ void k_first_loads(const float* __restrict__ p, size_t wg_floats, float* sink) {
  const float* q = p + (size_t)blockIdx.x * wg_floats;
  float a = 0.f;
  for (int r = 0; r < 28; ++r) a += q[(size_t)r * 512 + threadIdx.x];
  if (a == 12345.678f) sink[0] = a;
}

int main() {
  const size_t bytes = (size_t)8 << 30;
  float *buf, *other, *sink;
  hipMalloc(&buf, bytes); hipMalloc(&other, (size_t)1 << 30); hipMalloc(&sink, 4);
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, buf, bytes / 4);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t strides[] = {256, 4096, 65536, 2u << 20};
  for (size_t s : strides) {
    size_t n = bytes / s; if (n > (1u << 20)) n = 1u << 20;
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, other, ((size_t)1 << 30) / 4);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_touch, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, buf, s / 4, n, sink);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      std::printf("stride %8zu B: %8zu loads in %8.1f us = %6.2f ns per load (rep %d)\n", s, n, ms * 1e3, ms * 1e6 / n, rep);
    }
  }
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, other, ((size_t)1 << 30) / 4);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_first_loads, dim3(1024), dim3(128), 0, 0, buf, (size_t)16384, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::printf("1024 workgroups x 128 threads x 28 loads from their own 64 KiB: %.1f us (rep %d)\n", ms * 1e3, rep);
  }
  return 0;
}
