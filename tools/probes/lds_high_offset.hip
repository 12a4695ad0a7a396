// Are LDS reads at high offsets of a workgroup's allocation slower than at low ones?  (Round 6: k_taps_pfa510 read a 16-row
// table at byte offset 32768 of its 33 KB allocation and half of a launch's first-round workgroups spent 10 - 15 us instead of
// 2.6 in that loop; with the table at offset 32640 none did.)  A workgroup of 256 threads with ALLOC bytes of static LDS
// reads, REPS times, a dword whose address is the same for all lanes of a wave (a broadcast read, as the table's) or lane-
// consecutive, at byte offset OFF, and stamps the loop's duration; the launch is 1 round (<= resident) or 2 rounds of workgroups.
// hipcc --offload-arch=gfx950 -O2 tools/probes/lds_high_offset.hip -o /tmp/lds_hi && /tmp/lds_hi
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

template <int ALLOC>
__global__ void __launch_bounds__(256) k_read(int off_bytes, int broadcast, int reps, long long* dur, float* sink) {
  __shared__ __attribute__((aligned(16))) float buf[ALLOC / 4];
  for (int i = threadIdx.x; i < ALLOC / 4; i += 256) buf[i] = (float)i;
  __syncthreads();
  const int base = off_bytes / 4 + (broadcast ? ((threadIdx.x >> 6) & 3) : (threadIdx.x & 31));
  const volatile float* p = buf + base;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f, a5 = 0.f, a6 = 0.f, a7 = 0.f;
  const long long t0 = wall_clock64();
#pragma unroll 1
  for (int r = 0; r < reps; r += 8) {                      // eight independent reads in flight: throughput, not latency
    const float v0 = p[0], v1 = p[4], v2 = p[8], v3 = p[12], v4 = p[16], v5 = p[20], v6 = p[24], v7 = p[28];
    a0 += v0; a1 += v1; a2 += v2; a3 += v3; a4 += v4; a5 += v5; a6 += v6; a7 += v7;
  }
  const long long t1 = wall_clock64();
  if (threadIdx.x == 0) dur[blockIdx.x] = t1 - t0;
  const float acc = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
  if (acc == -1.f) sink[0] = acc;
}

template <int ALLOC>
void run(int off, int broadcast, int wgs, long long* d, float* sink) {
  const int reps = 4096;
  hipMemset(d, 0, wgs * sizeof(long long));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_read<ALLOC>, dim3(wgs), dim3(256), 0, 0, off, broadcast, reps, d, sink);
  hipDeviceSynchronize();
  std::vector<long long> h(wgs);
  hipMemcpy(h.data(), d, wgs * sizeof(long long), hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  std::printf("alloc %6d B  offset %6d  %-9s  %5d workgroups:  per read  p10 %6.1f  p50 %6.1f  p90 %6.1f  max %6.1f ns\n", ALLOC, off,
              broadcast ? "broadcast" : "lanes", wgs, 10.0 * h[wgs / 10] / reps, 10.0 * h[wgs / 2] / reps, 10.0 * h[wgs * 9 / 10] / reps,
              10.0 * h[wgs - 1] / reps);
}

int main() {
  long long* d; float* sink;
  hipMalloc(&d, 8192 * sizeof(long long)); hipMalloc(&sink, 4);
  for (int wgs : {1024, 1724}) {
    for (int bc : {1, 0}) {
      run<33088>(0, bc, wgs, d, sink);
      run<33088>(16384, bc, wgs, d, sink);
      run<33088>(32512, bc, wgs, d, sink);
      run<33088>(32768, bc, wgs, d, sink);
      run<33088>(32896, bc, wgs, d, sink);
      run<40960>(0, bc, wgs, d, sink);
      run<40960>(32512, bc, wgs, d, sink);
      run<40960>(32768, bc, wgs, d, sink);
      run<40960>(36864, bc, wgs, d, sink);
      run<40960>(40000, bc, wgs, d, sink);
      run<32768>(32512, bc, wgs, d, sink);
      run<24576>(24000, bc, wgs, d, sink);
      run<65536>(0, bc, wgs, d, sink);
      run<65536>(32768, bc, wgs, d, sink);
      run<65536>(65000, bc, wgs, d, sink);
    }
  }
  return 0;
}
