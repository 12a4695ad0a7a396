// How many 256-thread workgroups with a given static LDS size does a CU hold at once?  Each workgroup stamps its start
// time, then idles for ~30 us; the number of stamps in the first few microseconds / 256 CUs is the resident count.
// hipcc --offload-arch=gfx950 -O2 tools/probes/lds_occupancy.hip -o /tmp/lds_occ && /tmp/lds_occ
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int BYTES>
__global__ void __launch_bounds__(256) k_hold(long long* start, float* sink) {
  __shared__ float buf[BYTES / 4];
  const long long t0 = wall_clock64();
  if (threadIdx.x == 0) start[blockIdx.x] = t0;
  buf[threadIdx.x] = (float)threadIdx.x;
  buf[BYTES / 4 - 1 - threadIdx.x] = 1.f;
  __syncthreads();
  while (wall_clock64() - t0 < 3000) { }              // 30 us at 100 MHz
  if (buf[(threadIdx.x * 7) % (BYTES / 4)] < -1.f) sink[0] = 1.f;
}

template <int BYTES>
void run(long long* d, float* sink) {
  const int n = 256 * 8;
  hipMemset(d, 0, n * sizeof(long long));
  hipLaunchKernelGGL(k_hold<BYTES>, dim3(n), dim3(256), 0, 0, d, sink);
  hipDeviceSynchronize();
  std::vector<long long> h(n);
  hipMemcpy(h.data(), d, n * sizeof(long long), hipMemcpyDeviceToHost);
  const long long t0 = *std::min_element(h.begin(), h.end());
  int first = 0;
  for (long long t : h) first += (t - t0) < 1000;     // started within 10 us
  std::printf("LDS %6d B per workgroup: %4d workgroups resident at once = %.2f per CU\n", BYTES, first, first / 256.0);
}

int main() {
  long long* d; float* sink;
  hipMalloc(&d, 256 * 8 * sizeof(long long)); hipMalloc(&sink, 4);
  run<40960>(d, sink); run<39936>(d, sink); run<32768>(d, sink); run<32768>(d, sink); run<32704>(d, sink); run<32256>(d, sink); run<31744>(d, sink); run<30720>(d, sink);
  run<27000>(d, sink); run<23000>(d, sink); run<16384>(d, sink);
  return 0;
}
