// self-test kernels for the emulator itself (compiled by tests/hipemu/build.py, CPU only)
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k_reverse(const float* in, float* out, int n) {
  __shared__ float buf[256];
  int t = threadIdx.x;
  int base = blockIdx.x * 256;
  buf[t] = (base + t < n) ? in[base + t] : 0.f;
  __syncthreads();
  if (base + t < n) out[base + t] = buf[255 - t];
}

__global__ void k_wave_scan(const double* in, double* out) {
  int t = threadIdx.x + blockIdx.x * blockDim.x;
  int lane = threadIdx.x & 63;
  double v = in[t];
  for (int d = 1; d < 64; d <<= 1) {
    double u = __shfl_up(v, d);
    if (lane >= d) v += u;
  }
  out[t] = v;
}

// C[16x16] = A[16x4] * B[4x16] + C, one wave
__global__ void k_mfma(const float* A, const float* B, float* C) {
  int l = threadIdx.x;
  float a = A[(l & 15) * 4 + (l >> 4)];
  float b = B[(l >> 4) * 16 + (l & 15)];
  f32x4 c;
  for (int r = 0; r < 4; ++r) c[r] = C[((l >> 4) * 4 + r) * 16 + (l & 15)];
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) C[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}

__global__ void k_early_exit(float* out, int n) {
  __shared__ float s[128];
  int t = threadIdx.x;
  if (t >= n) return;                 // whole trailing work-items leave before the barrier
  s[t] = float(t);
  __syncthreads();
  out[t] = s[n - 1 - t];
}

extern "C" int emu_selftest() {
  int bad = 0;
  {
    const int n = 700;
    float in[768], out[768];
    for (int i = 0; i < n; ++i) in[i] = float(i);
    hipLaunchKernelGGL(k_reverse, dim3(3), dim3(256), 0, nullptr, (const float*)in, out, n);
    for (int i = 0; i < n; ++i) {
      int b = i / 256, t = i % 256;
      int src = b * 256 + 255 - t;
      float want = src < n ? float(src) : 0.f;
      if (out[i] != want) bad |= 1;
    }
  }
  {
    double in[256], out[256];
    for (int i = 0; i < 256; ++i) in[i] = 1.0 + i;
    hipLaunchKernelGGL(k_wave_scan, dim3(2), dim3(128), 0, nullptr, (const double*)in, out);
    for (int i = 0; i < 256; ++i) {
      double want = 0;
      for (int j = (i / 64) * 64; j <= i; ++j) want += in[j];
      if (out[i] != want) bad |= 2;
    }
  }
  {
    float A[64], B[64], C[256], R[256];
    for (int i = 0; i < 64; ++i) { A[i] = float(i % 7) - 3.f; B[i] = float((i * 5) % 11) - 4.f; }
    for (int i = 0; i < 256; ++i) C[i] = float(i % 3);
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        float acc = C[i * 16 + j];
        for (int k = 0; k < 4; ++k) acc += A[i * 4 + k] * B[k * 16 + j];
        R[i * 16 + j] = acc;
      }
    hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, nullptr, (const float*)A, (const float*)B, C);
    for (int i = 0; i < 256; ++i) if (C[i] != R[i]) bad |= 4;
  }
  {
    float out[128];
    for (int i = 0; i < 128; ++i) out[i] = -1.f;
    hipLaunchKernelGGL(k_early_exit, dim3(1), dim3(128), 0, nullptr, out, 100);
    for (int i = 0; i < 100; ++i) if (out[i] != float(99 - i)) bad |= 8;
    for (int i = 100; i < 128; ++i) if (out[i] != -1.f) bad |= 8;
  }
  return bad;
}
