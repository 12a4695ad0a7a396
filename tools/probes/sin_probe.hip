// accuracy + throughput probe for sine variants on gfx950 (diagnostics only, not part of the library)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__device__ __forceinline__ float sin_hw(float x) {      // 2-constant Cody-Waite to [-pi,pi], then v_sin_f32 (input in turns)
  const float inv2pi = 0.15915494309189535f;
  const float c_hi = 6.28318548202514648f;               // fl32(2pi)
  const float c_lo = -1.7484555e-7f;                     // 2pi - fl32(2pi)
  float n = rintf(x * inv2pi);
  float r = fmaf(-n, c_hi, x);
  r = fmaf(-n, c_lo, r);
  return __builtin_amdgcn_sinf(r * inv2pi);
}
__device__ __forceinline__ float sin_poly(float x) {    // quarter-turn reduction + minimax polynomials
  const float two_over_pi = 0.63661977236758138f;
  const float p_hi = 1.57079637050628662f;               // fl32(pi/2)
  const float p_lo = -4.37113883e-8f;                    // pi/2 - fl32(pi/2)
  float n = rintf(x * two_over_pi);
  float r = fmaf(-n, p_hi, x);
  r = fmaf(-n, p_lo, r);
  int q = (int)n;
  float r2 = r * r;
  // sin(r), |r| <= pi/4
  float s = fmaf(r2, fmaf(r2, fmaf(r2, 2.6083159809786593541503e-06f, -1.9810690719168633222580e-04f), 8.3330785855650901794434e-03f), -1.6666659712791442871094e-01f);
  s = fmaf(s * r2, r, r);
  // cos(r)
  float c = fmaf(r2, fmaf(r2, fmaf(r2, 2.4433157138526439666748e-05f, -1.3887316826730966567993e-03f), 4.1666645556688308715820e-02f), -0.5f);
  c = fmaf(c, r2, 1.0f);
  float v = (q & 1) ? c : s;
  return (q & 2) ? -v : v;
}
__global__ void k_eval(const float* x, float* a, float* b, float* c, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { a[i] = sinf(x[i]); b[i] = sin_hw(x[i]); c[i] = sin_poly(x[i]); }
}
template <int V>
__global__ void k_time(const float* ph, float* out, int H) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  float p[8], acc[8];
  for (int r = 0; r < 8; ++r) { p[r] = ph[i * 8 + r]; acc[r] = 0.f; }
  for (int k = 1; k <= H; ++k) {
    float kf = (float)k;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      float arg = p[r] * kf;
      float s = V == 0 ? sinf(arg) : V == 1 ? sin_hw(arg) : sin_poly(arg);
      acc[r] = fmaf(s, 0.37f, acc[r]);
    }
  }
  float t = 0.f;
  for (int r = 0; r < 8; ++r) t += acc[r];
  out[i] = t;
}
int main() {
  const int n = 1 << 22;
  std::vector<float> hx(n);
  srand(1);
  for (int i = 0; i < n; ++i) hx[i] = ((float)rand() / RAND_MAX * 2.f - 1.f) * 805.f;
  for (int i = 0; i < 4096; ++i) hx[i] = (float)(3.14159265358979 * (i - 2048) / 8.0);   // near multiples of pi/8
  float *dx, *da, *db, *dc;
  hipMalloc(&dx, n * 4); hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dc, n * 4);
  hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
  k_eval<<<n / 256, 256>>>(dx, da, db, dc, n);
  std::vector<float> ha(n), hb(n), hc(n);
  hipMemcpy(ha.data(), da, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(hb.data(), db, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(hc.data(), dc, n * 4, hipMemcpyDeviceToHost);
  double ma = 0, mb = 0, mc = 0, ra = 0, rb = 0, rc = 0;
  for (int i = 0; i < n; ++i) {
    double t = sin((double)hx[i]);
    double ea = fabs(ha[i] - t), eb = fabs(hb[i] - t), ec = fabs(hc[i] - t);
    ma = fmax(ma, ea); mb = fmax(mb, eb); mc = fmax(mc, ec);
    ra += ea * ea; rb += eb * eb; rc += ec * ec;
  }
  printf("abs err vs double  max / rms:  sinf %.3e %.3e   hw %.3e %.3e   poly %.3e %.3e\n", ma, sqrt(ra / n), mb, sqrt(rb / n), mc, sqrt(rc / n));
  const int threads = 32 * 862 * 64;   // one lane per 8 samples, as k_sins_bank
  float *ph, *out;
  hipMalloc(&ph, (size_t)threads * 8 * 4); hipMalloc(&out, (size_t)threads * 4);
  hipMemcpy(ph, hx.data(), (size_t)(threads * 8 < n ? threads * 8 : n) * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int v = 0; v < 3; ++v) {
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (v == 0) k_time<0><<<threads / 256, 256>>>(ph, out, 256);
      if (v == 1) k_time<1><<<threads / 256, 256>>>(ph, out, 256);
      if (v == 2) k_time<2><<<threads / 256, 256>>>(ph, out, 256);
      hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    printf("variant %d: %.3f ms for %.2f G sines\n", v, ms, threads * 8.0 * 256 / 1e9);
  }
  return 0;
}
