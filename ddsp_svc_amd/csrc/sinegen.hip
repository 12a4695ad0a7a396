// Harmonic source of NSF-HiFiGAN: nsf_hifigan/models.py:101-204, SourceModuleHnNSF.forward =
// tanh(Linear(SineGen(f0, upp))).  The reference materialises [B, T, dim] sine waves, the same amount of noise, the
// voiced mask and the noise amplitudes (dim = 9: ~0.5 GB each at B = 32 x 10 s) before the 9 -> 1 linear layer
// collapses them; here one pass reads the noise draw and writes the merged [B, T] excitation.
//
// The float32 recipe of _f02sine (models.py:140-154) is followed op for op: rad = (f0 / sr) * (i + 1) inside a
// frame, frame totals wrapped by fmod(. + 0.5, 1) - 0.5, cumulative sum over frames (ATen: float64 running sum,
// float32 outputs -- the terms are multiples of 2^-24 below 1, so a parallel float64 scan is bit-identical), fmod 1,
// shifted by one frame; per harmonic rad * h + rand_ini, sine of the float32 product 2 pi rad.
#include "ddsp_common.h"
#include "kernels.h"

namespace ddsp {

// frame-rate part: rad_acc[b, l] = fmod(float(cumsum_{l' <= l} rad2[l']), 1); one 256-thread workgroup per utterance
__global__ void __launch_bounds__(256) k_sinegen_scan(const float* __restrict__ f0, int L, int upp, float sr,
                                                      float* __restrict__ rad_acc) {
  __shared__ double wsum[4];
  __shared__ double carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long b = blockIdx.x;
  if (tid == 0) carry_s = 0.0;
  __syncthreads();
  for (int base = 0; base < L; base += 256) {
    const int l = base + tid;
    float rad2 = 0.f;
    if (l < L) {
      const float last = (f0[b * L + l] / sr) * (float)upp;                          // rad[..., -1], models.py:141-142
      rad2 = fmodf(last + 0.5f, 1.0f) - 0.5f;
    }
    const double v = (double)rad2;
    const double excl = wave_excl_scan(v, lane);
    if (lane == 63) wsum[wave] = excl + v;
    __syncthreads();
    double pre = carry_s;
    for (int w = 0; w < wave; ++w) pre += wsum[w];
    const double incl = pre + excl + v;
    if (l < L) rad_acc[b * L + l] = fmodf((float)incl, 1.0f);                         // models.py:143
    __syncthreads();
    if (tid == 255) carry_s = incl;
    __syncthreads();
  }
}

// per-sample part; DIM harmonics, one thread per sample
template <int DIM>
__global__ void __launch_bounds__(256) k_sinegen(const float* __restrict__ f0, const float* __restrict__ rad_acc,
                                                 const float* __restrict__ rand_ini, const float* __restrict__ noise,
                                                 const float* __restrict__ weight, const float* __restrict__ bias,
                                                 int L, int upp, float sr, float sine_amp, float noise_std,
                                                 float voiced_threshold, long total, float* __restrict__ out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const long T = (long)L * upp;
  const long b = i / T;
  const int t = (int)(i - b * T);
  const int l = t / upp, n = t - l * upp;
  const float f = f0[b * L + l];
  float rad = (f / sr) * (float)(n + 1);                                             // models.py:141
  rad = rad + (l > 0 ? rad_acc[b * L + l - 1] : 0.0f);                               // models.py:144
  const bool voiced = f > voiced_threshold;                                          // models.py:163-164
  const float namp = voiced ? noise_std : sine_amp / 3.0f;                           // models.py:165
  const float* nz = noise + i * DIM;
  float acc = bias[0];
#pragma unroll
  for (int h = 0; h < DIM; ++h) {
    const float r = rad * (float)(h + 1) + rand_ini[h];                              // models.py:146-149
    const float s = sin_turns(kTwoPiF * r) * sine_amp;                               // models.py:150, :162
    const float wave = (voiced ? s : 0.0f) + namp * nz[h];                           // models.py:166-167
    acc = fmaf(weight[h], wave, acc);                                                // models.py:203 (Linear)
  }
  out[i] = tanhf(acc);                                                               // models.py:203 (Tanh)
}

int launch_sine_source(const float* f0, int B, int L, int upp, double sr, const float* rand_ini, const float* noise,
                       const float* weight, const float* bias, int dim, float sine_amp, float noise_std,
                       float voiced_threshold, float* rad_acc, float* out, hipStream_t st) {
  if (dim != 9 && dim != 1) return -1;
  const long total = (long)B * L * upp;
  const long blocks = (total + 255) / 256;
  if (blocks > 0x7fffffffL) return -1;
  hipLaunchKernelGGL(k_sinegen_scan, dim3((unsigned)B), dim3(256), 0, st, f0, L, upp, (float)sr, rad_acc);
  if (dim == 9)
    hipLaunchKernelGGL(k_sinegen<9>, dim3((unsigned)blocks), dim3(256), 0, st, f0, rad_acc, rand_ini, noise, weight, bias, L,
                       upp, (float)sr, sine_amp, noise_std, voiced_threshold, total, out);
  else
    hipLaunchKernelGGL(k_sinegen<1>, dim3((unsigned)blocks), dim3(256), 0, st, f0, rad_acc, rand_ini, noise, weight, bias, L,
                       upp, (float)sr, sine_amp, noise_std, voiced_threshold, total, out);
  return 0;
}

}  // namespace ddsp
