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
#include "philox.h"

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

// per-sample part; DIM harmonics, one thread per sample.  DRAW: the standard-normal noise (models.py:168, randn_like of the
// [B, T, DIM] sine waves) is not read but drawn here (philox.h, keyed by (seed, offset)): the kernel then reads 4 bytes of f0 per
// frame and writes 4 bytes per sample, where the resident draw costs 4 DIM bytes per sample of HBM traffic in this kernel, the
// same again in the kernel that wrote it, and [B, T, DIM] floats of memory (1.0 GB at B = 64 x 10 s).
template <int DIM, bool DRAW>
__global__ void __launch_bounds__(256) k_sinegen(const float* __restrict__ f0, const float* __restrict__ rad_acc,
                                                 const float* __restrict__ rand_ini, const float* __restrict__ noise,
                                                 const float* __restrict__ weight, const float* __restrict__ bias,
                                                 int L, int upp, float sr, float sine_amp, float noise_std,
                                                 float voiced_threshold, long total, float* __restrict__ out, NoiseGen rng) {
  // grid (blocks of an utterance, utterance): no 64-bit division per thread (it was a fifth of the DRAW kernel's instructions)
  const long T = (long)L * upp;
  const long tl = (long)blockIdx.x * 256 + threadIdx.x;
  if (tl >= T) return;
  const long b = blockIdx.y;
  const long i = b * T + tl;
  const unsigned t = (unsigned)tl;                                                   // T < 2^31 (launcher)
  const int l = (int)(t / (unsigned)upp), n = (int)(t - (unsigned)l * (unsigned)upp);
  const float f = f0[b * L + l];
  float rad = (f / sr) * (float)(n + 1);                                             // models.py:141
  rad = rad + (l > 0 ? rad_acc[b * L + l - 1] : 0.0f);                               // models.py:144
  const bool voiced = f > voiced_threshold;                                          // models.py:163-164
  const float namp = voiced ? noise_std : sine_amp / 3.0f;                           // models.py:165
  float nz[DIM];
  if (DRAW) {
#pragma unroll
    for (int j = 0; j < (DIM + 3) / 4; ++j) {
      const Normal4 q = philox_normal4(rng, (unsigned)b, (unsigned)t, (unsigned)j);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (4 * j + e < DIM) nz[4 * j + e] = q.z[e];
    }
  } else {
#pragma unroll
    for (int h = 0; h < DIM; ++h) nz[h] = noise[i * DIM + h];
  }
  float acc = bias[0];
#pragma unroll
  for (int h = 0; h < DIM; ++h) {
    const float r = rad * (float)(h + 1) + rand_ini[h];                              // models.py:146-149
    const float s = sin_turns(kTwoPiF * r) * sine_amp;                               // models.py:150, :162
    const float wave = (voiced ? s : 0.0f) + namp * nz[h];                           // models.py:166-167
    acc = fmaf(weight[h], wave, acc);                                                // models.py:203 (Linear)
  }
  out[i] = tanh_hw(acc);                                                             // models.py:203 (Tanh; hardware exponential, <= 1.5e-7)
}

// the draw of k_sinegen<DIM, true> written out: z[B, T, dim] (tests; callers that need the numbers themselves)
__global__ void __launch_bounds__(256) k_normal_noise(NoiseGen rng, long T, int dim, long total, float* __restrict__ out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;           // one thread per (sample, group of four harmonics)
  const int groups = (dim + 3) / 4;
  if (i >= total * groups) return;
  const long s = i / groups;
  const int j = (int)(i - s * groups);
  const long b = s / T;
  const Normal4 q = philox_normal4(rng, (unsigned)b, (unsigned)(s - b * T), (unsigned)j);
  for (int e = 0; e < 4; ++e)
    if (4 * j + e < dim) out[s * dim + 4 * j + e] = q.z[e];
}

int launch_normal_noise(unsigned long long seed, unsigned long long offset, int B, long T, int dim, float* out, hipStream_t st) {
  // the counter holds t in 32 bits and 4 * offset_hi + j in 32: j = h / 4 must stay below 4 (16 harmonics), or call j of one
  // offset would be call j - 4 of the next
  if (T >= (1L << 32) || dim < 1 || dim > 16 || (offset >> 62) != 0) return -1;
  const long total = (long)B * T, threads = total * ((dim + 3) / 4);
  const long blocks = (threads + 255) / 256;
  if (blocks > 0x7fffffffL) return -1;
  if (total == 0) return 0;
  hipLaunchKernelGGL(k_normal_noise, dim3((unsigned)blocks), dim3(256), 0, st, NoiseGen{seed, offset, 1, 0u}, T, dim, total, out);
  return 0;
}

int launch_sine_source(const float* f0, int B, int L, int upp, double sr, const float* rand_ini, const float* noise,
                       const float* weight, const float* bias, int dim, float sine_amp, float noise_std,
                       float voiced_threshold, float* rad_acc, float* out, hipStream_t st, const NoiseGen* gen) {
  if (dim != 9 && dim != 1) return -1;
  const long total = (long)B * L * upp;
  const long T = (long)L * upp;
  if (T >= (1L << 31) || B > 65535) return -1;
  if (total == 0) return 0;
  const dim3 grid((unsigned)((T + 255) / 256), (unsigned)B);
  const bool draw = gen && gen->on;
  if (draw && ((long)L * upp >= (1L << 32) || (gen->offset >> 62) != 0)) return -1;
  if (!draw && !noise) return -1;
  NoiseGen rng{0ull, 0ull, 0, 0u};
  if (draw) rng = *gen;
  hipLaunchKernelGGL(k_sinegen_scan, dim3((unsigned)B), dim3(256), 0, st, f0, L, upp, (float)sr, rad_acc);
#define DDSP_SINEGEN(D, R)                                                                                                     \
  hipLaunchKernelGGL((k_sinegen<D, R>), grid, dim3(256), 0, st, f0, rad_acc, rand_ini, noise, weight, bias, L, \
                     upp, (float)sr, sine_amp, noise_std, voiced_threshold, total, out, rng)
  if (dim == 9) { if (draw) DDSP_SINEGEN(9, true); else DDSP_SINEGEN(9, false); }
  else { if (draw) DDSP_SINEGEN(1, true); else DDSP_SINEGEN(1, false); }
#undef DDSP_SINEGEN
  return 0;
}

}  // namespace ddsp
