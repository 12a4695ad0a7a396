// Spectral loss of the training loop: ddsp/loss.py:9-32, SSSLoss.forward, behind the STFT.
//
//   S = |X| / ||window||_2 + eps                                  (torchaudio Spectrogram(power=1, normalized=True), :20,23-24)
//   converge = mean_b ||S_true - S_pred||_F(b) / ||S_true + S_pred||_F(b)                        (:26)
//   log_term = mean |log S_true - log S_pred|                                                     (:28)
//   loss     = converge + alpha log_term                                                          (:30)
//
// The transform sizes of RSSLoss are arbitrary integers in [fft_min, fft_max) (:47), so the STFT itself stays with the
// host's FFT library; everything after it -- two magnitudes, three reductions, and in the backward pass the whole
// chain down to the gradient of the complex spectrum -- is one pass over the two spectra here, where the eager
// composition runs some ten elementwise / reduction kernels over [B, bins, frames] temporaries.
//   k_sss_partial : per (utterance, chunk) float64 partial sums of (St-Sp)^2, (St+Sp)^2, |log St - log Sp|
//   k_sss_final   : fixed-order sums of the partials -> per-utterance norms and the scalar loss (no atomics: a launch
//                   geometry is bit-reproducible)
//   k_sss_grad    : d loss / d X (complex, PyTorch's convention dRe + i dIm) for the predicted or the true spectrum
// HBM-bound: 16 B read per complex bin pair forward, 16 B read + 8 B written backward.
#include "ddsp_common.h"
#include "kernels.h"

namespace ddsp {

constexpr int SL_THREADS = 256;

// Hardware square root and logarithm (1 ulp; the spectra are far from the denormal range the library versions guard):
// with the library functions the pass is VALU-bound at 1.3 TB/s.
__device__ __forceinline__ float sl_mag(float2 z) { return __builtin_amdgcn_sqrtf(fmaf(z.x, z.x, z.y * z.y)); }
__device__ __forceinline__ float sl_log(float s) { return 0.6931471805599453f * __builtin_amdgcn_logf(s); }

__global__ void __launch_bounds__(SL_THREADS) k_sss_partial(const float2* __restrict__ xt, const float2* __restrict__ xp,
                                                            long per_utt, int chunks, float inv_wn, float eps,
                                                            double* __restrict__ partial) {
  __shared__ double red[3][SL_THREADS / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = blockIdx.x, b = blockIdx.y;
  const long span = (per_utt + chunks - 1) / chunks;
  const long lo = (long)c * span;
  long hi = lo + span;
  if (hi > per_utt) hi = per_utt;
  const float2* t = xt + (long)b * per_utt;
  const float2* p = xp + (long)b * per_utt;
  double d2 = 0.0, s2 = 0.0, l1 = 0.0;
  for (long base = lo; base < hi; base += 4 * SL_THREADS) {
    float a[4], q[4];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {                              // loads first, arithmetic after
      const long i = base + u * SL_THREADS + tid;
      ok[u] = i < hi;
      const long j = ok[u] ? i : lo;
      a[u] = sl_mag(t[j]);
      q[u] = sl_mag(p[j]);
    }
    float fd = 0.f, fs = 0.f, fl = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float st = fmaf(a[u], inv_wn, eps), sp = fmaf(q[u], inv_wn, eps);
      const float d = st - sp, s = st + sp;
      const float l = fabsf(sl_log(st) - sl_log(sp));
      fd += ok[u] ? d * d : 0.f;
      fs += ok[u] ? s * s : 0.f;
      fl += ok[u] ? l : 0.f;
    }
    d2 += (double)fd; s2 += (double)fs; l1 += (double)fl;
  }
  d2 = wave_sum(d2); s2 = wave_sum(s2); l1 = wave_sum(l1);
  if (lane == 0) { red[0][wave] = d2; red[1][wave] = s2; red[2][wave] = l1; }
  __syncthreads();
  if (tid < 3) {
    double v = 0.0;
    for (int w = 0; w < SL_THREADS / 64; ++w) v += red[tid][w];
    partial[((long)b * chunks + c) * 3 + tid] = v;
  }
}

__global__ void __launch_bounds__(SL_THREADS) k_sss_final(const double* __restrict__ partial, int B, int chunks,
                                                          double inv_B, double inv_n, float alpha,
                                                          float* __restrict__ norms, float* __restrict__ loss) {
  __shared__ double conv[SL_THREADS], logs[SL_THREADS];
  const int tid = threadIdx.x;
  double cv = 0.0, lg = 0.0;
  for (int b = tid; b < B; b += SL_THREADS) {
    double d2 = 0.0, s2 = 0.0, l1 = 0.0;
    for (int c = 0; c < chunks; ++c) {
      const double* q = partial + ((long)b * chunks + c) * 3;
      d2 += q[0]; s2 += q[1]; l1 += q[2];
    }
    const float nd = (float)sqrt(d2), ns = (float)sqrt(s2);    // torch.linalg.norm results are float32
    norms[2 * b] = nd;
    norms[2 * b + 1] = ns;
    cv += (double)(nd / ns);
    lg += l1;
  }
  conv[tid] = cv; logs[tid] = lg;
  __syncthreads();
  for (int d = SL_THREADS / 2; d > 0; d >>= 1) {
    if (tid < d) { conv[tid] += conv[tid + d]; logs[tid] += logs[tid + d]; }
    __syncthreads();
  }
  if (tid == 0) *loss = (float)(conv[0] * inv_B) + alpha * (float)(logs[0] * inv_n);
}

// gradient with respect to the complex spectrum of the predicted (WRT_TRUE = 0) or the true (1) signal
template <int WRT_TRUE>
__global__ void __launch_bounds__(SL_THREADS) k_sss_grad(const float2* __restrict__ xt, const float2* __restrict__ xp,
                                                         long per_utt, const float* __restrict__ norms, float inv_wn,
                                                         float eps, float alpha, float inv_B, float inv_n,
                                                         const float* __restrict__ grad_out, float2* __restrict__ dx) {
  const int b = blockIdx.y;
  const float go = grad_out[0];
  const float nd = norms[2 * b], ns = norms[2 * b + 1];
  const float k1 = nd > 0.f ? inv_B / (nd * ns) : 0.f;         // d ||d|| = d / ||d||, taken as 0 at the origin
  const float k2 = inv_B * nd / (ns * ns * ns);
  const float kl = alpha * inv_n;
  const float2* t = xt + (long)b * per_utt;
  const float2* p = xp + (long)b * per_utt;
  float2* o = dx + (long)b * per_utt;
  const long stride = (long)gridDim.x * SL_THREADS;
  for (long i = (long)blockIdx.x * SL_THREADS + threadIdx.x; i < per_utt; i += stride) {
    const float2 zt = t[i], zp = p[i];
    const float at = sl_mag(zt), ap = sl_mag(zp);
    const float st = fmaf(at, inv_wn, eps), sp = fmaf(ap, inv_wn, eps);
    const float d = st - sp, s = st + sp;
    const float l = sl_log(st) - sl_log(sp);
    const float sg = l > 0.f ? 1.f : (l < 0.f ? -1.f : 0.f);
    float g;
    float2 z;
    float a;
    if (WRT_TRUE) { g = k1 * d - k2 * s + kl * sg / st; z = zt; a = at; }
    else          { g = -k1 * d - k2 * s - kl * sg / sp; z = zp; a = ap; }
    const float r = a > 0.f ? go * g * inv_wn / a : 0.f;       // d|z| = z / |z|, 0 at the origin (as autograd)
    o[i] = float2{r * z.x, r * z.y};
  }
}

int sss_chunks(int B, long per_utt) {
  long c = (per_utt + 8191) / 8192;
  const long want = (2048 + (B > 0 ? B : 1) - 1) / (B > 0 ? B : 1);      // ~2 k workgroups on the chip, >= 8 k bins each
  if (c > want) c = want;
  if (c < 1) c = 1;
  return (int)c;
}

size_t sss_scratch_bytes(int B, long per_utt) { return (size_t)B * sss_chunks(B, per_utt) * 3 * sizeof(double); }

void launch_sss_final(const double* scratch, int B, int chunks, long per_utt, float alpha, float* norms, float* loss,
                      hipStream_t st) {
  hipLaunchKernelGGL(k_sss_final, dim3(1), dim3(SL_THREADS), 0, st, scratch, B, chunks, 1.0 / (double)B,
                     1.0 / ((double)B * (double)per_utt), alpha, norms, loss);
}

int launch_sss_loss(const float* xt, const float* xp, int B, long per_utt, float inv_wn, float eps, float alpha,
                    double* scratch, float* norms, float* loss, hipStream_t st) {
  if (B < 1 || B > 65535 || per_utt < 1) return -1;
  const int chunks = sss_chunks(B, per_utt);
  hipLaunchKernelGGL(k_sss_partial, dim3((unsigned)chunks, (unsigned)B), dim3(SL_THREADS), 0, st,
                     reinterpret_cast<const float2*>(xt), reinterpret_cast<const float2*>(xp), per_utt, chunks, inv_wn,
                     eps, scratch);
  launch_sss_final(scratch, B, chunks, per_utt, alpha, norms, loss, st);
  return 0;
}

int launch_sss_loss_bwd(const float* xt, const float* xp, int B, long per_utt, const float* norms, float inv_wn,
                        float eps, float alpha, const float* grad_out, int wrt_true, float* dx, hipStream_t st) {
  if (B < 1 || B > 65535 || per_utt < 1) return -1;
  long gx = (per_utt + 4 * SL_THREADS - 1) / (4 * SL_THREADS);
  if (gx > 1024) gx = 1024;
  const float inv_B = 1.0f / (float)B;
  const float inv_n = (float)(1.0 / ((double)B * (double)per_utt));
  const dim3 grid((unsigned)gx, (unsigned)B);
  if (wrt_true)
    hipLaunchKernelGGL(k_sss_grad<1>, grid, dim3(SL_THREADS), 0, st, reinterpret_cast<const float2*>(xt),
                       reinterpret_cast<const float2*>(xp), per_utt, norms, inv_wn, eps, alpha, inv_B, inv_n, grad_out,
                       reinterpret_cast<float2*>(dx));
  else
    hipLaunchKernelGGL(k_sss_grad<0>, grid, dim3(SL_THREADS), 0, st, reinterpret_cast<const float2*>(xt),
                       reinterpret_cast<const float2*>(xp), per_utt, norms, inv_wn, eps, alpha, inv_B, inv_n, grad_out,
                       reinterpret_cast<float2*>(dx));
  return 0;
}

}  // namespace ddsp
