// Spectral loss of the training loop straight from the two waveforms (SURVEY.md 8-f #3b): ddsp/loss.py:9-32 with the
// STFT of torchaudio's Spectrogram(n_fft, hop, power=1, normalized=True, center=False) (:20) inside the kernel.
//
// RSSLoss draws ARBITRARY transform sizes n in [fft_min, fft_max) (:47) -- most of them with large prime factors, where
// a library FFT falls back to Bluestein's algorithm over several passes through HBM.  Here the whole chain of one frame
// stays in one workgroup: the DFT of size n is the chirp-z identity
//     X[k] = c[k] sum_j (x[j] w[j] c[j]) conj(c)[k - j],     c[j] = exp(-i pi j^2 / n),
// a circular convolution of size N = 1024 / 2048 / 4096 >= 2n - 1 on the plans of fft_r.h (two transforms: the chirp
// filter's spectrum is a table).  The two REAL signals of the loss ride in ONE complex transform, z = x_true + i x_pred,
// and come apart through Z[k] and conj Z[n - k]; magnitudes, the three reductions of loss.hip and the stores of the two
// spectra (the backward pass reads them) follow in registers.
//   k_czt_tables  : c, w c and the filter spectrum for one n (float64 sums, once per size and device)
//   k_sss_czt     : frames (any hop) -> spectra + per (utterance, chunk) float64 partial sums, two frames per pass in
//                   lockstep; k_sss_final of loss.hip finishes
//   k_sss_czt_bwd : the gradient of the spectrum (k_sss_grad's formula) of TWO frames, Hermitian-extended, as one inverse
//                   chirp-z transform, times the window: d loss / d x when hop == n, else the frames' gradients in a
//                   workspace, which
//   k_frames_overlap_add gathers per sample in ascending frame order
// Bound: arithmetic (per frame pair 2 transforms of N points); HBM sees 8 B per sample pair in and 16 B per bin pair out.
// Measured and what was tried: DESIGN.md 7.1, EXPERIMENTS.md 3.7.
#include "ddsp_common.h"
#include "fft_r.h"
#include "kernels.h"
#include "occupancy.h"
#include "tuning.h"

namespace ddsp {
using fft::cconj;
using fft::cmul;

constexpr int CZ_TAB_THREADS = 256;

int czt_plan(int n) {
  if (n < 2) return 0;
  if (2 * n - 1 <= 1024) return 2;
  if (2 * n - 1 <= 2048) return 4;
  if (2 * n - 1 <= 4096) return 8;
  return 0;
}

size_t czt_table_bytes(int n) {
  const int R = czt_plan(n);
  return R ? (size_t)(2 * n + 512 * R) * sizeof(float2) : 0;
}

// tab: c[n] | w c[n] | bhat[N].  Blocks [0, N): one filter-spectrum bin each,
//     bhat[q] = sum_{|m| < n} exp(+i pi m^2 / n) exp(-2 pi i q m / N)
//             = 1 + 2 sum_{m = 1}^{n - 1} exp(i pi m^2 / n) cos(2 pi q m / N);
// the blocks behind them fill the chirps.  Phases are reduced in integers, so the float64 arguments are exact.
__global__ void __launch_bounds__(CZ_TAB_THREADS) k_czt_tables(int n, int N, float2* __restrict__ tab) {
  __shared__ double red[2][CZ_TAB_THREADS / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if ((int)blockIdx.x >= N) {
    const int j = ((int)blockIdx.x - N) * CZ_TAB_THREADS + tid;
    if (j < n) {
      double s, c;
      sincospi((double)(((long)j * j) % (2 * n)) / (double)n, &s, &c);
      const double w = 0.5 - 0.5 * cospi(2.0 * (double)j / (double)n);            // periodic Hann, torch.hann_window
      tab[j] = float2{(float)c, (float)-s};
      tab[n + j] = float2{(float)(w * c), (float)(-w * s)};
    }
    return;
  }
  const int q = blockIdx.x;
  double re = 0.0, im = 0.0;
  for (int m = 1 + tid; m < n; m += CZ_TAB_THREADS) {
    double s, c;
    sincospi((double)(((long)m * m) % (2 * n)) / (double)n, &s, &c);
    const double cq = cospi(2.0 * (double)(((long)q * m) % N) / (double)N);
    re += c * cq;
    im += s * cq;
  }
  re = wave_sum(re);
  im = wave_sum(im);
  if (lane == 0) { red[0][wave] = re; red[1][wave] = im; }
  __syncthreads();
  if (tid == 0) {
    double a = 0.0, b = 0.0;
    for (int w = 0; w < CZ_TAB_THREADS / 64; ++w) { a += red[0][w]; b += red[1][w]; }
    tab[2 * n + q] = float2{(float)(1.0 + 2.0 * a), (float)(2.0 * b)};
  }
}

__device__ __forceinline__ float cz_mag(f32x2 z) { return __builtin_amdgcn_sqrtf(fmaf(z.x, z.x, z.y * z.y)); }
__device__ __forceinline__ float cz_log(float s) { return 0.6931471805599453f * __builtin_amdgcn_logf(s); }

// One circular convolution with the chirp filter: v[0..3] (natural order, slot m <-> index P m + tid; the upper half of the
// input is zero padding) -> conj of the result in v[0..7], N times too large.  post[P m + tid] -- the table the caller
// multiplies the result with -- is fetched between the two transforms, so its latency hides behind the second one (reads
// past a table's n entries stay inside tab and are unused).
template <int R, class Between>
__device__ __forceinline__ void czt_convolve(f32x2 (&v)[8], const typename fft::Plan<R>::Tw& tw, const f32x2* __restrict__ bh,
                                             const f32x2* __restrict__ post, f32x2 (&pv)[4], f32x2* A, f32x2* B, int tid,
                                             Between&& between) {
  using PL = fft::Plan<R>;
  f32x2 g[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) g[m] = bh[PL::P * m + tid];
  PL::template forward<true>(v, tw, A, B, tid);              // n <= N / 2: the input fills the slots 0 .. 3 only
#pragma unroll
  for (int m = 0; m < 8; ++m) v[m] = cconj(cmul(v[m], g[m]));
#pragma unroll
  for (int m = 0; m < 4; ++m) pv[m] = post[PL::P * m + tid];   // the result is wanted below n <= N / 2 only
  between();                                                 // more loads of the caller's that the second transform hides
  __syncthreads();                                           // pass 4 of the first transform still reads A
  PL::forward(v, tw, A, B, tid);
}

// The same for two inputs in lockstep (Plan::forward2): v through A, B and u through A2, B2; one copy of the tables.
template <int R, class Between>
__device__ __forceinline__ void czt_convolve2(f32x2 (&v)[8], f32x2 (&u)[8], const typename fft::Plan<R>::Tw& tw,
                                              const f32x2* __restrict__ bh, const f32x2* __restrict__ post, f32x2 (&pv)[4],
                                              f32x2* A, f32x2* B, f32x2* A2, f32x2* B2, int tid, Between&& between) {
  using PL = fft::Plan<R>;
  f32x2 g[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) g[m] = bh[PL::P * m + tid];
  PL::template forward2<true>(v, u, tw, A, B, A2, B2, tid);
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    v[m] = cconj(cmul(v[m], g[m]));
    u[m] = cconj(cmul(u[m], g[m]));
  }
#pragma unroll
  for (int m = 0; m < 4; ++m) pv[m] = post[PL::P * m + tid];
  between();
  __syncthreads();
  PL::forward2(v, u, tw, A, B, A2, B2, tid);
}

// Two frames per pass, in lockstep.  A chunk with an odd number of frames runs its last one twice (the second copy is
// neither stored nor counted).
template <int R>
__global__ void __launch_bounds__(64 * R, 2) k_sss_czt(const float* __restrict__ xt, const float* __restrict__ xp, long ld,
                                                       int n, int hop, int frames, int chunks, int span,
                                                       const float2* __restrict__ tab, float inv_wn, float eps,
                                                       float2* __restrict__ spec_t, float2* __restrict__ spec_p,
                                                       double* __restrict__ partial) {
  using PL = fft::Plan<R>;
  constexpr int N = PL::N, P = PL::P;
  __shared__ __attribute__((aligned(16))) f32x2 ex[4][N];
  __shared__ double red[3][R];
  // per [pass parity][frame of the pair]: flags {true, pred: a nonzero sample behind the window; the two frames differ} and
  // the float bits of the largest windowed |sample| of {true, pred}
  __shared__ int live[2][2][3];
  __shared__ unsigned peak[2][2][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = blockIdx.x, b = blockIdx.y;
  const int bins = n / 2 + 1;
  const int f_lo = c * span;
  const int f_hi = f_lo + span < frames ? f_lo + span : frames;
  const f32x2* ch = reinterpret_cast<const f32x2*>(tab);
  const f32x2* wc = ch + n;
  const f32x2* bh = ch + 2 * n;
  typename PL::Tw tw;
  tw.init(tid);
  const float scale = 0.5f / (float)N;                       // the inverse transform's 1/N and the 1/2 of the split
  double d2 = 0.0, s2 = 0.0, l1 = 0.0;
  if (tid < 12) (&live[0][0][0])[tid] = 0;
  if (tid < 8) (&peak[0][0][0])[tid] = 0u;
  f32x2 wcr[4];                                              // window times chirp of this thread's four samples
#pragma unroll
  for (int n1 = 0; n1 < 4; ++n1) wcr[n1] = P * n1 + tid < n ? wc[P * n1 + tid] : f32x2{0.f, 0.f};
  float na[2][4], nq[2][4];                                  // the NEXT pair's samples: in flight during this pair's transforms
  auto fetch = [&](int f) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int ff = f + h < f_hi ? f + h : f_hi - 1;
      const float* pt = xt + (long)b * ld + (long)ff * hop;
      const float* pp = xp + (long)b * ld + (long)ff * hop;
#pragma unroll
      for (int n1 = 0; n1 < 4; ++n1) {
        const int j = P * n1 + tid;
        const int jj = j < n ? j : 0;
        na[h][n1] = pt[jj];
        nq[h][n1] = pp[jj];
      }
    }
  };
  // The two signals share one transform, so each spectrum carries the other's rounding noise -- 1e-7 of the LARGER one,
  // where separate transforms (the reference) err by 1e-7 of each signal's own size: a quiet prediction against a loud
  // target would lose digits exactly where the loss weighs 1 / S_pred.  So the prediction enters the transform times a
  // power of two that brings its peak to the target's (exact both ways), found per frame one pass ahead: the next pair's
  // samples are in registers long before they are used.
  auto note_peaks = [&](int par_next) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float mt = 0.f, mq = 0.f;
#pragma unroll
      for (int n1 = 0; n1 < 4; ++n1) {                       // peaks BEHIND the window (wcr = 0 behind the frame's end)
        const float wabs = fabsf(wcr[n1].x) + fabsf(wcr[n1].y);
        mt = fmaxf(mt, fabsf(na[h][n1]) * wabs);
        mq = fmaxf(mq, fabsf(nq[h][n1]) * wabs);
      }
      const unsigned ut = wave_max_dpp(__float_as_uint(mt)), uq = wave_max_dpp(__float_as_uint(mq));
      if (lane == 0) {
        atomicMax(&peak[par_next][h][0], ut);
        atomicMax(&peak[par_next][h][1], uq);
      }
    }
  };
  __syncthreads();                                           // the zeroed flags and peaks
  if (f_lo < f_hi) {
    fetch(f_lo);
    note_peaks(0);
  }
  __syncthreads();
  for (int f = f_lo, par = 0; f < f_hi; f += 2, par ^= 1) {
    f32x2 v[2][8], co[4];
    float up[2];                                             // 2^d, d = exponent of the target's peak - the prediction's
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const unsigned pt = peak[par][h][0], pq = peak[par][h][1];
      int d = (int)((pt >> 23) & 0xff) - (int)((pq >> 23) & 0xff);
      d = (pt == 0u || pq == 0u) ? 0 : (d > 60 ? 60 : (d < -60 ? -60 : d));
      up[h] = __uint_as_float((unsigned)(127 + d) << 23);
    }
    if (tid < 4) (&peak[par ^ 1][0][0])[tid] = 0u;           // the next pass's peaks; their writers are barriers away
    // The two signals share one transform, so each spectrum carries the other's rounding noise (1e-7 of ITS size).  A
    // frame that is all zero behind the window -- digital silence -- must come out as exact zeros (S = eps, and a zero
    // gradient at the origin, as the separate transforms of the reference give): such frames are flagged here and zeroed
    // behind the split.  Likewise two EQUAL frames must give equal spectra (loss 0 for identical signals), which the two
    // halves of the split do not.
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      bool any_t = false, any_p = false, differ = false;
#pragma unroll
      for (int n1 = 0; n1 < 4; ++n1) {
        const float a = na[h][n1], q = nq[h][n1], qs = q * up[h];
        const f32x2 w = wcr[n1];                             // 0 behind the frame's end
        v[h][n1] = f32x2{a * w.x - qs * w.y, a * w.y + qs * w.x};
        any_t |= a * w.x != 0.f || a * w.y != 0.f;           // the WINDOWED sample: hann[0] = 0
        any_p |= q * w.x != 0.f || q * w.y != 0.f;
        differ |= P * n1 + tid < n && a != q;
      }
      if (any_t) live[par][h][0] = 1;
      if (any_p) live[par][h][1] = 1;
      if (differ) live[par][h][2] = 1;
    }
    if (tid < 6) (&live[par ^ 1][0][0])[tid] = 0;            // next pass's flags; its writers are barriers away
    if (f + 2 < f_hi) fetch(f + 2);
    czt_convolve2<R>(v[0], v[1], tw, bh, ch, co, ex[0], ex[1], ex[2], ex[3], tid, [] {});
    if (f + 2 < f_hi) note_peaks(par ^ 1);                   // before the barrier below; read at the top of the next pass
    f32x2 z[2][3];
    float keep_t[2], keep_p[2];
    bool same[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {                            // read BEFORE the barrier below: the next pass resets these flags
      keep_t[h] = live[par][h][0] ? 1.f : 0.f;
      // ... and the prediction's spectrum back to its scale: 2^-d from 2^d's bits (one register less across the transforms)
      keep_p[h] = live[par][h][1] ? __uint_as_float(0x7f000000u - __float_as_uint(up[h])) : 0.f;
      same[h] = !live[par][h][2];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int k = P * m + tid;
        if (k < n) {
          const f32x2 zz = cmul(cconj(v[h][m]), co[m]) * scale;      // Z[k] / 2
          ex[1 + 2 * h][k] = zz;
          if (m < 3) z[h][m] = zz;
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (f + h >= f_hi) break;
      float fd = 0.f, fs = 0.f, fl = 0.f;
      const long row = ((long)b * frames + f + h) * bins;
#pragma unroll
      for (int m = 0; m < 3; ++m) {                          // n <= N / 2, so the bins 0 .. n/2 lie in the slots 0 .. 2
        const int k = P * m + tid;
        if (k < bins) {
          const f32x2 zk = z[h][m], zm = ex[1 + 2 * h][k == 0 ? 0 : n - k];
          const f32x2 Xt = f32x2{zk.x + zm.x, zk.y - zm.y} * keep_t[h];                     // (Z[k] + conj Z[n-k]) / 2
          const f32x2 Xp = same[h] ? Xt : f32x2{zk.y + zm.y, zm.x - zk.x} * keep_p[h];      // (Z[k] - conj Z[n-k]) / 2i
          spec_t[row + k] = float2{Xt.x, Xt.y};
          spec_p[row + k] = float2{Xp.x, Xp.y};
          const float st = fmaf(cz_mag(Xt), inv_wn, eps), sp = fmaf(cz_mag(Xp), inv_wn, eps);
          const float d = st - sp, s = st + sp;
          fd += d * d;
          fs += s * s;
          fl += fabsf(cz_log(st) - cz_log(sp));
        }
      }
      d2 += (double)fd; s2 += (double)fs; l1 += (double)fl;
    }
  }
  d2 = wave_sum(d2); s2 = wave_sum(s2); l1 = wave_sum(l1);
  if (lane == 0) { red[0][wave] = d2; red[1][wave] = s2; red[2][wave] = l1; }
  __syncthreads();
  if (tid < 3) {
    double v = 0.0;
    for (int w = 0; w < R; ++w) v += red[tid][w];
    partial[((long)b * chunks + c) * 3 + tid] = v;
  }
}

// The gradient of the loss with respect to one complex bin (k_sss_grad of loss.hip).
template <int WRT_TRUE>
__device__ __forceinline__ f32x2 cz_bin_grad(float2 zt, float2 zp, float k1, float k2, float kl, float go, float inv_wn,
                                             float eps) {
  const f32x2 a{zt.x, zt.y}, p{zp.x, zp.y};
  const float at = cz_mag(a), ap = cz_mag(p);
  const float st = fmaf(at, inv_wn, eps), sp = fmaf(ap, inv_wn, eps);
  const float d = st - sp, s = st + sp;
  const float l = cz_log(st) - cz_log(sp);
  const float sg = l > 0.f ? 1.f : (l < 0.f ? -1.f : 0.f);
  float g, m;
  f32x2 z;
  if (WRT_TRUE) { g = k1 * d - k2 * s + kl * sg / st; z = a; m = at; }
  else          { g = -k1 * d - k2 * s - kl * sg / sp; z = p; m = ap; }
  const float r = m > 0.f ? go * g * inv_wn / m : 0.f;
  return z * r;
}

template <int R, int WRT_TRUE>
__global__ void __launch_bounds__(64 * R, 2) k_sss_czt_bwd(const float2* __restrict__ spec_t, const float2* __restrict__ spec_p,
                                                        int n, int frames, int chunks, const float2* __restrict__ tab,
                                                        const float* __restrict__ norms, float inv_wn, float eps,
                                                        float alpha, float inv_B, float inv_n,
                                                        const float* __restrict__ grad_out, float* __restrict__ dx,
                                                        long ld_dx, int T, int accumulate, int knob_turns) {
  using PL = fft::Plan<R>;
  constexpr int N = PL::N, P = PL::P;
  __shared__ __attribute__((aligned(16))) f32x2 ex[2][N];
  const int tid = threadIdx.x;
  const int c = blockIdx.x, b = blockIdx.y;
  const int bins = n / 2 + 1;
  const int pairs = (frames + 1) / 2;
  const int span = (pairs + chunks - 1) / chunks;
  const int p_lo = c * span;
  const int p_hi = p_lo + span < pairs ? p_lo + span : pairs;
  const f32x2* ch = reinterpret_cast<const f32x2*>(tab);
  const f32x2* wc = ch + n;
  const f32x2* bh = ch + 2 * n;
  typename PL::Tw tw;
  tw.init(tid);
  const float go = grad_out[0];
  const float nd = norms[2 * b], ns = norms[2 * b + 1];
  const float k1 = nd > 0.f ? inv_B / (nd * ns) : 0.f;
  const float k2 = inv_B * nd / (ns * ns * ns);
  const float kl = alpha * inv_n;
  const float scale = 1.0f / (float)N;
  float* o = dx + (long)b * ld_dx;
  // One pair of frames per pass.  The NEXT pair's bins are fetched between the two transforms of this pair and turned into
  // its input (gradient of the bins, Hermitian extension, chirp) behind them, so neither the loads' latency nor 64 registers
  // of raw bins sit on the transforms.
  float2 rt0[4], rp0[4], rt1[4], rp1[4];
  auto fetch = [&](int pi) {
    const int f0 = 2 * pi;
    const float2* t0 = spec_t + ((long)b * frames + f0) * bins;
    const float2* q0 = spec_p + ((long)b * frames + f0) * bins;
    const long next = f0 + 1 < frames ? bins : 0;
#pragma unroll
    for (int n1 = 0; n1 < 4; ++n1) {
      const int k = P * n1 + tid;
      const int kk = k >= n ? 0 : (k < bins ? k : n - k);
      rt0[n1] = t0[kk]; rp0[n1] = q0[kk];
      rt1[n1] = t0[next + kk]; rp1[n1] = q0[next + kk];
    }
  };
  f32x2 chr[4];                                              // chirp of this thread's four bins
#pragma unroll
  for (int n1 = 0; n1 < 4; ++n1) chr[n1] = P * n1 + tid < n ? ch[P * n1 + tid] : f32x2{0.f, 0.f};
  f32x2 vn[4];
  auto prepare = [&](int pi) {
    const bool two = 2 * pi + 1 < frames;
#pragma unroll
    for (int n1 = 0; n1 < 4; ++n1) {
      const int k = P * n1 + tid;
      const int kk = k >= n ? 0 : (k < bins ? k : n - k);
      const f32x2 g0 = cz_bin_grad<WRT_TRUE>(rt0[n1], rp0[n1], k1, k2, kl, go, inv_wn, eps);
      f32x2 g1 = cz_bin_grad<WRT_TRUE>(rt1[n1], rp1[n1], k1, k2, kl, go, inv_wn, eps);
      if (!two) g1 = f32x2{0.f, 0.f};
      // Hermitian extension: the real part of the one-sided sum as a full inverse transform
      const bool edge = kk == 0 || 2 * kk == n;
      const float hr = edge ? 1.f : 0.5f;
      const float hi = edge ? 0.f : (k < bins ? 0.5f : -0.5f);
      const f32x2 a0{g0.x * hr, g0.y * hi}, a1{g1.x * hr, g1.y * hi};
      const f32x2 gc{a0.x - a1.y, a0.y + a1.x};              // frame f0 + i frame f0+1
      vn[n1] = cmul(cconj(gc), chr[n1]);                     // chr = 0 behind the last bin
    }
  };
  if (p_lo < p_hi) fetch(p_lo);
  const int turn = __builtin_amdgcn_s_getreg(0x1804) & 1;      // as the forward kernel
  for (int pi = p_lo; pi < p_hi; ++pi) {
    if (knob_turns && ((pi + turn) & 1)) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
    const int f0 = 2 * pi;
    const bool two = f0 + 1 < frames;
    prepare(pi);                                             // ONE call site (the bins were fetched between the previous pair's transforms): code that exists once is fetched once
    f32x2 v[8], co[4];
#pragma unroll
    for (int n1 = 0; n1 < 4; ++n1) v[n1] = vn[n1];
    float old0[4], old1[4];
    czt_convolve<R>(v, tw, bh, wc, co, ex[0], ex[1], tid, [&] {
      if (pi + 1 < p_hi) fetch(pi + 1);
      if (accumulate) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int j = P * m + tid;
          const int jj = j < n ? j : 0;
          old0[m] = o[(long)f0 * n + jj];
          old1[m] = o[(long)(two ? f0 + 1 : f0) * n + jj];
        }
      }
    });
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int j = P * m + tid;
      if (j < n) {
        const f32x2 dj = cmul(cconj(v[m]), co[m]) * scale;   // w[j] D[j]; the inverse transform is its conjugate
        o[(long)f0 * n + j] = accumulate ? old0[m] + dj.x : dj.x;
        if (two) o[(long)(f0 + 1) * n + j] = accumulate ? old1[m] - dj.y : -dj.y;
      }
    }
    __syncthreads();                                         // pass 4 of the last transform still reads ex[0]
  }
  if (c == chunks - 1 && !accumulate)                        // samples behind the last whole frame do not reach the loss
    for (long i = (long)frames * n + tid; i < T; i += P) o[i] = 0.f;
}

// Launch geometry: one round (knob CZT_ROUNDS: more) of what the chip holds (asked of the runtime once per kernel: it depends
// on the registers the compiler used), the frames of an utterance in equal spans -- of an even number of frames, both kernels
// take them in pairs.  Measured (profiles/r03_v25_loss_rounds.txt): a workgroup's start -- 24 sincospi for the twiddles,
// table loads, the first fetch -- is worth about one frame, so 2 / 4 / 8 rounds cost +3 / +12 / +19 % of the step.
struct WaveGeom { int chunks, span; };

static int czt_resident_of(int R, bool backward) {
  static ResidentCache cache[2][3];
  ResidentCache& c = cache[backward ? 1 : 0][R == 2 ? 0 : (R == 4 ? 1 : 2)];
  if (!backward) {
    if (R == 2) return resident_workgroups(k_sss_czt<2>, 128, c);
    if (R == 4) return resident_workgroups(k_sss_czt<4>, 256, c);
    return resident_workgroups(k_sss_czt<8>, 512, c);
  }
  if (R == 2) return resident_workgroups(k_sss_czt_bwd<2, 0>, 128, c);
  if (R == 4) return resident_workgroups(k_sss_czt_bwd<4, 0>, 256, c);
  return resident_workgroups(k_sss_czt_bwd<8, 0>, 512, c);
}

static WaveGeom sss_wave_geom(int B, int n, int frames, bool backward) {
  const int pairs = (frames + 1) / 2;
  const long rounds = knob(KNOB_CZT_ROUNDS) > 0 ? knob(KNOB_CZT_ROUNDS) : 1;
  int want = (int)((czt_resident_of(czt_plan(n), backward) * rounds + B - 1) / B);     // chunks per utterance
  if (want > pairs) want = pairs;
  if (want < 1) want = 1;
  const int span = 2 * ((pairs + want - 1) / want);
  return WaveGeom{(frames + span - 1) / span, span};
}

size_t sss_wave_scratch_bytes(int B, int n, int frames) {
  return (size_t)B * sss_wave_geom(B, n, frames, false).chunks * 3 * sizeof(double);
}

int launch_czt_tables(int n, float* tab, hipStream_t st) {
  const int R = czt_plan(n);
  if (!R) return -1;
  const int N = 512 * R;
  hipLaunchKernelGGL(k_czt_tables, dim3((unsigned)(N + (n + CZ_TAB_THREADS - 1) / CZ_TAB_THREADS)), dim3(CZ_TAB_THREADS),
                     0, st, n, N, reinterpret_cast<float2*>(tab));
  return 0;
}

int launch_sss_wave(const float* xt, const float* xp, int B, long ld, int n, int hop, int frames, const float* tab,
                    float inv_wn, float eps, float alpha, double* scratch, float* spec_t, float* spec_p, float* norms,
                    float* loss, hipStream_t st) {
  const int R = czt_plan(n);
  if (!R || B < 1 || B > 65535 || frames < 1 || hop < 1) return -1;
  const WaveGeom geo = sss_wave_geom(B, n, frames, false);
  const int chunks = geo.chunks, span = geo.span;
  const dim3 grid((unsigned)chunks, (unsigned)B);
  const float2* tb = reinterpret_cast<const float2*>(tab);
  float2* s_t = reinterpret_cast<float2*>(spec_t);
  float2* s_p = reinterpret_cast<float2*>(spec_p);
  if (R == 2)
    hipLaunchKernelGGL(k_sss_czt<2>, grid, dim3(128), 0, st, xt, xp, ld, n, hop, frames, chunks, span, tb, inv_wn, eps, s_t, s_p,
                       scratch);
  else if (R == 4)
    hipLaunchKernelGGL(k_sss_czt<4>, grid, dim3(256), 0, st, xt, xp, ld, n, hop, frames, chunks, span, tb, inv_wn, eps, s_t, s_p,
                       scratch);
  else
    hipLaunchKernelGGL(k_sss_czt<8>, grid, dim3(512), 0, st, xt, xp, ld, n, hop, frames, chunks, span, tb, inv_wn, eps, s_t, s_p,
                       scratch);
  const long per_utt = (long)frames * (n / 2 + 1);
  launch_sss_final(scratch, B, chunks, per_utt, alpha, norms, loss, st);
  return 0;
}

template <int R>
static void launch_bwd_r(dim3 grid, hipStream_t st, int wrt_true, const float2* s_t, const float2* s_p, int n, int frames,
                         int chunks, const float2* tb, const float* norms, float inv_wn, float eps, float alpha, float inv_B,
                         float inv_n, const float* grad_out, float* dx, long ld_dx, int T, int accumulate, int turns) {
  if (wrt_true)
    hipLaunchKernelGGL((k_sss_czt_bwd<R, 1>), grid, dim3(64 * R), 0, st, s_t, s_p, n, frames, chunks, tb, norms, inv_wn, eps,
                       alpha, inv_B, inv_n, grad_out, dx, ld_dx, T, accumulate, turns);
  else
    hipLaunchKernelGGL((k_sss_czt_bwd<R, 0>), grid, dim3(64 * R), 0, st, s_t, s_p, n, frames, chunks, tb, norms, inv_wn, eps,
                       alpha, inv_B, inv_n, grad_out, dx, ld_dx, T, accumulate, turns);
}

// Overlapping frames (hop < n): k_sss_czt_bwd leaves every frame's windowed gradient in a frame-major scratch, and each
// sample gathers its up to ceil(n / hop) frames here -- in ascending frame order, so the sum is reproducible.
__global__ void __launch_bounds__(256) k_frames_overlap_add(const float* __restrict__ fg, int frames, int n, int hop, int T,
                                                            float* __restrict__ dx, long ld_dx, int accumulate) {
  const int b = blockIdx.y;
  const float* src = fg + (long)b * frames * n;
  float* o = dx + (long)b * ld_dx;
  for (int t = blockIdx.x * 256 + threadIdx.x; t < T; t += gridDim.x * 256) {
    int f_hi = t / hop;
    if (f_hi > frames - 1) f_hi = frames - 1;
    int f_lo = t - n + 1 <= 0 ? 0 : (t - n + hop) / hop;          // ceil((t - n + 1) / hop)
    float acc = 0.f;
    for (int f = f_lo; f <= f_hi; ++f) acc += src[(long)f * n + (t - f * hop)];
    o[t] = accumulate ? o[t] + acc : acc;
  }
}

size_t sss_wave_bwd_ws_bytes(int B, int n, int hop, int frames) {
  return hop == n ? 0 : (size_t)B * frames * n * sizeof(float);
}

int launch_sss_wave_bwd(const float* spec_t, const float* spec_p, int B, int T, int n, int hop, int frames, const float* tab,
                        const float* norms, float inv_wn, float eps, float alpha, const float* grad_out, int wrt_true,
                        float* dx, long ld_dx, int accumulate, float* ws, hipStream_t st) {
  const int R = czt_plan(n);
  if (!R || B < 1 || B > 65535 || frames < 1 || hop < 1 || hop > n || (long)(frames - 1) * hop + n > T) return -1;
  if (hop != n) {                                            // per-frame gradients into ws, then the gather
    if (!ws || (long)frames * n >= (1L << 31)) return -1;
    const int rc = launch_sss_wave_bwd(spec_t, spec_p, B, frames * n, n, n, frames, tab, norms, inv_wn, eps, alpha, grad_out,
                                       wrt_true, ws, (long)frames * n, 0, nullptr, st);
    if (rc != 0) return rc;
    int gx = (T + 255) / 256;
    if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(k_frames_overlap_add, dim3((unsigned)gx, (unsigned)B), dim3(256), 0, st, (const float*)ws, frames, n, hop,
                       T, dx, ld_dx, accumulate);
    return 0;
  }
  const WaveGeom geo = sss_wave_geom(B, n, frames, true);
  // priority turns of the waves that share a SIMD (fir_blk.hip): in this kernel only -- measured, same box: off 1.058 ms per
  // four-scale step, on in both kernels 1.046 with the forward kernels alone 1 % slower (and, at 4096 points, one register
  // past their file), so the forward kernel does without (knob CZT_TURNS = 2: off here too)
  const int turns = knob(KNOB_CZT_TURNS) == 2 ? 0 : 1;
  const int chunks = geo.chunks;                                       // the kernel cuts the pairs of frames into as many spans
  const dim3 grid((unsigned)chunks, (unsigned)B);
  const float2* tb = reinterpret_cast<const float2*>(tab);
  const float2* s_t = reinterpret_cast<const float2*>(spec_t);
  const float2* s_p = reinterpret_cast<const float2*>(spec_p);
  const float inv_B = 1.0f / (float)B;
  const float inv_n = (float)(1.0 / ((double)B * (double)frames * (double)(n / 2 + 1)));
  if (R == 2)
    launch_bwd_r<2>(grid, st, wrt_true, s_t, s_p, n, frames, chunks, tb, norms, inv_wn, eps, alpha, inv_B, inv_n, grad_out,
                    dx, ld_dx, T, accumulate, turns);
  else if (R == 4)
    launch_bwd_r<4>(grid, st, wrt_true, s_t, s_p, n, frames, chunks, tb, norms, inv_wn, eps, alpha, inv_B, inv_n, grad_out,
                    dx, ld_dx, T, accumulate, turns);
  else
    launch_bwd_r<8>(grid, st, wrt_true, s_t, s_p, n, frames, chunks, tb, norms, inv_wn, eps, alpha, inv_B, inv_n, grad_out,
                    dx, ld_dx, T, accumulate, turns);
  return 0;
}

}  // namespace ddsp
