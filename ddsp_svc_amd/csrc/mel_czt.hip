// The log-mel front-end at ANY transform length (second half): nsf_hifigan/nvSTFT.py:73-117, STFT.get_mel
// with keyshift != 0 (the transform and its window stretched to round(n_fft 2^(k/12)) points, :83-85; the cascade's formant
// shift, main_diff.py:359, and the pitch augmentation of preprocess.py:88-92), speed != 1 (hop round(hop speed)), center = True
// (torch.stft's own reflect padding of n/2 on top of the manual one), or a configuration other than 2048 / 512.
//
//   pad (win' - hop')/2 left, max((win' - hop' + 1)/2, win' - T - left) right (reflect, zeros for short signals, :97-103) ->
//   [center: reflect n'/2 both sides] -> frames of n' every hop' -> periodic Hann of win' (centred in n') -> the first
//   min(n_fft/2, n'/2) + 1 bins of the n'-point DFT -> sqrt(re^2 + im^2 + 1e-9) (:108) -> bins above n'/2 are zeros,
//   times win / win' (:109-114, only when keyshift != 0) -> mel_basis @ spec -> log(clamp(., clip_val)).
//
// n' is any integer (2048 2^(2/12) = 2299 = 11 * 11 * 19; odd lengths happen), so the transform is the chirp-z identity
//     X[k] = c[k] sum_j (x[j] w[j] c[j]) conj(c)[k - j],      c[j] = exp(-i pi j^2 / n'),
// one circular convolution of N = 2048 / 4096 points on the plans of fft_r.h with the chirp filter's spectrum from a
// table.  Only K <= N/4 + 1 bins are wanted, so a frame is cut into chunks of L = N/2 samples whose transforms add up
// after a linear phase, X[k] = sum_c exp(-2 pi i k s_c / n') X_c[k], s_c = c L -- and TWO chunks of a frame share one
// convolution: they are real, so z = a + i b evaluated at the bins -K < k < K (L + 2K - 2 <= N filter taps) comes apart
// through Z[k] and conj Z[-k].  Both halves belong to the same frame and are added up again, so neither carries rounding noise
// of a foreign level.  A frame of up to N samples (4096: a shift of +12 semitones at n_fft = 2048) is ONE convolution.
// One workgroup walks a run of frames; the whole chain of a frame stays in registers and LDS: the waveform is read
// (overlapped, through L2) and [B, frames, n_mels] written.  Bound: arithmetic, 2 transforms of N points per chunk pair.
//   k_mel_czt_tables : bhat[N] | (w c)[2 Cp][L] | G1[Cp][K'] | G2[Cp][K'] for one (n', win', K), float64 phases reduced in
//                      integers; K' = N/4 + 1
//   k_mel_czt<R>     : the frames, two in lockstep
#include "fft_r.h"
#include "kernels.h"

namespace ddsp {
using fft::cconj;
using fft::cmul;

constexpr int MZ_TAB_THREADS = 256;
constexpr int MZ_MAX_CHUNKS = 4;                              // two convolutions per frame at most

// bins the shifted transform delivers to the basis (nvSTFT.py:110-114: the rest are zeros)
static int mel_czt_bins(int n_new, int n_bins) { return n_bins < n_new / 2 + 1 ? n_bins : n_new / 2 + 1; }

// the convolution plan: K <= N/4 + 1
int mel_czt_plan(int n_new, int n_bins) {
  if (n_bins < 2 || n_bins > 1025 || n_new < 2) return 0;
  const int K = mel_czt_bins(n_new, n_bins);
  const int R = K <= 513 ? 4 : 8;
  return (n_new + 256 * R - 1) / (256 * R) <= MZ_MAX_CHUNKS ? R : 0;
}

static int mel_czt_chunks(int n_new, int R) { return (n_new + 256 * R - 1) / (256 * R); }

size_t mel_czt_table_bytes(int n_new, int n_bins) {
  const int R = mel_czt_plan(n_new, n_bins);
  if (!R) return 0;
  const int Cp = (mel_czt_chunks(n_new, R) + 1) / 2;
  return (size_t)(512 * R + 2 * Cp * 256 * R + 2 * Cp * (128 * R + 1)) * sizeof(float2);
}

// Blocks [0, N): one bin of the chirp filter's spectrum each,
//     bhat[q] = sum_{m = -(L-1)-(K-1)}^{K-1} exp(+i pi m^2 / n) exp(-2 pi i q m / N);
// the blocks behind them:
//     (w c)[c][j] = hann_win(s_c + j - left) exp(-i pi j^2 / n), zero outside the window and behind the frame, c < 2 Cp;
//     G1[p][k] = c[k] (phi_a - i phi_b) / 2N,  G2[p][k] = conj(c[k]) (phi_a + i phi_b) / 2N,  phi_c = exp(-2 pi i k s_c / n)
// for the chunks a = 2p, b = 2p + 1 of pair p and 0 <= k <= N/4: with y the convolution of (a + i b) c with the filter,
//     X[k] += y[k] G1[p][k] + conj(y[-k]) G2[p][k]          (A = (Z[k] + conj Z[-k]) / 2, B = (Z[k] - conj Z[-k]) / 2i, Z = c y).
__global__ void __launch_bounds__(MZ_TAB_THREADS) k_mel_czt_tables(int n, int wn, int K, int N, int Cp, float2* __restrict__ tab) {
  __shared__ double red[2][MZ_TAB_THREADS / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int L = N / 2, KT = N / 4 + 1;
  if ((int)blockIdx.x >= N) {
    int e = ((int)blockIdx.x - N) * MZ_TAB_THREADS + tid;
    double s, co;
    if (e < 2 * Cp * L) {
      const int c = e / L, j = e - c * L;
      const int i = c * L + j, left = (n - wn) / 2;                                // torch.stft centres a short window
      double w = 0.0;
      if (i < n && i >= left && i < left + wn) w = 0.5 - 0.5 * cospi(2.0 * (double)(i - left) / (double)wn);   // periodic Hann
      sincospi((double)(((long)j * j) % (2 * n)) / (double)n, &s, &co);
      tab[N + e] = float2{(float)(w * co), (float)(-w * s)};
      return;
    }
    e -= 2 * Cp * L;
    if (e >= 2 * Cp * KT) return;
    const int which = e / (Cp * KT), r = e - which * (Cp * KT), p = r / KT, k = r - p * KT;
    const long sa = 2L * p * L, sb = sa + L;
    // c[k] phi_a = exp(-i pi (k^2 + 2 k s_a) / n); the conjugate chirp for G2
    double cas, cac, cbs, cbc;
    const long k2 = (long)k * k;
    const long pa = which == 0 ? (k2 + 2L * k * sa) % (2L * n) : ((2L * k * sa - k2) % (2L * n) + 2L * n) % (2L * n);
    const long pb = which == 0 ? (k2 + 2L * k * sb) % (2L * n) : ((2L * k * sb - k2) % (2L * n) + 2L * n) % (2L * n);
    sincospi((double)pa / (double)n, &cas, &cac);
    sincospi((double)pb / (double)n, &cbs, &cbc);
    // ea = exp(-i pi pa / n) = (cac, -cas), eb likewise;  G1 = (ea - i eb) / 2N,  G2 = (ea + i eb) / 2N
    const double h = 0.5 / (double)N;
    const double re = which == 0 ? cac - cbs : cac + cbs;       // -i (x + i y) = y - i x
    const double im = which == 0 ? -cas - cbc : -cas + cbc;
    tab[N + 2 * Cp * L + e] = float2{(float)(h * re), (float)(h * im)};
    return;
  }
  const int q = blockIdx.x;
  double re = 0.0, im = 0.0;
  for (int t = tid; t < L + 2 * K - 2; t += MZ_TAB_THREADS) {
    const int m = t - (L - 1) - (K - 1);
    const long qm = (((long)q * m) % N + N) % N;
    const double ph = (double)(((long)m * m) % (2 * n)) / (double)n - 2.0 * (double)qm / (double)N;
    double s, co;
    sincospi(ph, &s, &co);
    re += co;
    im += s;
  }
  re = wave_sum(re);
  im = wave_sum(im);
  if (lane == 0) { red[0][wave] = re; red[1][wave] = im; }
  __syncthreads();
  if (tid == 0) {
    double a = 0.0, b = 0.0;
    for (int w = 0; w < MZ_TAB_THREADS / 64; ++w) { a += red[0][w]; b += red[1][w]; }
    tab[q] = float2{(float)a, (float)b};
  }
}

struct MelCztGeom {
  int T, frames, span;
  int n, hop, center;         // transform length, hop, torch.stft's center flag
  int pad_left, T1, reflect;  // the manual padding (nvSTFT.py:97-103): left pad, padded length, mode
  int pairs, bins, bins_eff;  // chunk pairs per frame; bins of the mel basis; bins the shifted transform has (the rest are zeros)
  int n_mels;
  float clip, mag_scale;
  long sb, sm, sf;
};

constexpr int MZ_MAGS = 1032;                                // floats of a frame's magnitudes in LDS (a basis of <= 1025 bins)

// TWO frames in lockstep (Plan::forward2: the same arithmetic as two transforms, behind shared barriers -- twice the work between
// two barriers, one frame's exchange latency under the other's butterflies): 0.627 ms against 0.667 one frame at a time (B = 32 x
// 10 s, r06_v36_mel_shifted_lockstep.txt).  A run with an odd number of frames does its last one twice (the copy is not stored).
// Two waves per SIMD (237 registers, none spilled; 136 KB of LDS at N = 4096: one workgroup per CU).  The one-frame form cut to 128
// registers for four waves spilled 25 - 40 of them and took 0.96 ms against 0.69 (r06_v30_mel_shifted_forms.txt).
template <int R>
__global__ void __launch_bounds__(64 * R, 2) k_mel_czt(const float* __restrict__ audio, const float2* __restrict__ tab,
                                                        const int* __restrict__ band, const float* __restrict__ packed,
                                                        float* __restrict__ out, MelCztGeom g) {
  using PL = fft::Plan<R>;
  constexpr int N = PL::N, P = PL::P, L = N / 2, KT = N / 4 + 1;
  __shared__ __attribute__((aligned(16))) f32x2 ex[4][N];
  __shared__ float mags[2][MZ_MAGS];
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int f_lo = blockIdx.x * g.span;
  const int f_hi = f_lo + g.span < g.frames ? f_lo + g.span : g.frames;
  const f32x2* bh = reinterpret_cast<const f32x2*>(tab);
  const f32x2* wc = bh + N;
  const f32x2* g1 = wc + 2 * g.pairs * L;
  const f32x2* g2 = g1 + g.pairs * KT;
  const float* ab = audio + (long)b * g.T;
  typename PL::Tw tw;
  tw.init(tid);
  f32x2 gq[8];                                               // the chirp filter's spectrum: the same for every frame and pair
#pragma unroll
  for (int m = 0; m < 8; ++m) gq[m] = bh[P * m + tid];

  // sample i of the (centre-padded) manually padded signal; indices behind a frame's end are never used (w c = 0 there)
  auto sample = [&](int i) -> float {
    int t = i;
    if (g.center) {                                          // torch.stft(center=True, pad_mode='reflect')
      t -= g.n / 2;
      if (t < 0) t = -t;
      if (t >= g.T1) t = 2 * (g.T1 - 1) - t;
    }
    int u = t - g.pad_left;
    bool ok = true;
    if (g.reflect) {
      if (u < 0) u = -u;
      if (u >= g.T) u = 2 * (g.T - 1) - u;
    } else {
      ok = u >= 0 && u < g.T;
    }
    u = u < 0 ? 0 : (u >= g.T ? g.T - 1 : u);
    const float v = ab[u];
    return ok ? v : 0.f;
  };
  float nx[2][2][4];                                         // [frame of the pair][chunk of the pair][slot]: the NEXT convolution's samples
  auto fetch = [&](int f, int p) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int fe = f + e < f_hi ? f + e : f_hi - 1;
      const int base = fe * g.hop + 2 * p * L;
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int m = 0; m < 4; ++m) nx[e][h][m] = sample(base + h * L + P * m + tid);
    }
  };
  if (f_lo < f_hi) fetch(f_lo, 0);
  for (int f = f_lo; f < f_hi; f += 2) {
    f32x2 acc[2][3];                                         // X[k], k = P m + tid <= N/4: the slots 0, 1 and (thread 0) 2
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int m = 0; m < 3; ++m) acc[e][m] = f32x2{0.f, 0.f};
    for (int p = 0; p < g.pairs; ++p) {
      f32x2 v[2][8];
#pragma unroll
      for (int m = 0; m < 4; ++m) {                          // (a w + i b w') c: the two chunks' windows differ, the chirp does not
        const f32x2 wa = wc[(2 * p) * L + P * m + tid], wb = wc[(2 * p + 1) * L + P * m + tid];
#pragma unroll
        for (int e = 0; e < 2; ++e)
          v[e][m] = f32x2{nx[e][0][m] * wa.x - nx[e][1][m] * wb.y, nx[e][0][m] * wa.y + nx[e][1][m] * wb.x};
      }
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int m = 4; m < 8; ++m) v[e][m] = f32x2{0.f, 0.f};
      if (p + 1 < g.pairs) fetch(f, p + 1);
      else if (f + 2 < f_hi) fetch(f + 2, 0);
      __syncthreads();                                       // the last transforms' pass 4 still reads ex[0], ex[2]
      PL::template forward2<true>(v[0], v[1], tw, ex[0], ex[1], ex[2], ex[3], tid);
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        v[0][m] = cconj(cmul(v[0][m], gq[m]));
        v[1][m] = cconj(cmul(v[1][m], gq[m]));
      }
      __syncthreads();
      PL::forward2(v[0], v[1], tw, ex[0], ex[1], ex[2], ex[3], tid);   // v = conj(N y), natural order: index P m + tid
      // the bins -N/4 .. -1 sit at the indices 3N/4 .. N-1 (slots 6, 7): through ex[1] / ex[3] (free since pass 3) to their mirror
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        f32x2* Y = ex[1 + 2 * e];
        Y[6 * P + tid] = v[e][6];
        Y[7 * P + tid] = v[e][7];
        if (tid == 0) Y[0] = v[e][0];
      }
      f32x2 q1[3], q2[3];
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const int k = P * m + tid;
        const int kk = k < KT ? k : 0;
        q1[m] = g1[p * KT + kk];
        q2[m] = g2[p * KT + kk];
      }
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int m = 0; m < 3; ++m) {
          const int k = P * m + tid;
          if (k < KT) {
            const f32x2 ym = ex[1 + 2 * e][(N - k) & (N - 1)];   // conj(N y[-k])
            const f32x2 x = cmul(cconj(v[e][m]), q1[m]), z = cmul(ym, q2[m]);
            acc[e][m] = f32x2{acc[e][m].x + (x.x + z.x), acc[e][m].y + (x.y + z.y)};
          }
        }
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const int k = P * m + tid;
        if (k < KT && k < g.bins_eff)
          mags[e][k] = __builtin_amdgcn_sqrtf(fmaf(acc[e][m].x, acc[e][m].x, acc[e][m].y * acc[e][m].y) + 1e-9f) * g.mag_scale;   // (hardware root, 1 ulp)
      }
      for (int k = g.bins_eff + tid; k < g.bins; k += P) mags[e][k] = 0.f;    // nvSTFT.py:110-113: zeros above the shifted Nyquist
    }
    __syncthreads();
    // banded mel projection (nvSTFT.py:115-116): four lanes per filter, every fourth bin of its band each
    for (int e = 0; e < 2 && f + e < f_hi; ++e) {
      for (int c0 = 0; c0 < g.n_mels; c0 += P / 4) {
        const int c = c0 + (tid >> 2), part = tid & 3;
        float a = 0.f;
        if (c < g.n_mels) {
          const int lo = band[4 * c], hi = band[4 * c + 1];
          const float* wr = packed + band[4 * c + 2] - lo;
          for (int k = lo + part; k < hi; k += 4) a = fmaf(wr[k], mags[e][k], a);
        }
        a += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0xB1, 0xF, 0xF, false));     // quad_perm [1,0,3,2]
        a += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x4E, 0xF, 0xF, false));     // quad_perm [2,3,0,1]
        // (the hardware logarithm, 1 ulp of log2: the whole wave paid libm's ~25 instructions for its 16 storing lanes)
        if (c < g.n_mels && part == 0)
          out[(long)b * g.sb + (long)c * g.sm + (long)(f + e) * g.sf] = 0.6931471805599453f * __builtin_amdgcn_logf(fmaxf(a, g.clip));
      }
    }
    // no barrier: the next frames write mags (and ex[1], ex[3]) behind their transforms' barriers
  }
}

// frames of get_mel for a signal of T samples, or -1 where torch raises (a transform longer than the padded signal, a
// window longer than the transform, a reflection longer than the signal)
int mel_czt_frames(int T, int n_new, int win_new, int hop_new, int center) {
  if (T < 1 || n_new < 2 || win_new < 1 || win_new > n_new || hop_new < 1 || hop_new > win_new) return -1;
  const int pad_left = (win_new - hop_new) / 2;
  int pad_right = (win_new - hop_new + 1) / 2;
  if (win_new - T - pad_left > pad_right) pad_right = win_new - T - pad_left;
  const long T1 = (long)T + pad_left + pad_right;
  if (T1 >= (1L << 30)) return -1;
  if (center) {
    if (n_new / 2 >= T1) return -1;
    return (int)(1 + (T1 + 2 * (n_new / 2) - n_new) / hop_new);
  }
  if (T1 < n_new) return -1;
  return (int)(1 + (T1 - n_new) / hop_new);
}

int launch_mel_czt_tables(int n_new, int win_new, int n_bins, float* tab, hipStream_t st) {
  const int R = mel_czt_plan(n_new, n_bins);
  if (!R || win_new < 1 || win_new > n_new) return -1;
  const int N = 512 * R, L = N / 2, KT = N / 4 + 1, Cp = (mel_czt_chunks(n_new, R) + 1) / 2;
  const int fill = (2 * Cp * L + 2 * Cp * KT + MZ_TAB_THREADS - 1) / MZ_TAB_THREADS;
  hipLaunchKernelGGL(k_mel_czt_tables, dim3((unsigned)(N + fill)), dim3(MZ_TAB_THREADS), 0, st, n_new, win_new,
                     mel_czt_bins(n_new, n_bins), N, Cp, reinterpret_cast<float2*>(tab));
  return 0;
}

int launch_mel_czt(const float* audio, int B, int T, const float* tab, int n_new, int win_new, int hop_new, int center,
                   int n_bins, float mag_scale, const int* band, const float* packed, int n_mels, float clip, float* out,
                   long sb, long sm, long sf, hipStream_t st) {
  const int R = mel_czt_plan(n_new, n_bins);
  const int frames = mel_czt_frames(T, n_new, win_new, hop_new, center);
  if (!R || frames < 1 || B < 1 || n_mels < 1) return -1;
  MelCztGeom g;
  g.pairs = (mel_czt_chunks(n_new, R) + 1) / 2;
  g.T = T; g.frames = frames;
  g.n = n_new; g.hop = hop_new; g.center = center ? 1 : 0;
  g.pad_left = (win_new - hop_new) / 2;
  int pad_right = (win_new - hop_new + 1) / 2;
  if (win_new - T - g.pad_left > pad_right) pad_right = win_new - T - g.pad_left;
  g.T1 = T + g.pad_left + pad_right;
  g.reflect = pad_right < T ? 1 : 0;
  g.bins = n_bins;
  g.bins_eff = mel_czt_bins(n_new, n_bins);
  g.n_mels = n_mels; g.clip = clip; g.mag_scale = mag_scale;
  g.sb = sb; g.sm = sm; g.sf = sf;
  // A run of frames per workgroup amortises the twiddles and the filter spectrum (32 registers' worth of loads and sincos: about
  // 1.5 frames' time).  One workgroup per CU is resident (N = 4096; two at 2048): the run length that fills whole rounds of
  // 256 workgroups with the least work per round.
  const int per_round = R == 8 ? 256 : 512;
  long best = -1;
  g.span = 2;
  for (int sp = 2; sp <= 32; sp += 2) {                      // frames go through in pairs
    const long wgs = (long)((frames + sp - 1) / sp) * B;
    const long cost = ((wgs + per_round - 1) / per_round) * (2 * sp + 3);
    if (best < 0 || cost < best) { best = cost; g.span = sp; }
  }
  const float2* tb = reinterpret_cast<const float2*>(tab);
  for (int b0 = 0; b0 < B; b0 += 65535) {                    // grid.y holds 65535 utterances
    const int nb = B - b0 < 65535 ? B - b0 : 65535;
    const dim3 grid((unsigned)((frames + g.span - 1) / g.span), (unsigned)nb);
    const float* a = audio + (long)b0 * T;
    float* o = out + (long)b0 * sb;
    if (R == 4) hipLaunchKernelGGL(k_mel_czt<4>, grid, dim3(256), 0, st, a, tb, band, packed, o, g);
    else        hipLaunchKernelGGL(k_mel_czt<8>, grid, dim3(512), 0, st, a, tb, band, packed, o, g);
  }
  return 0;
}

}  // namespace ddsp
