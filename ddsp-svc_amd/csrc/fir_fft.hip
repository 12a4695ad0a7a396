// HOT-2c (FFT form): the time-varying FIR of ddsp/core.py:120-182 evaluated the way the reference itself
// does it -- per-frame block convolution in the frequency domain with overlap-add -- but as one fused
// kernel whose spectra never leave the CU.
//
// Frame j (0..F, row F re-uses taps F-1, core.py:167) convolves its taps (N <= 512 here) with the chunk
// (x * tri_j)[(j-1) hop .. (j+1) hop)  (the periodic Bartlett window of core.py:161 IS tri_j) and adds the
// 2 hop + N - 1 results at output position (j-1) hop - N/2 (crop of core.py:113-117).  With hop = 512 and
// N <= 512 the linear convolution (<= 1535 samples) fits a 2048-point transform without time aliasing
// (core.py:165 pads to 1533) and two consecutive frames fit the 2048-sample overlap-add ring.
//
// Per PAIR of frames (j, j+1) a 128-thread workgroup runs three 2048-point complex FFTs (fft2048.h):
//   Z_j   = FFT(chunk_j + i * s_j * taps_j)        two real sequences per transform; s_j = power of two that
//   Z_j+1 = FFT(chunk_j+1 + i * s_j+1 * taps_j+1)   balances their magnitudes (exact to undo)
//   G[k]  = (Z[k] + conj Z[-k]) (Z[k] - conj Z[-k]) / 4i  = X[k] H[k]          (Hermitian by construction)
//   out_j + i out_j+1 = IFFT(G_j / s_j + i G_j+1 / s_j+1)                       (inverse = conj, forward, conj)
// and adds both results into a 2048-sample overlap-add ring in LDS, from which finished samples are
// streamed out.  A workgroup walks a run of consecutive pairs of one utterance; it starts one pair early
// (whose output it discards) so the ring holds the tails of the frames before its first own pair.
//
// Cost: ~1.5 complex 2048-FFTs per frame (~170 kflop on the vector ALUs) against 2 hop N = 522 k
// multiply-adds (1.04 Mflop, x1.29 tile waste) for the direct form on the MFMA pipe.
#include "fft2048.h"
#include "kernels.h"
#include <stdlib.h>

namespace ddsp {

using fft::cconj;
using fft::cmul;

constexpr int FF_HOP = 512;

struct FirFftGeom {
  int F, N, T;            // frames, taps, samples per utterance
  int pairs;              // frame pairs per utterance: ceil((F + 1) / 2)
  int run;                // own pairs per workgroup
  int runs_per_utt;       // ceil(pairs / run)
};

// G = X H from the packed spectrum: a = Z[k], b = conj(Z[-k]);  X = (a + b)/2, H = (a - b)/2i
__device__ __forceinline__ f32x2 packed_product(f32x2 a, f32x2 zneg) {
  const f32x2 b = cconj(zneg);
  const f32x2 p = cmul(a + b, a - b);                // = 4i X H
  return f32x2{0.25f * p.y, -0.25f * p.x};           // / 4i
}

__global__ void __launch_bounds__(fft::THREADS) k_fir_fft(const float* __restrict__ x, int x_is_u01,
                                                          const float* __restrict__ taps,
                                                          const float* __restrict__ addend, float* __restrict__ out,
                                                          float* __restrict__ out_plain, FirFftGeom g) {
  __shared__ __attribute__((aligned(16))) f32x2 ex[fft::EX_WORDS];
  __shared__ float ring[fft::N];
  __shared__ float red[4];
  const int tid = threadIdx.x;
  const int b = blockIdx.x / g.runs_per_utt;
  const int run_no = blockIdx.x - b * g.runs_per_utt;
  const int p_first = run_no * g.run;                         // first own pair
  int p_last = p_first + g.run;                               // one past the last own pair
  if (p_last > g.pairs) p_last = g.pairs;
  const int D = g.N >> 1;
  const float* xb = x + (long)b * g.T;
  const float* tb = taps + (long)b * g.F * g.N;
  const long ob = (long)b * g.T;
  const float inv_hop = 1.0f / (float)FF_HOP;

  fft::Twiddles tw;
  tw.init(tid);
#pragma unroll
  for (int m = 0; m < 16; ++m) ring[128 * m + tid] = 0.f;

  for (int pr = (p_first > 0 ? p_first - 1 : 0); pr < p_last; ++pr) {
    f32x2 V[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) V[m] = f32x2{0.f, 0.f};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int j = 2 * pr + h;                               // frame index; j == F + 1 only pads an odd frame count
      f32x2 z[16];
      float mx = 0.f, mh = 0.f;
      if (j <= g.F) {
        const int s0 = (j - 1) * FF_HOP;
        const int row = j < g.F ? j : g.F - 1;
        const float* tr = tb + (long)row * g.N;
#pragma unroll
        for (int m = 0; m < 16; ++m) {
          const int n = 128 * m + tid;
          float xv = 0.f, hv = 0.f;
          if (m < 4 && n < g.N) hv = tr[n];                   // N <= 512
          if (m < 8) {                                        // chunk: 2 hop = 1024 samples
            const int s = s0 + n;
            if (s >= 0 && s < g.T) {
              xv = xb[s];
              if (x_is_u01) xv = fmaf(2.0f, xv, -1.0f);       // noise = rand*2-1 (vocoder.py:603,854)
              const float lam = (float)(n & (FF_HOP - 1)) * inv_hop;
              xv = (n < FF_HOP ? lam : 1.0f - lam) * xv;      // periodic Bartlett (core.py:161)
            }
          }
          z[m] = f32x2{xv, hv};
          mx = fmaxf(mx, fabsf(xv));
          mh = fmaxf(mh, fabsf(hv));
        }
      } else {
#pragma unroll
        for (int m = 0; m < 16; ++m) z[m] = f32x2{0.f, 0.f};
      }
      // workgroup-wide maxima -> power-of-two balance factor for the tap sequence
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) {
        mx = fmaxf(mx, __shfl_xor(mx, d));
        mh = fmaxf(mh, __shfl_xor(mh, d));
      }
      if ((tid & 63) == 0) { red[(tid >> 6) * 2] = mx; red[(tid >> 6) * 2 + 1] = mh; }
      __syncthreads();
      mx = fmaxf(red[0], red[2]);
      mh = fmaxf(red[1], red[3]);
      float sc = 1.0f, isc = 1.0f;
      if (mx > 0.f && mh > 0.f) {
        int e = ilogbf(mx) - ilogbf(mh);
        e = e < -60 ? -60 : (e > 60 ? 60 : e);
        sc = ldexpf(1.0f, e);
        isc = ldexpf(1.0f, -e);
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) z[m].y *= sc;               // N <= 512: the taps live in slots 0..3
      fft::forward(z, tw, ex, tid);
      // natural order to LDS, then G[k] from Z[k] and Z[-k]
#pragma unroll
      for (int m = 0; m < 16; ++m) ex[128 * m + tid] = z[m];
      __syncthreads();
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        const int k = 128 * m + tid;
        const f32x2 G = packed_product(z[m], ex[(fft::N - k) & (fft::N - 1)]) * isc;
        // V = G_j + i G_j+1, conjugated for the inverse-by-forward trick:  conj(V) = conj(G_j) - i conj(G_j+1)
        if (h == 0) V[m] = cconj(G);
        else V[m] = V[m] + f32x2{-G.y, -G.x};                 // -i * conj(G) = -i (Gx - i Gy) = (-Gy, -Gx)
      }
      __syncthreads();                                        // ex free for the next transform
    }
    fft::forward(V, tw, ex, tid);
    // ifft(V) = conj(FFT(conj V)) / 2048:  out_j = Re / 2048,  out_j+1 = -Im / 2048
    const int a0 = (2 * pr - 1) * FF_HOP - D;                 // output position of frame 2 pr's first sample
    const float scale = 1.0f / 2048.0f;
#pragma unroll
    for (int m = 0; m < 12; ++m) {                            // the linear convolution ends at 2 hop + N - 2 < 1536
      const int n = 128 * m + tid;
      ring[(a0 + n) & (fft::N - 1)] += V[m].x * scale;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 12; ++m) {
      const int n = 128 * m + tid;
      ring[(a0 + FF_HOP + n) & (fft::N - 1)] -= V[m].y * scale;
    }
    __syncthreads();
    // samples [a0, a0 + 2 hop) are complete once both frames are in; the last pair also flushes the tail
    const bool own = pr >= p_first;
    const int n_emit = (pr == g.pairs - 1) ? 16 : 8;
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      if (m < n_emit) {
        const int t = a0 + 128 * m + tid;
        const int ri = t & (fft::N - 1);
        const float v = ring[ri];
        ring[ri] = 0.f;
        if (own && t >= 0 && t < g.T) {
          if (out_plain) out_plain[ob + t] = v;
          out[ob + t] = addend ? v + addend[ob + t] : v;
        }
      }
    }
    __syncthreads();
  }
}

// returns the implementation id (4) or < 0 when the shape is outside this kernel
int launch_fir_fft(const float* x, int x_is_u01, const float* taps, const float* addend, float* out, float* out_plain,
                   int B, int F, int hop, int N, hipStream_t st) {
  if (hop != FF_HOP || N < 2 || (N & 1) || N > 512 || (long)F * hop >= (1L << 30)) return -1;
  FirFftGeom g;
  g.F = F; g.N = N; g.T = F * hop;
  g.pairs = (F + 2) / 2;
  // run length (own pairs per workgroup): as many workgroups as the chip holds at once (4 per CU at this
  // kernel's register budget), so all of them run in one round with equal work; every run pays one warm-up pair
  const long slots = 4 * 256;
  long per_utt = slots / (B > 0 ? B : 1);
  if (per_utt < 1) per_utt = 1;
  int run = (int)((g.pairs + per_utt - 1) / per_utt);
  if (run < 3) run = 3;
  if (const char* e = getenv("DDSP_HIP_FFT_RUN")) { int v = atoi(e); if (v >= 1) run = v; }
  if (run > g.pairs) run = g.pairs;
  g.run = run;
  g.runs_per_utt = (g.pairs + run - 1) / run;
  const long wgs = (long)B * g.runs_per_utt;
  if (wgs > 0x7fffffffL) return -1;
  hipLaunchKernelGGL(k_fir_fft, dim3((unsigned)wgs), dim3(fft::THREADS), 0, st, x, x_is_u01, taps, addend, out, out_plain, g);
  return 4;
}

}  // namespace ddsp
