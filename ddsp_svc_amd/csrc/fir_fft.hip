// HOT-2c (FFT form): the time-varying FIR of ddsp/core.py:120-182 evaluated the way the reference itself
// does it -- per-frame block convolution in the frequency domain with overlap-add -- but as one fused
// kernel whose spectra never leave the CU.
//
// Frame j (0..F, row F re-uses taps F-1, core.py:167) convolves its taps (N <= 512; 514 .. 1022 in the LONG variant) with the chunk
// (x * tri_j)[(j-1) hop .. (j+1) hop)  (the periodic Bartlett window of core.py:161 IS tri_j) and adds the
// 2 hop + N - 1 results at output position (j-1) hop - N/2 (crop of core.py:113-117).  With hop = 512 and
// N <= 512 the linear convolution (<= 1535 samples) fits a 2048-point transform without time aliasing
// (core.py:165 pads to 1533) and two consecutive frames fit the 2048-sample overlap-add ring.
//
// Per PAIR of frames (j, j+1) a 256-thread workgroup runs three 2048-point complex FFTs (fft2048.h):
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

using fft::cmul;

constexpr int FF_HOP = 512;

struct FirFftGeom {
  int F, N, T;            // frames, taps, samples per utterance
  int pairs;              // frame pairs per utterance: ceil((F + 1) / 2)
  int run;                // own pairs per workgroup
  int runs_per_utt;       // ceil(pairs / run)
};

// G = X H from the packed spectrum: a = Z[k], zneg = Z[-k], b = conj(zneg);  X = (a + b)/2, H = (a - b)/2i,
// so (a + b)(a - b) = 4i X H and G = p / 4i = (p.y, -p.x) / 4; q = (s/4, -s/4) also undoes the tap scaling.
__device__ __forceinline__ f32x2 packed_product(f32x2 a, f32x2 zneg, f32x2 q) {
  const f32x2 p = cmul(fft::add_conj(a, zneg), fft::sub_conj(a, zneg));
  return fft::swap_scale(p, q);
}

// LONG: tap counts from 514 to 1022 (n_mag up to 512: the harmonic filter of the classic CombSub configuration,
// n_mag_allpass 256 / n_mag_harmonic 512 / n_mag_noise 256).  A frame's linear convolution (2 hop + N - 1 <= 2045 samples)
// still fits the 2048-point transform; what grows is its reach: four taps per thread instead of two, all eight slots of the
// result are live, two consecutive frames span up to 2557 samples (a 4096-sample ring: 48 KB of LDS, three workgroups per
// CU), and a run needs TWO warm-up pairs for its ring to hold the tails of the four frames that reach into it.  Without this
// form such a filter ran on the direct MFMA form at ~3x the time.
template <bool LONG>
__global__ void __launch_bounds__(fft::THREADS, LONG ? 3 : 4) k_fir_fft(const float* __restrict__ x, int x_is_u01,
                                                                     const float* __restrict__ taps,
                                                                     const float* __restrict__ addend, float* __restrict__ out,
                                                                     float* __restrict__ out_plain, FirFftGeom g) {
  constexpr int S = fft::SLOTS;                               // 8 complex points per thread, point k = 256 m + tid
  constexpr int RING = LONG ? 4096 : 2048, TAPM = LONG ? 4 : 2, ACC_M = LONG ? 8 : 6;
  __shared__ __attribute__((aligned(16))) f32x2 exA[fft::EX_WORDS];
  __shared__ __attribute__((aligned(16))) f32x2 exB[fft::EX_WORDS];
  __shared__ float ring[RING];               // 2 x 16 KB + 8 KB = 40 KB exactly: four workgroups per CU (LONG: 48 KB, three)
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
  for (int m = 0; m < RING / 256; ++m) ring[256 * m + tid] = 0.f;

  // inputs of one frame: 4 chunk samples (already Bartlett-weighted) and 2 (LONG: 4) taps per thread
  struct Frame { float xv[4], hv[TAPM]; };
  auto load_frame = [&](int j) -> Frame {
    Frame f;
#pragma unroll
    for (int m = 0; m < 4; ++m) f.xv[m] = 0.f;
#pragma unroll
    for (int m = 0; m < TAPM; ++m) f.hv[m] = 0.f;
    if (j <= g.F) {                                           // j == F + 1 only pads an odd frame count
      const int s0 = (j - 1) * FF_HOP;
      const int row = j < g.F ? j : g.F - 1;                  // core.py:167
      const float* tr = tb + (long)row * g.N;
#pragma unroll
      for (int m = 0; m < TAPM; ++m)
        if (256 * m + tid < g.N) f.hv[m] = tr[256 * m + tid];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int n = 256 * m + tid;
        const int sidx = s0 + n;
        if (sidx >= 0 && sidx < g.T) {
          float xv = xb[sidx];
          if (x_is_u01) xv = fmaf(2.0f, xv, -1.0f);           // noise = rand*2-1 (vocoder.py:603,854)
          const float lam = (float)(n & (FF_HOP - 1)) * inv_hop;
          f.xv[m] = (n < FF_HOP ? lam : 1.0f - lam) * xv;     // periodic Bartlett (core.py:161)
        }
      }
    }
    return f;
  };

  const int pr0 = p_first > (LONG ? 1 : 0) ? p_first - (LONG ? 2 : 1) : 0;    // warm-up pairs: their output is discarded
  Frame nxt = load_frame(2 * pr0);
  for (int pr = pr0; pr < p_last; ++pr) {
    f32x2 V[S];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const Frame cur = nxt;
      // the next frame's global loads are issued now and land while this frame is transformed
      nxt = load_frame(2 * pr + h + 1);
      // energies of the two sequences -> power-of-two balance factor for the taps (their spectra then have the
      // same mean magnitude; a packed transform rounds relative to the larger one)
      float sx = 0.f, sh = 0.f;
#pragma unroll
      for (int m = 0; m < 4; ++m) sx = fmaf(cur.xv[m], cur.xv[m], sx);
      sh = fmaf(cur.hv[0], cur.hv[0], cur.hv[1] * cur.hv[1]);
      if (LONG) sh = fmaf(cur.hv[TAPM - 2], cur.hv[TAPM - 2], fmaf(cur.hv[TAPM - 1], cur.hv[TAPM - 1], sh));
      sx = wave_sum_dpp(sx);
      sh = wave_sum_dpp(sh);
      // the four waves meet through 8 ring slots at the far end of the window: they belong to samples this pair
      // only reaches with its second overlap-add, and are zeroed again before that (see below)
      float* red = ring;
      const int rbase = (2 * pr - 1) * FF_HOP - D + RING - 8;
      if ((tid & 63) == 0) {
        red[(rbase + (tid >> 6) * 2) & (RING - 1)] = sx;
        red[(rbase + (tid >> 6) * 2 + 1) & (RING - 1)] = sh;
      }
      __syncthreads();
      sx = (red[rbase & (RING - 1)] + red[(rbase + 2) & (RING - 1)]) +
           (red[(rbase + 4) & (RING - 1)] + red[(rbase + 6) & (RING - 1)]);
      sh = (red[(rbase + 1) & (RING - 1)] + red[(rbase + 3) & (RING - 1)]) +
           (red[(rbase + 5) & (RING - 1)] + red[(rbase + 7) & (RING - 1)]);
      float sc = 1.0f, isc = 1.0f;
      if (sx > 0.f && sh > 0.f && sx < 3e38f && sh < 3e38f) {
        int e = (ilogbf(sx) - ilogbf(sh)) >> 1;               // sqrt of the energy ratio, as a power of two
        e = e < -60 ? -60 : (e > 60 ? 60 : e);
        sc = ldexpf(1.0f, e);
        isc = ldexpf(1.0f, -e);
      }
      const f32x2 q = {0.25f * isc, -0.25f * isc};
      f32x2 z[S];
#pragma unroll
      for (int m = 0; m < S; ++m) z[m] = f32x2{m < 4 ? cur.xv[m] : 0.f, m < TAPM ? cur.hv[m < TAPM ? m : 0] * sc : 0.f};
      fft::forward(z, tw, exA, exB, tid);
      // natural order to LDS (buffer B is free), then G[k] from Z[k] and Z[-k]
#pragma unroll
      for (int m = 0; m < S; ++m) exB[256 * m + tid] = z[m];
      __syncthreads();
#pragma unroll
      for (int m = 0; m < S; ++m) {
        const int k = 256 * m + tid;
        const f32x2 G = packed_product(z[m], exB[(fft::N - k) & (fft::N - 1)], q);
        // V = G_j + i G_j+1, conjugated for the inverse-by-forward trick:  conj(V) = conj(G_j) - i conj(G_j+1)
        if (h == 0) V[m] = G;
        else V[m] = fft::conj_minus_i_conj(V[m], G);
      }
      // no barrier here: the next transform first writes A (whose pass-4 readers all passed the barrier above)
      // and touches B only after its own first barrier
    }
    fft::forward(V, tw, exA, exB, tid);
    // ifft(V) = conj(FFT(conj V)) / 2048:  out_j = Re / 2048,  out_j+1 = -Im / 2048
    const int a0 = (2 * pr - 1) * FF_HOP - D;                 // output position of frame 2 pr's first sample
    const float scale = 1.0f / 2048.0f;
#pragma unroll
    for (int m = 0; m < ACC_M; ++m) {                         // the linear convolution ends at 2 hop + N - 2 < 1536 (LONG: < 2048)
      const int n = 256 * m + tid;
      ring[(a0 + n) & (RING - 1)] += V[m].x * scale;
    }
    if (tid < 8) ring[(a0 + RING - 8 + tid) & (RING - 1)] = 0.f;         // the borrowed reduction slots
    __syncthreads();
#pragma unroll
    for (int m = 0; m < ACC_M; ++m) {
      const int n = 256 * m + tid;
      ring[(a0 + FF_HOP + n) & (RING - 1)] -= V[m].y * scale;
    }
    __syncthreads();
    // samples [a0, a0 + 2 hop) are complete once both frames are in; the last pair also flushes the tail
    const bool own = pr >= p_first;
    const int n_emit = (pr == g.pairs - 1) ? 8 : 4;
#pragma unroll
    for (int m = 0; m < S; ++m) {
      if (m < n_emit) {
        const int t = a0 + 256 * m + tid;
        const int ri = t & (RING - 1);
        const float v = ring[ri];
        ring[ri] = 0.f;
        if (own && t >= 0 && t < g.T) {
          if (out_plain) out_plain[ob + t] = v;
          out[ob + t] = addend ? v + addend[ob + t] : v;
        }
      }
    }
    // The zeroed ring slots are next written after the barriers inside the coming transforms -- except the eight at the end
    // of the emitted stretch, which the next pair borrows for its energy reduction right away: without this barrier a wave
    // that runs ahead puts its two sums there before a slower wave has emitted those samples (found by tools/flaky_probe.py:
    // two wrong output samples in about one launch of a hundred).
    __syncthreads();
  }
}

// returns the implementation id (4) or < 0 when the shape is outside this kernel
int launch_fir_fft(const float* x, int x_is_u01, const float* taps, const float* addend, float* out, float* out_plain,
                   int B, int F, int hop, int N, hipStream_t st) {
  if (hop != FF_HOP || N < 2 || (N & 1) || N > 1022 || (long)F * hop >= (1L << 30)) return -1;
#ifdef DDSP_AB_GENERATIONS
  const bool long_taps = N > 512;                 // A/B builds keep round 2's short form (two taps per thread, 2048-sample ring)
#else
  const bool long_taps = true;                    // ONE form: four taps per thread, 4096-sample ring -- every N up to 1022
#endif
  FirFftGeom g;
  g.F = F; g.N = N; g.T = F * hop;
  g.pairs = (F + 2) / 2;
  // run length (own pairs per workgroup): as many workgroups as the chip holds at once (4 per CU at this
  // kernel's LDS budget), so all of them run in one round with equal work; every run pays one warm-up pair
  const long slots = (long_taps ? 3 : 4) * 256;
  const int Bg = t_geometry_batch > 0 ? t_geometry_batch : B;          // kernels.h: a sub-batch keeps the whole call's split
  long per_utt = slots / (Bg > 0 ? Bg : 1);
  if (per_utt < 1) per_utt = 1;
  int run = (int)((g.pairs + per_utt - 1) / per_utt);
  if (run < 3) run = 3;
  if (const long v = knob(KNOB_FFT_RUN)) { if (v >= 1) run = (int)v; }
  if (run > g.pairs) run = g.pairs;
  g.run = run;
  g.runs_per_utt = (g.pairs + run - 1) / run;
  const long wgs = (long)B * g.runs_per_utt;
  if (wgs > 0x7fffffffL) return -1;
#ifdef DDSP_AB_GENERATIONS
  if (!long_taps) {
    hipLaunchKernelGGL(k_fir_fft<false>, dim3((unsigned)wgs), dim3(fft::THREADS), 0, st, x, x_is_u01, taps, addend, out, out_plain, g);
    return 4;
  }
#endif
  hipLaunchKernelGGL(k_fir_fft<true>, dim3((unsigned)wgs), dim3(fft::THREADS), 0, st, x, x_is_u01, taps, addend, out, out_plain, g);
  return 4;
}

}  // namespace ddsp
