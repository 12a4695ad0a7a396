// Waveform -> log-mel front-end of the cascade: nsf_hifigan/nvSTFT.py:73-117, STFT.get_mel with
// keyshift = 0, speed = 1, center = False, n_fft = win = 2048, hop = 512 (the 44.1 kHz NSF-HiFiGAN configuration the
// diffusion / reflow cascades extract their conditioning mel with, diffusion/vocoder.py:146-148,248,296).
//
//   pad (win-hop)/2 = 768 samples on both sides (reflect, or zeros when the signal is not longer than the right pad,
//   :97-103) -> frames of 2048 every 512 -> periodic Hann -> rfft (:106) -> sqrt(re^2 + im^2 + 1e-9) (:108) ->
//   mel_basis @ spec (:115) -> log(clamp(., clip_val)) (:116).
//
// The reference materialises a [B, 1025, F] complex spectrum and its magnitude.  Here a 256-thread workgroup walks a
// run of frame PAIRS: frame j rides in the real and frame j+1 in the imaginary part of ONE 2048-point complex
// transform (fft_r.h), the two spectra are separated with the mirrored bin Z[-k] through LDS, the magnitudes of
// both frames are parked in LDS, and four lanes per (filter, frame) reduce them against the filter's band -- a mel basis row
// is a short contiguous band (2..90 bins of 1025), so the projection reads ~2 k weights per frame instead of the
// 131 k of the dense matmul.  Only the waveform (4x overlapped, through L2) and the [B, F, n_mels] result touch HBM.
#include "fft_r.h"
#include "kernels.h"
#include <stdlib.h>

namespace ddsp {

constexpr int ME_HOP = 512;
constexpr int ME_WPACK = 4096;                                 // LDS floats for the packed band weights (Slaney-128: 1460)

struct MelGeom {
  int T, frames, pairs;     // samples per utterance, frames per utterance, ceil(frames / 2)
  int run, runs_per_utt;    // pairs per workgroup
  int reflect;              // padding mode (nvSTFT.py:99-102)
  int n_mels;
  int packed_len;           // floats of band weights staged in LDS (0: read the dense basis from global memory)
  float clip;
  long sb, sm, sf;          // output strides (floats): utterance, mel channel, frame
};

template <int WPS>
__global__ void __launch_bounds__(256, WPS) k_mel(const float* __restrict__ audio, const float* __restrict__ window,
                                                 const float* __restrict__ basis, const int* __restrict__ band,
                                                 const float* __restrict__ packed, float* __restrict__ out,
                                                 MelGeom g) {
  using PL = fft::Plan<4>;
  constexpr int N = PL::N, P = PL::P, S = 8;
  constexpr int BINS = N / 2 + 1;
  constexpr int PAD = (N - ME_HOP) / 2;                         // 768 (nvSTFT.py:97)
  constexpr int MROW = 1032;                                   // floats per magnitude row in LDS
  __shared__ __attribute__((aligned(16))) f32x2 ex[2][N];
  __shared__ float wl[ME_WPACK];                               // every filter's band weights, back to back
  const int tid = threadIdx.x;
  const int b = blockIdx.x / g.runs_per_utt;
  const int run_no = blockIdx.x - b * g.runs_per_utt;
  const int p_first = run_no * g.run;
  int p_last = p_first + g.run;
  if (p_last > g.pairs) p_last = g.pairs;
  const float* ab = audio + (long)b * g.T;

  typename PL::Tw tw;
  tw.init(tid);
  float w[S];
#pragma unroll
  for (int m = 0; m < S; ++m) w[m] = window[P * m + tid];
  for (int i = tid; i < g.packed_len; i += P) wl[i] = packed[i];     // visible after the first barrier below
  int cur = 0;

  // raw samples of a frame pair: issued unconditionally from clamped (reflected) addresses for the NEXT pair before
  // the current one is transformed, masked when used
  struct Raw { float v[S][2]; };
  // (an INTERIOR pair -- both frames inside the signal, all but the first and the last of an utterance -- takes its 16 samples from
  // one base address and instruction offsets; the reflected / clamped index arithmetic was ~130 of a pass's ~750 vector instructions)
  auto interior = [&](int pr) -> bool {
    const int s0 = 2 * pr * ME_HOP - PAD;
    return s0 >= 0 && s0 + ME_HOP + N <= g.T;
  };
  auto load_pair = [&](int pr) -> Raw {
    Raw r;
    const int s0 = 2 * pr * ME_HOP - PAD;
    if (interior(pr)) {                                        // workgroup-uniform
      const float* src = ab + s0 + tid;
#pragma unroll
      for (int m = 0; m < S; ++m) {
        r.v[m][0] = src[P * m];
        r.v[m][1] = src[ME_HOP + P * m];
      }
      return r;
    }
#pragma unroll
    for (int m = 0; m < S; ++m) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int i = s0 + h * ME_HOP + P * m + tid;
        if (g.reflect) {
          if (i < 0) i = -i;
          if (i >= g.T) i = 2 * (g.T - 1) - i;
        }
        i = i < 0 ? 0 : (i >= g.T ? g.T - 1 : i);
        r.v[m][h] = ab[i];
      }
    }
    return r;
  };
  // a thread's (filter, frame, quarter) tasks are the same for every pair: their band limits and weight offsets are fetched ONCE
  // (up to four rounds of 256 tasks: 128 filters; more filters read them per pair as before)
  constexpr int PRE = 4;
  const bool pre = 8 * g.n_mels <= PRE * P;
  int b_lo[PRE], b_hi[PRE], b_w[PRE];
#pragma unroll
  for (int r = 0; r < PRE; ++r) {
    const int c = (r * P + tid) >> 3;
    const bool in = pre && c < g.n_mels;
    b_lo[r] = in ? band[4 * c] : 0;
    b_hi[r] = in ? band[4 * c + 1] : 0;
    b_w[r] = in ? band[4 * c + 2] : 0;
  }
  Raw nxt = load_pair(p_first);
  for (int pr = p_first; pr < p_last; ++pr) {
    const int j0 = 2 * pr;
    const bool live1 = j0 + 1 < g.frames;
    const Raw cr = nxt;
    if (pr + 1 < p_last) nxt = load_pair(pr + 1);
    // the two windowed frames: j0 in the real, j0 + 1 in the imaginary part
    f32x2 z[S];
    const int s0 = j0 * ME_HOP - PAD;
    if (interior(pr)) {
#pragma unroll
      for (int m = 0; m < S; ++m) z[m] = f32x2{w[m] * cr.v[m][0], w[m] * cr.v[m][1]};
    } else {
#pragma unroll
      for (int m = 0; m < S; ++m) {
        float v[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          int i = s0 + h * ME_HOP + P * m + tid;
          // zero padding (constant mode) and the frame past the end of an odd count
          const bool ok = (g.reflect || (i >= 0 && i < g.T)) && (h == 0 || live1);
          v[h] = ok ? cr.v[m][h] : 0.f;
        }
        z[m] = f32x2{w[m] * v[0], w[m] * v[1]};
      }
    }
    f32x2* A = ex[cur];
    f32x2* Bx = ex[cur ^ 1];
    cur ^= 1;
    PL::forward(z, tw, A, Bx, tid);
#pragma unroll
    for (int m = 0; m < S; ++m) Bx[P * m + tid] = z[m];         // natural order
    __syncthreads();                                            // ... and A is free (everyone left the last pass)
    // magnitudes of bins k = 256 m + tid (m < 4) and the Nyquist bin (thread 0) of both frames -> LDS (in A)
    float* mags = reinterpret_cast<float*>(A);
#pragma unroll
    for (int m = 0; m < S / 2 + 1; ++m) {
      if (m < S / 2 || tid == 0) {
        const int k = P * m + tid;
        const f32x2 zneg = Bx[(N - k) & (N - 1)];
        const f32x2 a2 = fft::add_conj(z[m], zneg);             // 2 X_j0[k]
        const f32x2 b2 = fft::sub_conj(z[m], zneg);             // 2i X_j0+1[k]
        // (the hardware square root, 1 ulp: the correctly rounded one was ~20 instructions, ten times per thread and pair)
        mags[k] = __builtin_amdgcn_sqrtf(fmaf(0.25f * a2.x, a2.x, 0.25f * a2.y * a2.y) + 1e-9f);          // nvSTFT.py:108
        mags[MROW + k] = __builtin_amdgcn_sqrtf(fmaf(0.25f * b2.x, b2.x, 0.25f * b2.y * b2.y) + 1e-9f);
      }
    }
    __syncthreads();
    // banded mel projection (nvSTFT.py:115-116): four lanes per (filter, frame), every fourth bin of the band each; a round of
    // 256 lanes takes 32 consecutive filters of both frames, so its lanes' bands are about equally long (a Slaney band grows from
    // 2 to 90 bins across the filters: one thread per filter made every wave wait for the longest; 0.1217 -> 0.1209 ms, r06_v37_mel_quad.txt)
    if (pre) {
#pragma unroll
      for (int r = 0; r < PRE; ++r) {
        if (r * P >= 8 * g.n_mels) break;                       // workgroup-uniform
        const int task = r * P + tid, part = task & 3, fc = task >> 2;
        const int c = fc >> 1, h = fc & 1;
        const bool on = c < g.n_mels && (h == 0 || live1);
        float acc = 0.f;
        if (on) {
          const int lo = b_lo[r], hi = b_hi[r];
          const float* mg = mags + h * MROW;
          const float* wr = (g.packed_len > 0 ? wl + b_w[r] : basis + (long)c * BINS) - (g.packed_len > 0 ? lo : 0);
          for (int k = lo + part; k < hi; k += 4) acc = fmaf(wr[k], mg[k], acc);
        }
        acc += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
        acc += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc), 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
        if (on && part == 0) out[(long)b * g.sb + (long)c * g.sm + (long)(j0 + h) * g.sf] = 0.6931471805599453f * __builtin_amdgcn_logf(fmaxf(acc, g.clip));   // (hardware log2, 1 ulp)
      }
    } else
    for (int t0 = 0; t0 < 8 * g.n_mels; t0 += P) {
      const int task = t0 + tid, part = task & 3, fc = task >> 2;
      const int c = fc >> 1, h = fc & 1;
      const bool on = c < g.n_mels && (h == 0 || live1);
      float acc = 0.f;
      if (on) {
        const int lo = band[4 * c], hi = band[4 * c + 1];
        const float* mg = mags + h * MROW;
        if (g.packed_len > 0) {
          const float* wr = wl + band[4 * c + 2] - lo;
          for (int k = lo + part; k < hi; k += 4) acc = fmaf(wr[k], mg[k], acc);
        } else {
          const float* wr = basis + (long)c * BINS;
          for (int k = lo + part; k < hi; k += 4) acc = fmaf(wr[k], mg[k], acc);
        }
      }
      acc += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
      acc += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc), 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
      if (on && part == 0) out[(long)b * g.sb + (long)c * g.sm + (long)(j0 + h) * g.sf] = 0.6931471805599453f * __builtin_amdgcn_logf(fmaxf(acc, g.clip));   // (hardware log2, 1 ulp)
    }
    // no barrier: the next transform writes Bx first (its readers are behind the barrier above) and A -- the
    // magnitudes -- only after its own first barrier, which every thread reaches after its projection loop
  }
}

int launch_mel(const float* audio, int B, int T, const float* window, int n_fft, int hop, const float* basis,
               const int* band, const float* packed, int packed_len, int n_mels, float clip, float* out, long sb,
               long sm, long sf, hipStream_t st) {
  if (n_fft != 2048 || hop != ME_HOP || T < 1 || T >= (1 << 30) || n_mels < 1) return -1;
  MelGeom g;
  const int pad_left = (n_fft - hop) / 2;
  int pad_right = (n_fft - hop + 1) / 2;
  if (n_fft - T - pad_left > pad_right) pad_right = n_fft - T - pad_left;
  g.T = T;
  g.frames = (T + pad_left + pad_right - n_fft) / hop + 1;
  g.pairs = (g.frames + 1) / 2;
  g.reflect = pad_right < T ? 1 : 0;
  g.n_mels = n_mels; g.clip = clip;
  g.packed_len = (packed && packed_len > 0 && packed_len <= ME_WPACK) ? packed_len : 0;
  g.sb = sb; g.sm = sm; g.sf = sf;
  // pairs are independent; a run only amortises the twiddle set-up.  One round of workgroups on the chip.
  int wps = 3;
  if (const long v = knob(KNOB_MEL_WPS)) { if (v >= 1) wps = (int)v; }
  const long slots = (long)wps * 256;
  long per_utt = slots / (B > 0 ? B : 1);
  if (per_utt < 1) per_utt = 1;
  int run = (int)((g.pairs + per_utt - 1) / per_utt);
  if (run < 4) run = 4;
  if (const long v = knob(KNOB_MEL_RUN)) { if (v >= 1) run = (int)v; }
  if (run > g.pairs) run = g.pairs;
  g.run = run;
  g.runs_per_utt = (g.pairs + run - 1) / run;
  const long wgs = (long)B * g.runs_per_utt;
  if (wgs > 0x7fffffffL) return -1;
  if (wps >= 4)
    hipLaunchKernelGGL(k_mel<4>, dim3((unsigned)wgs), dim3(256), 0, st, audio, window, basis, band, packed, out, g);
  else if (wps == 3)
    hipLaunchKernelGGL(k_mel<3>, dim3((unsigned)wgs), dim3(256), 0, st, audio, window, basis, band, packed, out, g);
  else
    hipLaunchKernelGGL(k_mel<2>, dim3((unsigned)wgs), dim3(256), 0, st, audio, window, basis, band, packed, out, g);
  return 0;
}

int mel_frames(int T, int n_fft, int hop) {
  const int pad_left = (n_fft - hop) / 2;
  int pad_right = (n_fft - hop + 1) / 2;
  if (n_fft - T - pad_left > pad_right) pad_right = n_fft - T - pad_left;
  return (T + pad_left + pad_right - n_fft) / hop + 1;
}

}  // namespace ddsp
