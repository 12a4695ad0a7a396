// Short-time spectral filtering: the DSP tails of CombSubFast (ddsp/vocoder.py:758-784) and CombSubSuperFast
// (:661-708), plus the closed-form exciter of the latter (fast_source_gen, :639-651).
//
// Both models frame the exciter and a noise draw with 50 % / 75 % overlap (frames of `win` samples every hop = 512,
// win/2 samples of zero or reflect padding), window them, multiply the one-sided spectra by per-frame complex
// filters exp(mag + i pi phase) predicted by Unit2Control, inverse-transform, window again and overlap-add
// (CombSubSuperFast then divides by the summed squared window, as torch.istft does).  The reference materialises
// [B,F+1,win/2+1] complex spectra four times; here one workgroup walks a run of consecutive frame PAIRS of one
// utterance and nothing but the controls, the two input signals and the output touches HBM:
//   Z_j     = FFT(w e_j + i w u_j)                 exciter and noise frames packed in one complex transform
//   E = (Z[k] + conj Z[-k]) / 2,  U = (Z[k] - conj Z[-k]) / 2i
//   S_j[k]  = E Hs_j[k] + U Hn_j[k]                for k <= win/2, Hermitian-extended above (irfft semantics:
//                                                  the imaginary parts of the DC and Nyquist bins are dropped)
//   y_j + i y_j+1 = IFFT(S_j + i S_j+1)            two real frames per inverse transform
// i.e. 1.5 complex FFTs per frame (fft_r.h).  Overlap-add happens in an LDS ring of `win` samples; because every
// frame starts at a multiple of the thread count, each thread only ever touches ring slots congruent to its id,
// so the ring needs no barriers.  A run starts `WARM` pairs early (discarded) so the ring holds the tails of the
// frames before its first own pair: no atomics, bit-reproducible.
#include "fft_r.h"
#include "kernels.h"
#include <stdlib.h>

namespace ddsp {

using fft::cmul;

constexpr int ST_HOP = 512;

// ------------------------------------------------------------------------------------------------
// fast_source_gen, frame-rate part (vocoder.py:641-647,650): rad_acc[f] = fmod(cumsum_f' rad2[f'], 1) with
// rad2 = fmod(rad_last + 0.5, 1) - 0.5, and phase_frames = 2 pi rad[:, :, 0].  All float32 in the reference's
// operation order; the cumulative sum is ATen's (float64 running sum, float32 outputs) -- the terms are
// multiples of 2^-24 below 1, so the float64 sums are exact in any order and a parallel scan is bit-identical.
// One 256-thread workgroup per utterance.
// ------------------------------------------------------------------------------------------------
struct FastSrc {
  float sr;
  int F, hop;
  // s0 and ds0 of frame f (vocoder.py:641-642)
  __device__ __forceinline__ void frame(const float* __restrict__ f0_row, int f, float& s0, float& ds0) const {
    s0 = f0_row[f] / sr;
    ds0 = 0.f;
    if (f < F - 1) ds0 = f0_row[f + 1] / sr - s0;
  }
  // rad before the accumulated offset, at in-frame position n (vocoder.py:643)
  __device__ __forceinline__ float rad_local(float s0, float ds0, int n) const {
    const float nf = (float)n, n1 = (float)(n + 1);
    const float a = s0 * n1;
    const float b = (((0.5f * ds0) * nf) * n1) / (float)hop;
    return a + b;
  }
};

__global__ void __launch_bounds__(256) k_fast_source_scan(const float* __restrict__ f0_frames, FastSrc cfg,
                                                          float* __restrict__ rad_acc,
                                                          float* __restrict__ phase_frames) {
  __shared__ double wsum[4];
  __shared__ double carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long b = blockIdx.x;
  const float* f0_row = f0_frames + b * cfg.F;
  if (tid == 0) carry_s = 0.0;
  __syncthreads();
  for (int base = 0; base < cfg.F; base += 256) {
    const int f = base + tid;
    float s0 = 0.f, ds0 = 0.f, rad2 = 0.f;
    if (f < cfg.F) {
      cfg.frame(f0_row, f, s0, ds0);
      const float last = cfg.rad_local(s0, ds0, cfg.hop - 1);
      rad2 = fmodf(last + 0.5f, 1.0f) - 0.5f;                                      // :645
    }
    const double v = (double)rad2;
    const double excl = wave_excl_scan(v, lane);
    if (lane == 63) wsum[wave] = excl + v;
    __syncthreads();
    double pre = carry_s;
    for (int w = 0; w < wave; ++w) pre += wsum[w];
    const double incl = pre + excl + v;                                           // running sum including frame f
    const double before = pre + excl;                                             // ... up to frame f - 1
    if (f < cfg.F) {
      rad_acc[b * cfg.F + f] = fmodf((float)incl, 1.0f);                            // :646
      const float shifted = f > 0 ? fmodf((float)before, 1.0f) : 0.0f;             // F.pad(rad_acc[:, :-1]), :647
      float rad = cfg.rad_local(s0, ds0, 0) + shifted;
      rad = rad - rintf(rad);                                                      // :648
      if (phase_frames) phase_frames[b * cfg.F + f] = kTwoPiF * rad;               // :650
    }
    __syncthreads();
    if (tid == 255) carry_s = incl;
    __syncthreads();
  }
}

// combtooth = sinc(rad / (s0 + 1e-5)) (vocoder.py:643-649); 4 consecutive samples per thread.  When the hop is a
// multiple of 4 the four samples share their frame, so s0 / ds0 (two float32 divisions) are formed once.
__global__ void __launch_bounds__(256) k_fast_combtooth(const float* __restrict__ f0_frames,
                                                        const float* __restrict__ rad_acc, FastSrc cfg, long total,
                                                        float* __restrict__ out) {
  const long i0 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i0 >= total) return;
  const long T = (long)cfg.F * cfg.hop;
  float v[4];
  auto sample = [&](float s0, float ds0, float shift, int n) -> float {
    float rad = cfg.rad_local(s0, ds0, n);
    const float s0n = s0 + (ds0 * (float)n) / (float)cfg.hop;                      // :644
    rad = rad + shift;                                                             // :647
    rad = rad - rintf(rad);                                                        // :648
    return sinc_f32(div_pos(rad, s0n + 1e-5f));                                    // :649
  };
  if ((cfg.hop & 3) == 0 && i0 + 3 < total) {
    const long b = i0 / T;
    const int t = (int)(i0 - b * T);
    const int f = t / cfg.hop, n = t - f * cfg.hop;
    float s0, ds0;
    cfg.frame(f0_frames + b * cfg.F, f, s0, ds0);
    const float shift = f > 0 ? rad_acc[b * cfg.F + f - 1] : 0.0f;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = sample(s0, ds0, shift, n + r);
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long i = i0 + r;
      v[r] = 0.f;
      if (i < total) {
        const long b = i / T;
        const int t = (int)(i - b * T);
        const int f = t / cfg.hop, n = t - f * cfg.hop;
        float s0, ds0;
        cfg.frame(f0_frames + b * cfg.F, f, s0, ds0);
        v[r] = sample(s0, ds0, f > 0 ? rad_acc[b * cfg.F + f - 1] : 0.0f, n);
      }
    }
  }
  if (i0 + 3 < total && (reinterpret_cast<uintptr_t>(out + i0) & 15) == 0) {
    *reinterpret_cast<float4*>(out + i0) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
    for (int r = 0; r < 4 && i0 + r < total; ++r) out[i0 + r] = v[r];
  }
}

// The same for hop % 4 == 0 (every shipped configuration) without the index arithmetic of the general form: the grid's
// second dimension is the utterance (no 64-bit division of a flat sample index), the frame of a thread's four samples
// is a shift when the hop is a power of two, and the two divisions by float(hop) of the recipe (vocoder.py:643-644)
// are then exact multiplications by its reciprocal -- bit-identical, and 2.5 instead of 5.5 division sequences per
// sample.  The divisions that round (f0 / sr, rad / (s0 + 1e-5)) stay IEEE divisions.
template <bool POW2>
__global__ void __launch_bounds__(256) k_fast_combtooth4(const float* __restrict__ f0_frames,
                                                         const float* __restrict__ rad_acc, FastSrc cfg, int T,
                                                         int shift, float* __restrict__ out) {
  const int t = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (t >= T) return;
  const long b = blockIdx.y;
  const int f = POW2 ? t >> shift : (int)((unsigned)t / (unsigned)cfg.hop);
  const int n = t - f * cfg.hop;
  float s0, ds0;
  cfg.frame(f0_frames + b * cfg.F, f, s0, ds0);
  const float acc = f > 0 ? rad_acc[b * cfg.F + f - 1] : 0.0f;
  const float hopf = (float)cfg.hop, rhop = 1.0f / hopf;      // rhop is exact for POW2, unused otherwise
  float v[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float nf = (float)(n + r), n1 = (float)(n + r + 1);
    const float q = ((0.5f * ds0) * nf) * n1;
    float rad = s0 * n1 + (POW2 ? q * rhop : q / hopf);                              // :643
    const float dn = ds0 * nf;
    const float s0n = s0 + (POW2 ? dn * rhop : dn / hopf);                           // :644
    rad = rad + acc;                                                                 // :647
    rad = rad - rintf(rad);                                                          // :648
    v[r] = sinc_f32(div_pos(rad, s0n + 1e-5f));                                      // :649
  }
  *reinterpret_cast<float4*>(out + b * (long)T + t) = make_float4(v[0], v[1], v[2], v[3]);
}

// ------------------------------------------------------------------------------------------------
// the spectral filter itself
// ------------------------------------------------------------------------------------------------
// (cos, sin) of pi * p on v_cos_f32 / v_sin_f32, which take revolutions; fract keeps the argument in their range
__device__ __forceinline__ f32x2 cis_pi(float p) {
  const float rev = __builtin_amdgcn_fractf(0.5f * p);
  return f32x2{__builtin_amdgcn_cosf(rev), __builtin_amdgcn_sinf(rev)};
}
struct StftGeom {
  int F, T;               // control frames, samples per utterance (F * hop)
  int pairs;              // frame pairs per utterance: ceil((F + 1) / 2)   (frames 0..F)
  int run, runs_per_utt;  // own pairs per workgroup
  int reflect;            // padding: reflect (torch.stft, vocoder.py:667-670) or zeros (:766)
  int normalize;          // divide by the overlap-added squared window (torch.istft)
  int noise_u01;          // noise holds a U[0,1) draw: apply 2u - 1 on load (vocoder.py:771)
  float noise_scale;      // 1/128 (vocoder.py:663,760)
  long ld_hm, ld_hp, ld_nm, ld_np;
  // EXC kernels (streaming shapes): the exciter is not read but made where a frame's samples are fetched -- k_fast_combtooth4<true>'s
  // arithmetic, sample by sample (vocoder.py:643-649)
  const float* exc_f0; const float* exc_acc;
  float exc_sr;
};

// combtooth sample i of an utterance (hop 512), bit for bit what k_fast_combtooth4<true> stores at i
__device__ __forceinline__ float fast_combtooth_at(const float* __restrict__ f0_row, const float* __restrict__ acc_row, int F, float sr, int i) {
  const int f = i >> 9, n = i & (ST_HOP - 1);
  const float s0 = f0_row[f] / sr;
  float ds0 = 0.f;
  if (f < F - 1) ds0 = f0_row[f + 1] / sr - s0;
  const float acc = f > 0 ? acc_row[f - 1] : 0.0f;
  const float rhop = 1.0f / (float)ST_HOP;
  const float nf = (float)n, n1 = (float)(n + 1);
  const float q = ((0.5f * ds0) * nf) * n1;
  float rad = s0 * n1 + q * rhop;                                                      // :643
  const float s0n = s0 + (ds0 * nf) * rhop;                                            // :644
  rad = rad + acc;                                                                     // :647
  rad = rad - rintf(rad);                                                              // :648
  return sinc_f32(div_pos(rad, s0n + 1e-5f));                                          // :649
}

// WPS = waves per SIMD the kernel is compiled for (HIP's second __launch_bounds__ argument): register budget 512 / WPS
template <int R, int WPS, bool EXC = false>
__global__ void __launch_bounds__(64 * R, WPS)
k_stft_filter(const float* __restrict__ exc, const float* __restrict__ noise, const float* __restrict__ c_hmag,
              const float* __restrict__ c_hphase, const float* __restrict__ c_nmag,
              const float* __restrict__ c_nphase, const float* __restrict__ window, float* __restrict__ out,
              StftGeom g) {
  using PL = fft::Plan<R>;
  constexpr int N = PL::N, P = PL::P, S = 8;
  constexpr int PAD = N / 2;
  constexpr int EMIT = ST_HOP / P;                             // ring slots per thread that complete per frame
  constexpr int OVL = N / ST_HOP;                              // frames overlapping one sample
  constexpr int WARM = (OVL - 1 + 1) / 2;                      // warm-up pairs: OVL - 1 earlier frames reach in
  __shared__ __attribute__((aligned(16))) f32x2 ex[2][N];       // ping-pong exchange buffers; roles swap per transform
  // Overlap-add ring of `win` samples: a thread only ever touches the 8 slots congruent to its id, so they live in
  // registers.  Frame j's sample P m + tid is slot (aj / P + m) mod 8 with aj = j hop - win/2: a frame pair advances the
  // ring by 1024 samples = 8 slots for win 1024 (every index a compile-time constant) and 4 slots for win 2048 (the two
  // halves of the register array are swapped after every pair instead).
  float ring[S];
  const int tid = threadIdx.x;
  const int b = blockIdx.x / g.runs_per_utt;
  // an utterance's pairs are split evenly over its runs, and the run number is rotated by the utterance number so that
  // the longer and the shorter runs spread over XCDs and CUs (as in k_fir_blk, fir_blk.hip)
  const int run_no = (int)((blockIdx.x - b * g.runs_per_utt + b + (b >> 4) + (b >> 8)) % g.runs_per_utt);
  const int p_first = (int)(((long)run_no * g.pairs) / g.runs_per_utt);
  const int p_last = (int)(((long)(run_no + 1) * g.pairs) / g.runs_per_utt);
  const long ob = (long)b * g.T;
  const float* eb = EXC ? nullptr : exc + ob;
  const float* nb = noise + ob;
  const float* xf0 = EXC ? g.exc_f0 + (long)b * g.F : nullptr;
  const float* xacc = EXC ? g.exc_acc + (long)b * g.F : nullptr;

  typename PL::Tw tw;
  tw.init(tid);
  float w[S];
#pragma unroll
  for (int m = 0; m < S; ++m) {
    w[m] = window[P * m + tid];
    ring[m] = 0.f;
  }
  // istft envelope of the thread's EMIT samples per frame when all OVL frames that reach them exist (same fma order as the
  // general form below), and its reciprocal
  float env_all[EMIT], renv_all[EMIT];
#pragma unroll
  for (int m = 0; m < EMIT; ++m) {
    float env = 0.f;
#pragma unroll
    for (int q = 0; q < OVL; ++q) env = fmaf(w[m + q * EMIT], w[m + q * EMIT], env);
    env_all[m] = env;
    renv_all[m] = 1.0f / env;
  }
  const float cs = 0.5f / (float)N;                            // E = (..)/2, U = (..)/2i and the 1/N of the inverse
  const float cn = cs * g.noise_scale;
  int cur = 0;                                                 // ex[cur] plays "A" of the next transform

  const int pr0 = p_first > WARM ? p_first - WARM : 0;
  // slot of sample index 0 of the pair's first frame, as seen by the register array: aj / P = (2 pr hop - win/2) / P.
  // win 1024: 8 pr - 4 -> 4 for every pair.  win 2048: 4 pr - 4 -> 4 or 0; the array is kept rotated so that it is 4
  // at the top of every pair (an odd first pair starts rotated).
  constexpr int BASE = 4;
  constexpr int STEP = ST_HOP / P;                             // slots between the two frames of a pair
  // the waves that share a SIMD (of different workgroups) take turns at its arbiter's priority, pair by pair (fir_blk.hip)
  const int turn = __builtin_amdgcn_s_getreg(0x1804) & 1;      // HW_ID[3:0]: wave slot within the SIMD
  // raw samples of the NEXT frame (exciter, noise): fetched while the current frame is transformed -- consumed where they
  // were issued, their latency sat on every frame with only the SIMD's other wave to hide it
  float ne[S], nu[S];
  auto fetch_frame = [&](int j) {
    const int s0 = j * ST_HOP - PAD;
    if (s0 >= 0 && s0 + N <= g.T) {                              // interior frame (wave-uniform)
      const float* ef = EXC ? nullptr : eb + s0 + tid;
      const float* nf = nb + s0 + tid;
#pragma unroll
      for (int m = 0; m < S; ++m) {
        ne[m] = EXC ? fast_combtooth_at(xf0, xacc, g.F, g.exc_sr, s0 + P * m + tid) : ef[P * m];
        nu[m] = nf[P * m];
      }
    } else {                                                     // clamped (and reflected) addresses; masked where they are used
#pragma unroll
      for (int m = 0; m < S; ++m) {
        int i = s0 + P * m + tid;
        if (g.reflect) {
          if (i < 0) i = -i;
          if (i >= g.T) i = 2 * (g.T - 1) - i;
        }
        i = i < 0 ? 0 : (i >= g.T ? g.T - 1 : i);
        ne[m] = EXC ? fast_combtooth_at(xf0, xacc, g.F, g.exc_sr, i) : eb[i];
        nu[m] = nb[i];
      }
    }
  };
  fetch_frame(2 * pr0);
  for (int pr = pr0; pr < p_last; ++pr) {
    if ((pr + turn) & 1) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
    f32x2 V[S];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int j = 2 * pr + h;                                // frame index, 0..F (F + 1 only pads an odd count)
      const bool live = j <= g.F;
      const int row = j < g.F ? j : g.F - 1;                   // last filter frame repeated (vocoder.py:662,664)
      const long rb = (long)b * g.F + row;
      // raw controls of the bins this thread evaluates: k = P m + tid for m < 4, plus the Nyquist bin on thread 0
      // (the upper half of the spectrum is the conjugate mirror, handed over through LDS below)
      // (loads are unconditional, from clamped addresses, and masked where they are used: a load feeding a select
      // right away is waited for on the spot instead of staying in flight across the transform)
      constexpr int NB = S / 2 + 1;
      float hm[NB], hp[NB], nm[NB], np_[NB];
#pragma unroll
      for (int m = 0; m < NB; ++m) {
        int k = P * m + tid;
        k = k > N / 2 ? N / 2 : k;                             // m = NB - 1 is the Nyquist bin, used by thread 0 only
        hm[m] = c_hmag[rb * g.ld_hm + k];
        hp[m] = c_hphase[rb * g.ld_hp + k];
        nm[m] = c_nmag[rb * g.ld_nm + k];
        np_[m] = 0.f;
      }
      if (c_nphase) {
#pragma unroll
        for (int m = 0; m < NB; ++m) {
          int k = P * m + tid;
          k = k > N / 2 ? N / 2 : k;
          np_[m] = c_nphase[rb * g.ld_np + k];
        }
      }
      // windowed input frame: exciter in the real, noise in the imaginary part -- from the samples fetched one frame ago
      f32x2 z[S];
      const int s0 = j * ST_HOP - PAD;
      if (s0 >= 0 && s0 + N <= g.T) {                            // interior frame (wave-uniform): no edge handling
#pragma unroll
        for (int m = 0; m < S; ++m) {
          float u = nu[m];
          if (g.noise_u01) u = fmaf(2.0f, u, -1.0f);
          z[m] = f32x2{w[m] * ne[m], w[m] * u};
        }
      } else {
#pragma unroll
        for (int m = 0; m < S; ++m) {
          const int i = s0 + P * m + tid;
          const bool ok = live && (g.reflect || (i >= 0 && i < g.T));
          float u = nu[m];
          if (g.noise_u01) u = fmaf(2.0f, u, -1.0f);
          z[m] = f32x2{ok ? w[m] * ne[m] : 0.f, ok ? w[m] * u : 0.f};
        }
      }
      fetch_frame(j + 1);                                        // in flight during this frame's transform
      f32x2* A = ex[cur];
      f32x2* Bx = ex[cur ^ 1];
      cur ^= 1;
      PL::forward(z, tw, A, Bx, tid);
#pragma unroll
      for (int m = 0; m < S; ++m) Bx[P * m + tid] = z[m];       // natural order, buffer B is free
      __syncthreads();                                          // ... and now A is free too (everyone left pass 4)
#pragma unroll
      for (int m = 0; m < NB; ++m) {
        const int k = P * m + tid;
        const f32x2 zneg = Bx[(N - k) & (N - 1)];
        const f32x2 e2 = fft::add_conj(z[m], zneg);             // 2 E[k]
        const f32x2 u2 = fft::sub_conj(z[m], zneg);             // 2i U[k]
        // filters exp(mag) (cos(pi ph) + i sin(pi ph)); the noise one carries U's factor -i
        const f32x2 ch = cis_pi(hp[m]), cz = cis_pi(np_[m]);
        const float ah = cs * exp_hw(hm[m]);
        const float an = cn * exp_hw(nm[m]);
        const f32x2 Hs = {ah * ch.x, ah * ch.y};
        const f32x2 Hn = {an * cz.y, -an * cz.x};
        f32x2 s = cmul(e2, Hs) + cmul(u2, Hn);
        if (!live) s = f32x2{0.f, 0.f};
        if (m < NB - 1) {
          if (m == 0 && tid == 0) s.y = 0.f;                    // irfft ignores Im(DC)
          A[(N - k) & (N - 1)] = fft::cconj(s);                 // S[N - k] = conj S[k] for its owner (slot 7 - m)
          if (h == 0) V[m] = s;
          else V[m] = fft::conj_minus_i_conj(V[m], s);          // V = S_j + i S_j+1, conjugated for the inverse
        } else if (tid == 0) {
          s.y = 0.f;                                            // ... and Im(Nyquist)
          if (h == 0) V[m] = s;
          else V[m] = fft::conj_minus_i_conj(V[m], s);
        }
      }
      __syncthreads();
#pragma unroll
      for (int m = NB - 1; m < S; ++m) {
        if (m == NB - 1 && tid == 0) continue;                  // the Nyquist bin was evaluated above
        const f32x2 s = A[P * m + tid];
        if (h == 0) V[m] = s;
        else V[m] = fft::conj_minus_i_conj(V[m], s);
      }
      // no barrier: the next transform writes Bx in its first pass (its readers are behind the barrier above) and
      // A only after its own first barrier
    }
    {
      f32x2* A = ex[cur];
      f32x2* Bx = ex[cur ^ 1];
      cur ^= 1;
      PL::forward(V, tw, A, Bx, tid);
    }
    // ifft(V) = conj(FFT(conj V)): y_j = Re, y_j+1 = -Im (1/N folded into the filters)
    const int a0 = 2 * pr * ST_HOP - PAD;                      // output position of frame 2 pr's first sample
    const bool own = pr >= p_first;
    if (R == 4 && pr != g.pairs - 1) {                         // (win 1024 keeps the general form: the split measured 19 % slower there)
      // Every pair but an utterance's last: a frame completes the EMIT samples per thread at [aj, aj + hop), which lie
      // inside [0, T) for every lane or (aj < 0: the first frames) for none -- workgroup-uniform, so no per-lane range
      // checks; and away from the utterance's start every frame that reaches a sample exists, so the istft envelope is
      // the thread's constant (env_all) and the division one reciprocal-multiply with a fused residual correction.
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int j = 2 * pr + h;
        const int aj = a0 + h * ST_HOP;
#pragma unroll
        for (int m = 0; m < S; ++m) {
          const float y = h == 0 ? V[m].x : -V[m].y;
          ring[(BASE + h * STEP + m) & 7] += y * w[m];
        }
        const bool store = own && aj >= 0;
        const bool all_frames = j >= OVL - 1;                   // frames j - q, q < OVL, all exist (j <= F holds here)
#pragma unroll
        for (int m = 0; m < EMIT; ++m) {
          const int ri = (BASE + h * STEP + m) & 7;
          float v = ring[ri];
          ring[ri] = 0.f;
          if (store) {
            if (g.normalize) {
              if (all_frames) {
                const float q0 = v * renv_all[m];
                v = fmaf(fmaf(-q0, env_all[m], v), renv_all[m], q0);
              } else {
                float env = 0.f;
#pragma unroll
                for (int q = 0; q < OVL; ++q) {
                  const float wq = w[m + q * EMIT];
                  if (j - q >= 0) env = fmaf(wq, wq, env);
                }
                v = v / env;
              }
            }
            out[ob + aj + P * m + tid] = v;
          }
        }
      }
    } else {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int j = 2 * pr + h;
        const int aj = a0 + h * ST_HOP;
#pragma unroll
        for (int m = 0; m < S; ++m) {
          const float y = h == 0 ? V[m].x : -V[m].y;
          ring[(BASE + h * STEP + m) & 7] += y * w[m];
        }
        // samples [aj, aj + hop) have now seen every frame that reaches them; the last pair flushes the rest
        const int n_emit = (pr == g.pairs - 1 && h == 1) ? S : EMIT;
#pragma unroll
        for (int m = 0; m < S; ++m) {
          if (m < n_emit) {
            const int t = aj + P * m + tid;
            const int ri = (BASE + h * STEP + m) & 7;
            float v = ring[ri];
            ring[ri] = 0.f;
            if (own && t >= 0 && t < g.T) {
              if (g.normalize) {
                // summed squared window over the frames jf = j + (m / EMIT) - q that exist (0..F) and cover t
                float env = 0.f;
#pragma unroll
                for (int q = 0; q < OVL; ++q) {
                  const int jf = j + m / EMIT - q;
                  const float wq = w[(m % EMIT) + q * EMIT];
                  if (jf >= 0 && jf <= g.F) env = fmaf(wq, wq, env);
                }
                v = v / env;
              }
              out[ob + t] = v;
            }
          }
        }
      }
    }
    if (R == 4) {                                              // win 2048: the next pair's first frame starts 4 slots further on
#pragma unroll
      for (int m = 0; m < 4; ++m) { const float tmp = ring[m]; ring[m] = ring[m + 4]; ring[m + 4] = tmp; }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// gradient of the spectral filter w.r.t. its four control streams (what autograd returns for the controls of
// CombSubFast / CombSubSuperFast.forward given dL/dsignal):
//   gamma_j = w * (grad_out / env)[frame j]     adjoint of crop, envelope division, synthesis window and overlap-add
//   G_j[k]  = c_k / N * rfft(gamma_j)[k]        adjoint of irfft (c_0 = c_N/2 = 1 with the imaginary part dropped, else 2)
//   dL/dmag = Re(conj(G) E H),  dL/dphase = -pi Im(conj(G) E H)     per filter (E: exciter or noise frame spectrum)
// Same machinery as the forward kernel: exciter and noise frames share one transform, the cotangent frames of a
// pair share another (1.5 transforms per frame), no inverse transform and no overlap-add, so pairs are independent.
// The repeated last filter frame (vocoder.py:662,664) adds frame F's gradient onto control row F-1: the workgroup
// that owns frame F-1 processes frame F right after it and accumulates into the values it just wrote.
// ------------------------------------------------------------------------------------------------
template <int R, int WPS>
__global__ void __launch_bounds__(64 * R, WPS)
k_stft_filter_bwd(const float* __restrict__ exc, const float* __restrict__ noise, const float* __restrict__ c_hmag,
                  const float* __restrict__ c_hphase, const float* __restrict__ c_nmag,
                  const float* __restrict__ c_nphase, const float* __restrict__ window,
                  const float* __restrict__ grad_out, float* d_hmag, float* d_hphase, float* d_nmag, float* d_nphase,
                  StftGeom g) {
  using PL = fft::Plan<R>;
  constexpr int N = PL::N, P = PL::P, S = 8;
  constexpr int PAD = N / 2;
  constexpr int EMIT = ST_HOP / P;
  constexpr int OVL = N / ST_HOP;
  constexpr int NB = S / 2 + 1;                                // bins per thread: k = P m + tid (m < 4) + Nyquist on thread 0
  constexpr int NBINS = N / 2 + 1;
  __shared__ __attribute__((aligned(16))) f32x2 ex[2][N];
  const int tid = threadIdx.x;
  const int b = blockIdx.x / g.runs_per_utt;
  const int run_no = blockIdx.x - b * g.runs_per_utt;
  const int p_first = run_no * g.run;
  int p_last = p_first + g.run;
  if (p_last > g.pairs) p_last = g.pairs;                      // pairs cover frames 0..F-1 here
  const long ob = (long)b * g.T;
  const float* eb = exc + ob;
  const float* nb = noise + ob;
  const float* gb = grad_out + ob;

  typename PL::Tw tw;
  tw.init(tid);
  float w[S];
#pragma unroll
  for (int m = 0; m < S; ++m) w[m] = window[P * m + tid];
  f32x2* A = ex[0];
  f32x2* Bx = ex[1];

  // natural-order copy of a transform in Bx, one barrier: the next transform may write A at once (its last-pass
  // readers are through) and Bx after its own first barrier
  auto transform = [&](f32x2 (&z)[S]) {
    PL::forward(z, tw, A, Bx, tid);
#pragma unroll
    for (int m = 0; m < S; ++m) Bx[P * m + tid] = z[m];
    __syncthreads();
  };
  // Raw inputs of the NEXT stage -- the two cotangent frames of a pair, or a frame's exciter and noise samples -- fetched
  // while the current stage's transform runs (consumed where they were issued, their latency sat on each of the three
  // stages of a pair).  Clamped (and reflected) addresses; masked where they are used.
  float pf0[S], pf1[S];
  auto fetch_cotangent = [&](int ja) {
#pragma unroll
    for (int m = 0; m < S; ++m) {
      const int t = ja * ST_HOP - PAD + P * m + tid, t1 = t + ST_HOP;
      pf0[m] = gb[t < 0 ? 0 : (t >= g.T ? g.T - 1 : t)];
      pf1[m] = gb[t1 < 0 ? 0 : (t1 >= g.T ? g.T - 1 : t1)];
    }
  };
  auto fetch_frame = [&](int jj) {
    const int s0 = jj * ST_HOP - PAD;
#pragma unroll
    for (int m = 0; m < S; ++m) {
      int i = s0 + P * m + tid;
      if (g.reflect) {
        if (i < 0) i = -i;
        if (i >= g.T) i = 2 * (g.T - 1) - i;
      }
      i = i < 0 ? 0 : (i >= g.T ? g.T - 1 : i);
      pf0[m] = eb[i];
      pf1[m] = nb[i];
    }
  };
  // windowed cotangent of frame jj at slot m from its raw value: w * grad / env, zero outside the cropped range
  auto gamma = [&](int jj, int m, float v) -> float {
    const int t = jj * ST_HOP - PAD + P * m + tid;
    const bool ok = jj <= g.F && t >= 0 && t < g.T;
    if (g.normalize) {
      float env = 0.f;
#pragma unroll
      for (int q = 0; q < OVL; ++q) {
        const int jf = jj + m / EMIT - q;
        const float wq = w[(m % EMIT) + q * EMIT];
        if (jf >= 0 && jf <= g.F) env = fmaf(wq, wq, env);
      }
      v = v / (ok ? env : 1.0f);
    }
    return ok ? w[m] * v : 0.f;
  };
  // Gs = c_k / (4 N) * 2 Gamma for the thread's bins of frames ja (real part of the packed transform) and ja + 1; the
  // raw cotangent is in pf0 / pf1, and next() issues the following stage's fetch
  auto cotangent_pair = [&](int ja, f32x2 (&G0)[NB], f32x2 (&G1)[NB], auto&& next) {
    f32x2 z[S];
#pragma unroll
    for (int m = 0; m < S; ++m) z[m] = f32x2{gamma(ja, m, pf0[m]), gamma(ja + 1, m, pf1[m])};
    next();
    transform(z);
#pragma unroll
    for (int m = 0; m < NB; ++m) {
      const int k = P * m + tid;
      const f32x2 zneg = Bx[(N - k) & (N - 1)];
      const f32x2 p = fft::add_conj(z[m], zneg);                // 2 Gamma_ja
      const f32x2 d = fft::sub_conj(z[m], zneg);                // 2i Gamma_ja+1
      const bool edge = (m == 0 && tid == 0) || m == NB - 1;    // DC and Nyquist bins
      const float c = (edge ? 0.25f : 0.5f) / (float)N;
      G0[m] = f32x2{p.x * c, edge ? 0.f : p.y * c};
      G1[m] = f32x2{d.y * c, edge ? 0.f : -d.x * c};            // d / i
    }
  };
  // one frame: spectra of exciter and noise (raw samples in pf0 / pf1), filters, products, store (accumulate: add onto what
  // this thread wrote)
  auto frame = [&](int jj, const f32x2 (&Gs)[NB], bool accumulate, auto&& next) {
    const int row = jj < g.F ? jj : g.F - 1;
    const long rb = (long)b * g.F + row;
    float hm[NB], hp[NB], nm[NB], np_[NB];
#pragma unroll
    for (int m = 0; m < NB; ++m) {                             // unconditional, clamped; masked at use (see the forward kernel)
      int k = P * m + tid;
      k = k > N / 2 ? N / 2 : k;
      hm[m] = c_hmag[rb * g.ld_hm + k];
      hp[m] = c_hphase[rb * g.ld_hp + k];
      nm[m] = c_nmag[rb * g.ld_nm + k];
      np_[m] = 0.f;
    }
    if (c_nphase) {
#pragma unroll
      for (int m = 0; m < NB; ++m) {
        int k = P * m + tid;
        k = k > N / 2 ? N / 2 : k;
        np_[m] = c_nphase[rb * g.ld_np + k];
      }
    }
    f32x2 z[S];
    const int s0 = jj * ST_HOP - PAD;
#pragma unroll
    for (int m = 0; m < S; ++m) {
      const int i = s0 + P * m + tid;
      const bool ok = g.reflect || (i >= 0 && i < g.T);
      float u = pf1[m];
      if (g.noise_u01) u = fmaf(2.0f, u, -1.0f);
      z[m] = f32x2{ok ? w[m] * pf0[m] : 0.f, ok ? w[m] * u : 0.f};
    }
    next();
    transform(z);
#pragma unroll
    for (int m = 0; m < NB; ++m) {
      if (m < NB - 1 || tid == 0) {
        const int k = P * m + tid;
        const f32x2 zneg = Bx[(N - k) & (N - 1)];
        const f32x2 e2 = fft::add_conj(z[m], zneg);             // 2 E[k]
        const f32x2 u2 = fft::sub_conj(z[m], zneg);             // 2i U[k]
        const f32x2 ch = cis_pi(hp[m]), cz = cis_pi(np_[m]);
        const float ah = exp_hw(hm[m]);
        const float an = g.noise_scale * exp_hw(nm[m]);
        const f32x2 a = cmul(e2, f32x2{ah * ch.x, ah * ch.y});    // 2 E Hs
        const f32x2 n = cmul(u2, f32x2{an * cz.y, -an * cz.x});   // 2 U Hn  (U = u2 / 2i)
        // conj(G) x: real part G.x x.x + G.y x.y, imaginary part G.x x.y - G.y x.x
        const float gx = Gs[m].x, gy = Gs[m].y;
        float v0 = fmaf(gx, a.x, gy * a.y);
        float v1 = -kPiF * fmaf(gx, a.y, -(gy * a.x));
        float v2 = fmaf(gx, n.x, gy * n.y);
        float v3 = -kPiF * fmaf(gx, n.y, -(gy * n.x));
        const long o = rb * NBINS + k;
        if (accumulate) {
          v0 += d_hmag[o];
          v1 += d_hphase[o];
          v2 += d_nmag[o];
          if (d_nphase) v3 += d_nphase[o];
        }
        d_hmag[o] = v0;
        d_hphase[o] = v1;
        d_nmag[o] = v2;
        if (d_nphase) d_nphase[o] = v3;
      }
    }
  };

  if (p_first < p_last) fetch_cotangent(2 * p_first);
  for (int pr = p_first; pr < p_last; ++pr) {
    const int j0 = 2 * pr;
    const bool last = pr == g.pairs - 1;                         // the utterance's last pair: frame F follows (see the head comment)
    const bool more = pr + 1 < p_last;
    f32x2 G0[NB], G1[NB];
    cotangent_pair(j0, G0, G1, [&] { fetch_frame(j0); });
    if (j0 + 1 < g.F) {
      frame(j0, G0, false, [&] { fetch_frame(j0 + 1); });
      frame(j0 + 1, G1, false, [&] {
        if (last) fetch_cotangent(g.F);
        else if (more) fetch_cotangent(j0 + 2);
      });
    } else {                                                     // F odd: j0 = F - 1 is the last frame of its own
      frame(j0, G0, false, [&] { fetch_frame(g.F); });
    }
    if (last) {
      // the repeated last frame F: its cotangent is G1 when F is odd (frame F = j0 + 1), else a transform of its own
      if (j0 + 1 == g.F) {
        frame(g.F, G1, true, [] {});
      } else {
        f32x2 GF[NB], Gdrop[NB];
        cotangent_pair(g.F, GF, Gdrop, [&] { fetch_frame(g.F); });
        frame(g.F, GF, true, [] {});
      }
    }
  }
}

// ---- launchers -----------------------------------------------------------------------------------
int launch_fast_combtooth(const float* f0_frames, const float* rad_acc, int B, int F, int hop, double sr, float* out,
                          hipStream_t st) {
  if ((long)F * hop >= (1L << 30)) return -1;
  FastSrc cfg;
  cfg.sr = (float)sr; cfg.F = F; cfg.hop = hop;
  const long total = (long)B * F * hop;
  if ((hop & 3) == 0 && B <= 65535 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
    const int T = F * hop;
    int shift = 0;
    while ((1 << shift) < hop) ++shift;
    const bool pow2 = (1 << shift) == hop;
    const dim3 grid((unsigned)((T + 1023) / 1024), (unsigned)B);
    if (pow2) hipLaunchKernelGGL(k_fast_combtooth4<true>, grid, dim3(256), 0, st, f0_frames, rad_acc, cfg, T, shift, out);
    else hipLaunchKernelGGL(k_fast_combtooth4<false>, grid, dim3(256), 0, st, f0_frames, rad_acc, cfg, T, shift, out);
    return 0;
  }
  const long blocks = (total + 1023) / 1024;
  if (blocks > 0x7fffffffL) return -1;
  hipLaunchKernelGGL(k_fast_combtooth, dim3((unsigned)blocks), dim3(256), 0, st, f0_frames, rad_acc, cfg, total, out);
  return 0;
}

int launch_fast_source(const float* f0_frames, int B, int F, int hop, double sr, float* rad_acc, float* phase_frames,
                       float* combtooth, hipStream_t st) {
  if ((long)F * hop >= (1L << 30)) return -1;
  FastSrc cfg;
  cfg.sr = (float)sr; cfg.F = F; cfg.hop = hop;
  hipLaunchKernelGGL(k_fast_source_scan, dim3((unsigned)B), dim3(256), 0, st, f0_frames, cfg, rad_acc, phase_frames);
  if (combtooth) return launch_fast_combtooth(f0_frames, rad_acc, B, F, hop, sr, combtooth, st);
  return 0;
}

int launch_stft_filter(const float* exc, const float* noise, int noise_is_u01, const float* c_hmag, long ld_hm,
                       const float* c_hphase, long ld_hp, const float* c_nmag, long ld_nm, const float* c_nphase,
                       long ld_np, float noise_scale, const float* window, int win, int reflect, int normalize, int B,
                       int F, int hop, float* out, hipStream_t st, const float* exc_f0, const float* exc_acc, double exc_sr) {
  if (hop != ST_HOP || (win != 1024 && win != 2048) || (long)F * hop >= (1L << 30)) return -1;
  if (exc_f0 && (win != 2048 || !exc_acc)) return -1;          // the inline exciter: the 2048-point kernel only
  StftGeom g;
  g.exc_f0 = exc_f0; g.exc_acc = exc_acc; g.exc_sr = (float)exc_sr;
  g.F = F; g.T = F * hop;
  g.pairs = (F + 2) / 2;
  g.reflect = reflect; g.normalize = normalize; g.noise_u01 = noise_is_u01; g.noise_scale = noise_scale;
  g.ld_hm = ld_hm; g.ld_hp = ld_hp; g.ld_nm = ld_nm; g.ld_np = ld_np;
  // waves per SIMD the kernel variant is compiled for (register budget = 512 / wps): the natural register demand is
  // ~190, and a spilled control value has to be waited for the moment it is loaded instead of staying in flight
  // across the transform, so 2 waves per SIMD for both sizes (win 2048: 0.30 ms against 0.39 ms at 168 VGPRs with
  // 16 spills, 0.61 ms at 128; profiles/r01_v5_stft_variants.json)
  int wps = 2;
  if (const long v = knob(KNOB_STFT_WPS)) { if (v >= 1) wps = (int)v; }
  const int wg_per_cu = wps * 4 / (win == 2048 ? 4 : 2);
  const int warm = win == 2048 ? 2 : 1;
  // run length: as many workgroups as the chip holds at once, so all of them run in one round with equal work (a
  // partial second round costs more than the longer runs; every run pays its warm-up)
  const long slots = (long)wg_per_cu * 256;
  long per_utt = slots / (B > 0 ? B : 1);
  if (per_utt < 1) per_utt = 1;
  int run = (int)((g.pairs + per_utt - 1) / per_utt);
  // at least three times the warm-up per workgroup -- except at streaming shapes (gui.py:118-133: B = 1, a fraction of a second),
  // where every workgroup is resident at once anyway and the launch is as long as its longest workgroup: one pair each (round 6;
  // six pairs + two of warm-up were 19 us of a 57 us step there).  The overlap-add order per sample does not depend on the
  // split (a run's warm-up pairs rebuild the ring in frame order): same bits, tests/test_small_shapes.py; knob SMALL_PATH = 1: off
  if (run < 3 * warm && ((long)B * F >= kSmallRows || knob(KNOB_SMALL_PATH) == 1)) run = 3 * warm;
  if (const long v = knob(KNOB_STFT_RUN)) { if (v >= 1) run = (int)v; }
  if (run > g.pairs) run = g.pairs;
  g.run = run;
  g.runs_per_utt = (g.pairs + run - 1) / run;
  const long wgs = (long)B * g.runs_per_utt;
  if (wgs > 0x7fffffffL) return -1;
#define DDSP_STFT_LAUNCH(R_, WPS_)                                                                                  \
  hipLaunchKernelGGL((k_stft_filter<R_, WPS_>), dim3((unsigned)wgs), dim3(64 * R_), 0, st, exc, noise, c_hmag, c_hphase, \
                     c_nmag, c_nphase, window, out, g)
  if (exc_f0) {
    hipLaunchKernelGGL((k_stft_filter<4, 2, true>), dim3((unsigned)wgs), dim3(256), 0, st, exc, noise, c_hmag, c_hphase, c_nmag,
                       c_nphase, window, out, g);
    return 0;
  }
  if (win == 2048) {
    if (wps == 4) DDSP_STFT_LAUNCH(4, 4);
    else if (wps == 2) DDSP_STFT_LAUNCH(4, 2);
    else DDSP_STFT_LAUNCH(4, 3);
  } else {
    if (wps == 3) DDSP_STFT_LAUNCH(2, 3);
    else DDSP_STFT_LAUNCH(2, 2);
  }
#undef DDSP_STFT_LAUNCH
  return 0;
}

int launch_stft_filter_bwd(const float* exc, const float* noise, int noise_is_u01, const float* c_hmag, long ld_hm,
                           const float* c_hphase, long ld_hp, const float* c_nmag, long ld_nm, const float* c_nphase,
                           long ld_np, float noise_scale, const float* window, int win, int reflect, int normalize,
                           const float* grad_out, int B, int F, int hop, float* d_hmag, float* d_hphase, float* d_nmag,
                           float* d_nphase, hipStream_t st) {
  if (hop != ST_HOP || (win != 1024 && win != 2048) || (long)F * hop >= (1L << 30)) return -1;
  StftGeom g;
  g.F = F; g.T = F * hop;
  g.pairs = (F + 1) / 2;                                       // frames 0..F-1; frame F rides with the last pair
  g.reflect = reflect; g.normalize = normalize; g.noise_u01 = noise_is_u01; g.noise_scale = noise_scale;
  g.ld_hm = ld_hm; g.ld_hp = ld_hp; g.ld_nm = ld_nm; g.ld_np = ld_np;
  // waves per SIMD of the variant launched (knob STFT_WPS: 2 or 3 at win 2048).  With the next stage's inputs prefetched the
  // register file is the limit: 2 waves per SIMD (25 spilled dwords) 0.745 ms per CombSubSuperFast training step, 3 waves
  // (107 spilled) 0.93; before the prefetch 3 waves (54 spilled) were the faster ones, 0.778
  int wps = 2;
  if (win == 2048) { if (const long v = knob(KNOB_STFT_WPS)) { if (v == 2 || v == 3) wps = (int)v; } }
  const int wg_per_cu = win == 2048 ? wps : 4;
  const long slots = (long)wg_per_cu * 256;
  long per_utt = slots / (B > 0 ? B : 1);
  if (per_utt < 1) per_utt = 1;
  int run = (int)((g.pairs + per_utt - 1) / per_utt);
  if (run < 2) run = 2;
  if (const long v = knob(KNOB_STFT_RUN)) { if (v >= 1) run = (int)v; }
  if (run > g.pairs) run = g.pairs;
  g.run = run;
  g.runs_per_utt = (g.pairs + run - 1) / run;
  const long wgs = (long)B * g.runs_per_utt;
  if (wgs > 0x7fffffffL) return -1;
  if (win == 2048 && wps == 3)
    hipLaunchKernelGGL((k_stft_filter_bwd<4, 3>), dim3((unsigned)wgs), dim3(256), 0, st, exc, noise, c_hmag, c_hphase,
                       c_nmag, c_nphase, window, grad_out, d_hmag, d_hphase, d_nmag, d_nphase, g);
  else if (win == 2048)
    hipLaunchKernelGGL((k_stft_filter_bwd<4, 2>), dim3((unsigned)wgs), dim3(256), 0, st, exc, noise, c_hmag, c_hphase,
                       c_nmag, c_nphase, window, grad_out, d_hmag, d_hphase, d_nmag, d_nphase, g);
  else
    hipLaunchKernelGGL((k_stft_filter_bwd<2, 2>), dim3((unsigned)wgs), dim3(128), 0, st, exc, noise, c_hmag, c_hphase,
                       c_nmag, c_nphase, window, grad_out, d_hmag, d_hphase, d_nmag, d_nphase, g);
  return 0;
}

}  // namespace ddsp
