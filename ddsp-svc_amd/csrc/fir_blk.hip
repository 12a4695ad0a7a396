// HOT-2c (hop-block FFT form): the time-varying FIR of ddsp/core.py:120-182, regrouped by INPUT hop block.
//
// The reference windows 50 %-overlapping frames of 2 hop samples with a periodic Bartlett window and convolves
// frame j with taps j (core.py:155-177).  The two Bartlett halves that cover hop block b (samples [b hop, (b+1)
// hop)) belong to frames b and b+1, so the same operator reads
//     y = sum_b  conv(x_b (1 - lambda), taps_b)  +  conv(x_b lambda, taps_min(b+1, F-1)),    lambda = s / hop,
// with the result of block b placed at output position b hop - N/2 (last tap row held,
// core.py:167).  With hop = 512 and N <= 512 every one of these linear convolutions (<= 1023 samples) fits a
// 1024-point transform, so per hop block the work is
//     Zx = FFT(x_b (1-lambda) + i x_b lambda)      -> X1, X2          one transform per block
//     Zh = FFT(taps'_j + i taps'_j+1)              -> H_j, H_j+1      half a transform per block
//     y_b + i y_b+1 = IFFT(Y_b + i Y_b+1),  Y_b = X1 H_b + X2 H_b+1   half a transform per block
// = two 1024-point complex FFTs per 512 output samples, against 1.5 2048-point ones in k_fir_fft (fir_fft.hip):
// 40 % fewer flops and a third less LDS exchange traffic.
//
// A 128-thread workgroup (2 waves, fft_r.h with R = 2) walks a run of consecutive block pairs of one utterance
// and keeps the spectra of three tap rows in registers.  The taps enter the transform circularly shifted by
// 512 - N/2, which moves the (circular, but alias-free: support <= 1023) result of block b to the output range
// starting at (b-1) hop -- a multiple of the thread count -- so every thread only ever touches overlap-add
// ring slots congruent to its id and the ring needs no barriers.  A run starts one pair early (discarded) so
// the ring holds its predecessor's tail: no atomics, so a launch geometry is bit-reproducible (different run splits
// agree to rounding: the first tap spectrum of a run comes out of a differently packed transform).
#include "fft_r.h"
#include "kernels.h"
#include <stdlib.h>
#include <type_traits>

namespace ddsp {

using fft::cmul;

constexpr int FB_HOP = 512;

struct FirBlkGeom {
  int F, N, T;            // frames, taps, samples per utterance
  int pairs;              // block pairs per utterance: ceil(F / 2)
  int run, runs_per_utt;  // own pairs per workgroup
};

template <int WPS>
__global__ void __launch_bounds__(128, WPS) k_fir_blk(const float* __restrict__ x, int x_is_u01,
                                                     const float* __restrict__ taps,
                                                     const float* __restrict__ addend, float* __restrict__ out,
                                                     float* __restrict__ out_plain, FirBlkGeom g) {
  using PL = fft::Plan<2>;
  constexpr int NF = PL::N, P = PL::P, S = 8;                  // 1024 points, 128 threads, 8 points per thread
  __shared__ __attribute__((aligned(16))) f32x2 ex[2][NF];
  const int tid = threadIdx.x;
  const int b = blockIdx.x / g.runs_per_utt;
  const int run_no = blockIdx.x - b * g.runs_per_utt;
  const int q_first = run_no * g.run;
  int q_last = q_first + g.run;
  if (q_last > g.pairs) q_last = g.pairs;
  const int SH = FB_HOP - (g.N >> 1);                          // circular tap shift
  const float* xb = x + (long)b * g.T;
  const float* tb = taps + (long)b * g.F * g.N;
  const long ob = (long)b * g.T;
  const float inv_hop = 1.0f / (float)FB_HOP;

  typename PL::Tw tw;
  tw.init(tid);
  // Overlap-add ring: 1024 samples, of which a thread only ever touches the 8 congruent to its id -- they live in
  // registers.  Transform index n = 128 m + tid of block bb is time (bb - 1) hop + n, i.e. ring slot (4 (bb - 1) + m) mod 8:
  // a pair advances the ring by exactly one revolution, so every slot index below is a compile-time constant.
  float ring[S];
#pragma unroll
  for (int m = 0; m < S; ++m) ring[m] = 0.f;
  int cur = 0;                                                 // ex[cur]: the exchange buffer no wave is reading any more

  // Global loads are issued unconditionally from clamped addresses and masked when they are USED: a load whose
  // result feeds a select or the 2u-1 map right away would be waited for on the spot instead of staying in flight
  // across the transforms (and a predicated load costs an exec-mask branch).
  // One tap row, shifted: the value at transform index n = 128 m + tid is taps[row][n - SH]; only m >= 2 can be live.
  struct TapRow { float v[6]; };
  auto load_taps = [&](int j) -> TapRow {
    TapRow r;
    const int row = j < g.F ? j : g.F - 1;                     // core.py:167
    const float* tr = tb + (long)row * g.N;
#pragma unroll
    for (int m = 2; m < S; ++m) {
      int i = P * m + tid - SH;
      i = i < 0 ? 0 : (i >= g.N ? g.N - 1 : i);
      r.v[m - 2] = tr[i];
    }
    return r;
  };
  auto tap_at = [&](const TapRow& r, int m) -> float {          // m >= 2
    const int i = P * m + tid - SH;
    return (i >= 0 && i < g.N) ? r.v[m - 2] : 0.f;
  };
  // one hop block of the input: 4 samples per thread (s = 128 m + tid); blocks beyond the utterance read block F-1
  // and are zeroed at use
  struct Blk { float v[4]; };
  auto load_blk = [&](int bi) -> Blk {
    Blk r;
    const float* src = xb + (long)(bi < g.F ? bi : g.F - 1) * FB_HOP + tid;
#pragma unroll
    for (int m = 0; m < 4; ++m) r.v[m] = src[P * m];
    return r;
  };
  // FFT of a packed pair (real + i imaginary) into the scrambled layout S of fft_r.h (two LDS exchanges; every
  // spectral array of this kernel lives in S), then the same values parked in LDS at their natural bin index for the
  // mirrored read Z[-k].  ex[cur] is the buffer no wave reads any more; the mirror copy goes there as well (its
  // readers of the first pass are behind the second barrier), so afterwards the OTHER buffer is the free one.
  const int kS0 = PL::s_index(tid, 0);                          // bin of slot 0; slot m holds bin kS0 + 64 m
  auto transform = [&](f32x2 (&z)[S], auto hi_zero) -> f32x2* {
    f32x2* X = ex[cur];
    f32x2* Y = ex[cur ^ 1];
    PL::template forward_s<decltype(hi_zero)::value>(z, tw, X, Y, tid);
#pragma unroll
    for (int m = 0; m < S; ++m) X[kS0 + 64 * m] = z[m];
    __syncthreads();
    cur ^= 1;
    return X;
  };
  const float ch = 0.25f / (float)NF;                          // the 1/2 of both splits and the 1/N of the inverse
  // Ga = ch * H_j, Gb = ch * H_j+1 from Z = FFT(h_j + i h_j+1):  H_j = (Z[k] + conj Z[-k]) / 2,
  // H_j+1 = (Z[k] - conj Z[-k]) / 2i
  auto split_taps = [&](const TapRow& ta, const TapRow& tb2, f32x2 (&Ga)[S], f32x2 (&Gb)[S]) {
    f32x2 z[S];
    z[0] = z[1] = f32x2{0.f, 0.f};
#pragma unroll
    for (int m = 2; m < S; ++m) z[m] = f32x2{tap_at(ta, m), tap_at(tb2, m)};
    const f32x2* Zn = transform(z, std::false_type{});
#pragma unroll
    for (int m = 0; m < S; ++m) {
      const int k = kS0 + 64 * m;
      const f32x2 zneg = Zn[(NF - k) & (NF - 1)];
      const f32x2 p = fft::add_conj(z[m], zneg);                // 2 H_j
      const f32x2 d = fft::sub_conj(z[m], zneg);                // 2i H_j+1
      Ga[m] = p * ch;
      Gb[m] = f32x2{d.y * ch, -d.x * ch};                      // d / i
    }
  };

  const int q0 = q_first > 0 ? q_first - 1 : 0;
  // prologue: spectrum of the first block's own tap row (packed with the row after it, which the loop recomputes
  // together with its successor -- one extra half transform per run)
  f32x2 Gc[S];
  {
    f32x2 Gdrop[S];
    split_taps(load_taps(2 * q0), load_taps(2 * q0 + 1), Gc, Gdrop);
  }
  TapRow t1 = load_taps(2 * q0 + 1), t2 = load_taps(2 * q0 + 2);
  Blk x0 = load_blk(2 * q0), x1 = load_blk(2 * q0 + 1);

  for (int q = q0; q < q_last; ++q) {
    const int b0 = 2 * q;
    const TapRow ct1 = t1, ct2 = t2;
    const Blk cx0 = x0, cx1 = x1;
    // the next pair's global loads are issued now and land while this pair is transformed
    t1 = load_taps(b0 + 3);
    t2 = load_taps(b0 + 4);
    x0 = load_blk(b0 + 2);
    x1 = load_blk(b0 + 3);

    f32x2 Ga[S], Gb[S];
    split_taps(ct1, ct2, Ga, Gb);                               // H_b0+1, H_b0+2

    f32x2 V[S];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const Blk& cx = h == 0 ? cx0 : cx1;
      const bool live = b0 + h < g.F;
      f32x2 z[S];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        float xv = cx.v[m];
        if (x_is_u01) xv = fmaf(2.0f, xv, -1.0f);               // noise = rand*2-1 (vocoder.py:603,854)
        if (!live) xv = 0.f;
        const float lam = (float)(P * m + tid) * inv_hop;
        z[m] = f32x2{(1.0f - lam) * xv, lam * xv};               // the two Bartlett halves (core.py:161)
      }
#pragma unroll
      for (int m = 4; m < S; ++m) z[m] = f32x2{0.f, 0.f};
      const f32x2* Zn = transform(z, std::true_type{});          // the block fills the lower half of the transform
#pragma unroll
      for (int m = 0; m < S; ++m) {
        const int k = kS0 + 64 * m;
        const f32x2 zneg = Zn[(NF - k) & (NF - 1)];
        const f32x2 p = fft::add_conj(z[m], zneg);              // 2 X1
        const f32x2 d = fft::sub_conj(z[m], zneg);              // 2i X2
        // Y = X1 H_b + X2 H_b+1 = p G_b - i d G_b+1
        const f32x2 y = h == 0 ? fft::add_mi(cmul(p, Gc[m]), cmul(d, Ga[m]))
                               : fft::add_mi(cmul(p, Ga[m]), cmul(d, Gb[m]));
        // V = Y_b0 + i Y_b0+1, conjugated for the inverse-by-forward trick
        if (h == 0) V[m] = y;
        else V[m] = fft::conj_minus_i_conj(V[m], y);
      }
    }
    // the addend of the 1024 samples this pair emits is fetched now and lands during the inverse transform
    const bool own = q >= q_first;
    const int e0 = (b0 - 1) * FB_HOP + 256 + tid;               // first emitted time of this thread
    float add[S];
#pragma unroll
    for (int i = 0; i < S; ++i) add[i] = 0.f;
    if (addend) {                                               // uniform; clamped addresses, masked by the store
#pragma unroll
      for (int i = 0; i < S; ++i) {
        int t = e0 + P * i;
        t = t < 0 ? 0 : (t >= g.T ? g.T - 1 : t);
        add[i] = addend[ob + t];
      }
    }
    // back to time order: the transposed factorisation takes layout S and leaves slot m, lane tid = sample 128 m + tid.
    // No barrier follows (the overlap-add ring is thread-private): its second buffer may still be read by the slower
    // wave, its first -- ex[cur] -- is free again, which is what the next transform expects
    PL::transposed(V, tw, ex[cur], ex[cur ^ 1], tid);
    // ifft = conj(FFT(conj V)): y_b0 = Re, y_b0+1 = -Im.  Transform index n of block bb is time (bb-1) hop + n.
    const bool last = q == g.pairs - 1;
    const bool interior = own && !last && (b0 - 1) * FB_HOP + 256 >= 0 && (b0 + 1) * FB_HOP + 256 <= g.T;   // uniform
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int base = (b0 + h - 1) * FB_HOP;                   // time of transform index 0; ring slot of index 128 m: (4 (h+1) + m) & 7
      constexpr int kRot[2] = {4, 0};
#pragma unroll
      for (int m = 0; m < S; ++m) ring[(kRot[h] + m) & 7] += h == 0 ? V[m].x : -V[m].y;
      // times below (bb+1) hop - N/2 are final once block bb is in: emit [base + 256, base + 768) = indices 128 (2 + m),
      // m = 0..3; the last pair also flushes what is left
      if (interior) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int t = base + 256 + P * m + tid;
          const int ri = (kRot[h] + 2 + m) & 7;
          const float v = ring[ri];
          ring[ri] = 0.f;
          if (out_plain) out_plain[ob + t] = v;
          out[ob + t] = v + add[4 * h + m];
        }
      } else {
        const int n_emit = (last && h == 1) ? 8 : 4;
#pragma unroll
        for (int m = 0; m < S; ++m) {
          if (m < n_emit) {
            const int t = base + 256 + P * m + tid;
            const int ri = (kRot[h] + 2 + m) & 7;
            const float v = ring[ri];
            ring[ri] = 0.f;
            if (own && t >= 0 && t < g.T) {
              if (out_plain) out_plain[ob + t] = v;
              float a = 0.f;
              if (m < 4) a = add[4 * h + m];                    // t == e0 + 128 (4 h + m)
              else if (addend) a = addend[ob + t];              // the flush of the last pair
              out[ob + t] = v + a;
            }
          }
        }
      }
    }
    // hand the spectrum of tap row b0 + 2 to the next pair
#pragma unroll
    for (int m = 0; m < S; ++m) Gc[m] = Gb[m];
  }
}

// returns the implementation id (5) or < 0 when the shape is outside this kernel
int launch_fir_blk(const float* x, int x_is_u01, const float* taps, const float* addend, float* out, float* out_plain,
                   int B, int F, int hop, int N, hipStream_t st) {
  if (hop != FB_HOP || N < 2 || (N & 1) || N > 512 || (long)F * hop >= (1L << 30)) return -1;
  FirBlkGeom g;
  g.F = F; g.N = N; g.T = F * hop;
  g.pairs = (F + 1) / 2;
  int wps = 2;
  if (const long v = knob(KNOB_BLK_WPS)) { if (v >= 1) wps = (int)v; }
  // run length: as many workgroups as the chip holds at once (2 waves each), one round, equal work; every run
  // pays one warm-up pair and one extra half transform
  const long slots = (long)wps * 2 * 256;
  long per_utt = slots / (B > 0 ? B : 1);
  if (per_utt < 1) per_utt = 1;
  int run = (int)((g.pairs + per_utt - 1) / per_utt);
  if (run < 3) run = 3;
  if (const long v = knob(KNOB_BLK_RUN)) { if (v >= 1) run = (int)v; }
  if (run > g.pairs) run = g.pairs;
  g.run = run;
  g.runs_per_utt = (g.pairs + run - 1) / run;
  const long wgs = (long)B * g.runs_per_utt;
  if (wgs > 0x7fffffffL) return -1;
  size_t pad = 0;                                               // occupancy probe: extra dynamic LDS per workgroup
  if (const long v = knob(KNOB_BLK_PADLDS)) { if (v > 0) pad = (size_t)v; }
  if (pad > 0) {
    hipLaunchKernelGGL(k_fir_blk<2>, dim3((unsigned)wgs), dim3(128), pad, st, x, x_is_u01, taps, addend, out, out_plain, g);
    return 5;
  }
  if (wps >= 4)
    hipLaunchKernelGGL(k_fir_blk<4>, dim3((unsigned)wgs), dim3(128), 0, st, x, x_is_u01, taps, addend, out, out_plain, g);
  else if (wps == 3)
    hipLaunchKernelGGL(k_fir_blk<3>, dim3((unsigned)wgs), dim3(128), 0, st, x, x_is_u01, taps, addend, out, out_plain, g);
  else
    hipLaunchKernelGGL(k_fir_blk<2>, dim3((unsigned)wgs), dim3(128), 0, st, x, x_is_u01, taps, addend, out, out_plain, g);
  return 5;
}

}  // namespace ddsp
