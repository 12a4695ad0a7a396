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
#include "philox.h"
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

// RNG: the input is not read but drawn in the load path (philox.h; the uniform draw of the noise branch, mapped to 2u-1)
template <int WPS, bool RNG = false>
__global__ void __launch_bounds__(128, WPS) k_fir_blk(const float* __restrict__ x, int x_is_u01,
                                                     const float* __restrict__ taps,
                                                     const float* __restrict__ addend, float* __restrict__ out,
                                                     float* __restrict__ out_plain, FirBlkGeom g, NoiseGen rng) {
  using PL = fft::Plan<2>;
  constexpr int NF = PL::N, P = PL::P, S = 8;                  // 1024 points, 128 threads, 8 points per thread
  __shared__ __attribute__((aligned(16))) f32x2 ex[4][NF];   // two ping-pong pairs: ex[0..1] every transform, ex[2..3] the second of a lockstep pair
  const int tid = threadIdx.x;
  const int b = blockIdx.x / g.runs_per_utt;
  const int run_no = blockIdx.x - b * g.runs_per_utt;
  const int q_first = run_no * g.run;
  int q_last = q_first + g.run;
  if (q_last > g.pairs) q_last = g.pairs;
  const int SH = FB_HOP - (g.N >> 1);                          // circular tap shift
  // Every global access goes through a buffer descriptor whose byte count bounds it (BufF32, ddsp_common.h): positions
  // outside a tap row, blocks beyond the utterance and output times outside [0, T) are dropped by the address unit, so the
  // loop carries no clamps, selects or exec-mask branches for them.  Descriptors are built from workgroup-uniform values.
  const int bu = __builtin_amdgcn_readfirstlane(b);
  const float* xb = x + (long)bu * g.T;
  const float* tb = taps + (long)bu * g.F * g.N;
  const long ob = (long)bu * g.T;
  const BufF32 out_buf = BufF32::make(out + ob, g.T);
  const BufF32 plain_buf = BufF32::make(out_plain ? out_plain + ob : out + ob, out_plain ? g.T : 0);
  const BufF32 add_buf = BufF32::make(addend ? addend + ob : out + ob, addend ? g.T : 0);
  const float inv_hop = 1.0f / (float)FB_HOP;
  const int tid4 = 4 * tid;

  typename PL::Tw tw;
  tw.init(tid);
  // Overlap-add ring: 1024 samples, of which a thread only ever touches the 8 congruent to its id -- they live in
  // registers.  Transform index n = 128 m + tid of block bb is time (bb - 1) hop + n, i.e. ring slot (4 (bb - 1) + m) mod 8:
  // a pair advances the ring by exactly one revolution, so every slot index below is a compile-time constant.
  float ring[S];
#pragma unroll
  for (int m = 0; m < S; ++m) ring[m] = 0.f;
  int cur = 0;                                                 // ex[cur]: the exchange buffer no wave is reading any more

  // Global loads are issued at the top of a pair for the NEXT pair and stay in flight across the transforms.
  // One tap row, shifted: the value at transform index n = 128 m + tid is taps[row][n - SH]; only m >= 2 can be live.
  // The six byte offsets are loop invariants; a position before the row start gets the out-of-range constant (a
  // position beyond the row end is out of range by itself: the descriptor spans exactly one row).
  struct TapRow { float v[6]; };
  int tap_off[6];
#pragma unroll
  for (int m = 2; m < S; ++m) {
    const int i = P * m + tid - SH;
    tap_off[m - 2] = i >= 0 ? 4 * i : BufF32::kOutOfRange;
  }
  auto load_taps = [&](int j) -> TapRow {
    TapRow r;
    const int row = j < g.F ? j : g.F - 1;                     // core.py:167
    const BufF32 tr = BufF32::make(tb + (long)row * g.N, g.N);
#pragma unroll
    for (int m = 2; m < S; ++m) r.v[m - 2] = tr.ld(tap_off[m - 2]);
    return r;
  };
  // one hop block of the input: 4 samples per thread (s = 128 m + tid); a block beyond the utterance reads zeros
  struct Blk { float v[4]; };
  auto load_blk = [&](int bi) -> Blk {
    Blk r;
    if (RNG) {                                                  // drawn, not read: 4 uniforms of (utterance, block, lane)
      const Quad q = philox_uniform4(rng, (unsigned)bu, (unsigned)bi, (unsigned)tid);
#pragma unroll
      for (int m = 0; m < 4; ++m) r.v[m] = bi < g.F ? q.u[m] : 0.f;
      return r;
    }
    const BufF32 xr = BufF32::make(xb + (long)bi * FB_HOP, bi < g.F ? FB_HOP : 0);
#pragma unroll
    for (int m = 0; m < 4; ++m) r.v[m] = xr.ld(tid4 + 4 * P * m);
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
    for (int m = 2; m < S; ++m) z[m] = f32x2{ta.v[m - 2], tb2.v[m - 2]};
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

    // The two block transforms of the pair run in lockstep (Plan::forward_s2): one set of barriers for both, and each
    // wave has the other block's butterflies to issue while one block's LDS round trip is in flight.
    f32x2 z0[S], z1[S];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const Blk& cx = h == 0 ? cx0 : cx1;
      const bool live = b0 + h < g.F;
      f32x2 (&z)[S] = h == 0 ? z0 : z1;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        float xv = cx.v[m];
        if ((RNG || x_is_u01) && live) xv = fmaf(2.0f, xv, -1.0f);   // noise = rand*2-1 (vocoder.py:603,854); uniform condition
        const float lam = (float)(P * m + tid) * inv_hop;
        z[m] = f32x2{(1.0f - lam) * xv, lam * xv};               // the two Bartlett halves (core.py:161)
      }
#pragma unroll
      for (int m = 4; m < S; ++m) z[m] = f32x2{0.f, 0.f};
    }
    f32x2* const Xa = ex[cur];
    f32x2* const Xb = ex[2 + cur];
    PL::template forward_s2<true>(z0, z1, tw, Xa, ex[cur ^ 1], Xb, ex[2 + (cur ^ 1)], tid);
#pragma unroll
    for (int m = 0; m < S; ++m) { Xa[kS0 + 64 * m] = z0[m]; Xb[kS0 + 64 * m] = z1[m]; }
    __syncthreads();
    cur ^= 1;
    f32x2 V[S];
#pragma unroll
    for (int m = 0; m < S; ++m) {
      const int k = kS0 + 64 * m;
      const int kn = (NF - k) & (NF - 1);
      // Y = X1 H_b + X2 H_b+1 = p G_b - i d G_b+1 with p = 2 X1, d = 2i X2 from the packed block transform
      const f32x2 za = Xa[kn], zb = Xb[kn];
      const f32x2 y0 = fft::add_mi(cmul(fft::add_conj(z0[m], za), Gc[m]), cmul(fft::sub_conj(z0[m], za), Ga[m]));
      const f32x2 y1 = fft::add_mi(cmul(fft::add_conj(z1[m], zb), Ga[m]), cmul(fft::sub_conj(z1[m], zb), Gb[m]));
      V[m] = fft::conj_minus_i_conj(y0, y1);                    // V = Y_b0 + i Y_b0+1, conjugated for the inverse-by-forward trick
    }
    // the addend of the 1024 samples this pair emits is fetched now and lands during the inverse transform.  Emitted times
    // of this thread: t = e0 + 128 i, i = 0..7.  From the second pair of an utterance on e0 >= 0, so the byte offset is
    // 4 e0 plus an instruction immediate; the first pair (e0 < 0 for some lanes) forms every offset in full.
    const bool own = q >= q_first;
    const int e0 = (b0 - 1) * FB_HOP + 256 + tid;               // first emitted time of this thread
    auto t_off = [&](int i) -> int {
      if (b0 > 0) return 4 * e0 + 4 * P * i;
      const int t = e0 + P * i;
      return t >= 0 ? 4 * t : BufF32::kOutOfRange;
    };
    float add[S];
#pragma unroll
    for (int i = 0; i < S; ++i) add[i] = add_buf.ld(t_off(i));   // 0 without an addend (empty descriptor)
    // back to time order: the transposed factorisation takes layout S and leaves slot m, lane tid = sample 128 m + tid.
    // No barrier follows (the overlap-add ring is thread-private): its second buffer may still be read by the slower
    // wave, its first -- ex[cur] -- is free again, which is what the next transform expects
    PL::transposed(V, tw, ex[cur], ex[cur ^ 1], tid);
    // ifft = conj(FFT(conj V)): y_b0 = Re, y_b0+1 = -Im.  Transform index n of block bb is time (bb-1) hop + n.
    const bool last = q == g.pairs - 1;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      constexpr int kRot[2] = {4, 0};                           // ring slot of transform index 128 m: (4 (h+1) + m) & 7
#pragma unroll
      for (int m = 0; m < S; ++m) ring[(kRot[h] + m) & 7] += h == 0 ? V[m].x : -V[m].y;
      // times below (bb+1) hop - N/2 are final once block bb is in: emit indices 128 (2 + m), m = 0..3, i.e. emitted
      // sample i = 4 h + m of the pair; stores outside [0, T) are dropped by the descriptor
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int ri = (kRot[h] + 2 + m) & 7;
        const float v = ring[ri];
        ring[ri] = 0.f;
        if (own) {
          const int off = t_off(4 * h + m);
          plain_buf.st(v, off);                                  // empty descriptor unless out_plain was asked for
          out_buf.st(v + add[4 * h + m], off);
        }
      }
    }
    if (last && own) {                                          // the last pair also flushes what is left of the ring
#pragma unroll
      for (int m = 4; m < S; ++m) {
        const int ri = (2 + m) & 7;
        const int off = t_off(4 + m);                           // t = (b0 + 1 - 1) hop + 256 + 128 m + tid
        const float v = ring[ri];
        plain_buf.st(v, off);
        out_buf.st(v + add_buf.ld(off), off);
      }
    }
    // hand the spectrum of tap row b0 + 2 to the next pair (unrolling the loop by two with swapped roles instead of these
    // 16 moves needs 256 registers and spills 9)
#pragma unroll
    for (int m = 0; m < S; ++m) Gc[m] = Gb[m];
  }
}

// the same draw written out as a [B,T] tensor of u in [0,1) (tests, the oracle comparison, and callers whose noise filter
// is not served by the in-kernel form); any T: block = t / 512, lane = t % 128, output word = (t % 512) / 128
__global__ void __launch_bounds__(128) k_uniform_noise(NoiseGen rng, int B, long T, float* __restrict__ out) {
  const long blocks_per_utt = (T + FB_HOP - 1) / FB_HOP;
  const long wg = blockIdx.x;
  const unsigned b = (unsigned)(wg / blocks_per_utt);
  const unsigned bi = (unsigned)(wg - (long)b * blocks_per_utt);
  const Quad q = philox_uniform4(rng, b, bi, threadIdx.x);
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const long t = (long)bi * FB_HOP + 128 * m + threadIdx.x;
    if (t < T) out[(long)b * T + t] = q.u[m];
  }
}

int launch_uniform_noise(unsigned long long seed, unsigned long long offset, int B, long T, float* out, hipStream_t st) {
  const long wgs = (long)B * ((T + FB_HOP - 1) / FB_HOP);
  if (wgs <= 0 || wgs > 0x7fffffffL) return -1;
  hipLaunchKernelGGL(k_uniform_noise, dim3((unsigned)wgs), dim3(128), 0, st, NoiseGen{seed, offset, 1}, B, T, out);
  return 0;
}

// returns the implementation id (5) or < 0 when the shape is outside this kernel
int launch_fir_blk(const float* x, int x_is_u01, const float* taps, const float* addend, float* out, float* out_plain,
                   int B, int F, int hop, int N, hipStream_t st, const NoiseGen* noise_gen) {
  if (hop != FB_HOP || N < 2 || (N & 1) || N > 512 || (long)F * hop >= (1L << 29)) return -1;   // byte offsets of one utterance stay below 2^31 (buffer descriptors)
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
  NoiseGen rng{0ull, 0ull, 0};
  if (noise_gen && noise_gen->on) {                             // the input is drawn in the kernel (x may be null)
    rng = *noise_gen;
    hipLaunchKernelGGL((k_fir_blk<2, true>), dim3((unsigned)wgs), dim3(128), 0, st, x, 0, taps, addend, out, out_plain, g, rng);
    return 5;
  }
  size_t pad = 0;                                               // occupancy probe: extra dynamic LDS per workgroup
  if (const long v = knob(KNOB_BLK_PADLDS)) { if (v > 0) pad = (size_t)v; }
  if (pad > 0) {
    hipLaunchKernelGGL((k_fir_blk<2, false>), dim3((unsigned)wgs), dim3(128), pad, st, x, x_is_u01, taps, addend, out, out_plain, g, rng);
    return 5;
  }
  if (wps >= 4)
    hipLaunchKernelGGL((k_fir_blk<4, false>), dim3((unsigned)wgs), dim3(128), 0, st, x, x_is_u01, taps, addend, out, out_plain, g, rng);
  else if (wps == 3)
    hipLaunchKernelGGL((k_fir_blk<3, false>), dim3((unsigned)wgs), dim3(128), 0, st, x, x_is_u01, taps, addend, out, out_plain, g, rng);
  else
    hipLaunchKernelGGL((k_fir_blk<2, false>), dim3((unsigned)wgs), dim3(128), 0, st, x, x_is_u01, taps, addend, out, out_plain, g, rng);
  return 5;
}

}  // namespace ddsp
