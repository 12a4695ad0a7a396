// HOT-2c (hop-block FFT form): the time-varying FIR of ddsp/core.py:120-182, regrouped by INPUT hop block.
//
// The reference windows 50 %-overlapping frames of 2 hop samples with a periodic Bartlett window and convolves
// frame j with taps j (core.py:155-177).  The two Bartlett halves that cover hop block b (samples [b hop, (b+1)
// hop)) belong to frames b and b+1, so the same operator reads
//     y = sum_b  conv(x_b (1 - lambda), taps_b)  +  conv(x_b lambda, taps_min(b+1, F-1)),    lambda = s / hop,
// with the result of block b placed at output position b hop - N/2 (last tap row held,
// core.py:167).  With hop = 512 and N <= 512 every one of these linear convolutions (<= 1023 samples) fits a
// 1024-point transform, so per hop block the work is
//     Z_b = FFT(x_b (1-lambda) + i x_b lambda)                        one transform per block
//     T   = FFT(taps'_j + i taps'_j+1)  -> H_j, H_j+1                 half a transform per block
//     y_b = Re IFFT(Z_b G_b),  G_b = H_b - i H_b+1                    half a transform per block (two blocks per inverse)
// = two 1024-point complex FFTs per 512 output samples, against 1.5 2048-point ones in k_fir_fft (fir_fft.hip):
// 40 % fewer flops and a third less LDS exchange traffic.  The product needs no separation of the packed block
// transform: (x1 + i x2) * (h_b - i h_b+1) has x1 * h_b + x2 * h_b+1 as its real part, so W_b = Z_b G_b is ONE complex
// product per bin, and taking real parts of two blocks at once is the Hermitian split of the PRODUCTS,
//     V = Y_b + i Y_b+1 = (A[k] + conj B[-k]) / 2,    A = W_b + i W_b+1,  B = W_b - i W_b+1,
// one parked array and one mirrored read per pair (round 2 separated X1, X2 of both blocks first: 15 instead of 7 packed
// instructions per bin pair, two parked arrays).  Of the tap transform, G_j = conj T[-k] comes straight out of the
// mirrored read; G_j-1 = H_j-1 - i H_j takes the carried H_j-1.
//
// A 128-thread workgroup (2 waves, fft_r.h with R = 2) walks a run of consecutive block pairs of one utterance
// and keeps the spectra of three tap rows in registers.  The taps enter the transform shifted by 256 - N/2: they
// then sit in the lower half of the transform (its first pass is pruned like the block's) and the result of block b
// (support <= 1023: no aliasing) starts at output time (b - 1/2) hop -- a multiple of the thread count -- so every
// thread only ever touches overlap-add ring slots congruent to its id and the ring needs no barriers.  A run starts
// one BLOCK early (discarded) so the ring holds its predecessor's tail: no atomics, so a launch geometry is
// bit-reproducible (different run splits agree to rounding: the first tap spectrum of a run comes out of a differently
// packed transform).  A pair of blocks is two lockstep stages (fft_r.h): the two block transforms, then the pair's
// inverse beside the transform of the next pair's tap rows.
#include "fft_1024p.h"
#include "kernels.h"
#include "philox.h"
#include <stdlib.h>
#include <type_traits>

namespace ddsp {

using fft::cmul;

constexpr int FB_HOP = 512;

#ifdef DDSP_HIP_TIMELINE
// diagnostics build only (tools/fir_blk_timeline.py): per-workgroup wall-clock stamps (100 MHz, one epoch for the chip)
__device__ long long* g_blk_timeline = nullptr;
__global__ void k_set_blk_timeline(long long* p) { g_blk_timeline = p; }
#define BLK_STAMP(slot) do { if (g_blk_timeline && threadIdx.x == 0 && (slot) < 32) g_blk_timeline[(long)blockIdx.x * 32 + (slot)] = (long long)wall_clock64(); } while (0)
#define BLK_STAMP_CYC(slot) do { if (g_blk_timeline && threadIdx.x == 0 && (slot) < 32) g_blk_timeline[(long)blockIdx.x * 32 + (slot)] = (long long)__builtin_readcyclecounter(); } while (0)
#define BLK_STAMP_WHERE(slot) do { if (g_blk_timeline && threadIdx.x == 0) g_blk_timeline[(long)blockIdx.x * 32 + (slot)] = ((long long)__builtin_amdgcn_s_getreg(0xF814) << 32) | (unsigned)__builtin_amdgcn_s_getreg(0xF804); } while (0)
#else
#define BLK_STAMP(slot) do { } while (0)
#define BLK_STAMP_CYC(slot) do { } while (0)
#define BLK_STAMP_WHERE(slot) do { } while (0)
#endif

struct FirBlkGeom {
  int F, N, T;            // frames, taps, samples per utterance
  int pairs;              // block pairs per utterance: ceil(F / 2)
  int run, runs_per_utt;  // most pairs a workgroup owns; workgroups per utterance
  int turns;              // waves sharing a SIMD alternate priority (knob BLK_TURNS: 1 = off)
};

// (round 3's two-wave kernel: compiled only into the A/B builds of tools/build_variant.sh, -DDDSP_AB_GENERATIONS; the product ships
// ONE generation per kernel)
#ifdef DDSP_AB_GENERATIONS
// RNG: the input is not read but drawn in the load path (philox.h; the uniform draw of the noise branch, mapped to 2u-1)
// An addend added to the stored result and a second, plain output (the options of a step's last filter) are run-time,
// workgroup-uniform switches: ONE code object serves every filter of a step, so that it stays in the instruction cache
// from one launch to the next
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
  // which run of the utterance: rotated by the utterance number (and by its higher digits), so that the longer and the
  // shorter runs of the even split below line up neither with the XCD round-robin of the dispatcher (blockIdx % 8) nor
  // with the workgroups a CU collects (blockIdx 256 apart at the headline shape): every CU gets a mix
  const int run_no = (int)((blockIdx.x - b * g.runs_per_utt + b + (b >> 4) + (b >> 8)) % g.runs_per_utt);
  // an utterance's pairs are split evenly over its runs (lengths differ by at most one: workgroups that share a SIMD
  // then finish together instead of leaving it half empty)
  const int q_first = (int)(((long)run_no * g.pairs) / g.runs_per_utt);
  const int q_last = (int)(((long)(run_no + 1) * g.pairs) / g.runs_per_utt);
  const int SH = FB_HOP / 2 - (g.N >> 1);                      // tap shift: row centred on transform index 256
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

  BLK_STAMP(0);
  BLK_STAMP_WHERE(31);
  // Overlap-add ring: 1024 samples, of which a thread only ever touches the 8 congruent to its id.  Transform index
  // n = 128 m + tid of block bb is time (bb - 1/2) hop + n, i.e. ring slot (4 bb - 2 + m) mod 8: a pair advances the ring by
  // exactly one revolution and every slot receives exactly two contributions -- the upper half (m >= 4) of one block's result,
  // then the lower half of the next block's, which completes it.  So nothing is accumulated in place: the first
  // contribution just stays where the transform left it (within a pair: the registers of V; across pairs: the four values
  // carried in `tail`), and a sample is formed when its second contribution arrives.
  float tail[4];                                                // -sigma times the upper half of the previous pair's second block
#pragma unroll
  for (int m = 0; m < 4; ++m) tail[m] = 0.f;
  // Global loads are issued at the top of a pair: the blocks of the NEXT pair, and the tap rows whose transform rides
  // beside this pair's inverse (see the loop).
  // One tap row, shifted: the value at transform index n = 128 m + tid is taps[row][n - SH]; only m < 4 can be live.
  // The four byte offsets are loop invariants; a position before the row start gets the out-of-range constant (a
  // position beyond the row end is out of range by itself: the descriptor spans exactly one row).
  struct TapRow { float v[4]; };
  int tap_off[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int i = P * m + tid - SH;
    tap_off[m] = i >= 0 ? 4 * i : BufF32::kOutOfRange;
  }
  auto load_taps = [&](int j) -> TapRow {
    TapRow r;
    const int row = j < g.F ? j : g.F - 1;                     // core.py:167
    const BufF32 tr = BufF32::make(tb + (long)row * g.N, g.N);
#pragma unroll
    for (int m = 0; m < 4; ++m) r.v[m] = tr.ld(tap_off[m]);
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
  // transform inputs: two tap rows packed as real + i imaginary; one block as its two Bartlett halves (core.py:161).
  // Both occupy the lower half of the transform (pruned first pass).
  auto pack_taps = [&](const TapRow& ta, const TapRow& tb2, f32x2 (&z)[S]) {
#pragma unroll
    for (int m = 0; m < 4; ++m) z[m] = f32x2{ta.v[m], tb2.v[m]};
#pragma unroll
    for (int m = 4; m < S; ++m) z[m] = f32x2{0.f, 0.f};
  };
  auto pack_blk = [&](const Blk& cx, bool live, f32x2 (&z)[S]) {
    // noise = rand * 2 - 1 (vocoder.py:603,854) as one multiply-add with workgroup-uniform coefficients: (2, -1) for a
    // uniform draw inside the signal, (1, 0) otherwise (a block beyond the signal reads zeros and must stay zero)
    const bool u01 = (RNG || x_is_u01) && live;
    const float ua = u01 ? 2.0f : 1.0f, ub = u01 ? -1.0f : 0.0f;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const float xv = fmaf(ua, cx.v[m], ub);
      const float lam = (float)(P * m + tid) * inv_hop;
      z[m] = f32x2{(1.0f - lam) * xv, lam * xv};
    }
#pragma unroll
    for (int m = 4; m < S; ++m) z[m] = f32x2{0.f, 0.f};
  };
  // Every spectral array of this kernel lives in the scrambled layout S of fft_r.h (slot m of thread tid holds bin
  // kS0 + 64 m) -- the transforms' outputs in its sign-carrying form S- (odd threads hold the negated bin, fft_r.h
  // lane_pair_dft2<FLIP>: one instruction per value for the lane-pair step).  Block spectra Z and filter spectra G both
  // carry that sign, so their product W is plain layout S; for the mirrored read [-k] an array is parked in LDS by
  // (swizzled) bin index.
  const int kS0 = PL::s_index(tid, 0);
  const int kP0 = PL::parked(kS0);                              // where slot 0 is parked (bits 3, 4, 9 are not touched by 64 m)
  auto mirrored = [&](const f32x2* X, int m) -> f32x2 { return X[PL::parked((NF - (kS0 + 64 * m)) & (NF - 1))]; };
  const float sg = (tid & 1) ? -1.0f : 1.0f;                   // the sign layout S- puts on this thread's bins / samples
  const float nsg = -sg;
  // A tap transform T' (S-) is parked as it is, except bins 0 and 512 -- slot 0 of threads 0, 1 --, which are their own
  // mirror images: every other bin's mirror image lives on a thread of the OTHER parity and so comes back with the
  // opposite sign, and negating these two makes the split below one formula for all bins.
  const float self_mirror = tid < 2 ? -1.0f : 1.0f;
  auto park_taps = [&](const f32x2 (&z)[S], f32x2* X) {
    X[kP0] = z[0] * f32x2{self_mirror, self_mirror};
#pragma unroll
    for (int m = 1; m < S; ++m) X[kP0 + 64 * m] = z[m];
  };
  // From T = FFT(h_j + i h_j+1) (z in S-, parked in Zp; Tn = the parked mirror image, sign opposite to z's) and the carried
  // Hc = c H_j-1, with c = 1 / 2N (the 1/2 of the Hermitian split of the products and the 1/N of the inverse):
  //     G1 = c (H_j - i H_j+1)   = c conj T[-k]                 -> - c conj Tn
  //     G0 = c (H_j-1 - i H_j)   = Hc - i (c/2) (T + conj T[-k])  -> Hc - i (c/2) (z - conj Tn)
  //     Hc' = c H_j+1            = (c/2) (T - conj T[-k]) / i     -> -i (c/2) (z + conj Tn)
  const float ch = 0.5f / (float)NF;
  const f32x2 kG1 = {-ch, ch};
  const f32x2 kMi = {0.5f * ch, -0.5f * ch};                   // times (-i) after the half swap of swap_scale
  auto split_taps = [&](const f32x2 (&z)[S], const f32x2* Zp, f32x2 (&Hc)[S], f32x2 (&G0)[S], f32x2 (&G1)[S]) {
#pragma unroll
    for (int m = 0; m < S; m += 2) {                            // two bins interleaved: no instruction consumes its predecessor's result
      const f32x2 tn0 = mirrored(Zp, m), tn1 = mirrored(Zp, m + 1);
      const f32x2 p0 = fft::sub_conj(z[m], tn0), p1 = fft::sub_conj(z[m + 1], tn1);
      const f32x2 d0 = fft::add_conj(z[m], tn0), d1 = fft::add_conj(z[m + 1], tn1);
      G1[m] = tn0 * kG1;
      G1[m + 1] = tn1 * kG1;
      G0[m] = fft::swap_scale_add(p0, kMi, Hc[m]);
      G0[m + 1] = fft::swap_scale_add(p1, kMi, Hc[m + 1]);
      Hc[m] = fft::swap_scale(d0, kMi);
      Hc[m + 1] = fft::swap_scale(d1, kMi);
    }
  };
  // W0 = Z0 G0, W1 = Z1 G1 (plain layout S);  A = W0 + i W1 stays in a, B = W0 - i W1 is parked for the mirrored read
  auto products = [&](const f32x2 (&za)[S], const f32x2 (&zb)[S], const f32x2 (&G0)[S], const f32x2 (&G1)[S], f32x2 (&a)[S], f32x2* Bp) {
#pragma unroll
    for (int m = 0; m < S; m += 2) {                            // four products in flight
      const f32x2 t0 = fft::cmul_lo(za[m], G0[m]), t1 = fft::cmul_lo(zb[m], G1[m]);
      const f32x2 t2 = fft::cmul_lo(za[m + 1], G0[m + 1]), t3 = fft::cmul_lo(zb[m + 1], G1[m + 1]);
      const f32x2 w0 = fft::cmul_hi(za[m], G0[m], t0), w1 = fft::cmul_hi(zb[m], G1[m], t1);
      const f32x2 w2 = fft::cmul_hi(za[m + 1], G0[m + 1], t2), w3 = fft::cmul_hi(zb[m + 1], G1[m + 1], t3);
      a[m] = fft::sub_mi(w0, w1);
      a[m + 1] = fft::sub_mi(w2, w3);
      Bp[kP0 + 64 * m] = fft::add_mi(w0, w1);
      Bp[kP0 + 64 * (m + 1)] = fft::add_mi(w2, w3);
    }
  };
  // conj V = conj(Y_b + i Y_b+1) = B[-k] + conj A[k]  (the scale is in G), conjugated for the inverse-by-forward trick
  auto hermitian = [&](f32x2 (&a)[S], const f32x2* Bp) {
#pragma unroll
    for (int m = 0; m < S; ++m) a[m] = fft::add_conj(mirrored(Bp, m), a[m]);
  };

  // The four exchange buffers have fixed roles.  A pair is two lockstep stages (fft_r.h), three barriers each:
  //   forward_s2:  the two block transforms; first exchange through A / B, second through C / D, bins parked in A / B
  //   transposed_and_forward_s:  the pair's inverse beside the transform of the NEXT pair's two tap rows; first exchange
  //       through C / D (free since the park barrier), second through A / B (free at the first barrier: the product is
  //       done with them), tap bins parked in C (last read before the second barrier)
  // so when a stage writes a buffer, every wave is past the barrier that followed its last read.
  f32x2* const bA = ex[0];
  f32x2* const bB = ex[1];
  f32x2* const bC = ex[2];
  f32x2* const bD = ex[3];

  // One loop body serves the warm-up and the run (the code of a pass exists once: 12 KB instead of 24 -- on part of the
  // pool an instruction fetch that misses the 64 KB instruction cache is slow enough to cost a cold launch 35 us,
  // DESIGN.md).  The run's passes want the filter spectra of their pair (G0, G1, and Hc for the pair after) and the
  // predecessor's tail.  Of the pair before the run only its SECOND block b_w = 2 q_first - 1 reaches into the run's first
  // emitted sample (the result of block b ends before (b + 3/2) hop), so the warm-up pass q = q_first - 1 transforms
  // [rows b_w, b_w+1 | block b_w] where a pass of the run has its two blocks, splits those rows at once, filters the one
  // block (it rides in the imaginary part, where the odd block of a pair does) and emits nothing; everything else --
  // the loads for the next pass, [inverse | next rows], the split -- is what every pass does.  An utterance's first run
  // has no predecessor: a zero block, and row 0 twice (Hc = c H_0).
  const int bw = 2 * q_first - 1;
  const TapRow pa = load_taps(bw > 0 ? bw : 0), pb = load_taps(bw + 1);
  Blk x0 = load_blk(bw >= 0 ? bw : g.F), x1 = x0;               // the warm-up pass reads x0 only
  TapRow t1 = pa, t2 = pb;
  typename PL::Tw tw;                                           // the twiddles are formed while those loads are in flight
  tw.init(tid);
  BLK_STAMP(1);
  f32x2 Hc[S], G0[S], G1[S];
#pragma unroll
  for (int m = 0; m < S; ++m) { Hc[m] = f32x2{0.f, 0.f}; G0[m] = f32x2{0.f, 0.f}; G1[m] = f32x2{0.f, 0.f}; }
  const bool has_add = addend != nullptr, has_plain = out_plain != nullptr;   // workgroup-uniform: scalar branches

  // The two waves that share a SIMD belong to different workgroups, and its arbiter serves the older one first: left
  // alone, one workgroup of each such pair runs ahead and finishes early, and its partner does the rest of its run at
  // single-wave throughput (tools/fir_blk_timeline.py: lifetimes of 83 / 105 us inside one CU).  The waves take turns
  // instead: priority 1 on alternate pairs, the phase taken from the wave slot.
  const int turn = __builtin_amdgcn_s_getreg(0x1804) & 1;      // HW_ID[3:0]: wave slot within the SIMD
  const int turns_mask = g.turns ? 1 : 0;
  for (int q = q_first - 1; q < q_last; ++q) {
    const bool warm = q < q_first;                              // workgroup-uniform
    if ((q + turn) & turns_mask) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
    const int b0 = 2 * q;
    const bool stamp_it = q - q_first == 5;
    if (stamp_it) BLK_STAMP_CYC(24);
    f32x2 z0[S], z1[S];
    if (warm) {
      pack_taps(t1, t2, z0);                                    // rows b_w, b_w + 1
      pack_blk(x0, bw >= 0, z1);                                // block b_w
    } else {
      pack_blk(x0, b0 < g.F, z0);
      pack_blk(x1, b0 + 1 < g.F, z1);
    }
    // fetched now: the tap rows of the next pass (their transform rides beside this pass's inverse) and its blocks
    t1 = load_taps(b0 + 3);
    t2 = load_taps(b0 + 4);
    x0 = load_blk(b0 + 2);
    x1 = load_blk(b0 + 3);
    PL::template forward_s2<true, true>(z0, z1, tw, bA, bC, bB, bD, tid);
    if (warm) {
      park_taps(z0, bB);
      __syncthreads();
      split_taps(z0, bB, Hc, G0, G1);                           // G1 = c (H_b_w - i H_b_w+1), Hc = c H_b_w+1; G0 is not used:
#pragma unroll
      for (int m = 0; m < S; ++m) z0[m] = f32x2{0.f, 0.f};     // the even block of this pass is absent
    }
    f32x2 V[S];
    products(z0, z1, G0, G1, V, bA);
    __syncthreads();
    if (stamp_it) BLK_STAMP_CYC(25);
    hermitian(V, bA);                                           // V = conj(Y_b0 + i Y_b0+1)
    // the addend of the 1024 samples this pair emits is fetched now and lands during the inverse transform.  Emitted times
    // of this thread: t = e0 + 128 i, i = 0..7 (and 8..11 for the flush).  e0 >= -256, and negative times are exactly
    // i = 0, 1 of an utterance's first pair -- for every lane -- so those two take a base that a scalar select turns into
    // the out-of-range constant, and the rest a base that is never negative: no per-lane clamps, no branches, and no
    // negative offset that an instruction immediate could carry back into range.
    const int e0 = b0 * FB_HOP - 256 + tid;                     // first emitted time of this thread
    const int off_a = b0 > 0 ? 4 * e0 : BufF32::kOutOfRange;
    const int off_b = 4 * e0 + 8 * P;
    auto t_off = [&](int i) -> int { return i < 2 ? off_a + 4 * P * i : off_b + 4 * P * (i - 2); };
    float add[S];
#pragma unroll
    for (int i = 0; i < S; ++i) add[i] = 0.f;
    if (has_add && !warm) {
#pragma unroll
      for (int i = 0; i < S; ++i) add[i] = add_buf.ld(t_off(i));
    }
    if (stamp_it) BLK_STAMP_CYC(26);
    // back to time order (the transposed factorisation takes layout S and leaves slot m, lane tid = sample 128 m + tid),
    // beside the transform of the next pass's tap rows b0 + 3, b0 + 4
    f32x2 zt[S];
    pack_taps(t1, t2, zt);
    PL::template transposed_and_forward_s<true>(V, zt, tw, bC, bA, bD, bB, tid);
    park_taps(zt, bC);
    __syncthreads();
    if (stamp_it) BLK_STAMP_CYC(27);
    split_taps(zt, bC, Hc, G0, G1);                             // the spectrum of tap row b0 + 2 moves on in Hc
    if (stamp_it) BLK_STAMP_CYC(28);
    // ifft = conj(FFT(conj V)): y_b0 = sigma Re V, y_b0+1 = -sigma Im V (sigma: this thread's sign, layout S- through the
    // transposed factorisation).  Emitted sample i of the pair, time e0 + 128 i:
    //     i = 0..3   lower half of block b0 on top of the previous pair's tail:      sigma (Re V[i] - tail[i])
    //     i = 4..7   lower half of block b0 + 1 on top of the upper half of b0:      sigma (Re V[i] - Im V[i - 4])
    // times below (bb + 1/2) hop are final once block bb is in; stores outside [0, T) are dropped by the descriptor
    if (!warm) {
      float d[S];
#pragma unroll
      for (int i = 0; i < S; ++i) d[i] = i < 4 ? V[i].x - tail[i] : V[i].x - V[i - 4].y;
      if (has_plain) {
#pragma unroll
        for (int i = 0; i < S; ++i) plain_buf.st(sg * d[i], t_off(i));
      }
#pragma unroll
      for (int i = 0; i < S; ++i) out_buf.st(fmaf(sg, d[i], add[i]), t_off(i));   // no addend: + 0
      if (q == g.pairs - 1) {                                   // the last pair also emits the upper half of its second block
#pragma unroll
        for (int m = 4; m < S; ++m) {
          const int off = t_off(4 + m);                         // t = (b0 + 1/2) hop + 128 m + tid
          const float v = nsg * V[m].y;
          if (has_plain) plain_buf.st(v, off);
          out_buf.st(v + (has_add ? add_buf.ld(off) : 0.f), off);
        }
      }
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) tail[m] = V[4 + m].y;
    if (stamp_it) BLK_STAMP_CYC(29);
    BLK_STAMP(3 + q - q_first);
  }
}
#endif  // DDSP_AB_GENERATIONS

// ---- the same operator at three waves per SIMD (six workgroups per CU) ---------------------------------------------------
// k_fir_blk above holds 226 registers and 32 KB of LDS per workgroup: four workgroups per CU, two waves per SIMD, and at two
// waves a SIMD is bound by each wave's in-order issue.  A third wave needs <= 168 registers and <= 26.6 KB.
// What moves, against the kernel above:
//   * the filter spectra G0, G1 (32 registers through the whole first stage) are not kept: the tap transform T stays PARKED
//     in its LDS buffer Cp from the end of one pass to the product of the next, and the product forms G0[m], G1[m], Hc'[m]
//     from T[k] (its own parked value) and T[-k] (the mirrored read) bin by bin where it consumes them; only Hc is carried;
//   * three exchange buffers instead of four (24 KB): the two block transforms run one after the other through X / Y
//     (Cp is busy holding T), the pair's inverse and the next tap transform run staggered through Y / X / Cp
//     (fft_r.h, transposed_then_forward_s); 8 barriers per pass instead of 6;
//   * the loads of a pass's two blocks are issued half a pass ahead (at the top of its predecessor's second stage) instead of
//     a whole pass: with six workgroups per CU there are other waves to cover what is left of their latency.
// Buffer roles per pass (every wave is past the barrier that followed a buffer's last read before anyone writes it):
//     z0: X -> Y      z1: X -> Y      product: reads Cp (T), parks B in X      barrier      mirrored read of X
//     inverse: Y -> X beside taps: Cp -> Y      T parked in Cp (last read: the product, four barriers ago)
// and the next pass's first write is to X, last read before the stage's third barrier.
// A launch takes one or two JOBS (grid.y): filters of the same shape that do not depend on each other -- the all-pass and the
// noise filter of a CombSub step at a streaming shape, where the chain of dependent launches is the latency.
struct FirJob {
  const float* x; int x_is_u01;
  const float* taps; const float* addend;
  float* out; float* out_plain;
  int rng;                      // k_fir_blk6<true> only: THIS job's input is drawn in the load path (the other job of a launch may read its)
  int taps_half;                // rows of N/2 + 1 taps of an even response (kernels.h): tap N - j is read from j
};
struct FirJobs { FirJob j[2]; };

// SEQ: ONE row of workgroups runs job 0 and then job 1 of a launch over the same run of block pairs -- for two filters of which the
// second takes the first one's result as its addend (Sins: signal = all-pass(sinusoids) + filtered noise, vocoder.py:597-609).  A
// thread reads back as addend exactly the samples it stored itself a sub-run earlier (the same kernel code maps the same times to
// the same thread), so the hand-over needs no flag and no other workgroup: one barrier and one wait for the stores between the two.
template <bool RNG = false, bool SEQ = false>
// Cache policy of the kernel's input streams: every block, tap row and addend byte is read once by one CU, so they are loaded with the
// non-temporal policy (aux bit 1 of the buffer instructions) instead of pushing each other and the next launch's inputs out of the L2;
// the results keep the default policy (the next launch reads them).  [MI355X] CombSub step, same box, three interleaved repetitions
// each: default policy 0.2884 ms | block loads 0.2838 | tap rows 0.2856 | stores 0.2877 | all three 0.2815 (r06_v41_bench_*.json);
// on a second box 0.2972 | all three 0.2925 | + the addend 0.2890 | loads only 0.2901 (r06_v42_*); on a third 0.2862 | everything 0.2791 |
// every load, default stores 0.2784 (r06_v43_*).  (A/B builds: -DDDSP_FIR_NT=<mask>: 1 block loads, 2 tap rows, 4 stores, 8 the addend.)
#ifndef DDSP_FIR_NT
#define DDSP_FIR_NT 11
#endif
#if DDSP_FIR_NT & 1
#define FB_LD_X ld_nt
#else
#define FB_LD_X ld
#endif
#if DDSP_FIR_NT & 2
#define FB_LD_T ld_nt
#else
#define FB_LD_T ld
#endif
#if DDSP_FIR_NT & 4
#define FB_ST st_nt
#else
#define FB_ST st
#endif
#if DDSP_FIR_NT & 8
#define FB_LD_A ld_nt
#else
#define FB_LD_A ld
#endif
__global__ void __launch_bounds__(128, 3) k_fir_blk6(FirJobs jobs, FirBlkGeom g, NoiseGen rng) {
  int jb = SEQ ? 0 : (int)blockIdx.y;
seq_next:
 {
  const FirJob& J = jobs.j[jb];
  const float* __restrict__ x = J.x;
  const int x_is_u01 = J.x_is_u01;
  const bool draw = RNG && J.rng != 0;                      // workgroup-uniform
  const float* __restrict__ taps = J.taps;
  const float* __restrict__ addend = J.addend;
  float* __restrict__ out = J.out;
  float* __restrict__ out_plain = J.out_plain;
  using PL = fft::Plan1024P;
  constexpr int NF = PL::N, P = PL::P, S = 8;
  __shared__ __attribute__((aligned(16))) f32x2 ex[3][PL::WORDS];
  f32x2* const bX = ex[0];
  f32x2* const bY = ex[1];
  f32x2* const bC = ex[2];
  const int tid = threadIdx.x;
  const int b = blockIdx.x / g.runs_per_utt;
  const int run_no = (int)((blockIdx.x - b * g.runs_per_utt + b + (b >> 4) + (b >> 8)) % g.runs_per_utt);
  const int q_first = (int)(((long)run_no * g.pairs) / g.runs_per_utt);
  const int q_last = (int)(((long)(run_no + 1) * g.pairs) / g.runs_per_utt);
  const int SH = FB_HOP / 2 - (g.N >> 1);
  const int bu = __builtin_amdgcn_readfirstlane(b);
  const float* xb = x + (long)bu * g.T;
  const int tap_ld = J.taps_half ? (g.N >> 1) + 1 : g.N;    // floats per tap row (workgroup-uniform)
  const float* tb = taps + (long)bu * g.F * tap_ld;
  const long ob = (long)bu * g.T;
  const BufF32 out_buf = BufF32::make(out + ob, g.T);
  const BufF32 plain_buf = BufF32::make(out_plain ? out_plain + ob : out + ob, out_plain ? g.T : 0);
  const BufF32 add_buf = BufF32::make(addend ? addend + ob : out + ob, addend ? g.T : 0);
  const int tid4 = 4 * tid;
  // the Bartlett weights of this thread's four samples of a block: lambda_m = (128 m + tid) / 512 = lam0 + m / 4 and
  // 1 - lambda_m = (1 - m / 4) - lam0, all exact.  Only lam0 is kept; the pairs are formed where they are used (one packed
  // add each) -- as loop invariants they would hold eight registers that this kernel does not have
  const float lam0 = (float)tid * (1.0f / (float)FB_HOP);
  float tail[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) tail[m] = 0.f;
  struct TapRow { float v[4]; };
  int tap_off[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int i = P * m + tid - SH;
    const int ih = i >= g.N ? -1 : ((J.taps_half && i > (g.N >> 1)) ? g.N - i : i);   // an even response: tap N - j is tap j
    tap_off[m] = ih >= 0 ? 4 * ih : BufF32::kOutOfRange;
  }
  // live == false (workgroup-uniform): the run ends before this row / block would be used -- the descriptor then spans 0 bytes
  // and the loads touch no memory (a load is issued a pass ahead of its use, so every run used to fetch 8 - 12 KB past its end)
  auto load_taps = [&](int j, bool live = true) -> TapRow {
    TapRow r;
    const int row = j < g.F ? j : g.F - 1;                     // core.py:167
    const BufF32 tr = BufF32::make(tb + (long)row * tap_ld, live ? tap_ld : 0);
#pragma unroll
    for (int m = 0; m < 4; ++m) r.v[m] = tr.FB_LD_T(tap_off[m]);
    return r;
  };
  struct Blk { float v[4]; };
  auto load_blk = [&](int bi, bool live = true) -> Blk {
    Blk r;
    if (draw) {
      const Quad q = philox_uniform4(rng, (unsigned)bu, (unsigned)bi, (unsigned)tid);
#pragma unroll
      for (int m = 0; m < 4; ++m) r.v[m] = bi < g.F ? q.u[m] : 0.f;
      return r;
    }
    const BufF32 xr = BufF32::make(xb + (long)bi * FB_HOP, bi < g.F && live ? FB_HOP : 0);
#pragma unroll
    for (int m = 0; m < 4; ++m) r.v[m] = xr.FB_LD_X(tid4 + 4 * P * m);
    return r;
  };
  auto pack_taps = [&](const TapRow& ta, const TapRow& tb2, f32x2 (&z)[S]) {
#pragma unroll
    for (int m = 0; m < 4; ++m) z[m] = f32x2{ta.v[m], tb2.v[m]};
#pragma unroll
    for (int m = 4; m < S; ++m) z[m] = f32x2{0.f, 0.f};
  };
  auto pack_blk = [&](const Blk& cx, bool live, f32x2 (&z)[S]) {
    const bool u01 = (draw || x_is_u01) && live;                // noise = rand * 2 - 1 (vocoder.py:603,854)
    const float ua = u01 ? 2.0f : 1.0f, ub = u01 ? -1.0f : 0.0f;
    float l0 = lam0;
    asm volatile("" : "+v"(l0));                                // not a loop invariant (see lam0)
    const f32x2 lp = {-l0, l0};
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const float xv = fmaf(ua, cx.v[m], ub);
      const f32x2 w = f32x2{1.0f - 0.25f * (float)m, 0.25f * (float)m} + lp;   // (1 - lambda, lambda)
      z[m] = w * f32x2{xv, xv};
    }
#pragma unroll
    for (int m = 4; m < S; ++m) z[m] = f32x2{0.f, 0.f};
  };
  const int kS0 = PL::s_index(tid, 0);
  const int kP0 = PL::parked(kS0);
  // the mirror image of slot m, parked, is at mb - 64 m (fft_1024p.h); not for slot 0 of threads 0, 1 (own_mirror)
  const int mb = PL::mirror_base(tid);
  auto mirrored = [&](const f32x2* X, int m) -> f32x2 { return PL::rd_parked(X + mb - 64 * m); };
  const bool own_mirror = tid < 2;
  const float sg = (tid & 1) ? -1.0f : 1.0f;
  const float nsg = -sg;
  // bins 0 and 512 (slot 0 of threads 0, 1) are their own mirror images: every other bin's mirror image lives on a thread of
  // the other parity and comes back with the opposite sign (layout S-); for these two the thread's own value, negated, takes
  // the place of the read, so that the split is one formula for all bins
  auto park = [&](const f32x2 (&z)[S], f32x2* X) {
#pragma unroll
    for (int m = 0; m < S; ++m) X[kP0 + 64 * m] = z[m];
  };
  const float ch = 0.5f / (float)NF;
  const f32x2 kG1 = {-ch, ch};
  const f32x2 kMi = {0.5f * ch, -0.5f * ch};
  // the split of k_fir_blk's split_taps and its products in one sweep over the bins: from T (own value z, mirror image tn,
  // both read from Cp) and the carried Hc:  G1 = -c conj tn,  G0 = Hc - i (c/2)(z - conj tn),  Hc' = -i (c/2)(z + conj tn);
  // W0 = Z0 G0, W1 = Z1 G1;  A = W0 + i W1 replaces za, B = W0 - i W1 is parked for the mirrored read
  auto split_and_products = [&](f32x2 (&za)[S], const f32x2 (&zb)[S], f32x2 (&Hc)[S], const f32x2* Tp, f32x2* Bp, f32x2& B0) {
#pragma unroll
    for (int m = 0; m < S; m += 2) {
      const f32x2 o0 = PL::rd_parked(Tp + kP0 + 64 * m), o1 = PL::rd_parked(Tp + kP0 + 64 * (m + 1));
      f32x2 tn0 = mirrored(Tp, m);
      const f32x2 tn1 = mirrored(Tp, m + 1);
      if (m == 0) tn0 = own_mirror ? -o0 : tn0;
      const f32x2 p0 = fft::sub_conj(o0, tn0), p1 = fft::sub_conj(o1, tn1);
      const f32x2 d0 = fft::add_conj(o0, tn0), d1 = fft::add_conj(o1, tn1);
      const f32x2 g10 = tn0 * kG1, g11 = tn1 * kG1;
      const f32x2 g00 = fft::swap_scale_add(p0, kMi, Hc[m]), g01 = fft::swap_scale_add(p1, kMi, Hc[m + 1]);
      Hc[m] = fft::swap_scale(d0, kMi);
      Hc[m + 1] = fft::swap_scale(d1, kMi);
      const f32x2 t0 = fft::cmul_lo(za[m], g00), t1 = fft::cmul_lo(zb[m], g10);
      const f32x2 t2 = fft::cmul_lo(za[m + 1], g01), t3 = fft::cmul_lo(zb[m + 1], g11);
      const f32x2 w0 = fft::cmul_hi(za[m], g00, t0), w1 = fft::cmul_hi(zb[m], g10, t1);
      const f32x2 w2 = fft::cmul_hi(za[m + 1], g01, t2), w3 = fft::cmul_hi(zb[m + 1], g11, t3);
      za[m] = fft::sub_mi(w0, w1);
      za[m + 1] = fft::sub_mi(w2, w3);
      const f32x2 b0v = fft::add_mi(w0, w1);
      if (m == 0) B0 = b0v;
      Bp[kP0 + 64 * m] = b0v;
      Bp[kP0 + 64 * (m + 1)] = fft::add_mi(w2, w3);
    }
  };
  auto hermitian = [&](f32x2 (&a)[S], const f32x2* Bp, f32x2 B0) {
#pragma unroll
    for (int m = 0; m < S; ++m) {
      f32x2 bm = mirrored(Bp, m);
      if (m == 0) bm = own_mirror ? B0 : bm;                    // B is plain layout S: the own value as it is
      a[m] = fft::add_conj(bm, a[m]);
    }
  };

  // the warm-up pass q_first - 1 is a pass of the same loop (see k_fir_blk): [rows b_w, b_w+1 | block b_w] where a pass of
  // the run has its two blocks; its tap transform is parked in Cp where every other pass finds its predecessor's
  const int bw = 2 * q_first - 1;
  TapRow t1 = load_taps(bw > 0 ? bw : 0), t2 = load_taps(bw + 1);
  Blk x0 = load_blk(bw >= 0 ? bw : g.F), x1 = x0;
  typename PL::Tw tw;
  tw.init(tid);
  typename PL::Ix ix;
  ix.init(tid);
  f32x2 Hc[S];
#pragma unroll
  for (int m = 0; m < S; ++m) Hc[m] = f32x2{0.f, 0.f};
  const bool has_add = addend != nullptr, has_plain = out_plain != nullptr;
  const int turn = __builtin_amdgcn_s_getreg(0x1804) & 1;
  const int turns_mask = g.turns ? 1 : 0;
  float add[S];                                                 // the addend of the 1024 samples a pass emits (zeros without one)
#pragma unroll
  for (int i = 0; i < S; ++i) add[i] = 0.f;
  for (int q = q_first - 1; q < q_last; ++q) {
    const bool warm = q < q_first;                              // workgroup-uniform
    if ((q + turn) & turns_mask) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
    const int b0 = 2 * q;
    f32x2 z0[S], z1[S];
#if defined(DDSP_B6_HALF_PASS_AHEAD)
    if (warm) pack_taps(t1, t2, z0);                            // rows b_w, b_w + 1
    else pack_blk(x0, b0 < g.F, z0);
    // the tap rows of the next pass: their transform rides behind this pass's inverse
    t1 = load_taps(b0 + 3);
    t2 = load_taps(b0 + 4);
    PL::template forward_s<true, true>(z0, tw, bX, bY, ix);
    if (warm) pack_blk(x0, bw >= 0, z1);                        // block b_w
    else pack_blk(x1, b0 + 1 < g.F, z1);
    PL::template forward_s<true, true>(z1, tw, bX, bY, ix);
#else
    // Every global load is issued a whole pass ahead of its use, into the registers its predecessor has just left (inside
    // a step the blocks and tap rows come from HBM, written by the kernel before: half a pass does not cover that --
    // measured alone, on inputs that sit in the memory-side cache, 76 us; inside the step 88; profiles/r04_v5_*)
    const bool next_pass = q + 1 < q_last, pass_after = q + 2 < q_last;      // workgroup-uniform
    if (warm) {
      pack_taps(t1, t2, z0);                                    // rows b_w, b_w + 1
      t1 = load_taps(b0 + 3, next_pass);                        // the rows whose transform rides behind THIS pass's inverse
      t2 = load_taps(b0 + 4, next_pass);
    } else {
      pack_blk(x0, b0 < g.F, z0);
      x0 = load_blk(b0 + 2, next_pass);                         // the next pass's first block
    }
    PL::template forward_s<true, true>(z0, tw, bX, bY, ix);
    if (warm) {
      pack_blk(x0, bw >= 0, z1);                                // block b_w
      x0 = load_blk(b0 + 2, next_pass);
    } else {
      pack_blk(x1, b0 + 1 < g.F, z1);
    }
    x1 = load_blk(b0 + 3, next_pass);
    PL::template forward_s<true, true>(z1, tw, bX, bY, ix);
#endif
    if (warm) {
      park(z0, bC);                                             // Cp has no readers yet
      __syncthreads();
#pragma unroll
      for (int m = 0; m < S; ++m) z0[m] = f32x2{0.f, 0.f};     // the even block of this pass is absent
    }
    f32x2 B0;
    split_and_products(z0, z1, Hc, bC, bX, B0);                 // z0 := A
    __syncthreads();
    hermitian(z0, bX, B0);                                          // z0 = conj(Y_b0 + i Y_b0+1)
#if defined(DDSP_B6_HALF_PASS_AHEAD)
    x0 = load_blk(b0 + 2);
    x1 = load_blk(b0 + 3);
#endif
    // fetched now: this pass's addend
    const int e0 = b0 * FB_HOP - 256 + tid;                     // first emitted time of this thread
    const int off_a = b0 > 0 ? 4 * e0 : BufF32::kOutOfRange;
    const int off_b = 4 * e0 + 8 * P;
    auto t_off = [&](int i) -> int { return i < 2 ? off_a + 4 * P * i : off_b + 4 * P * (i - 2); };
#if defined(DDSP_B6_HALF_PASS_AHEAD)
    if (has_add && !warm) {
#pragma unroll
      for (int i = 0; i < S; ++i) add[i] = add_buf.FB_LD_A(t_off(i));
    }
#endif
    f32x2 zt[S];
    pack_taps(t1, t2, zt);
#if !defined(DDSP_B6_HALF_PASS_AHEAD)
    t1 = load_taps(b0 + 5, pass_after);                         // the rows of the NEXT pass's second stage (its transform serves the pass after)
    t2 = load_taps(b0 + 6, pass_after);
#endif
    PL::template transposed_then_forward_s<true>(z0, zt, tw, bY, bX, bC, ix);
    park(zt, bC);                                               // read by the next pass's product, behind its first stage's barriers
    if (!warm) {
      float d[S];
#pragma unroll
      for (int i = 0; i < S; ++i) d[i] = i < 4 ? z0[i].x - tail[i] : z0[i].x - z0[i - 4].y;
      if (has_plain) {
#pragma unroll
        for (int i = 0; i < S; ++i) plain_buf.FB_ST(sg * d[i], t_off(i));
      }
#pragma unroll
      for (int i = 0; i < S; ++i) out_buf.FB_ST(fmaf(sg, d[i], add[i]), t_off(i));
      if (q == g.pairs - 1) {                                   // the last pair also emits the upper half of its second block
#pragma unroll
        for (int m = 4; m < S; ++m) {
          const int off = t_off(4 + m);
          const float v = nsg * z0[m].y;
          if (has_plain) plain_buf.FB_ST(v, off);
          out_buf.FB_ST(v + (has_add ? add_buf.FB_LD_A(off) : 0.f), off);
        }
      }
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) tail[m] = z0[4 + m].y;
#if !defined(DDSP_B6_HALF_PASS_AHEAD)
    if (has_add && q + 1 < q_last) {                            // the NEXT pass's addend (times b0 + 2 on: never negative), a pass ahead like every load
      const int n0 = e0 + 2 * FB_HOP;                           // the next pass's first emitted time; negative only for b0 + 2 = 0
      const int nx_a = b0 + 2 > 0 ? 4 * n0 : BufF32::kOutOfRange, nx_b = 4 * n0 + 8 * P;
#pragma unroll
      for (int i = 0; i < S; ++i) add[i] = add_buf.FB_LD_A(i < 2 ? nx_a + 4 * P * i : nx_b + 4 * P * (i - 2));
    }
#endif
  }
 }
  if (SEQ && jb == 0) {                                        // job 1 over the same run: its addend is what this thread has just stored
    jb = 1;
    __builtin_amdgcn_s_waitcnt(0);                              // the stores have reached the L2
    __syncthreads();                                            // every wave has left job 0's last exchange
    goto seq_next;
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
                   int B, int F, int hop, int N, hipStream_t st, const NoiseGen* noise_gen, const FirSecond* second, int taps_half) {
  if (hop != FB_HOP || N < 2 || (N & 1) || N > 512 || (long)F * hop >= (1L << 28)) return -1;   // one utterance stays below 2^30 bytes (buffer descriptors, BufF32::kOutOfRange)
  FirBlkGeom g;
  g.F = F; g.N = N; g.T = F * hop;
  g.pairs = (F + 1) / 2;
  // three waves per SIMD (k_fir_blk6, six workgroups per CU) unless knob BLK_WPS = 2 asks for the two-wave kernel
  // (k_fir_blk, four per CU: the round-3 form, kept for same-box A/B runs)
  int wps = 3;
#ifdef DDSP_AB_GENERATIONS
  if (const long v = knob(KNOB_BLK_WPS)) { if (v >= 1) wps = (int)v; }
#endif
  // run length: as many workgroups as the chip holds at once (2 waves each), one round, equal work; every run
  // but an utterance's first pays one warm-up block (three transforms; a pair costs four)
  const long slots = (long)wps * 2 * 256;
  const int Bg = t_geometry_batch > 0 ? t_geometry_batch : B;          // kernels.h: a sub-batch keeps the whole call's split
  long per_utt = slots / (Bg > 0 ? Bg : 1);
  // two jobs share the one round of resident workgroups: runs twice as long, half the warm-up passes (at streaming shapes every
  // workgroup is resident anyway and the split stays what the one-job launches of the same shape use: same bits)
  if (second && !second->seq && (long)Bg * F >= kSmallRows) per_utt = slots / (2 * (Bg > 0 ? Bg : 1));
  if (per_utt < 1) per_utt = 1;
  int run = (int)((g.pairs + per_utt - 1) / per_utt);
  // at least three pairs per workgroup (each pays a warm-up block and its twiddles) -- except at streaming shapes, where every
  // workgroup is resident at once anyway and the launch is as long as its longest workgroup: one pair each
  if (run < 3 && (long)Bg * F >= kSmallRows) run = 3;
  if (const long v = knob(KNOB_BLK_RUN)) { if (v >= 1) run = (int)v; }
  if (run > g.pairs) run = g.pairs;
  g.run = run;
  g.runs_per_utt = (g.pairs + run - 1) / run;
  g.turns = knob(KNOB_BLK_TURNS) == 1 ? 0 : 1;
  const long wgs = (long)B * g.runs_per_utt;
  if (wgs > 0x7fffffffL) return -1;
  NoiseGen rng{0ull, 0ull, 0};
  FirJobs jobs;
#ifdef DDSP_AB_GENERATIONS
  if (taps_half || (second && second->taps_half)) return -1;   // (the two-wave kernel reads whole rows)
#endif
  jobs.j[0] = FirJob{x, x_is_u01, taps, addend, out, out_plain, 0, taps_half};
  jobs.j[1] = jobs.j[0];
  if (second && second->seq && noise_gen && noise_gen->on) return -1;   // (the chained form has no in-kernel draw)
  if (noise_gen && noise_gen->on && second) {                   // two jobs, the SECOND one's input drawn in the kernel (its x may be null)
#ifdef DDSP_AB_GENERATIONS
    if (wps < 3) return -1;
#endif
    rng = *noise_gen;
    jobs.j[1] = FirJob{second->x, 0, second->taps, second->addend, second->out, second->out_plain, 1, second->taps_half};
    hipLaunchKernelGGL((k_fir_blk6<true>), dim3((unsigned)wgs, 2u), dim3(128), 0, st, jobs, g, rng);
    return 5;
  }
  if (noise_gen && noise_gen->on) {                             // the input is drawn in the kernel (x may be null)
    rng = *noise_gen;
    jobs.j[0].x_is_u01 = 0;
    jobs.j[0].rng = jobs.j[1].rng = 1;
#ifdef DDSP_AB_GENERATIONS
    if (wps < 3) {
      hipLaunchKernelGGL((k_fir_blk<2, true>), dim3((unsigned)wgs), dim3(128), 0, st, x, 0, taps, addend, out, out_plain, g, rng);
      return 5;
    }
#endif
    hipLaunchKernelGGL((k_fir_blk6<true>), dim3((unsigned)wgs), dim3(128), 0, st, jobs, g, rng);
    return 5;
  }
  if (second && second->seq) {                                  // job 0, then job 1 (which reads job 0's result as its addend) in ONE row of workgroups
    if (wps < 3 || (noise_gen && noise_gen->on)) return -1;
    jobs.j[1] = FirJob{second->x, second->x_is_u01, second->taps, second->addend, second->out, second->out_plain, 0, second->taps_half};
    hipLaunchKernelGGL((k_fir_blk6<false, true>), dim3((unsigned)wgs), dim3(128), 0, st, jobs, g, rng);
    return 5;
  }
  if (second && wps >= 3) {                                     // two independent filters of this shape in one launch
    jobs.j[1] = FirJob{second->x, second->x_is_u01, second->taps, second->addend, second->out, second->out_plain, 0, second->taps_half};
    hipLaunchKernelGGL((k_fir_blk6<false>), dim3((unsigned)wgs, 2u), dim3(128), 0, st, jobs, g, rng);
    return 5;
  }
  if (second) return -1;
#ifdef DDSP_AB_GENERATIONS
  size_t pad = 0;                                               // occupancy probe: extra dynamic LDS per workgroup
  if (const long v = knob(KNOB_BLK_PADLDS)) { if (v > 0) pad = (size_t)v; }
  if (wps < 3 || pad != 0) {
    hipLaunchKernelGGL((k_fir_blk<2, false>), dim3((unsigned)wgs), dim3(128), pad, st, x, x_is_u01, taps, addend, out, out_plain, g, rng);
    return 5;
  }
#endif
  hipLaunchKernelGGL((k_fir_blk6<false>), dim3((unsigned)wgs), dim3(128), 0, st, jobs, g, rng);
  return 5;
}

}  // namespace ddsp

#ifdef DDSP_HIP_TIMELINE
extern "C" int ddsp_hip_debug_set_blk_timeline(long long* p, void* stream) {
  hipLaunchKernelGGL(ddsp::k_set_blk_timeline, dim3(1), dim3(1), 0, (hipStream_t)stream, p);
  return (int)hipGetLastError();
}
#endif
