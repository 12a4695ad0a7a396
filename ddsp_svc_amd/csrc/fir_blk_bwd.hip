// Adjoints of the time-varying FIR (what autograd returns for ddsp/core.py:120-182 fft_convolve): gradients w.r.t.
// the tap rows and, optionally, the input signal, in the hop-block form and with the geometry of fir_blk.hip.
//
// Forward, per hop block b (1024-point circular, alias-free):  r_b = Re IFFT(Z_b G_b),
//   Z_b = FFT(x_b (1 - lambda) + i x_b lambda),  G_b = H_b - i H_b+1  (H_j: tap row min(j, F-1) shifted by 256 - N/2),
//   and r_b[n] is added to the output at time (b - 1/2) hop + n.  With seg_b[n] = grad_out[(b - 1/2) hop + n] (zero
//   outside the signal; no masking to the block's support is needed: for a tap position inside its row and an input
//   position inside its block the circular correlations below never wrap) and S_b = FFT(seg_b):
//     U_b = conj(Z_b) S_b   <->  u_b = (x1_b star seg_b) - i (x2_b star seg_b):   d h_b += Re u_b,  d h_min(b+1,F-1) -= Im u_b
//     D_b = conj(G_b) S_b   <->  (h_b star seg_b) + i (h_b+1 star seg_b):         d x_b[s] = (1 - lambda_s) Re + lambda_s Im
//   -- ONE complex product per bin each, no separation of the packed block transform (round 2 separated X1, X2 of both
//   blocks and formed four conjugate products per bin with scalar arithmetic: 256 registers, 84 spilled dwords, 0.43 ms).
// Tap rows j, j + 1 of a pair come out of ONE inverse, V = R_j + i R_j+1 with the Hermitian spectra
//     R_j = (U_j + ~U_j) / 2 + i (B_j - ~B_j) / 2,   ~X[k] = conj X[-k],   B_j = U_j-1 (+ U_j on the last row, which also
//     takes its own block's second term: core.py:167 holds the last row),
//   which is V[k] = N[k] + conj M[-k],  N = (U_j - B_j+1) / 2 + i (B_j + U_j+1) / 2,  M = (U_j + B_j+1) / 2 + i (B_j - U_j+1) / 2:
//   one parked array and one mirrored read per pair, as in the forward kernel.  The cotangent segments of the two blocks
//   ride in one transform C = FFT(seg_b0 + i seg_b0+1) and are split (S_b0, S_b0+1) like a pair of tap rows there.
// Per pair of blocks: [Z_b0 | Z_b0+1] in lockstep, then [inverse of V | C of the NEXT pair] in lockstep -- four
//   transforms, two stages, six barriers, the forward kernel's shape.  With the input gradient: [inverse of D_b0 | tap rows
//   of the next pair] and the inverse of D_b0+1 in between (seven transforms).  U_b0+1 is carried to the next pair; a run
//   starts one pair early to rebuild it (and the tap / cotangent spectra), storing nothing.  No overlap-add, no atomics:
//   every tap row and every input block is written by exactly one workgroup.  Spectra live in the sign-carrying layout S-
//   of fft_r.h; products of two of them are plain layout S.
#include "fft_1024p.h"
#include "kernels.h"
#include "tuning.h"
#include <stdlib.h>

namespace ddsp {

constexpr int FBW_HOP = 512;

struct FirBwdGeom {
  int F, N, T;
  int pairs;              // block pairs per utterance: ceil(F / 2)
  int run, runs_per_utt;
};

template <bool WITH_DX>
__global__ void __launch_bounds__(128, 2) k_fir_blk_bwd(const float* __restrict__ x, int x_is_u01,
                                                       const float* __restrict__ taps,
                                                       const float* __restrict__ grad_out, float* __restrict__ d_x,
                                                       float* __restrict__ d_taps, FirBwdGeom g) {
  using PL = fft::Plan<2>;
  constexpr int NF = PL::N, P = PL::P, S = 8;
  __shared__ __attribute__((aligned(16))) f32x2 ex[4][NF];
  // the carried tap spectrum c H_j+1 of split_taps (input gradient only): a thread's own eight values, parked here between
  // the one read and the one write a pass makes -- sixteen registers the first stage does not have to spare
  __shared__ __attribute__((aligned(16))) f32x2 hc_park[WITH_DX ? S * 128 : 1];
  f32x2* const bA = ex[0];
  f32x2* const bB = ex[1];
  f32x2* const bC = ex[2];
  f32x2* const bD = ex[3];
  const int tid = threadIdx.x;
  const int b = __builtin_amdgcn_readfirstlane(blockIdx.x / g.runs_per_utt);
  const int run_no = blockIdx.x - b * g.runs_per_utt;
  const int q_first = run_no * g.run;
  int q_last = q_first + g.run;
  if (q_last > g.pairs) q_last = g.pairs;
  const int SH = FBW_HOP / 2 - (g.N >> 1);                     // tap shift of fir_blk.hip: row centred on transform index 256
  const float* xb = x + (long)b * g.T;
  const float* tb = taps + (long)b * g.F * g.N;
  float* dtb = d_taps + (long)b * g.F * g.N;
  const BufF32 g_buf = BufF32::make(grad_out + (long)b * g.T, g.T);
  const float inv_hop = 1.0f / (float)FBW_HOP;
  const float sg = (tid & 1) ? -1.0f : 1.0f;                   // the sign layout S- puts on this thread's bins / samples
  const int tid4 = 4 * tid;

  // ---- loads (buffer descriptors: whatever lies outside a row, a block or the signal reads zero / is not stored) ----
  struct Four { float v[4]; };
  int tap_off[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int i = P * m + tid - SH;
    tap_off[m] = i >= 0 ? 4 * i : BufF32::kOutOfRange;
  }
  auto load_taps = [&](int j) -> Four {                          // row min(j, F-1), clamped below at 0; only m < 4 can be live
    Four r;
    const int row = j < 0 ? 0 : (j < g.F ? j : g.F - 1);
    const BufF32 tr = BufF32::make(tb + (long)row * g.N, g.N);
#pragma unroll
    for (int m = 0; m < 4; ++m) r.v[m] = tr.ld(tap_off[m]);
    return r;
  };
  auto load_blk = [&](int bi) -> Four {
    Four r;
    const bool live = bi >= 0 && bi < g.F;
    const BufF32 xr = BufF32::make(xb + (long)(live ? bi : 0) * FBW_HOP, live ? FBW_HOP : 0);
#pragma unroll
    for (int m = 0; m < 4; ++m) r.v[m] = xr.ld(tid4 + 4 * P * m);
    return r;
  };
  // the cotangent of pair qn: twelve values L[i] = grad_out[(2 qn - 1/2) hop + 128 i + tid]; block 2 qn takes i = 0..7, block
  // 2 qn + 1 takes i = 4..11.  The sign of a time is the same for every lane (multiples of 128 plus tid), so times before
  // the signal get the out-of-range constant from a scalar select.
  struct Seg { float v[12]; };
  auto load_seg = [&](int qn) -> Seg {
    Seg r;
    const int t0 = 2 * qn * FBW_HOP - FBW_HOP / 2;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const int t = t0 + P * i;
      r.v[i] = g_buf.ld(t >= 0 ? 4 * (t + tid) : BufF32::kOutOfRange);
    }
    return r;
  };
  auto pack_seg = [&](const Seg& s, f32x2 (&z)[S]) {
#pragma unroll
    for (int m = 0; m < S; ++m) z[m] = f32x2{s.v[m], s.v[m + 4]};
  };
  auto pack_taps = [&](const Four& ta, const Four& tb2, f32x2 (&z)[S]) {
#pragma unroll
    for (int m = 0; m < 4; ++m) z[m] = f32x2{ta.v[m], tb2.v[m]};
#pragma unroll
    for (int m = 4; m < S; ++m) z[m] = f32x2{0.f, 0.f};
  };
  auto pack_blk = [&](const Four& cx, bool live, f32x2 (&z)[S]) {
    const float ua = x_is_u01 ? 2.0f : 1.0f, ub = (x_is_u01 && live) ? -1.0f : 0.0f;   // noise = rand * 2 - 1, uniform
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const float xv = fmaf(ua, cx.v[m], ub);
      const float lam = (float)(P * m + tid) * inv_hop;
      z[m] = f32x2{(1.0f - lam) * xv, lam * xv};
    }
#pragma unroll
    for (int m = 4; m < S; ++m) z[m] = f32x2{0.f, 0.f};
  };

  // ---- spectra (layout S-, fft_r.h; parking and the self-mirrored bins as in fir_blk.hip) ----
  const int kS0 = PL::s_index(tid, 0);
  const int kP0 = PL::parked(kS0);
  auto mirrored = [&](const f32x2* X, int m) -> f32x2 { return X[PL::parked((NF - (kS0 + 64 * m)) & (NF - 1))]; };
  const float self_mirror = tid < 2 ? -1.0f : 1.0f;
  auto park_pair = [&](const f32x2 (&z)[S], f32x2* X) {          // a packed transform of two real sequences, to be split
    X[kP0] = z[0] * f32x2{self_mirror, self_mirror};
#pragma unroll
    for (int m = 1; m < S; ++m) X[kP0 + 64 * m] = z[m];
  };
  // From P = FFT(a + i b) (z in S-, parked in Zp): Sa = c FFT(a), Sb = c FFT(b), both in S-
  const float cs = 0.5f / (float)NF;                           // the 1/2 of the Hermitian combination N + conj M and the 1/N of the inverse
  const f32x2 kHa = {0.5f * cs, 0.5f * cs};
  const f32x2 kMi = {0.5f * cs, -0.5f * cs};                   // times (-i) after the half swap of swap_scale
  auto split_pair = [&](const f32x2 (&z)[S], const f32x2* Zp, f32x2 (&Sa)[S], f32x2 (&Sb)[S]) {
#pragma unroll
    for (int m = 0; m < S; m += 2) {
      const f32x2 n0 = mirrored(Zp, m), n1 = mirrored(Zp, m + 1);
      const f32x2 p0 = fft::sub_conj(z[m], n0), p1 = fft::sub_conj(z[m + 1], n1);     // 2 sigma FFT(a)
      const f32x2 d0 = fft::add_conj(z[m], n0), d1 = fft::add_conj(z[m + 1], n1);     // 2 i sigma FFT(b)
      Sa[m] = p0 * kHa;
      Sa[m + 1] = p1 * kHa;
      Sb[m] = fft::swap_scale(d0, kMi);
      Sb[m + 1] = fft::swap_scale(d1, kMi);
    }
  };
  // tap rows: G1 = c (H_j - i H_j+1) = c conj T[-k], G0 = c (H_j-1 - i H_j), Hc' = c H_j+1 (fir_blk.hip, split_taps); c = 2:
  // the input gradient's inverse has no Hermitian combination, and the cotangent spectra it meets carry 1 / 2N
  const float ct = 2.0f;
  const f32x2 kG1 = {-ct, ct};
  const f32x2 kTi = {0.5f * ct, -0.5f * ct};
  auto split_taps = [&](const f32x2 (&z)[S], const f32x2* Zp, f32x2 (&G0)[S], f32x2 (&G1)[S]) {
    f32x2 Hc[S];
#pragma unroll
    for (int m = 0; m < S; ++m) Hc[m] = hc_park[128 * m + tid];
#pragma unroll
    for (int m = 0; m < S; m += 2) {
      const f32x2 n0 = mirrored(Zp, m), n1 = mirrored(Zp, m + 1);
      const f32x2 p0 = fft::sub_conj(z[m], n0), p1 = fft::sub_conj(z[m + 1], n1);
      const f32x2 d0 = fft::add_conj(z[m], n0), d1 = fft::add_conj(z[m + 1], n1);
      G1[m] = n0 * kG1;
      G1[m + 1] = n1 * kG1;
      G0[m] = fft::swap_scale_add(p0, kTi, Hc[m]);
      G0[m + 1] = fft::swap_scale_add(p1, kTi, Hc[m + 1]);
      hc_park[128 * m + tid] = fft::swap_scale(d0, kTi);
      hc_park[128 * (m + 1) + tid] = fft::swap_scale(d1, kTi);
    }
  };

  // ---- preamble: what the warm-up pass q_first - 1 needs -- the cotangent spectra of its pair and (input gradient) the
  // spectrum of tap row 2 q_first - 2 ... see the loop; an utterance's first run warms up on blocks and times before the
  // signal, which read zero ----
  typename PL::Tw tw;
  Seg sgm = load_seg(q_first - 1);
  Four t1 = load_taps(2 * q_first - 3), t2 = load_taps(2 * q_first - 2);
  Four x0 = load_blk(2 * q_first - 2), x1 = load_blk(2 * q_first - 1);
  tw.init(tid);
  f32x2 S0[S], S1[S], Uc[S], G0[S], G1[S];
#pragma unroll
  for (int m = 0; m < S; ++m) {
    Uc[m] = f32x2{0.f, 0.f};
    if (WITH_DX) hc_park[128 * m + tid] = f32x2{0.f, 0.f};
    G0[m] = f32x2{0.f, 0.f};
    G1[m] = f32x2{0.f, 0.f};
  }
  {
    f32x2 zc[S];
    pack_seg(sgm, zc);
    if (WITH_DX) {
      f32x2 zt[S];
      pack_taps(t1, t2, zt);                                     // rows 2 q_first - 3, 2 q_first - 2: leaves Hc = c H_(2 q_first - 2)
      PL::template forward_s2<false, true>(zc, zt, tw, bA, bC, bB, bD, tid);
      park_pair(zc, bA);
      park_pair(zt, bB);
      __syncthreads();
      split_taps(zt, bB, G0, G1);
    } else {
      PL::template forward_s<false, true>(zc, tw, bA, bC, tid);
      park_pair(zc, bA);
      __syncthreads();
    }
    split_pair(zc, bA, S0, S1);
  }
  __syncthreads();                                              // every wave is done with bA / bB (the splits above) before they are written again
  if (WITH_DX) {                                                // rows 2 q_first - 1, 2 q_first: G0, G1 of the warm-up pair
    f32x2 zt[S];
    const Four ta = load_taps(2 * q_first - 1), tb2 = load_taps(2 * q_first);
    pack_taps(ta, tb2, zt);
    PL::template forward_s<true, true>(zt, tw, bB, bD, tid);
    park_pair(zt, bB);
    __syncthreads();
    split_taps(zt, bB, G0, G1);
    __syncthreads();
  }

  for (int q = q_first - 1; q < q_last; ++q) {
    const bool warm = q < q_first;                              // workgroup-uniform
    const int b0 = 2 * q;
    f32x2 z0[S], z1[S];
    pack_blk(x0, b0 >= 0 && b0 < g.F, z0);
    pack_blk(x1, b0 + 1 >= 0 && b0 + 1 < g.F, z1);
    // fetched now: what the later stages of this pass transform (the next pair's cotangent, tap rows) and the next pass's
    // blocks.  With the input gradient the register file is the limit (three more spectra are live through the first
    // stage): there only the tap rows are fetched here, the rest behind the stage that frees the registers.
    if (WITH_DX) {
      t1 = load_taps(b0 + 3);
      t2 = load_taps(b0 + 4);
    } else {
      x0 = load_blk(b0 + 2);
      x1 = load_blk(b0 + 3);
      sgm = load_seg(q + 1);
    }
    PL::template forward_s2<true, true>(z0, z1, tw, bA, bC, bB, bD, tid);
    // U_b = conj(Z_b) S_b (plain layout S); the tap-gradient spectrum of rows b0, b0 + 1
    f32x2 V[S];
#pragma unroll
    for (int m = 0; m < S; m += 2) {
      const f32x2 l0 = fft::cmul_lo(z0[m], S0[m]), l1 = fft::cmul_lo(z1[m], S1[m]);
      const f32x2 l2 = fft::cmul_lo(z0[m + 1], S0[m + 1]), l3 = fft::cmul_lo(z1[m + 1], S1[m + 1]);
      z0[m] = fft::cmulc_hi(z0[m], S0[m], l0);
      z1[m] = fft::cmulc_hi(z1[m], S1[m], l1);
      z0[m + 1] = fft::cmulc_hi(z0[m + 1], S0[m + 1], l2);
      z1[m + 1] = fft::cmulc_hi(z1[m + 1], S1[m + 1], l3);
    }
    if (WITH_DX) {
      // conj D_b = G_b conj(S_b) (inverse by the forward transform; d x_b = sigma ((1 - lambda) Re - lambda Im)), formed at once
      // and in place: the cotangent and filter spectra are dead from here on
#pragma unroll
      for (int m = 0; m < S; m += 2) {
        const f32x2 l0 = fft::cmul_lo(S0[m], G0[m]), l1 = fft::cmul_lo(S1[m], G1[m]);
        const f32x2 l2 = fft::cmul_lo(S0[m + 1], G0[m + 1]), l3 = fft::cmul_lo(S1[m + 1], G1[m + 1]);
        S0[m] = fft::cmulc_hi(S0[m], G0[m], l0);
        S1[m] = fft::cmulc_hi(S1[m], G1[m], l1);
        S0[m + 1] = fft::cmulc_hi(S0[m + 1], G0[m + 1], l2);
        S1[m + 1] = fft::cmulc_hi(S1[m + 1], G1[m + 1], l3);
      }
      sgm = load_seg(q + 1);
    }
    {
      const bool last0 = b0 == g.F - 1, last1 = b0 + 1 == g.F - 1;      // the held last row (core.py:167) also takes its own block's second term
#pragma unroll
      for (int m = 0; m < S; ++m) {
        const f32x2 Bj = last0 ? Uc[m] + z0[m] : Uc[m];
        const f32x2 Bj1 = last1 ? z0[m] + z1[m] : z0[m];
        const f32x2 Pp = Bj + z1[m], Qq = Bj - z1[m];
        const f32x2 T1 = z0[m] - Bj1, T2 = z0[m] + Bj1;
        V[m] = fft::sub_mi(T1, Pp);                              // N = (U_j - B_j+1) + i (B_j + U_j+1)   (the 1/2 is in S)
        bA[kP0 + 64 * m] = fft::sub_mi(T2, Qq);                  // M = (U_j + B_j+1) + i (B_j - U_j+1), parked
        Uc[m] = z1[m];                                           // carried: U of the pair's second block
      }
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < S; ++m) V[m] = fft::add_conj(mirrored(bA, m), V[m]);   // conj V = M[-k] + conj N[k]: inverse by the forward transform

    if (WITH_DX) {
      f32x2 zt[S];
      pack_taps(t1, t2, zt);                                     // rows b0 + 3, b0 + 4: the next pair's
      PL::template transposed_and_forward_s<true, true>(S0, zt, tw, bC, bA, bD, bB, tid);
      park_pair(zt, bC);
      __syncthreads();
      split_taps(zt, bC, G0, G1);
      x0 = load_blk(b0 + 2);
      x1 = load_blk(b0 + 3);
      auto store_dx = [&](const f32x2 (&W)[S], int bi) {
        const bool live = !warm && bi < g.F;
        const BufF32 dr = BufF32::make(d_x + (long)b * g.T + (long)(live ? bi : 0) * FBW_HOP, live ? FBW_HOP : 0);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const float lam = (float)(P * m + tid) * inv_hop;
          dr.st(sg * fmaf(1.0f - lam, W[m].x, -(lam * W[m].y)), tid4 + 4 * P * m);
        }
      };
      store_dx(S0, b0);
      if (!warm) {
        PL::template transposed<true>(S1, tw, bD, bA, tid);
        store_dx(S1, b0 + 1);
      } else {
        __syncthreads();                                        // (the barriers of that transform also keep the next stage's writes to
      }                                                         //  bC behind the split's reads of the parked tap spectrum)
    }

    // the tap rows' inverse beside the transform of the next pair's cotangent (a full 1024-point input)
    f32x2 zc[S];
    pack_seg(sgm, zc);
    PL::template transposed_and_forward_s<true, false>(V, zc, tw, bC, bA, bD, bB, tid);
    park_pair(zc, bC);
    __syncthreads();
    split_pair(zc, bC, S0, S1);
    // d_taps[j][i] = d h_j[i + SH]: transform index n = 128 m + tid < 512; row b0 = sigma Re, row b0 + 1 = -sigma Im
    {
      const bool own0 = !warm && b0 < g.F, own1 = !warm && b0 + 1 < g.F;
      const BufF32 r0 = BufF32::make(dtb + (long)(own0 ? b0 : 0) * g.N, own0 ? g.N : 0);
      const BufF32 r1 = BufF32::make(dtb + (long)(own1 ? b0 + 1 : 0) * g.N, own1 ? g.N : 0);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        r0.st(sg * V[m].x, tap_off[m]);
        r1.st(-sg * V[m].y, tap_off[m]);
      }
    }
  }
}

// ---- the tap gradient alone at three waves per SIMD (round 6) -------------------------------------------------------------
// Two of a training step's three filter adjoints need no input gradient (the noise filter's input is data, the all-pass filter's
// the exciter): four transforms per block pair, the forward kernel's count -- and k_fir_blk_bwd<false> above, round 3's form (226+
// registers, four 1024-word exchange buffers: two waves per SIMD), took 116 us where the forward filter takes 72.  This is
// k_fir_blk6's shape (fir_blk.hip) applied to it: the padded plan of fft_1024p.h, three exchange buffers, the two block transforms
// one after the other, the pair's inverse and the NEXT pair's cotangent transform staggered on three buffers, and the cotangent
// spectra S_b0, S_b0+1 NOT kept in registers: the packed transform C = FFT(seg_b0 + i seg_b0+1) stays parked in its buffer and
// the product sweep splits it bin by bin where it consumes it (own value and mirror image, as the forward kernel's tap
// transform).  Carried between passes: U of the pair's second block (16 registers, the forward kernel's Hc).  Every load is
// issued a pass ahead.  Same operator, same geometry rules, rounding-level differences to the two-wave kernel (knob BWD_WPS = 2).
// A launch takes one or two JOBS (grid.y): tap gradients of filters of the same shape that do not depend on each other -- the
// all-pass and the noise filter of a CombSub training step (ddsp_hip_combsub_tail_backward).
struct FirBwdJob { const float* x; int x_is_u01; const float* grad_out; float* d_taps; };
struct FirBwdJobs { FirBwdJob j[2]; };

template <int DUMMY = 0>
__global__ void __launch_bounds__(128, 3) k_fir_blk_bwd6(FirBwdJobs jobs, FirBwdGeom g) {
  const FirBwdJob& J = jobs.j[blockIdx.y];
  const float* __restrict__ x = J.x;
  const int x_is_u01 = J.x_is_u01;
  const float* __restrict__ grad_out = J.grad_out;
  float* __restrict__ d_taps = J.d_taps;
  using PL = fft::Plan1024P;
  constexpr int NF = PL::N, P = PL::P, S = 8;
  __shared__ __attribute__((aligned(16))) f32x2 ex[3][PL::WORDS];
  f32x2* const bX = ex[0];
  f32x2* const bY = ex[1];
  f32x2* const bC = ex[2];
  const int tid = threadIdx.x;
  const int b = __builtin_amdgcn_readfirstlane(blockIdx.x / g.runs_per_utt);
  const int run_no = blockIdx.x - b * g.runs_per_utt;
  const int q_first = (int)(((long)run_no * g.pairs) / g.runs_per_utt);          // an utterance's pairs split evenly over its runs
  const int q_last = (int)(((long)(run_no + 1) * g.pairs) / g.runs_per_utt);
  const int SH = FBW_HOP / 2 - (g.N >> 1);
  const float* xb = x + (long)b * g.T;
  float* dtb = d_taps + (long)b * g.F * g.N;
  const BufF32 g_buf = BufF32::make(grad_out + (long)b * g.T, g.T);
  const float sg = (tid & 1) ? -1.0f : 1.0f;
  const int tid4 = 4 * tid;
  const float lam0 = (float)tid * (1.0f / (float)FBW_HOP);
  int tap_off[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int i = P * m + tid - SH;
    tap_off[m] = i >= 0 ? 4 * i : BufF32::kOutOfRange;
  }
  struct Four { float v[4]; };
  auto load_blk = [&](int bi, bool live = true) -> Four {
    Four r;
    const bool in = live && bi >= 0 && bi < g.F;
    const BufF32 xr = BufF32::make(xb + (long)(in ? bi : 0) * FBW_HOP, in ? FBW_HOP : 0);
#pragma unroll
    for (int m = 0; m < 4; ++m) r.v[m] = xr.ld(tid4 + 4 * P * m);
    return r;
  };
  struct Seg { float v[12]; };
  auto load_seg = [&](int qn, bool live = true) -> Seg {       // L[i] = grad_out[(2 qn - 1/2) hop + 128 i + tid], i = 0..11
    Seg r;
    const int t0 = 2 * qn * FBW_HOP - FBW_HOP / 2;
    const BufF32 gb = live ? g_buf : BufF32::make(grad_out, 0);
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const int t = t0 + P * i;
      r.v[i] = gb.ld(t >= 0 ? 4 * (t + tid) : BufF32::kOutOfRange);
    }
    return r;
  };
  auto pack_seg = [&](const Seg& s, f32x2 (&z)[S]) {
#pragma unroll
    for (int m = 0; m < S; ++m) z[m] = f32x2{s.v[m], s.v[m + 4]};
  };
  auto pack_blk = [&](const Four& cx, bool live, f32x2 (&z)[S]) {
    const bool u01 = x_is_u01 && live;                          // noise = rand * 2 - 1 (vocoder.py:603,854)
    const float ua = u01 ? 2.0f : 1.0f, ub = u01 ? -1.0f : 0.0f;
    float l0 = lam0;
    asm volatile("" : "+v"(l0));                                // not a loop invariant (k_fir_blk6)
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
  const int mb = PL::mirror_base(tid);
  auto mirrored = [&](const f32x2* X, int m) -> f32x2 { return PL::rd_parked(X + mb - 64 * m); };
  const bool own_mirror = tid < 2;
  auto park = [&](const f32x2 (&z)[S], f32x2* X) {
#pragma unroll
    for (int m = 0; m < S; ++m) X[kP0 + 64 * m] = z[m];
  };
  const float cs = 0.5f / (float)NF;                           // the 1/2 of the Hermitian combination N + conj M and the 1/N of the inverse
  const f32x2 kHa = {0.5f * cs, 0.5f * cs};
  const f32x2 kMi = {0.5f * cs, -0.5f * cs};

  typename PL::Tw tw;
  typename PL::Ix ix;
  // the warm-up pass q_first - 1 rebuilds the carried U of block 2 q_first - 1; its cotangent pair is transformed here
  Seg sgm = load_seg(q_first - 1);
  Four x0 = load_blk(2 * q_first - 2), x1 = load_blk(2 * q_first - 1);
  tw.init(tid);
  ix.init(tid);
  {
    f32x2 zc[S];
    pack_seg(sgm, zc);
    PL::template forward_s<false, true>(zc, tw, bX, bY, ix);
    park(zc, bC);
  }
  sgm = load_seg(q_first);
  __syncthreads();
  f32x2 Uc[S];
#pragma unroll
  for (int m = 0; m < S; ++m) Uc[m] = f32x2{0.f, 0.f};
  for (int q = q_first - 1; q < q_last; ++q) {
    const bool warm = q < q_first;                              // workgroup-uniform
    const bool next_pass = q + 1 < q_last, pass_after = q + 2 < q_last;
    const int b0 = 2 * q;
    f32x2 z0[S], z1[S];
    pack_blk(x0, b0 >= 0 && b0 < g.F, z0);
    x0 = load_blk(b0 + 2, next_pass);
    PL::template forward_s<true, true>(z0, tw, bX, bY, ix);
    pack_blk(x1, b0 + 1 >= 0 && b0 + 1 < g.F, z1);
    x1 = load_blk(b0 + 3, next_pass);
    PL::template forward_s<true, true>(z1, tw, bX, bY, ix);
    // one sweep over the bins: split the parked cotangent transform (own value o, mirror image tn: S_b0 = c FFT(seg_b0) =
    // (o - conj tn) c/2, S_b0+1 = -i (o + conj tn) c/2), U_b = conj(Z_b) S_b, and the two Hermitian combinations of
    // k_fir_blk_bwd: N stays in z0, M is parked for the mirrored read
    const bool last0 = b0 == g.F - 1, last1 = b0 + 1 == g.F - 1;      // the held last row (core.py:167) also takes its own block's second term
    f32x2 M0;
#pragma unroll
    for (int m = 0; m < S; m += 2) {
      const f32x2 o0 = PL::rd_parked(bC + kP0 + 64 * m), o1 = PL::rd_parked(bC + kP0 + 64 * (m + 1));
      f32x2 tn0 = mirrored(bC, m);
      const f32x2 tn1 = mirrored(bC, m + 1);
      if (m == 0) tn0 = own_mirror ? -o0 : tn0;
      const f32x2 p0 = fft::sub_conj(o0, tn0), p1 = fft::sub_conj(o1, tn1);
      const f32x2 d0 = fft::add_conj(o0, tn0), d1 = fft::add_conj(o1, tn1);
      const f32x2 sa0 = p0 * kHa, sa1 = p1 * kHa;
      const f32x2 sb0 = fft::swap_scale(d0, kMi), sb1 = fft::swap_scale(d1, kMi);
      const f32x2 l0 = fft::cmul_lo(z0[m], sa0), l1 = fft::cmul_lo(z1[m], sb0);
      const f32x2 l2 = fft::cmul_lo(z0[m + 1], sa1), l3 = fft::cmul_lo(z1[m + 1], sb1);
      const f32x2 u00 = fft::cmulc_hi(z0[m], sa0, l0), u10 = fft::cmulc_hi(z1[m], sb0, l1);
      const f32x2 u01 = fft::cmulc_hi(z0[m + 1], sa1, l2), u11 = fft::cmulc_hi(z1[m + 1], sb1, l3);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const f32x2 U0 = e ? u01 : u00, U1 = e ? u11 : u10;
        const f32x2 Bj = last0 ? Uc[m + e] + U0 : Uc[m + e];
        const f32x2 Bj1 = last1 ? U0 + U1 : U0;
        const f32x2 Pp = Bj + U1, Qq = Bj - U1;
        const f32x2 T1 = U0 - Bj1, T2 = U0 + Bj1;
        z0[m + e] = fft::sub_mi(T1, Pp);                        // N = (U_j - B_j+1) + i (B_j + U_j+1)   (the 1/2 is in S)
        const f32x2 Mv = fft::sub_mi(T2, Qq);                   // M = (U_j + B_j+1) + i (B_j - U_j+1), parked
        if (m + e == 0) M0 = Mv;
        bX[kP0 + 64 * (m + e)] = Mv;
        Uc[m + e] = U1;                                         // carried: U of the pair's second block
      }
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < S; ++m) {
      f32x2 mm = mirrored(bX, m);
      if (m == 0) mm = own_mirror ? M0 : mm;                    // M is plain layout S: the own value as it is
      z0[m] = fft::add_conj(mm, z0[m]);                         // conj V = M[-k] + conj N[k]: inverse by the forward transform
    }
    // the tap rows' inverse, and one exchange behind it the transform of the next pair's cotangent (a full 1024-point input)
    f32x2 zc[S];
    pack_seg(sgm, zc);
    sgm = load_seg(q + 2, pass_after);
    PL::template transposed_then_forward_s<true, true>(z0, zc, tw, bY, bX, bC, ix);
    park(zc, bC);                                               // read by the next pass's sweep, behind its first stage's barriers
    // d_taps[j][i] = d h_j[i + SH]: transform index n = 128 m + tid < 512; row b0 = sigma Re, row b0 + 1 = -sigma Im
    {
      const bool own0 = !warm && b0 < g.F, own1 = !warm && b0 + 1 < g.F;
      const BufF32 r0 = BufF32::make(dtb + (long)(own0 ? b0 : 0) * g.N, own0 ? g.N : 0);
      const BufF32 r1 = BufF32::make(dtb + (long)(own1 ? b0 + 1 : 0) * g.N, own1 ? g.N : 0);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        r0.st(sg * z0[m].x, tap_off[m]);
        r1.st(-sg * z0[m].y, tap_off[m]);
      }
    }
  }
}

int launch_fir_blk_bwd(const float* x, int x_is_u01, const float* taps, const float* grad_out, float* d_x, float* d_taps,
                       int B, int F, int hop, int N, hipStream_t st, const FirBwdSecond* second) {
  if (hop != FBW_HOP || N < 2 || (N & 1) || N > 512 || (long)F * hop >= (1L << 28)) return -1;   // one utterance below 2^30 bytes (buffer descriptors)
  FirBwdGeom g;
  g.F = F; g.N = N; g.T = F * hop;
  g.pairs = (F + 1) / 2;
  const long slots = 4L * 256;                                  // 2 waves per SIMD, two waves per workgroup
  long per_utt = slots / (B > 0 ? B : 1);
  if (per_utt < 1) per_utt = 1;
  int run = (int)((g.pairs + per_utt - 1) / per_utt);
  if (run < 3) run = 3;
  if (const long v = knob(KNOB_BLK_RUN)) { if (v >= 1) run = (int)v; }
  if (run > g.pairs) run = g.pairs;
  g.run = run;
  g.runs_per_utt = (g.pairs + run - 1) / run;
  if (second && (d_x || knob(KNOB_BWD_WPS) == 2)) return -1;    // a second job rides in the three-wave tap-gradient kernel only
  if (!d_x && knob(KNOB_BWD_WPS) != 2) {                         // the tap gradient alone: three waves per SIMD, six workgroups per CU
    const long slots6 = 6L * 256;
    long pu = slots6 / ((second ? 2L : 1L) * (B > 0 ? B : 1));    // two jobs share the one round: runs twice as long, half the warm-ups
    if (pu < 1) pu = 1;
    int run6 = (int)((g.pairs + pu - 1) / pu);
    if (run6 < 3) run6 = 3;
    if (const long v = knob(KNOB_BLK_RUN)) { if (v >= 1) run6 = (int)v; }
    if (run6 > g.pairs) run6 = g.pairs;
    g.run = run6;
    g.runs_per_utt = (g.pairs + run6 - 1) / run6;
    const long wgs6 = (long)B * g.runs_per_utt;
    if (wgs6 > 0x7fffffffL) return -1;
    FirBwdJobs jobs;
    jobs.j[0] = FirBwdJob{x, x_is_u01, grad_out, d_taps};
    jobs.j[1] = second ? FirBwdJob{second->x, second->x_is_u01, second->grad_out, second->d_taps} : jobs.j[0];
    hipLaunchKernelGGL(k_fir_blk_bwd6<0>, dim3((unsigned)wgs6, second ? 2u : 1u), dim3(128), 0, st, jobs, g);
    return 0;
  }
  const long wgs = (long)B * g.runs_per_utt;
  if (wgs > 0x7fffffffL) return -1;
  if (d_x)
    hipLaunchKernelGGL(k_fir_blk_bwd<true>, dim3((unsigned)wgs), dim3(128), 0, st, x, x_is_u01, taps, grad_out, d_x, d_taps, g);
  else
    hipLaunchKernelGGL(k_fir_blk_bwd<false>, dim3((unsigned)wgs), dim3(128), 0, st, x, x_is_u01, taps, grad_out, d_x, d_taps, g);
  return 0;
}

}  // namespace ddsp
