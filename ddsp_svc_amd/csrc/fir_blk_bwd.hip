// Adjoints of the time-varying FIR (what autograd returns for ddsp/core.py:120-182 fft_convolve): gradients w.r.t.
// the tap rows and, optionally, the input signal, in the hop-block form of fir_blk.hip.
//
// Forward, per hop block b (1024-point circular, alias-free):  r_b = x1_b (*) h'_b + x2_b (*) h'_min(b+1,F-1),
//   x1_b = x_b (1 - lambda), x2_b = x_b lambda, h'_j = tap row j circularly shifted by 512 - N/2, and r_b[n] is added
//   to the output at time (b-1) hop + u, u = n for n >= 512 - N/2 and n + 1024 for the wrapped head.  With
//   seg_b[n] = grad_out[(b-1) hop + u(n)] (zero outside the signal and outside the support):
//     d h'_b           += x1_b (star) seg_b      <->  conj(X1_b) S_b        (circular cross-correlation)
//     d h'_min(b+1,..) += x2_b (star) seg_b      <->  conj(X2_b) S_b
//     d x_b[s] = (1 - lambda_s) (h'_b (star) seg_b)[s] + lambda_s (h'_b+1 (star) seg_b)[s]   <->  conj(H) S_b
// Per pair of blocks: one transform for the two cotangent segments, one per block for (x1, x2), one inverse for the
// two finished tap-gradient rows (4 transforms per 1024 samples); with the input gradient also one for the tap
// spectra and one inverse per block (7).  A tap row collects from two consecutive blocks, so the second block's
// x2-term is carried in registers to the next pair; a run starts one pair early to rebuild that carry.  No
// overlap-add: every tap row and every input block is written by exactly one workgroup.
#include "fft_r.h"
#include "kernels.h"
#include <stdlib.h>

namespace ddsp {

using fft::cmul;

constexpr int FBW_HOP = 512;

struct FirBwdGeom {
  int F, N, T;
  int pairs;              // block pairs per utterance: ceil(F / 2)
  int run, runs_per_utt;
};

// conj(a) * b
__device__ __forceinline__ f32x2 cmulc(f32x2 a, f32x2 b) {
  return f32x2{fmaf(a.x, b.x, a.y * b.y), fmaf(a.x, b.y, -(a.y * b.x))};
}
__device__ __forceinline__ f32x2 mul_i(f32x2 a) { return f32x2{-a.y, a.x}; }
__device__ __forceinline__ f32x2 mul_mi(f32x2 a) { return f32x2{a.y, -a.x}; }

template <bool WITH_DX>
__global__ void __launch_bounds__(128, 2) k_fir_blk_bwd(const float* __restrict__ x, int x_is_u01,
                                                       const float* __restrict__ taps,
                                                       const float* __restrict__ grad_out, float* __restrict__ d_x,
                                                       float* __restrict__ d_taps, FirBwdGeom g) {
  using PL = fft::Plan<2>;
  constexpr int NF = PL::N, P = PL::P, S = 8;
  __shared__ __attribute__((aligned(16))) f32x2 ex[2][NF];
  const int tid = threadIdx.x;
  const int b = blockIdx.x / g.runs_per_utt;
  const int run_no = blockIdx.x - b * g.runs_per_utt;
  const int q_first = run_no * g.run;
  int q_last = q_first + g.run;
  if (q_last > g.pairs) q_last = g.pairs;
  const int SH = FBW_HOP - (g.N >> 1);
  const float* xb = x + (long)b * g.T;
  const float* tb = taps + (long)b * g.F * g.N;
  const float* gb = grad_out + (long)b * g.T;
  float* dtb = d_taps + (long)b * g.F * g.N;
  const float inv_hop = 1.0f / (float)FBW_HOP;
  const float cs = 0.25f / (float)NF;                          // the 1/2 of both splits and the 1/N of the inverse

  typename PL::Tw tw;
  tw.init(tid);
  // Transforms as in fir_blk.hip: forward into the scrambled bin layout S of fft_r.h (all spectra of this kernel live
  // in S; the products are pointwise), the transposed factorisation back to time order, two LDS exchanges each.
  // ex[cur] is the buffer no wave reads any more.
  int cur = 0;
  const int kS0 = PL::s_index(tid, 0);                          // slot m holds bin kS0 + 64 m
  const f32x2* mir = ex[0];                                     // where the last transform parked its bins by index
  auto transform = [&](f32x2 (&z)[S]) {
    f32x2* X = ex[cur];
    PL::forward_s(z, tw, X, ex[cur ^ 1], tid);
#pragma unroll
    for (int m = 0; m < S; ++m) X[kS0 + 64 * m] = z[m];
    __syncthreads();
    mir = X;
    cur ^= 1;
  };
  // p = Z[k] + conj Z[-k], d = Z[k] - conj Z[-k] for the thread's 8 bins
  auto split = [&](const f32x2 (&z)[S], f32x2 (&p)[S], f32x2 (&d)[S]) {
#pragma unroll
    for (int m = 0; m < S; ++m) {
      const int k = kS0 + 64 * m;
      const f32x2 zneg = mir[(NF - k) & (NF - 1)];
      p[m] = fft::add_conj(z[m], zneg);
      d[m] = fft::sub_conj(z[m], zneg);
    }
  };
  // inverse of V = Va + i Vb (both Hermitian) by conj / transform / conj: re -> ra, im -> rb, in time order
  auto inverse_pair = [&](const f32x2 (&Va)[S], const f32x2 (&Vb)[S], float (&ra)[S], float (&rb)[S]) {
    f32x2 v[S];
#pragma unroll
    for (int m = 0; m < S; ++m) v[m] = fft::conj_minus_i_conj(Va[m], Vb[m]);
    PL::transposed(v, tw, ex[cur], ex[cur ^ 1], tid);           // leaves ex[cur] free again
#pragma unroll
    for (int m = 0; m < S; ++m) { ra[m] = v[m].x; rb[m] = -v[m].y; }
  };
  // (x1, x2) of block bi packed in one transform -> p = 2 X1, d = 2i X2
  auto block_spectra = [&](int bi, f32x2 (&p)[S], f32x2 (&d)[S]) {
    f32x2 z[S];
    const float* src = xb + (long)(bi < g.F ? bi : g.F - 1) * FBW_HOP + tid;
    float raw[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) raw[m] = src[P * m];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      float xv = raw[m];
      if (x_is_u01) xv = fmaf(2.0f, xv, -1.0f);
      if (bi >= g.F) xv = 0.f;
      const float lam = (float)(P * m + tid) * inv_hop;
      z[m] = f32x2{(1.0f - lam) * xv, lam * xv};
    }
#pragma unroll
    for (int m = 4; m < S; ++m) z[m] = f32x2{0.f, 0.f};
    transform(z);
    split(z, p, d);
  };
  // cotangent segments of blocks b0, b0 + 1 packed in one transform.  The forward result of block bb occupies the
  // unwrapped indices u in [SH, SH + hop + N - 2] of its circular buffer (time (bb-1) hop + u); transform index n
  // holds u = n for n >= SH and u = n + 1024 for the wrapped head.  ps = cs (Z + conj Z-), ms = cs (Z - conj Z-).
  auto cotangent = [&](int b0, f32x2 (&ps)[S], f32x2 (&ms)[S]) {
    f32x2 z[S];
    float r0[S], r1[S];
    const int u_max = SH + FBW_HOP + g.N - 2;
#pragma unroll
    for (int m = 0; m < S; ++m) {
      const int n = P * m + tid;
      const int u = n >= SH ? n : n + NF;
      const int t0 = (b0 - 1) * FBW_HOP + u, t1 = t0 + FBW_HOP;
      r0[m] = gb[t0 < 0 ? 0 : (t0 >= g.T ? g.T - 1 : t0)];
      r1[m] = gb[t1 < 0 ? 0 : (t1 >= g.T ? g.T - 1 : t1)];
    }
#pragma unroll
    for (int m = 0; m < S; ++m) {
      const int n = P * m + tid;
      const int u = n >= SH ? n : n + NF;
      const int t0 = (b0 - 1) * FBW_HOP + u, t1 = t0 + FBW_HOP;
      const bool in = u <= u_max;
      z[m] = f32x2{(in && t0 >= 0 && t0 < g.T) ? r0[m] : 0.f, (in && t1 >= 0 && t1 < g.T) ? r1[m] : 0.f};
    }
    transform(z);
    split(z, ps, ms);
#pragma unroll
    for (int m = 0; m < S; ++m) { ps[m] = ps[m] * cs; ms[m] = ms[m] * cs; }
  };
  // tap rows j, j + 1 (shifted) packed -> h0 = 2 H_j, h1 = 2 H_j+1
  auto tap_spectra = [&](int j, f32x2 (&h0)[S], f32x2 (&h1)[S]) {
    f32x2 z[S];
    float ra[6], rb[6];
    const float* tr0 = tb + (long)(j < g.F ? j : g.F - 1) * g.N;
    const float* tr1 = tb + (long)(j + 1 < g.F ? j + 1 : g.F - 1) * g.N;
#pragma unroll
    for (int m = 2; m < S; ++m) {
      int i = P * m + tid - SH;
      i = i < 0 ? 0 : (i >= g.N ? g.N - 1 : i);
      ra[m - 2] = tr0[i];
      rb[m - 2] = tr1[i];
    }
    z[0] = z[1] = f32x2{0.f, 0.f};
#pragma unroll
    for (int m = 2; m < S; ++m) {
      const int i = P * m + tid - SH;
      const bool ok = i >= 0 && i < g.N;
      z[m] = f32x2{ok ? ra[m - 2] : 0.f, ok ? rb[m - 2] : 0.f};
    }
    transform(z);
    f32x2 d[S];
    split(z, h0, d);
#pragma unroll
    for (int m = 0; m < S; ++m) h1[m] = mul_mi(d[m]);          // d / i
  };

  f32x2 carry[S];
#pragma unroll
  for (int m = 0; m < S; ++m) carry[m] = f32x2{0.f, 0.f};
  f32x2 Hc[S];                                                 // 2 H_b0 (input gradient only)
  const int q0 = q_first > 0 ? q_first - 1 : 0;
  if (WITH_DX) {
    f32x2 hdrop[S];
    tap_spectra(2 * q_first, Hc, hdrop);
  }
  for (int q = q0; q < q_last; ++q) {
    const int b0 = 2 * q;
    const bool own = q >= q_first;
    f32x2 ps[S], ms[S];
    cotangent(b0, ps, ms);                                     // cs * 2 S_b0,  cs * 2i S_b0+1
    f32x2 p0[S], d0[S], p1[S], d1[S];
    block_spectra(b0, p0, d0);
    block_spectra(b0 + 1, p1, d1);
    // tap-gradient spectra (header): conj(X1) S and conj(X2) S of both blocks
    f32x2 DH0[S], DH1[S];
#pragma unroll
    for (int m = 0; m < S; ++m) {
      const f32x2 a0 = cmulc(p0[m], ps[m]);                    // conj(X1_b0) S_b0 / NF
      const f32x2 c0 = mul_i(cmulc(d0[m], ps[m]));             // conj(X2_b0) S_b0 / NF
      const f32x2 a1 = mul_mi(cmulc(p1[m], ms[m]));            // conj(X1_b1) S_b1 / NF
      const f32x2 c1 = cmulc(d1[m], ms[m]);                    // conj(X2_b1) S_b1 / NF
      DH0[m] = carry[m] + a0;
      DH1[m] = c0 + a1;
      if (b0 == g.F - 1) DH0[m] = DH0[m] + c0;                 // the last row also takes its own block's x2 term
      if (b0 + 1 == g.F - 1) DH1[m] = DH1[m] + c1;
      carry[m] = c1;
    }
    if (own) {
      float r0[S], r1[S];
      inverse_pair(DH0, DH1, r0, r1);
      // d_taps[j][i] = d h'_j[i + SH]
#pragma unroll
      for (int m = 2; m < S; ++m) {
        const int i = P * m + tid - SH;
        if (i >= 0 && i < g.N) {
          dtb[(long)b0 * g.N + i] = r0[m];
          if (b0 + 1 < g.F) dtb[(long)(b0 + 1) * g.N + i] = r1[m];
        }
      }
      if (WITH_DX) {
        f32x2 Ha[S], Hb[S];
        tap_spectra(b0 + 1, Ha, Hb);                           // 2 H_b0+1, 2 H_b0+2 (rows clamp to F-1)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          f32x2 V1[S], V2[S];
#pragma unroll
          for (int m = 0; m < S; ++m) {
            if (h == 0) {
              V1[m] = cmulc(Hc[m], ps[m]);                     // conj(H_b0) S_b0 / NF
              V2[m] = cmulc(Ha[m], ps[m]);
            } else {
              V1[m] = mul_mi(cmulc(Ha[m], ms[m]));             // conj(H_b1) S_b1 / NF
              V2[m] = mul_mi(cmulc(Hb[m], ms[m]));
            }
          }
          float dx1[S], dx2[S];
          inverse_pair(V1, V2, dx1, dx2);
          if (b0 + h < g.F) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
              const float lam = (float)(P * m + tid) * inv_hop;
              d_x[(long)b * g.T + (long)(b0 + h) * FBW_HOP + P * m + tid] = fmaf(1.0f - lam, dx1[m], lam * dx2[m]);
            }
          }
        }
#pragma unroll
        for (int m = 0; m < S; ++m) Hc[m] = Hb[m];
      }
    }
  }
}

int launch_fir_blk_bwd(const float* x, int x_is_u01, const float* taps, const float* grad_out, float* d_x, float* d_taps,
                       int B, int F, int hop, int N, hipStream_t st) {
  if (hop != FBW_HOP || N < 2 || (N & 1) || N > 512 || (long)F * hop >= (1L << 30)) return -1;
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
  const long wgs = (long)B * g.runs_per_utt;
  if (wgs > 0x7fffffffL) return -1;
  if (d_x)
    hipLaunchKernelGGL(k_fir_blk_bwd<true>, dim3((unsigned)wgs), dim3(128), 0, st, x, x_is_u01, taps, grad_out, d_x, d_taps, g);
  else
    hipLaunchKernelGGL(k_fir_blk_bwd<false>, dim3((unsigned)wgs), dim3(128), 0, st, x, x_is_u01, taps, grad_out, d_x, d_taps, g);
  return 0;
}

}  // namespace ddsp
