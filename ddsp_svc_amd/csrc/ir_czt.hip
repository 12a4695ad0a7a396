// Tap synthesis (core.py:254-270: irfft of the one-sided response, roll, window) for bin counts other than 256 as an inverse
// chirp-z transform -- the general-size companion of the prime-factor kernel (ir_pfa.hip, N = 510 = 2 * 3 * 5 * 17), on the
// machinery of the loss kernels (loss_czt.hip): NT = 2 (n - 1) taps are an inverse DFT of ANY even size NT,
//     y[j] = sum_k G'[k] exp(+2 pi i j k / NT) = conj( c[j] sum_k (conj(G'[k]) c[k]) conj(c)[j - k] ),   c[j] = exp(-i pi j^2 / NT),
// one circular convolution of N = 512 / 1024 / 2048 / 4096 >= 2 NT - 1 points (two transforms of fft_r.h).  TWO rows ride in one
// complex transform: G' = H'_a + i H'_b with H' the Hermitian extension of a row's response (irfft drops Im(DC) and
// Im(Nyquist), core.py:259), so the taps of row a are the real part and those of row b the imaginary part.  The chirp and
// the filter's spectrum are made once per workgroup (float64 sine / cosine of integer-reduced phases; one transform), the
// window is applied where the taps are stored.  Replaces the dense float32 MFMA contraction k_ir_gemm (ir.hip), which ran
// at 0.17 - 0.47 of the matrix pipe's roof on a path that is no contraction.
#include "ddsp_common.h"
#include "fft_r.h"
#include "kernels.h"
#include "occupancy.h"

namespace ddsp {
using fft::cconj;
using fft::cmul;

enum { TC_MODE_ROLL = 0, TC_MODE_HANN = 1, TC_MODE_DYNAMIC = 2 };

__device__ __forceinline__ float tc_cos_turns(float a) {          // cos(a), a in radians: k_ir_gemm's reduction (ir.hip, cos_turns)
  const float inv_hi = 0.15915494f, inv_lo = 6.4206383e-9f;
  const float nn = rintf(a * inv_hi);
  float r = fmaf(a, inv_hi, -nn);
  r = fmaf(a, inv_lo, r);
  return __builtin_amdgcn_cosf(r);
}

template <int R>
__global__ void __launch_bounds__(64 * R, 2) k_taps_czt(const float* __restrict__ a_re, long ld_re,
                                                        const float* __restrict__ a_im, long ld_im, int act, float scale,
                                                        const float* __restrict__ hann, int mode,
                                                        const float* __restrict__ half_width, float hw_sr, long rows, int n,
                                                        float* __restrict__ taps, long pairs_per_wg) {
  using PL = fft::Plan<R>;
  constexpr int N = PL::N, P = PL::P;
  __shared__ __attribute__((aligned(16))) f32x2 ex[2][N];
  const int tid = threadIdx.x;
  const int NT = 2 * (n - 1), half = NT / 2;
  const long pairs = (rows + 1) / 2;
  const long p_lo = (long)blockIdx.x * pairs_per_wg;
  const long p_hi = p_lo + pairs_per_wg < pairs ? p_lo + pairs_per_wg : pairs;
  if (p_lo >= p_hi) return;
  typename PL::Tw tw;
  tw.init(tid);
  // chirp of the thread's four indices j = P m + tid (NT <= N / 2: the upper slots hold padding), 0 behind NT
  f32x2 chr[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int j = P * m + tid;
    double s, c;
    sincospi((double)(((long)j * j) % (2 * NT)) / (double)NT, &s, &c);
    chr[m] = j < NT ? f32x2{(float)c, (float)-s} : f32x2{0.f, 0.f};
  }
  // spectrum of the chirp filter b[m] = conj(c[m]), |m| < NT, laid out circularly
  f32x2 bh[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const int q = P * m + tid;
    const int d = q < NT ? q : (q > N - NT ? N - q : -1);
    double s = 0.0, c = 0.0;
    if (d >= 0) sincospi((double)(((long)d * d) % (2 * NT)) / (double)NT, &s, &c);
    bh[m] = d >= 0 ? f32x2{(float)c, (float)s} : f32x2{0.f, 0.f};
  }
  PL::forward(bh, tw, ex[0], ex[1], tid);
  __syncthreads();
  const float fold = 1.0f / ((float)N * (float)NT);            // the inverse transform's 1 / N and irfft's 1 / NT
  const bool has_im = a_im != nullptr;

  // raw response of a pair of rows at the thread's four bins (mirrored above the Nyquist bin): fetched one pair ahead
  float re0[4], im0[4], re1[4], im1[4];
  auto fetch = [&](long pr) {
    const long r0 = 2 * pr, r1 = r0 + 1 < rows ? r0 + 1 : r0;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int k = P * m + tid;
      const int kk = k >= NT ? 0 : (k < n ? k : NT - k);
      re0[m] = a_re[r0 * ld_re + kk];
      re1[m] = a_re[r1 * ld_re + kk];
      im0[m] = has_im ? a_im[r0 * ld_im + kk] : 0.f;
      im1[m] = has_im ? a_im[r1 * ld_im + kk] : 0.f;
    }
  };
  fetch(p_lo);
  for (long pr = p_lo; pr < p_hi; ++pr) {
    const long r0 = 2 * pr;
    const bool two = r0 + 1 < rows;
    f32x2 v[8];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int k = P * m + tid;
      const int kk = k >= NT ? 0 : (k < n ? k : NT - k);
      const bool edge = kk == 0 || kk == n - 1;                   // Im(DC) and Im(Nyquist) are dropped
      const float sg = k < n ? 1.f : -1.f;                        // the mirrored half is the conjugate
      float x0 = re0[m], x1 = re1[m];
      if (act == 1) { x0 = expf(x0); x1 = expf(x1); }
      const f32x2 h0{x0 * scale, edge ? 0.f : sg * (im0[m] * scale)};
      const f32x2 h1 = two ? f32x2{x1 * scale, edge ? 0.f : sg * (im1[m] * scale)} : f32x2{0.f, 0.f};
      const f32x2 gc{h0.x - h1.y, h0.y + h1.x};                   // H'_a + i H'_b
      v[m] = cmul(cconj(gc), chr[m]);                              // chr = 0 behind NT
    }
    if (pr + 1 < p_hi) fetch(pr + 1);
    PL::template forward<true>(v, tw, ex[0], ex[1], tid);
#pragma unroll
    for (int m = 0; m < 8; ++m) v[m] = cconj(cmul(v[m], bh[m]));
    __syncthreads();
    PL::forward(v, tw, ex[0], ex[1], tid);
    // D[j] = conj(v) c[j] / N; the taps are Re D (row a) and -Im D (row b); roll by NT / 2 and window where they land
    float hw0 = 1.f, hw1 = 1.f;
    if (mode == TC_MODE_DYNAMIC) {
      hw0 = half_width[r0];
      hw1 = half_width[two ? r0 + 1 : r0];
      if (hw_sr > 0.f) {                                           // `half_width` holds f0: the width is formed here (vocoder.py:851)
        hw0 = (1.5f * hw_sr) / (hw0 + 1e-3f);
        hw1 = (1.5f * hw_sr) / (hw1 + 1e-3f);
      }
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int j = P * m + tid;
      if (j < NT) {
        const f32x2 d = cmul(cconj(v[m]), chr[m]) * fold;
        const int p = j + half < NT ? j + half : j + half - NT;    // ir[j] lands at tap (j + NT / 2) mod NT
        float w0 = 1.f, w1 = 1.f;
        if (mode == TC_MODE_HANN) {
          w0 = w1 = hann[p];
        } else if (mode == TC_MODE_DYNAMIC) {
          float u0 = (float)(p - half) / hw0, u1 = (float)(p - half) / hw1;      // core.py:244
          if (u0 > 1.0f) u0 = 0.0f;                                                  // core.py:245 -- only the upper side is clamped
          if (u1 > 1.0f) u1 = 0.0f;
          w0 = (1.0f + tc_cos_turns(kPiF * u0)) / 2.0f;                              // core.py:246
          w1 = (1.0f + tc_cos_turns(kPiF * u1)) / 2.0f;
        }
        taps[r0 * NT + p] = d.x * w0;
        if (two) taps[(r0 + 1) * NT + p] = -d.y * w1;
      }
    }
    __syncthreads();                                             // pass 4 of the last transform still reads ex[0]
  }
}

// Adjoint (what autograd returns for core.py:254-270 + the window helpers): the forward is window . roll . irfft, so the
// adjoint is a FORWARD real DFT of the windowed, un-rolled tap gradients (ir_pfa.hip, k_taps_pfa510_bwd),
//     D[k] = sum_m dz[m] exp(-2 pi i k m / NT),   dz[m] = w[j] d_taps[j],  j = (m + NT / 2) mod NT,
//     d re_k = (c_k / NT) Re D[k],   d im_k = (2 / NT) Im D[k]  (0 at DC and Nyquist),
// as a forward chirp-z transform of two rows at once, z = dz_a + i dz_b, separated through Z[k] and conj Z[NT - k].
template <int R>
__global__ void __launch_bounds__(64 * R, 2) k_taps_czt_bwd(const float* __restrict__ d_taps, const float* __restrict__ ctrl,
                                                            long ld_ctrl, int act, float scale, const float* __restrict__ hann,
                                                            int mode, const float* __restrict__ half_width, long rows, int n,
                                                            float* __restrict__ d_re, float* __restrict__ d_im,
                                                            long pairs_per_wg) {
  using PL = fft::Plan<R>;
  constexpr int N = PL::N, P = PL::P;
  __shared__ __attribute__((aligned(16))) f32x2 ex[2][N];
  const int tid = threadIdx.x;
  const int NT = 2 * (n - 1), half = NT / 2;
  const long pairs = (rows + 1) / 2;
  const long p_lo = (long)blockIdx.x * pairs_per_wg;
  const long p_hi = p_lo + pairs_per_wg < pairs ? p_lo + pairs_per_wg : pairs;
  if (p_lo >= p_hi) return;
  typename PL::Tw tw;
  tw.init(tid);
  f32x2 chr[4];
  int jtap[4];                                                   // the tap a slot's sample comes from
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int j = P * m + tid;
    double s, c;
    sincospi((double)(((long)j * j) % (2 * NT)) / (double)NT, &s, &c);
    chr[m] = j < NT ? f32x2{(float)c, (float)-s} : f32x2{0.f, 0.f};
    jtap[m] = j >= NT ? 0 : (j + half < NT ? j + half : j + half - NT);
  }
  f32x2 bh[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const int q = P * m + tid;
    const int d = q < NT ? q : (q > N - NT ? N - q : -1);
    double s = 0.0, c = 0.0;
    if (d >= 0) sincospi((double)(((long)d * d) % (2 * NT)) / (double)NT, &s, &c);
    bh[m] = d >= 0 ? f32x2{(float)c, (float)s} : f32x2{0.f, 0.f};
  }
  PL::forward(bh, tw, ex[0], ex[1], tid);
  __syncthreads();
  const float fold = 0.5f / (float)N;                            // the inverse transform's 1 / N and the 1 / 2 of the split
  const float inv_nt = 1.0f / (float)NT;
  float ga[4], gb[4];
  auto fetch = [&](long pr) {
    const long r0 = 2 * pr, r1 = r0 + 1 < rows ? r0 + 1 : r0;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      ga[m] = d_taps[r0 * NT + jtap[m]];
      gb[m] = d_taps[r1 * NT + jtap[m]];
    }
  };
  fetch(p_lo);
  for (long pr = p_lo; pr < p_hi; ++pr) {
    const long r0 = 2 * pr;
    const bool two = r0 + 1 < rows;
    float hw0 = 1.f, hw1 = 1.f;
    if (mode == TC_MODE_DYNAMIC) {
      hw0 = half_width[r0];
      hw1 = half_width[two ? r0 + 1 : r0];
    }
    f32x2 v[8];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int p = jtap[m];
      float w0 = 1.f, w1 = 1.f;
      if (mode == TC_MODE_HANN) {
        w0 = w1 = hann[p];
      } else if (mode == TC_MODE_DYNAMIC) {
        float u0 = (float)(p - half) / hw0, u1 = (float)(p - half) / hw1;
        if (u0 > 1.0f) u0 = 0.0f;
        if (u1 > 1.0f) u1 = 0.0f;
        w0 = (1.0f + tc_cos_turns(kPiF * u0)) / 2.0f;
        w1 = (1.0f + tc_cos_turns(kPiF * u1)) / 2.0f;
      }
      const f32x2 z{ga[m] * w0, two ? gb[m] * w1 : 0.f};
      v[m] = cmul(z, chr[m]);                                      // chr = 0 behind NT
    }
    if (pr + 1 < p_hi) fetch(pr + 1);
    PL::template forward<true>(v, tw, ex[0], ex[1], tid);
#pragma unroll
    for (int m = 0; m < 8; ++m) v[m] = cconj(cmul(v[m], bh[m]));
    __syncthreads();
    PL::forward(v, tw, ex[0], ex[1], tid);
    f32x2 z[3];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int k = P * m + tid;
      if (k < NT) {
        const f32x2 zz = cmul(cconj(v[m]), chr[m]) * fold;         // Z[k] / 2
        ex[1][k] = zz;
        if (m < 3) z[m] = zz;
      }
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 3; ++m) {                                  // NT <= N / 2: the bins 0 .. NT / 2 lie in the slots 0 .. 2
      const int k = P * m + tid;
      if (k < n) {
        const f32x2 zk = z[m], zm = ex[1][k == 0 ? 0 : NT - k];
        const f32x2 Da{zk.x + zm.x, zk.y - zm.y};                  // (Z[k] + conj Z[NT-k]) / 2
        const f32x2 Db{zk.y + zm.y, zm.x - zk.x};                  // (Z[k] - conj Z[NT-k]) / 2i
        const bool edge = k == 0 || k == n - 1;
        const float ce = edge ? inv_nt : 2.0f * inv_nt, ci = edge ? 0.f : 2.0f * inv_nt;
        float g0 = ce * Da.x, g1 = ce * Db.x;
        if (act == 1) {
          g0 *= scale * expf(ctrl[r0 * ld_ctrl + k]);
          if (two) g1 *= scale * expf(ctrl[(r0 + 1) * ld_ctrl + k]);
        }
        d_re[r0 * n + k] = g0;
        if (d_im) d_im[r0 * n + k] = ci * Da.y;
        if (two) {
          d_re[(r0 + 1) * n + k] = g1;
          if (d_im) d_im[(r0 + 1) * n + k] = ci * Db.y;
        }
      }
    }
    // no barrier: the next pass writes ex[0] first (its readers are behind the barrier above) and ex[1] only after its own
  }
}

int taps_czt_plan(int n) {
  const int NT = 2 * (n - 1);
  if (n < 2) return 0;
  if (2 * NT - 1 <= 512) return 1;
  if (2 * NT - 1 <= 1024) return 2;
  if (2 * NT - 1 <= 2048) return 4;
  if (2 * NT - 1 <= 4096) return 8;
  return 0;
}

// 0 = taken, -1 = not its shape (use launch_ir_gemm)
int launch_taps_czt(const float* a_re, long ld_re, const float* a_im, long ld_im, int act, float scale, const float* hann,
                    int mode, const float* half_width, long rows, int n, float* taps, hipStream_t st, float hw_from_f0_sr) {
  const int R = taps_czt_plan(n);
  if (!R || rows < 0 || (mode == TC_MODE_HANN && !hann) || (mode == TC_MODE_DYNAMIC && !half_width)) return -1;
  if (act == 1 && a_im) return -1;          // exp of a complex response: only the dense form defines it (as launch_taps_pfa510 declines it)
  if (rows == 0) return 0;
  static ResidentCache cache[4];
  const int resident = R == 1 ? resident_workgroups(k_taps_czt<1>, 64, cache[3])
                     : R == 2 ? resident_workgroups(k_taps_czt<2>, 128, cache[0])
                              : (R == 4 ? resident_workgroups(k_taps_czt<4>, 256, cache[1]) : resident_workgroups(k_taps_czt<8>, 512, cache[2]));
  const long pairs = (rows + 1) / 2;
  long per = (pairs + resident - 1) / resident;                  // one round of what the chip holds (loss_czt.hip)
  if (per < 1) per = 1;
  const long wgs = (pairs + per - 1) / per;
  if (wgs > 0x7fffffffL) return -1;
  const dim3 grid((unsigned)wgs);
  if (R == 1)
    hipLaunchKernelGGL(k_taps_czt<1>, grid, dim3(64), 0, st, a_re, ld_re, a_im, ld_im, act, scale, hann, mode, half_width,
                       hw_from_f0_sr, rows, n, taps, per);
  else if (R == 2)
    hipLaunchKernelGGL(k_taps_czt<2>, grid, dim3(128), 0, st, a_re, ld_re, a_im, ld_im, act, scale, hann, mode, half_width,
                       hw_from_f0_sr, rows, n, taps, per);
  else if (R == 4)
    hipLaunchKernelGGL(k_taps_czt<4>, grid, dim3(256), 0, st, a_re, ld_re, a_im, ld_im, act, scale, hann, mode, half_width,
                       hw_from_f0_sr, rows, n, taps, per);
  else
    hipLaunchKernelGGL(k_taps_czt<8>, grid, dim3(512), 0, st, a_re, ld_re, a_im, ld_im, act, scale, hann, mode, half_width,
                       hw_from_f0_sr, rows, n, taps, per);
  return 0;
}

// the adjoint; 0 = taken, -1 = not its shape (use launch_ir_gemm_bwd)
int launch_taps_czt_bwd(const float* d_taps, const float* ctrl, long ld_ctrl, int act, float scale, const float* hann, int mode,
                        const float* half_width, long rows, int n, float* d_re, float* d_im, hipStream_t st) {
  const int R = taps_czt_plan(n);
  if (!R || rows < 0 || (mode == TC_MODE_HANN && !hann) || (mode == TC_MODE_DYNAMIC && !half_width)) return -1;
  if (act == 1 && (!ctrl || d_im)) return -1;
  if (rows == 0) return 0;
  static ResidentCache cache[4];
  const int resident = R == 1 ? resident_workgroups(k_taps_czt_bwd<1>, 64, cache[3])
                     : R == 2 ? resident_workgroups(k_taps_czt_bwd<2>, 128, cache[0])
                              : (R == 4 ? resident_workgroups(k_taps_czt_bwd<4>, 256, cache[1]) : resident_workgroups(k_taps_czt_bwd<8>, 512, cache[2]));
  const long pairs = (rows + 1) / 2;
  long per = (pairs + resident - 1) / resident;
  if (per < 1) per = 1;
  const long wgs = (pairs + per - 1) / per;
  if (wgs > 0x7fffffffL) return -1;
  const dim3 grid((unsigned)wgs);
  if (R == 1)
    hipLaunchKernelGGL(k_taps_czt_bwd<1>, grid, dim3(64), 0, st, d_taps, ctrl, ld_ctrl, act, scale, hann, mode, half_width, rows,
                       n, d_re, d_im, per);
  else if (R == 2)
    hipLaunchKernelGGL(k_taps_czt_bwd<2>, grid, dim3(128), 0, st, d_taps, ctrl, ld_ctrl, act, scale, hann, mode, half_width, rows,
                       n, d_re, d_im, per);
  else if (R == 4)
    hipLaunchKernelGGL(k_taps_czt_bwd<4>, grid, dim3(256), 0, st, d_taps, ctrl, ld_ctrl, act, scale, hann, mode, half_width, rows,
                       n, d_re, d_im, per);
  else
    hipLaunchKernelGGL(k_taps_czt_bwd<8>, grid, dim3(512), 0, st, d_taps, ctrl, ld_ctrl, act, scale, hann, mode, half_width, rows,
                       n, d_re, d_im, per);
  return 0;
}

}  // namespace ddsp
