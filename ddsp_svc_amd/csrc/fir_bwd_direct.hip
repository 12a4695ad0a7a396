// Adjoints of the time-varying FIR (ddsp/core.py:120-182 under autograd) for EVERY hop and tap count: the shape-agnostic
// form behind k_fir_blk_bwd (hop 512, N <= 512), as k_fir_simple is behind the hop-block forward kernels.  Training the classic
// CombSub configuration (512 harmonic bins: N = 1022) or a model at another block size reaches these.
//
// Hop-block form of the operator (oracle/ddsp_oracle.py ltv_fir_backward states the same sums in float64): block b of the
// input, x_b[s] = x[b hop + s], meets tap row b with weight (1 - lambda_s) and row min(b + 1, F - 1) with lambda_s = s / hop
// (core.py:161-167), and lands at output offset b hop - N/2.  With seg_b[n] = grad_out[b hop - N/2 + n] (zero outside the signal):
//     d_taps[j][m]      = sum over the (block, weight) pairs that use row j of  sum_s x_b[s] w_s seg_b[s + m]
//     d_x[b hop + s]    = (1 - lambda_s) sum_m seg_b[s + m] h_b[m]  +  lambda_s sum_m seg_b[s + m] h_min(b+1,F-1)[m]
// Both are correlations against a window of the cotangent held in LDS; a thread owns four consecutive outputs and walks
// the other index four at a time: one broadcast read and one 16-byte read per 16 (32) multiply-adds.  Direct sums in
// float32 -- 2 hop N multiply-adds per frame and gradient, ~1 ms per launch at B = 32 x 10 s, N = 1022 -- no atomics:
// every tap row and every input block is written by exactly one workgroup.
#include "ddsp_common.h"
#include "kernels.h"

namespace ddsp {

namespace fbd {
constexpr int TPB = 256;                  // threads
constexpr int OUT_TILE = 4 * TPB;         // outputs per tile (tap positions m / input positions s)
constexpr int CH = 512;                   // walked index per staging
constexpr int SEG = OUT_TILE + CH + 8;    // cotangent window of one (tile, chunk)
}  // namespace fbd

// the cotangent window seg_b[n0 .. n0 + count) -> LDS (zeros outside the signal)
__device__ __forceinline__ void stage_seg(const float* __restrict__ gb, long T, long t0, int count, float* __restrict__ dst) {
  for (int i = threadIdx.x; i < count; i += fbd::TPB) {
    const long t = t0 + i;
    dst[i] = (t >= 0 && t < T) ? gb[t] : 0.f;
  }
}

__global__ void __launch_bounds__(fbd::TPB) k_fir_bwd_taps_direct(const float* __restrict__ x, int x_is_u01,
                                                                 const float* __restrict__ grad_out,
                                                                 float* __restrict__ d_taps, int F, int hop, int N) {
  using namespace fbd;
  __shared__ __attribute__((aligned(16))) float xw[CH];
  __shared__ __attribute__((aligned(16))) float seg[SEG];
  const int j = blockIdx.x;                                   // tap row
  const long b = blockIdx.y;
  const long T = (long)F * hop;
  const float* xb = x + b * T;
  const float* gb = grad_out + b * T;
  float* out = d_taps + (b * F + j) * (long)N;
  const int D = N / 2;
  const float inv_hop = 1.0f / (float)hop;
  const int tid = threadIdx.x;
  // (block, weight) pairs of row j: (j, 1 - lambda); (j - 1, lambda); and (F - 1, lambda) once more on the held last row
  int blk[3], kind[3], n_pairs = 0;
  blk[n_pairs] = j; kind[n_pairs++] = 0;
  if (j > 0) { blk[n_pairs] = j - 1; kind[n_pairs++] = 1; }
  if (j == F - 1) { blk[n_pairs] = j; kind[n_pairs++] = 1; }
  for (int m0 = 0; m0 < N; m0 += OUT_TILE) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int p = 0; p < n_pairs; ++p) {
      const long base = (long)blk[p] * hop;
      for (int s0 = 0; s0 < hop; s0 += CH) {
        const int cnt = hop - s0 < CH ? hop - s0 : CH;
        __syncthreads();                                      // the previous chunk has been read
        for (int i = tid; i < CH; i += TPB) {
          float v = 0.f;
          if (i < cnt) {
            v = xb[base + s0 + i];
            if (x_is_u01) v = fmaf(2.0f, v, -1.0f);
            const float lam = (float)(s0 + i) * inv_hop;
            v *= kind[p] ? lam : 1.0f - lam;
          }
          xw[i] = v;
        }
        stage_seg(gb, T, base - D + s0 + m0, SEG, seg);       // seg[i] = seg_b[s0 + m0 + i]
        __syncthreads();
        const float* sg = seg + 4 * tid;
        float4 ga = *reinterpret_cast<const float4*>(sg);
        const int cnt4 = (cnt + 3) & ~3;                      // xw is zero past cnt
        for (int so = 0; so < cnt4; so += 4) {
          const float4 xv = *reinterpret_cast<const float4*>(xw + so);
          const float4 gn = *reinterpret_cast<const float4*>(sg + so + 4);
          const float w[8] = {ga.x, ga.y, ga.z, ga.w, gn.x, gn.y, gn.z, gn.w};
          const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = fmaf(xs[i], w[i + e], acc[e]);
          ga = gn;
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int m = m0 + 4 * tid + e;
      if (m < N) out[m] = acc[e];
    }
  }
}

__global__ void __launch_bounds__(fbd::TPB) k_fir_bwd_x_direct(const float* __restrict__ taps, const float* __restrict__ grad_out,
                                                              float* __restrict__ d_x, int F, int hop, int N) {
  using namespace fbd;
  __shared__ __attribute__((aligned(16))) float h0[CH];
  __shared__ __attribute__((aligned(16))) float h1[CH];
  __shared__ __attribute__((aligned(16))) float seg[SEG];
  const int blkno = blockIdx.x;                               // input block
  const long b = blockIdx.y;
  const long T = (long)F * hop;
  const float* gb = grad_out + b * T;
  const float* r0 = taps + (b * F + blkno) * (long)N;
  const float* r1 = taps + (b * F + (blkno + 1 < F ? blkno + 1 : F - 1)) * (long)N;
  const int D = N / 2;
  const float inv_hop = 1.0f / (float)hop;
  const int tid = threadIdx.x;
  const long base = (long)blkno * hop;
  for (int s0 = 0; s0 < hop; s0 += OUT_TILE) {
    float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f};
    for (int m0 = 0; m0 < N; m0 += CH) {
      const int cnt = N - m0 < CH ? N - m0 : CH;
      __syncthreads();
      for (int i = tid; i < CH; i += TPB) {
        h0[i] = i < cnt ? r0[m0 + i] : 0.f;
        h1[i] = i < cnt ? r1[m0 + i] : 0.f;
      }
      stage_seg(gb, T, base - D + s0 + m0, SEG, seg);         // seg[i] = seg_b[s0 + m0 + i]
      __syncthreads();
      const float* sg = seg + 4 * tid;
      float4 ga = *reinterpret_cast<const float4*>(sg);
      const int cnt4 = (cnt + 3) & ~3;
      for (int mo = 0; mo < cnt4; mo += 4) {
        const float4 u = *reinterpret_cast<const float4*>(h0 + mo);
        const float4 v = *reinterpret_cast<const float4*>(h1 + mo);
        const float4 gn = *reinterpret_cast<const float4*>(sg + mo + 4);
        const float w[8] = {ga.x, ga.y, ga.z, ga.w, gn.x, gn.y, gn.z, gn.w};
        const float us[4] = {u.x, u.y, u.z, u.w}, vs[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            a0[e] = fmaf(us[i], w[i + e], a0[e]);
            a1[e] = fmaf(vs[i], w[i + e], a1[e]);
          }
        ga = gn;
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int s = s0 + 4 * tid + e;
      if (s < hop) {
        const float lam = (float)s * inv_hop;
        d_x[b * T + base + s] = fmaf(lam, a1[e], (1.0f - lam) * a0[e]);
      }
    }
  }
}

int launch_fir_bwd_direct(const float* x, int x_is_u01, const float* taps, const float* grad_out, float* d_x, float* d_taps,
                          int B, int F, int hop, int N, hipStream_t st) {
  if (hop < 1 || N < 2 || (N & 1) || B > 65535 || F < 1) return -1;
  const dim3 grid((unsigned)F, (unsigned)B), block(fbd::TPB);
  hipLaunchKernelGGL(k_fir_bwd_taps_direct, grid, block, 0, st, x, x_is_u01, grad_out, d_taps, F, hop, N);
  if (d_x) hipLaunchKernelGGL(k_fir_bwd_x_direct, grid, block, 0, st, taps, grad_out, d_x, F, hop, N);
  return 0;
}

}  // namespace ddsp
