// HOT-2c: the time-varying FIR ("frequency_filter" after tap synthesis).
// Reference: ddsp/core.py:120-182 (fft_convolve) -- pad, 50%-overlap frames, periodic Bartlett
// window, zero-padded FFT product with per-frame taps, overlap-add, crop.  Because the block
// convolution is linear (zero padded) and the two Bartlett halves sum to one, that whole chain
// equals (checked in tests against the oracle's block-FFT form)
//
//     y[t] = sum_j sum_m taps_j[m] * (x * tri_j)[t + N/2 - m],   tri_j[s] = max(0, 1 - |s - j*hop|/hop)
//
// i.e. every frame j contributes an ordinary convolution of its taps with the input weighted by
// a triangle centred on the frame start.  For a block of 256 consecutive outputs t = t0 + i + 16*c
// (i, c in 0..15) this is a Toeplitz contraction
//
//     Y[i][c] = sum_u Hj[i][u] * Xj[u][c],   Hj[i][u] = taps_j[u + i - 15],   Xj[u][c] = (x*tri_j)[t0 + 16c + N/2 + 15 - u]
//
// with inner dimension N+15 >= 16: it runs on the f32 MFMA pipe (v_mfma_f32_16x16x4_f32, exact
// f32 fmaf chains), one 16x16 accumulator per wave, both operands read from LDS.
#include "ddsp_common.h"

namespace ddsp {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// simple per-sample form (any hop / N): y[t] = sum_m h_s[m] x[s], s = t + N/2 - m, taps linearly
// interpolated between frame floor(s/hop) and the next (last held), indexed by the input sample.
// Slow; kept as the shape-agnostic path and as an on-GPU cross-check of the MFMA kernel.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_fir_simple(const float* __restrict__ x, int x_is_u01,
                                                    const float* __restrict__ taps, const float* __restrict__ addend,
                                                    float* __restrict__ out, float* __restrict__ out_plain, int F,
                                                    int hop, int N, long T) {
  const long b = blockIdx.y;
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const float* xb = x + b * T;
  const float* tb = taps + b * (long)F * N;
  const int D = N / 2;
  const float inv_hop = 1.0f / (float)hop;
  float acc = 0.f;
  for (int m = 0; m < N; ++m) {
    long s = t + D - m;
    if (s < 0 || s >= T) continue;
    int k = (int)(s / hop);
    float lam = (float)(s - (long)k * hop) * inv_hop;
    int k1 = k + 1 < F ? k + 1 : F - 1;
    float h = fmaf(lam, tb[(long)k1 * N + m], (1.0f - lam) * tb[(long)k * N + m]);
    float xv = xb[s];
    if (x_is_u01) xv = fmaf(2.0f, xv, -1.0f);
    acc = fmaf(h, xv, acc);
  }
  if (out_plain) out_plain[b * T + t] = acc;
  out[b * T + t] = addend ? acc + addend[b * T + t] : acc;
}

// ------------------------------------------------------------------------------------------------
// MFMA form.  Workgroup = 4 waves = 1024 consecutive outputs of one utterance (256 per wave).
// LDS: the input window it touches, pre-weighted twice (XA = (1-lam)*x for the frame's own taps,
// XB = lam*x for the next frame's taps), and the zero-padded tap rows of every frame in reach.
// ------------------------------------------------------------------------------------------------
constexpr int FIR_TILE = 1024;

__device__ __forceinline__ int fir_pad(int q) { return q + 2 * (q >> 4); }   // 18-word stride per 16: conflict-free B reads

struct FirGeom {
  int F, hop, N, D;
  int KU;        // inner extent rounded up to the MFMA K step: roundup4(N + 15)
  int HLEN;      // padded tap row: 15 zeros | N taps | zeros, KU + 16 words
  int SLEN;      // staged input samples: FIR_TILE + N + 8
  int XOFF;      // words between the XA and XB arrays
  int NJ;        // tap rows held per workgroup
  long T;
};

__global__ void __launch_bounds__(256) k_fir_mfma(const float* __restrict__ x, int x_is_u01,
                                                  const float* __restrict__ taps, const float* __restrict__ addend,
                                                  float* __restrict__ out, float* __restrict__ out_plain, FirGeom g) {
  HIP_DYNAMIC_SHARED(float, lds)                    // XA | XB | taps[NJ][HLEN]
  float* XA = lds;
  float* XB = lds + g.XOFF;
  float* HS = lds + 2 * g.XOFF;

  const int tid = threadIdx.x;
  const int wave = tid >> 6, l = tid & 63;
  const long b = blockIdx.y;
  const long T0 = (long)blockIdx.x * FIR_TILE;
  const long S_lo = T0 - g.D - 4;
  const float* xb = x + b * g.T;
  const float inv_hop = 1.0f / (float)g.hop;

  // ---- stage the weighted input window -----------------------------------------------------------
  for (int q = tid; q < g.SLEN; q += 256) {
    const long s = S_lo + q;
    float v = 0.f, lam = 0.f;
    if (s >= 0 && s < g.T) {
      v = xb[s];
      if (x_is_u01) v = fmaf(2.0f, v, -1.0f);          // noise = rand*2-1 (vocoder.py:603,854)
      const long k = s / g.hop;
      lam = (float)(s - k * g.hop) * inv_hop;
    }
    const int p = fir_pad(q);
    XA[p] = (1.0f - lam) * v;
    XB[p] = lam * v;
  }
  // ---- stage the tap rows of frames j_lo .. j_lo+NJ-1 (row F duplicates row F-1, core.py:167) ---------
  const long s_first = S_lo < 0 ? 0 : S_lo;
  const int j_lo = (int)(s_first / g.hop);
  for (int e = tid; e < g.NJ * g.HLEN; e += 256) {
    const int jr = e / g.HLEN, idx = e - jr * g.HLEN;
    const int j = j_lo + jr;
    const int m = idx - 15;
    float v = 0.f;
    if (j <= g.F && m >= 0 && m < g.N) {
      const int row = j < g.F ? j : g.F - 1;
      v = taps[(b * g.F + row) * (long)g.N + m];
    }
    HS[e] = v;
  }
  __syncthreads();

  // ---- contraction ---------------------------------------------------------------------------------
  const int i = l & 15;                 // A: row (fine output offset);  B: column c (coarse offset, x16)
  const int kq = l >> 4;                // K sub-index 0..3
  const long t0w = T0 + 256 * wave;
  if (t0w >= g.T) return;               // wave-uniform; no barrier below
  // s(u0) for this lane's B element: t0w + 16*c + D + 15 - (u0 + kq)
  const long sb = t0w + 16 * i + g.D + 15 - kq;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int jr = 0; jr < g.NJ; ++jr) {
    const long j = j_lo + jr;
    if (j > g.F) break;
    const long sup_lo = (j - 1) * g.hop;            // frames j-1 and j carry a non-zero triangle of frame j
    const long sup_hi = (j + 1) * g.hop;            // exclusive
    const long mid = j * g.hop;
    // K steps for which at least one lane of the wave reads inside the support
    long ulo = t0w + g.D + 13 - sup_hi;             // u0 must exceed  t0w + D + 12 - sup_hi
    long uhi = t0w + g.D + 255 - sup_lo + 1;        // u0 at most      t0w + D + 255 - sup_lo
    if (ulo < 0) ulo = 0;
    ulo &= ~3L;
    if (uhi > g.KU) uhi = g.KU;
    const float* hrow = HS + jr * g.HLEN + kq + i;
    for (long u0 = ulo; u0 < uhi; u0 += 4) {
      const float av = hrow[u0];
      const long s = sb - u0;
      const int p = fir_pad((int)(s - S_lo));
      float bv = (s < mid) ? XB[p] : XA[p];
      if (s < sup_lo || s >= sup_hi) bv = 0.f;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
    }
  }

  // ---- epilogue: D layout col = l&15 (c), row = 4*(l>>4) + reg (i)  ->  4 consecutive samples per lane ----
  const long t = t0w + 16 * (l & 15) + 4 * (l >> 4);
  const long base = b * g.T + t;
  if (t + 3 < g.T && (base & 3) == 0) {
    float4 r = make_float4(acc[0], acc[1], acc[2], acc[3]);
    if (out_plain) *reinterpret_cast<float4*>(out_plain + base) = r;
    if (addend) {
      const float4 a = *reinterpret_cast<const float4*>(addend + base);
      r.x += a.x; r.y += a.y; r.z += a.z; r.w += a.w;
    }
    *reinterpret_cast<float4*>(out + base) = r;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (t + e < g.T) {
        float r = acc[e];
        if (out_plain) out_plain[base + e] = r;
        out[base + e] = addend ? r + addend[base + e] : r;
      }
    }
  }
}

// ---- launchers -----------------------------------------------------------------------------------
static FirGeom fir_geom(int F, int hop, int N) {
  FirGeom g;
  g.F = F; g.hop = hop; g.N = N; g.D = N / 2;
  g.T = (long)F * hop;
  g.KU = (N + 15 + 3) & ~3;
  g.HLEN = g.KU + 16;
  g.SLEN = FIR_TILE + N + 8;
  int padded = g.SLEN + 2 * ((g.SLEN >> 4) + 1);
  g.XOFF = (padded + 31) & ~31;
  g.NJ = (g.SLEN + hop - 1) / hop + 2;
  return g;
}

size_t fir_mfma_lds_bytes(int F, int hop, int N) {
  FirGeom g = fir_geom(F, hop, N);
  return ((size_t)2 * g.XOFF + (size_t)g.NJ * g.HLEN) * sizeof(float);
}

// impl: 0 = auto, 1 = simple, 2 = mfma.  Returns the implementation used, or <0 on error.
int launch_fir(const float* x, int x_is_u01, const float* taps, const float* addend, float* out, float* out_plain,
               int B, int F, int hop, int N, int impl, hipStream_t st) {
  const long T = (long)F * hop;
  if (B == 0 || T == 0) return 0;
  size_t lds = fir_mfma_lds_bytes(F, hop, N);
  const bool mfma_ok = lds <= 64 * 1024 && (N % 2 == 0) && B <= 65535;
  if (impl == 2 && !mfma_ok) return -1;
  if (impl == 0) impl = mfma_ok ? 2 : 1;
  if (impl == 2) {
    FirGeom g = fir_geom(F, hop, N);
    dim3 grid((unsigned)((T + FIR_TILE - 1) / FIR_TILE), (unsigned)B), block(256);
    hipLaunchKernelGGL(k_fir_mfma, grid, block, lds, st, x, x_is_u01, taps, addend, out, out_plain, g);
  } else {
    if (B > 65535) return -1;
    dim3 grid((unsigned)((T + 255) / 256), (unsigned)B), block(256);
    hipLaunchKernelGGL(k_fir_simple, grid, block, 0, st, x, x_is_u01, taps, addend, out, out_plain, F, hop, N, T);
  }
  return impl;
}

}  // namespace ddsp
