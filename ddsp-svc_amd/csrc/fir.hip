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
// MFMA form.  Workgroup = WAVES waves = WAVES*256 consecutive outputs of one utterance.
//
// LDS holds, for every frame j in reach, the chunk  Wj[o] = (x * tri_j)[(j-1)*hop + o], o in [0, 2*hop)
// (first half = lam*x of frame j-1, second half = (1-lam)*x of frame j), chunks separated by FIR_G
// zeros, plus the zero-padded tap rows.  Lanes that fall outside a frame's support read guard
// zeros, so the inner loop has no selects or masks: per MFMA one ds_read for each operand.
// Chunk storage is skewed by 2 words per 16 (fir_pad) so the 16-sample-strided B reads of a wave
// hit 32 distinct banks; the K loop is phased so that 4 consecutive K steps stay inside one
// 16-word block and use immediate offsets.
// ------------------------------------------------------------------------------------------------
constexpr int FIR_G = 320;     // guard zeros between chunks: >= 16*15 + 3 + 2*15 (lane spread + loop phase slack) + 32 (prefetch overrun)
constexpr int FIR_TAIL = 64;   // slack words after the last tap row (prefetch overrun)
constexpr int FIR_FP = 16;     // zeros in front of a tap row: the phased K loop may start at u0 = -15

__device__ __forceinline__ int fir_pad(int q) { return q + 2 * (q >> 4); }

struct FirGeom {
  int F, hop, N, D;
  int KU;        // inner extent: u in [0, N+15)
  int HLEN;      // words per padded tap row
  int CH, CS;    // chunk length (2*hop) and chunk stride (CH + FIR_G), before the bank skew
  int NJ;        // chunks / tap rows a workgroup can need
  int WWORDS;    // words reserved for the chunk area (skewed)
  long T;
  long TPU, NTILES;   // tiles per utterance, tiles in the launch
};

template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64) k_fir_mfma(const float* __restrict__ x, int x_is_u01,
                                                         const float* __restrict__ taps,
                                                         const float* __restrict__ addend, float* __restrict__ out,
                                                         float* __restrict__ out_plain, FirGeom g) {
  constexpr int TILE = WAVES * 256;
  constexpr int NT = WAVES * 64;
  HIP_DYNAMIC_SHARED(float, lds)
  float* W = lds;
  float* HS = lds + g.WWORDS;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63;
  // XCD-aware tile order: workgroup w is dispatched to XCD w % 8; give every XCD a contiguous run of
  // tiles so neighbouring tiles (which share tap rows and the input halo) meet in the same L2
  const long per_xcd = gridDim.x >> 3;
  const long tile = (long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (tile >= g.NTILES) return;                      // grid is padded to a multiple of 8
  const long b = tile / g.TPU;
  const long T0 = (tile - b * g.TPU) * TILE;

  // input samples this tile can touch, and the frames (tap rows) they belong to
  long s_min = T0 - g.D + 1;
  if (s_min < 0) s_min = 0;
  long s_max = T0 + TILE - 1 + g.D;
  if (s_max > g.T - 1) s_max = g.T - 1;
  const int j_lo = (int)(s_min / g.hop);
  int j_hi = (int)(s_max / g.hop) + 1;               // row F duplicates row F-1 (core.py:167)
  if (j_hi > g.F) j_hi = g.F;
  int nj = j_hi - j_lo + 1;
  if (nj > g.NJ) nj = g.NJ;                          // cannot happen (host bound); keeps LDS accesses in range

  // ---- staging ------------------------------------------------------------------------------------------
  // A: every global load of the window is issued up front (one batch, independent)
  constexpr int XMAX = (TILE + 1022 + 32 + NT - 1) / NT;      // window words per thread for N <= 1022
  const float* xb = x + b * g.T;
  const long st_lo = T0 - g.D + 1 - 16;
  const int swin = TILE + 2 * g.D + 30;
  float xv[XMAX];
#pragma unroll
  for (int i = 0; i < XMAX; ++i) {
    const int q = tid + i * NT;
    const long s = st_lo + q;
    float v = 0.f;
    if (q < swin && s >= 0 && s < g.T) v = xb[s];
    xv[i] = v;
  }
  // B: zero the chunk area (guards + everything outside the window) and the tap-row pads; copy the taps
  {
    float4* W4 = reinterpret_cast<float4*>(W);
    const int n4 = (fir_pad(nj * g.CS + FIR_G) + 4) >> 2;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = tid; t < n4; t += NT) W4[t] = z;
  }
  for (int jr = 0; jr < nj; ++jr) {
    const int j = j_lo + jr;
    const int row = j < g.F ? j : g.F - 1;
    const float* trow = taps + (b * g.F + row) * (long)g.N;
    float* hrow = HS + jr * g.HLEN;
    for (int idx = tid; idx < FIR_FP + 15; idx += NT) hrow[idx] = 0.f;
    for (int idx = FIR_FP + 15 + g.N + tid; idx < g.HLEN; idx += NT) hrow[idx] = 0.f;
#pragma unroll 4
    for (int m = tid; m < g.N; m += NT) hrow[FIR_FP + 15 + m] = trow[m];
  }
  __syncthreads();
  // C: scatter each window sample into the two chunks that weight it
  {
    const float inv_hop = 1.0f / (float)g.hop;
#pragma unroll
    for (int i = 0; i < XMAX; ++i) {
      const int q = tid + i * NT;
      const long s = st_lo + q;
      if (q < swin && s >= 0 && s < g.T) {
        float v = xv[i];
        if (x_is_u01) v = fmaf(2.0f, v, -1.0f);        // noise = rand*2-1 (vocoder.py:603,854)
        const int k = (int)(s / g.hop);
        const int rr = (int)(s - (long)k * g.hop);
        const float lam = (float)rr * inv_hop;
        const int jr = k - j_lo;                        // chunk of frame k holds it as (1-lam)*x, second half
        if (jr >= 0 && jr < nj) W[fir_pad(FIR_G + jr * g.CS + g.hop + rr)] = (1.0f - lam) * v;
        if (jr + 1 >= 0 && jr + 1 < nj) W[fir_pad(FIR_G + (jr + 1) * g.CS + rr)] = lam * v;
      }
    }
  }
  __syncthreads();

  // ---- contraction ---------------------------------------------------------------------------------------
  const int c = l & 15;                 // A: row i (fine output offset);  B: column c (coarse offset, x16)
  const int kq = l >> 4;                // K sub-index 0..3
  const long t0w = T0 + 256 * wave;
  if (t0w >= g.T) return;               // wave-uniform; no barrier below
  f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
  for (int jr = 0; jr < nj; ++jr) {
    const long j = j_lo + jr;
    const long lo = (j - 1) * g.hop;                // support of tri_j: [lo, hi)
    const long hi = (j + 1) * g.hop;
    // K steps u0 for which at least one lane reads inside the support
    const long ulo_l = t0w + g.D + 13 - hi;
    const long uhi_l = t0w + g.D + 256 - lo;
    const int uhi = uhi_l > g.KU ? g.KU : (int)uhi_l;
    if (ulo_l >= uhi || uhi <= 0) continue;         // no lane of this wave reaches frame j
    const int ulo = ulo_l < 0 ? 0 : (int)ulo_l;
    // chunk-local index of lane (c=0,kq=0) at u0 = 0; phase the loop so that index % 16 == 15 at group start
    const int W0 = FIR_G + jr * g.CS + (int)(t0w - lo) + g.D + 15;
    const int r = (((W0 - 15) % 16) + 16) % 16;
    const int ug0 = ulo - ((((ulo - r) % 16) + 16) % 16);
    const int x0 = W0 - kq + 16 * c - ug0;
    const float* bp = W + fir_pad(x0) - 12;
    const float* ap = HS + jr * g.HLEN + FIR_FP + ug0 + kq + c;
    const int ng = (uhi - ug0 + 15) >> 4;
    // software pipeline, two register sets: while the 4 MFMAs of one group issue, the operands of the
    // group after next are already in flight.  Prefetches may run up to two groups past the end of the
    // pass (never used; FIR_G and the tail slack of the tap area keep them inside the LDS allocation).
    // Two accumulators break the 40-cycle dependent-accumulator latency of v_mfma_f32_16x16x4_f32.
    float a0 = ap[0], a1 = ap[4], a2 = ap[8], a3 = ap[12];
    float b0 = bp[12], b1 = bp[8], b2 = bp[4], b3 = bp[0];
    float c0 = ap[16], c1 = ap[20], c2 = ap[24], c3 = ap[28];
    float d0 = bp[-6], d1 = bp[-10], d2 = bp[-14], d3 = bp[-18];
    int gi = 0;
    for (; gi + 2 <= ng; gi += 2) {
      ap += 32;
      bp -= 36;
      // sched_barrier(0): keep hipcc from sinking the prefetches back next to their consumers
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc2, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b2, acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, b3, acc2, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      a0 = ap[0]; a1 = ap[4]; a2 = ap[8]; a3 = ap[12];
      b0 = bp[12]; b1 = bp[8]; b2 = bp[4]; b3 = bp[0];
      __builtin_amdgcn_sched_barrier(0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(c0, d0, acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(c1, d1, acc2, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(c2, d2, acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(c3, d3, acc2, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      c0 = ap[16]; c1 = ap[20]; c2 = ap[24]; c3 = ap[28];
      d0 = bp[-6]; d1 = bp[-10]; d2 = bp[-14]; d3 = bp[-18];
      __builtin_amdgcn_sched_barrier(0);
    }
    if (gi < ng) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc2, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b2, acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, b3, acc2, 0, 0, 0);
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) acc[e] += acc2[e];

  // ---- epilogue: D layout col = l&15 (c), row = 4*(l>>4) + reg (i)  ->  4 consecutive samples per lane ----
  const long t = t0w + 16 * (l & 15) + 4 * (l >> 4);
  const long base = b * g.T + t;
  if (t + 3 < g.T && (base & 3) == 0) {
    float4 r4 = make_float4(acc[0], acc[1], acc[2], acc[3]);
    if (out_plain) *reinterpret_cast<float4*>(out_plain + base) = r4;
    if (addend) {
      const float4 a = *reinterpret_cast<const float4*>(addend + base);
      r4.x += a.x; r4.y += a.y; r4.z += a.z; r4.w += a.w;
    }
    *reinterpret_cast<float4*>(out + base) = r4;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (t + e < g.T) {
        float rv = acc[e];
        if (out_plain) out_plain[base + e] = rv;
        out[base + e] = addend ? rv + addend[base + e] : rv;
      }
    }
  }
}

// ---- launchers -----------------------------------------------------------------------------------
static FirGeom fir_geom(int F, int hop, int N, int tile) {
  FirGeom g;
  g.F = F; g.hop = hop; g.N = N; g.D = N / 2;
  g.T = (long)F * hop;
  g.KU = N + 15;
  g.HLEN = (FIR_FP + g.KU + 32 + 3) & ~3;
  g.CH = 2 * hop;
  g.CS = g.CH + FIR_G;
  g.NJ = (tile + 2 * g.D - 2) / hop + 3;
  int last = g.NJ * g.CS + FIR_G;
  g.WWORDS = (last + 2 * (last >> 4) + 8 + 31) & ~31;
  return g;
}

size_t fir_mfma_lds_bytes(int F, int hop, int N, int waves) {
  FirGeom g = fir_geom(F, hop, N, waves * 256);
  return ((size_t)g.WWORDS + (size_t)g.NJ * g.HLEN + FIR_TAIL) * sizeof(float);
}

// impl: 0 = auto, 1 = simple, 2 = mfma with 4 waves (1024 outputs) per workgroup, 3 = mfma with 8 waves.
// Returns the implementation used, or <0 when the requested kernel cannot take the shape.
int launch_fir(const float* x, int x_is_u01, const float* taps, const float* addend, float* out, float* out_plain,
               int B, int F, int hop, int N, int impl, hipStream_t st) {
  const long T = (long)F * hop;
  if (B == 0 || T == 0) return 0;
  if (N & 1) return -1;
  auto fits = [&](int waves) { return N <= 1022 && fir_mfma_lds_bytes(F, hop, N, waves) <= 64 * 1024; };
  if (impl == 0) impl = fits(4) ? 2 : 1;
  if ((impl == 2 && !fits(4)) || (impl == 3 && !fits(8))) return -1;
  if (impl == 2) {
    FirGeom g = fir_geom(F, hop, N, 1024);
    g.TPU = (T + 1023) / 1024;
    g.NTILES = g.TPU * B;
    dim3 grid((unsigned)(((g.NTILES + 7) / 8) * 8));
    hipLaunchKernelGGL(k_fir_mfma<4>, grid, dim3(256), fir_mfma_lds_bytes(F, hop, N, 4), st, x, x_is_u01, taps, addend,
                       out, out_plain, g);
  } else if (impl == 3) {
    FirGeom g = fir_geom(F, hop, N, 2048);
    g.TPU = (T + 2047) / 2048;
    g.NTILES = g.TPU * B;
    dim3 grid((unsigned)(((g.NTILES + 7) / 8) * 8));
    hipLaunchKernelGGL(k_fir_mfma<8>, grid, dim3(512), fir_mfma_lds_bytes(F, hop, N, 8), st, x, x_is_u01, taps, addend,
                       out, out_plain, g);
  } else if (impl == 1) {
    if (B > 65535) return -1;
    dim3 grid((unsigned)((T + 255) / 256), (unsigned)B), block(256);
    hipLaunchKernelGGL(k_fir_simple, grid, block, 0, st, x, x_is_u01, taps, addend, out, out_plain, F, hop, N, T);
  } else {
    return -1;
  }
  return impl;
}

}  // namespace ddsp
