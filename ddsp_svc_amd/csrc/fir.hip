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
#include "kernels.h"
#include "philox.h"
#include <stdlib.h>

namespace ddsp {

#ifdef DDSP_HIP_TIMELINE
// diagnostics build only (tools/fir_timeline.py): per-workgroup phase timestamps (s_memtime)
__device__ long long* g_fir_timeline = nullptr;
__global__ void k_set_timeline(long long* p) { g_fir_timeline = p; }
#define FIR_STAMP(slot) do { if (g_fir_timeline && threadIdx.x == 0 && tile_no < 32) g_fir_timeline[((long)blockIdx.x * 32 + tile_no) * 8 + (slot)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define FIR_STAMP(slot) do { } while (0)
#endif

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
// MFMA form.  One tile = WAVES*256 consecutive outputs of one utterance, one wave per 256 outputs.
//
// LDS holds, for every frame j in reach, the chunk  Wj[o] = (x * tri_j)[(j-1)*hop + o], o in [0, 2*hop)
// (first half = lam*x of frame j-1, second half = (1-lam)*x of frame j), chunks separated by FIR_G
// zeros, plus the zero-padded tap rows.  Lanes that fall outside a frame's support read guard
// zeros, so the inner loop has no selects or masks: per MFMA one ds_read for each operand.
// Chunk storage is skewed by 2 words per 16 (fir_pad) so the 16-sample-strided B reads of a wave
// hit 32 distinct banks; the K loop is phased so that 4 consecutive K steps stay inside one
// 16-word block and use immediate offsets.
//
// Workgroups are persistent: each walks a strided list of tiles of "its" XCD, and while the MFMAs
// of tile i run, the global loads of tile i+1 (input window as float4, tap rows as float2) are
// already in flight into registers.  Staging competes for issue slots with the MFMA streams of the
// co-resident workgroups, so it is kept to a few wide instructions and 32-bit index arithmetic.
// ------------------------------------------------------------------------------------------------
constexpr int FIR_G = 320;     // guard zeros between chunks: >= 16*15 + 3 + 2*15 (lane spread + loop phase) + 32 (prefetch overrun)
constexpr int FIR_FP = 17;     // zeros in front of a tap row (the phased K loop may start at u0 = -15); FIR_FP + 15 = 32 keeps rows 8-byte aligned
constexpr int FIR_TAIL = 64;   // slack words after the last tap row (operand prefetch overrun)

__device__ __forceinline__ int fir_pad(int q) { return q + 2 * (q >> 4); }

struct FirGeom {
  int F, hop, N, D;
  int T;              // samples per utterance (< 2^31)
  int KU;             // inner extent: u in [0, N+15)
  int HLEN;           // words per padded tap row
  int CS;             // chunk stride (2*hop + FIR_G), before the bank skew
  int NJ;             // chunks / tap rows a workgroup can need
  int WWORDS;         // words reserved for the chunk area (skewed)
  int TPU;            // tiles per utterance
  long NTILES;        // tiles in the launch
  unsigned hop_magic; // ceil(2^32 / hop)
  unsigned hN_magic;  // ceil(2^32 / (N/2))
  int taps_in_regs;   // NJ*N fits the per-thread register prefetch
};

struct FirTile {
  int b, T0;           // utterance, first output sample
  int st_lo;           // first sample of the staged window, multiple of 4 (may be negative)
  int j_lo, nj;        // first frame in reach and number of chunks / tap rows
};

template <int TILE>
__device__ __forceinline__ FirTile fir_tile(int b, int tiu, const FirGeom& g) {
  FirTile t;
  t.b = b;
  t.T0 = tiu * TILE;
  int s_min = t.T0 - g.D + 1;
  if (s_min < 0) s_min = 0;
  int s_max = t.T0 + TILE - 1 + g.D;
  if (s_max > g.T - 1) s_max = g.T - 1;
  t.j_lo = (int)((unsigned)s_min / (unsigned)g.hop);
  int j_hi = (int)((unsigned)s_max / (unsigned)g.hop) + 1;      // row F duplicates row F-1 (core.py:167)
  if (j_hi > g.F) j_hi = g.F;
  t.nj = j_hi - t.j_lo + 1;
  if (t.nj > g.NJ) t.nj = g.NJ;                      // cannot happen (host bound); keeps LDS accesses in range
  t.st_lo = (t.T0 - g.D - 15) & ~3;
  return t;
}

template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64) k_fir_mfma(const float* __restrict__ x, int x_is_u01,
                                                         const float* __restrict__ taps,
                                                         const float* __restrict__ addend, float* __restrict__ out,
                                                         float* __restrict__ out_plain, FirGeom g) {
  constexpr int TILE = WAVES * 256;
  constexpr int NT = WAVES * 64;
  constexpr int XV = (TILE + 1022 + 40 + 4 * NT - 1) / (4 * NT);   // float4 window loads per thread (N <= 1022)
  constexpr int TP = WAVES == 4 ? 6 : 4;                            // float2 tap loads per thread held in registers
  HIP_DYNAMIC_SHARED(float, lds)
  float* W = lds;
  float* HS = lds + g.WWORDS;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63;
  const int c = l & 15;                 // A: row i (fine output offset);  B: column c (coarse offset, x16)
  const int kq = l >> 4;                // K sub-index 0..3
  const int swin = TILE + 2 * g.D + 36;
  const int hN = g.N >> 1;
  const float inv_hop = 1.0f / (float)g.hop;

  // tile schedule: workgroup w runs on XCD w % 8 (dispatch order); every XCD owns a contiguous range of
  // tiles and its resident workgroups sweep it together, so neighbouring tiles (shared tap rows, input
  // halo) meet in that XCD's L2
  const int slots = gridDim.x >> 3;
  const long per_xcd = (g.NTILES + 7) >> 3;
  const long x_begin = (long)(blockIdx.x & 7) * per_xcd;
  long x_end = x_begin + per_xcd;
  if (x_end > g.NTILES) x_end = g.NTILES;
  long tile = x_begin + (blockIdx.x >> 3);
  if (tile >= x_end) return;
  int ub = (int)(tile / g.TPU);                       // utterance / tile-in-utterance, advanced incrementally
  int tiu = (int)(tile - (long)ub * g.TPU);

  float4 xv[XV];
  float2 tv[TP];
  FirTile cur = fir_tile<TILE>(ub, tiu, g);

  // all addressing is a wave-uniform base pointer plus a 32-bit per-lane offset
  auto prefetch = [&](const FirTile& t) {
    const float* xw = x + (long)t.b * g.T + t.st_lo;    // window base (may point before the utterance)
    const int q_lo = t.st_lo < 0 ? -t.st_lo : 0;        // multiples of 4: groups are wholly in or out
    const int q_end = g.T - t.st_lo;
    const int q_hi = q_end < swin ? q_end : swin;
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      const int q = 4 * (tid + i * NT);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (q >= q_lo && q < q_hi) v = *reinterpret_cast<const float4*>(xw + q);
      xv[i] = v;
    }
    if (g.taps_in_regs) {
      // rows j_lo .. j_lo+nj-1 are contiguous in memory; a row index of F re-reads row F-1 (core.py:167)
      const float2* tb = reinterpret_cast<const float2*>(taps + ((long)t.b * g.F + t.j_lo) * g.N);
      const int total = t.nj * hN;
      const int real = (t.j_lo + t.nj > g.F ? g.F - t.j_lo : t.nj) * hN;
#pragma unroll
      for (int i = 0; i < TP; ++i) {
        const int e = tid + i * NT;
        float2 v = make_float2(0.f, 0.f);
        if (e < total) v = tb[e < real ? e : e - hN];
        tv[i] = v;
      }
    }
  };

  prefetch(cur);
  int tile_no = 0;
  (void)tile_no;
  for (;;) {
    FIR_STAMP(0);
    // staging is a short VALU/LDS burst competing with the other workgroups' MFMA streams on the same
    // SIMDs: let it through first so this workgroup gets back to its own MFMAs quickly
    __builtin_amdgcn_s_setprio(3);
    // ---- zero the chunk area (guards, everything outside the window) and the tap rows -----------------------
    {
      float4* L4 = reinterpret_cast<float4*>(lds);
      const int n4 = (g.WWORDS + cur.nj * g.HLEN) >> 2;
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int t = tid; t < n4; t += NT) L4[t] = z;
    }
    __syncthreads();
    FIR_STAMP(1);
    {
      // scatter each group of 4 window samples into the two chunks that weight it.  Frame index / offset
      // come from 32-bit arithmetic: hop_magic = ceil(2^32/hop) makes umulhi an exact divide here.
      int kb = cur.st_lo >= 0 ? (int)((unsigned)cur.st_lo / (unsigned)g.hop)
                              : -(int)(((unsigned)(-cur.st_lo) + g.hop - 1) / (unsigned)g.hop);   // floor
      const int r0 = cur.st_lo - kb * g.hop;            // 0 <= r0 < hop, multiple of 4
      const int jr0 = kb - cur.j_lo;
      const int q_lo = cur.st_lo < 0 ? -cur.st_lo : 0;
      const int q_end = g.T - cur.st_lo;
      const int q_hi = q_end < swin ? q_end : swin;
#pragma unroll
      for (int i = 0; i < XV; ++i) {
        const int q = 4 * (tid + i * NT);
        if (q >= q_lo && q < q_hi) {
          float4 v = xv[i];
          if (x_is_u01) {                               // noise = rand*2-1 (vocoder.py:603,854)
            v.x = fmaf(2.0f, v.x, -1.0f); v.y = fmaf(2.0f, v.y, -1.0f);
            v.z = fmaf(2.0f, v.z, -1.0f); v.w = fmaf(2.0f, v.w, -1.0f);
          }
          const unsigned rel = (unsigned)(r0 + q);
          const int dk = (int)__umulhi(rel, g.hop_magic);
          const int rr = (int)rel - dk * g.hop;         // multiple of 4, rr + 3 < hop
          const float l0 = (float)rr * inv_hop, l1 = (float)(rr + 1) * inv_hop;
          const float l2 = (float)(rr + 2) * inv_hop, l3 = (float)(rr + 3) * inv_hop;
          const int jr = jr0 + dk;                      // chunk of frame k holds it as (1-lam)*x, second half
          if (jr >= 0 && jr < cur.nj) {
            float* d = W + fir_pad(FIR_G + jr * g.CS + g.hop + rr);
            *reinterpret_cast<float2*>(d) = make_float2((1.0f - l0) * v.x, (1.0f - l1) * v.y);
            *reinterpret_cast<float2*>(d + 2) = make_float2((1.0f - l2) * v.z, (1.0f - l3) * v.w);
          }
          if (jr + 1 >= 0 && jr + 1 < cur.nj) {
            float* d = W + fir_pad(FIR_G + (jr + 1) * g.CS + rr);
            *reinterpret_cast<float2*>(d) = make_float2(l0 * v.x, l1 * v.y);
            *reinterpret_cast<float2*>(d + 2) = make_float2(l2 * v.z, l3 * v.w);
          }
        }
      }
      // tap rows: HS[jr][FIR_FP + 15 + m] = taps_j[m]
      if (g.taps_in_regs) {
        const int total = cur.nj * hN;
#pragma unroll
        for (int i = 0; i < TP; ++i) {
          const int e = tid + i * NT;
          if (e < total) {
            const int jr = (int)__umulhi((unsigned)e, g.hN_magic);
            const int m2 = e - jr * hN;
            *reinterpret_cast<float2*>(HS + jr * g.HLEN + FIR_FP + 15 + 2 * m2) = tv[i];
          }
        }
      } else {                                          // tap rows too large for the register prefetch
        for (int jr = 0; jr < cur.nj; ++jr) {
          const int j = cur.j_lo + jr;
          const int row = j < g.F ? j : g.F - 1;
          const float* trow = taps + ((long)cur.b * g.F + row) * g.N;
          float* hrow = HS + jr * g.HLEN + FIR_FP + 15;
#pragma unroll 4
          for (int m = tid; m < g.N; m += NT) hrow[m] = trow[m];
        }
      }
    }
    __syncthreads();
    FIR_STAMP(2);

    // ---- loads of the next tile go out now and land while this tile's MFMAs run --------------------------
    const long next = tile + slots;
    const bool has_next = next < x_end;
    FirTile nxt = cur;
    if (has_next) {
      tiu += slots;
      while (tiu >= g.TPU) { tiu -= g.TPU; ++ub; }
      nxt = fir_tile<TILE>(ub, tiu, g);
      prefetch(nxt);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(0);

    FIR_STAMP(3);
    // ---- contraction --------------------------------------------------------------------------------------
    const int t0w = cur.T0 + 256 * wave;
    if (t0w < g.T) {                                    // wave-uniform
      f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
      for (int jr = 0; jr < cur.nj; ++jr) {
        const int j = cur.j_lo + jr;
        const int lo = (j - 1) * g.hop;                 // support of tri_j: [lo, hi)
        const int hi = lo + 2 * g.hop;
        // K steps u0 for which at least one lane reads inside the support
        const int ulo_l = t0w + g.D + 13 - hi;
        const int uhi_l = t0w + g.D + 256 - lo;
        const int uhi = uhi_l > g.KU ? g.KU : uhi_l;
        if (ulo_l >= uhi || uhi <= 0) continue;         // no lane of this wave reaches frame j
        const int ulo = ulo_l < 0 ? 0 : ulo_l;
        // chunk-local index of lane (c=0,kq=0) at u0 = 0; phase the loop so that index % 16 == 15 at group start
        const int W0 = FIR_G + jr * g.CS + (t0w - lo) + g.D + 15;
        const int r = (W0 - 15) & 15;
        const int ug0 = ulo - ((ulo - r) & 15);
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
      FIR_STAMP(4);

      // epilogue: D layout col = l&15 (c), row = 4*(l>>4) + reg (i)  ->  4 consecutive samples per lane
      // (T % 4 == 0 and 16-byte aligned buffers are launch preconditions of this kernel)
      const int t = t0w + 16 * c + 4 * kq;
      if (t < g.T) {
        const long base = (long)cur.b * g.T + t;
        float4 r4 = make_float4(acc[0], acc[1], acc[2], acc[3]);
        if (out_plain) *reinterpret_cast<float4*>(out_plain + base) = r4;
        if (addend) {
          const float4 a = *reinterpret_cast<const float4*>(addend + base);
          r4.x += a.x; r4.y += a.y; r4.z += a.z; r4.w += a.w;
        }
        *reinterpret_cast<float4*>(out + base) = r4;
      }
    }
    FIR_STAMP(5);
    if (!has_next) break;
    __syncthreads();                                    // every wave is done reading this tile's LDS
    FIR_STAMP(6);
    ++tile_no;
    cur = nxt;
    tile = next;
  }
}

// ---- launchers -----------------------------------------------------------------------------------
static FirGeom fir_geom(int F, int hop, int N, int tile) {
  FirGeom g;
  g.F = F; g.hop = hop; g.N = N; g.D = N / 2;
  g.T = F * hop;
  g.KU = N + 15;
  g.HLEN = (FIR_FP + g.KU + 32 + 3) & ~3;
  g.CS = 2 * hop + FIR_G;
  g.hop_magic = (unsigned)((0x100000000ull + (unsigned long long)hop - 1) / (unsigned long long)hop);
  g.hN_magic = (unsigned)((0x100000000ull + (unsigned long long)(N / 2) - 1) / (unsigned long long)(N / 2));
  g.NJ = (tile + 2 * g.D - 2) / hop + 3;
  int last = g.NJ * g.CS + FIR_G;
  g.WWORDS = (last + 2 * (last >> 4) + 8 + 31) & ~31;
  g.TPU = 0; g.NTILES = 0; g.taps_in_regs = 0;
  return g;
}

size_t fir_mfma_lds_bytes(int F, int hop, int N, int waves) {
  FirGeom g = fir_geom(F, hop, N, waves * 256);
  return ((size_t)g.WWORDS + (size_t)g.NJ * g.HLEN + FIR_TAIL) * sizeof(float);
}

// impl: 0 = auto, 1 = simple, 2 = mfma with 4 waves (1024 outputs) per tile, 3 = mfma with 8 waves,
// 4 = FFT-domain block convolution (fir_fft.hip; hop 512, N <= 512).
// Returns the implementation used, or <0 when the requested kernel cannot take the shape.
int launch_fir(const float* x, int x_is_u01, const float* taps, const float* addend, float* out, float* out_plain,
               int B, int F, int hop, int N, int impl, hipStream_t st, const NoiseGen* noise_gen) {
  const long T = (long)F * hop;
  if (B == 0 || T == 0) return 0;
  if (N & 1) return -1;
  if (noise_gen && noise_gen->on) {                   // only the hop-block form draws its input itself
    if (!(hop == 512 && N <= 512 && (impl == 0 || impl == 5))) return -2;
    return launch_fir_blk(x, x_is_u01, taps, addend, out, out_plain, B, F, hop, N, st, noise_gen);
  }
  auto al = [](const void* p, uintptr_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; };
  // the MFMA kernels move float4 / float2: hop % 4 == 0 (so T % 4 == 0), 16-byte aligned signals, 8-byte aligned taps
  const bool vec_ok = hop >= 16 && (hop & 3) == 0 && N >= 4 && N <= 1022 && T < (1L << 30) && al(x, 16) && al(out, 16) &&
                      al(taps, 8) && (!addend || al(addend, 16)) && (!out_plain || al(out_plain, 16));
  auto fits = [&](int waves) { return vec_ok && fir_mfma_lds_bytes(F, hop, N, waves) <= 64 * 1024; };
  // auto: the hop-block FFT form where it applies (hop 512, N <= 512: 0.137 ms against 0.188 ms for the per-frame
  // FFT form and 0.32 ms for the direct form at B = 32 x 10 s, N = 510), else the MFMA direct form, else the simple kernel
  if (impl == 0 && hop == 512 && N <= 512) {
    // auto: the hop-block form unless it declines the shape (an utterance of 2^28 samples or more: its buffer descriptors
    // span less than 2^30 bytes) -- then the direct forms below, not an error; only an explicit impl = 5 fails there
    const int r = launch_fir_blk(x, x_is_u01, taps, addend, out, out_plain, B, F, hop, N, st);
    if (r >= 0) return r;
  }
  // 514 .. 1022 taps (n_mag up to 512: the harmonic filter of the classic CombSub configuration): the per-frame 2048-point
  // form in its LONG variant -- the direct form on the matrix pipe costs ~3x as much at these tap counts
  if (impl == 0 && hop == 512 && N > 512 && N <= 1022 && (long)F * hop < (1L << 30)) impl = 4;
  if (impl == 4) return launch_fir_fft(x, x_is_u01, taps, addend, out, out_plain, B, F, hop, N, st);
  if (impl == 5) return launch_fir_blk(x, x_is_u01, taps, addend, out, out_plain, B, F, hop, N, st);
  if (impl == 0) impl = fits(8) ? 3 : fits(4) ? 2 : 1;
  if ((impl == 2 && !fits(4)) || (impl == 3 && !fits(8))) return -1;
  if (impl == 2 || impl == 3) {
    const int waves = impl == 2 ? 4 : 8;
    const int tile = waves * 256;
    FirGeom g = fir_geom(F, hop, N, tile);
    g.TPU = (int)((T + tile - 1) / tile);
    g.NTILES = (long)g.TPU * B;
    g.taps_in_regs = (long)g.NJ * (N / 2) <= (long)(waves == 4 ? 6 : 4) * waves * 64;
    const size_t lds = fir_mfma_lds_bytes(F, hop, N, waves);
    // persistent grid: as many workgroups as are resident at once (32 CUs per XCD, LDS- and wave-limited)
    long occ = (long)((160 * 1024) / lds);
    const long wave_occ = 32 / waves;
    if (occ > wave_occ) occ = wave_occ;
    if (occ < 1) occ = 1;
    long slots = 32 * occ;
    if (const long v = knob(KNOB_FIR_MAX_SLOTS)) {         // tuning / test knob: workgroups per XCD
      if (v >= 1 && v < slots) slots = v;
    }
    const long per_xcd = (g.NTILES + 7) / 8;
    if (slots > per_xcd) slots = per_xcd;
    dim3 grid((unsigned)(slots * 8));
    if (waves == 4)
      hipLaunchKernelGGL(k_fir_mfma<4>, grid, dim3(256), lds, st, x, x_is_u01, taps, addend, out, out_plain, g);
    else
      hipLaunchKernelGGL(k_fir_mfma<8>, grid, dim3(512), lds, st, x, x_is_u01, taps, addend, out, out_plain, g);
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

#ifdef DDSP_HIP_TIMELINE
extern "C" int ddsp_hip_debug_set_timeline(long long* p, void* stream) {
  hipLaunchKernelGGL(ddsp::k_set_timeline, dim3(1), dim3(1), 0, (hipStream_t)stream, p);
  return (int)hipGetLastError();
}
#endif
