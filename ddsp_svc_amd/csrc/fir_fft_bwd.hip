// Adjoints of the time-varying FIR (ddsp/core.py:120-182 under autograd; solver.py:93-103 back-propagates through it) for
// 514 .. 1022 taps at hop 512 -- the harmonic filter of the classic CombSub configuration (n_mag 256 / 512 / 256) -- in the
// form of the forward kernel that takes those shapes, k_fir_fft<LONG> (fir_fft.hip): per-frame 2048-point transforms.  Until
// round 5 such a filter's gradients were direct correlations (fir_bwd_direct.hip: 1.7 ms per launch, 3.6 with the input
// gradient, at B = 32 x 10 s against 0.22 ms for the forward pass).
//
// Frame j (0..F; row min(j, F-1), core.py:167) convolves its taps h_j with the chunk c_j[m] = (x tri)[(j-1) hop + m],
// m < 2 hop, and adds the result at a_j = (j-1) hop - N/2.  With gs_j[u] = grad_out[a_j + u], u < 2048 (zero outside the signal):
//     d_taps[row j][n]      += sum_m c_j[m] gs_j[n + m]                 n < N
//     d_x[(j-1) hop + m]    += tri[m] sum_n h_j[n] gs_j[n + m]          m < 2 hop
// two cross-correlations against the SAME window of the cotangent (n + m <= 2044: a 2048-point circular correlation does not
// wrap), and both results are REAL sequences.  So with the forward kernel's packed transform Z_j = FFT(c_j + i s h_j),
//     IFFT(conj(Z_j) GS_j) = corr(c_j, gs_j) - i s corr(h_j, gs_j):
// the real part IS the tap gradient and the imaginary part the chunk gradient -- no separation of Z_j, one inverse per frame --
// and two frames share the transform of their cotangent windows, Q = FFT(gs_j + i gs_j+1), GS_j = (Q[k] + conj Q[-k]) / 2,
// GS_j+1 = (Q[k] - conj Q[-k]) / 2i.  Per PAIR of frames: five 2048-point transforms (forward pass: three).
//
// A 256-thread workgroup walks a run of consecutive pairs of one utterance.  The chunk gradients of neighbouring frames
// overlap by half (every input sample gets a rising and a falling contribution): the falling half of a pair's second frame is
// carried in registers to the next pair, and a run starts with the second frame of the pair before it to have that carry.
// The tap gradient of a frame is stored as it is -- except on the held last row, which frames F-1 and F both feed: that row
// is zeroed by the launcher and takes two atomic adds (two addends: the sum does not depend on their order).
#include "fft2048.h"
#include "kernels.h"

namespace ddsp {

using fft::cmulc;

constexpr int FFB_HOP = 512;
#ifndef DDSP_FFB_WGS
#define DDSP_FFB_WGS 3                    // resident workgroups per CU the register budget is set for: 168 registers with 10 spilled
                                          // dwords beat 180 without at two (0.320 against 0.341 ms, same box, r05_v6_fir_bwd_long.txt)
#endif

struct FirFftBwdGeom {
  int F, N, T;            // frames, taps, samples per utterance
  int pairs;              // frame pairs per utterance: ceil((F + 1) / 2)
  int run;                // own pairs per workgroup
  int runs_per_utt;       // ceil(pairs / run)
};

template <bool WITH_DX>
__global__ void __launch_bounds__(fft::THREADS, DDSP_FFB_WGS) k_fir_fft_bwd(const float* __restrict__ x, int x_is_u01,
                                                                const float* __restrict__ taps,
                                                                const float* __restrict__ grad_out, float* __restrict__ d_x,
                                                                float* __restrict__ d_taps, FirFftBwdGeom g) {
  constexpr int S = fft::SLOTS;                               // 8 complex points per thread, point k = 256 m + tid
  __shared__ __attribute__((aligned(16))) f32x2 exA[fft::EX_WORDS];
  __shared__ __attribute__((aligned(16))) f32x2 exB[fft::EX_WORDS];
  __shared__ float red[8];
  const int tid = threadIdx.x;
  const int b = blockIdx.x / g.runs_per_utt;
  const int run_no = blockIdx.x - b * g.runs_per_utt;
  const int p_first = run_no * g.run;                         // first own pair
  int p_last = p_first + g.run;                               // one past the last own pair
  if (p_last > g.pairs) p_last = g.pairs;
  const int D = g.N >> 1;
  const float* xb = x + (long)b * g.T;
  const float* tb = taps + (long)b * g.F * g.N;
  const float* gb = grad_out + (long)b * g.T;
  const float inv_hop = 1.0f / (float)FFB_HOP;

  fft::Twiddles tw;
  tw.init(tid);

  // one frame's operands: chunk and taps (4 slots each), cotangent window (8 slots).  Each array is fetched for the NEXT pair right
  // after this pair's transform has consumed it, into the registers it just left: one copy of everything is live, not two
  struct ChunkTaps { float cv[4], hv[4]; };
  struct Window { float gv[S]; };
  auto load_ct = [&](int j) -> ChunkTaps {
    ChunkTaps f;
#pragma unroll
    for (int m = 0; m < 4; ++m) { f.cv[m] = 0.f; f.hv[m] = 0.f; }
    if (j <= g.F) {                                           // j == F + 1 only pads an odd frame count
      const int s0 = (j - 1) * FFB_HOP;
      const int row = j < g.F ? j : g.F - 1;                  // core.py:167
      const float* tr = tb + (long)row * g.N;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int n = 256 * m + tid;
        if (n < g.N) f.hv[m] = tr[n];
        const int sidx = s0 + n;
        if (sidx >= 0 && sidx < g.T) {
          float xv = xb[sidx];
          if (x_is_u01) xv = fmaf(2.0f, xv, -1.0f);           // noise = rand*2-1 (vocoder.py:603,854)
          const float lam = (float)(n & (FFB_HOP - 1)) * inv_hop;
          f.cv[m] = (n < FFB_HOP ? lam : 1.0f - lam) * xv;    // periodic Bartlett (core.py:161)
        }
      }
    }
    return f;
  };
  auto load_win = [&](int j) -> Window {
    Window w;
#pragma unroll
    for (int m = 0; m < S; ++m) w.gv[m] = 0.f;
    if (j <= g.F) {
      const int a0 = (j - 1) * FFB_HOP - D;                   // gs_j[u] = grad_out[a_j + u]
#pragma unroll
      for (int m = 0; m < S; ++m) {
        const int t = a0 + 256 * m + tid;
        if (t >= 0 && t < g.T) w.gv[m] = gb[t];
      }
    }
    return w;
  };

  float carry[2] = {0.f, 0.f};                                // falling half of the previous frame's chunk gradient (block 2 pr - 1)
  const int pr0 = (WITH_DX && p_first > 0) ? p_first - 1 : p_first;    // the pair before the run: only its second frame, for the carry
  Window w0 = load_win(2 * pr0), w1 = load_win(2 * pr0 + 1);
  ChunkTaps ct[2] = {load_ct(2 * pr0), load_ct(2 * pr0 + 1)};
  for (int pr = pr0; pr < p_last; ++pr) {
    const bool own = pr >= p_first;
    const bool more = pr + 1 < p_last;
    // ---- Q = FFT(gs_j0 + i gs_j1) -> GS_j0, GS_j1 ----
    f32x2 GS0[S], GS1[S];
    {
      f32x2 q[S];
#pragma unroll
      for (int m = 0; m < S; ++m) q[m] = f32x2{w0.gv[m], w1.gv[m]};
      if (more) { w0 = load_win(2 * pr + 2); w1 = load_win(2 * pr + 3); }      // lands while this pair is transformed
      fft::forward(q, tw, exA, exB, tid);
#pragma unroll
      for (int m = 0; m < S; ++m) exB[256 * m + tid] = q[m];   // natural order (B is free), then the mirrored read
      __syncthreads();
      const f32x2 half = {0.5f, 0.5f}, mih = {0.5f, -0.5f};
#pragma unroll
      for (int m = 0; m < S; ++m) {
        const int k = 256 * m + tid;
        const f32x2 zneg = exB[(fft::N - k) & (fft::N - 1)];
        GS0[m] = fft::add_conj(q[m], zneg) * half;                            // (Q[k] + conj Q[-k]) / 2
        GS1[m] = fft::swap_scale(fft::sub_conj(q[m], zneg), mih);             // (Q[k] - conj Q[-k]) / 2i = -i/2 (..)
      }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (h == 0 && !own) {                                    // the warm-up pair: only its second frame feeds the carry
        if (more) ct[0] = load_ct(2 * pr + 2);
        continue;
      }
      const ChunkTaps cur = ct[h];
      const int j = 2 * pr + h;
      // energies of the two sequences -> power-of-two balance factor for the taps, as the forward kernel
      float sx = 0.f, sh = 0.f;
#pragma unroll
      for (int m = 0; m < 4; ++m) { sx = fmaf(cur.cv[m], cur.cv[m], sx); sh = fmaf(cur.hv[m], cur.hv[m], sh); }
      sx = wave_sum_dpp(sx);
      sh = wave_sum_dpp(sh);
      if ((tid & 63) == 0) { red[(tid >> 6) * 2] = sx; red[(tid >> 6) * 2 + 1] = sh; }
      __syncthreads();                                         // (also: every wave has left the previous transform's last pass)
      sx = (red[0] + red[2]) + (red[4] + red[6]);
      sh = (red[1] + red[3]) + (red[5] + red[7]);
      float sc = 1.0f, isc = 1.0f;
      if (sx > 0.f && sh > 0.f && sx < 3e38f && sh < 3e38f) {
        int e = (ilogbf(sx) - ilogbf(sh)) >> 1;
        e = e < -60 ? -60 : (e > 60 ? 60 : e);
        sc = ldexpf(1.0f, e);
        isc = ldexpf(1.0f, -e);
      }
      f32x2 z[S];
#pragma unroll
      for (int m = 0; m < S; ++m) z[m] = f32x2{m < 4 ? cur.cv[m < 4 ? m : 0] : 0.f, m < 4 ? cur.hv[m < 4 ? m : 0] * sc : 0.f};
      if (more) ct[h] = load_ct(2 * pr + 2 + h);               // this frame's successor, into the registers it leaves
      fft::forward(z, tw, exA, exB, tid);
      // conj(conj(Z) GS) = Z conj(GS); its forward transform R gives IFFT(conj(Z) GS) = conj(R) / 2048
#pragma unroll
      for (int m = 0; m < S; ++m) z[m] = cmulc(h == 0 ? GS0[m] : GS1[m], z[m]);
      __syncthreads();                                         // slower waves may still read A (last pass of the transform above)
      fft::forward(z, tw, exA, exB, tid);
      const float scale = 1.0f / 2048.0f;
      // ---- tap gradient: Re(conj R) / 2048 ----
      if (own && j <= g.F) {
        const int row = j < g.F ? j : g.F - 1;
        float* dr = d_taps + ((long)b * g.F + row) * g.N;
        const bool shared_row = j >= g.F - 1;                  // frames F - 1 and F both feed the held last row (zeroed by the launcher)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int n = 256 * m + tid;
          if (n < g.N) {
            const float v = z[m].x * scale;
            if (shared_row) atomicAdd(dr + n, v);
            else dr[n] = v;
          }
        }
      }
      // ---- chunk gradient: Im(conj R) / 2048 = -s corr(h, gs)  ->  corr = R.y / (2048 s), times the Bartlett weight ----
      if (WITH_DX) {
        const float cs = scale * isc;
        float dc[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int n = 256 * m + tid;
          const float lam = (float)(n & (FFB_HOP - 1)) * inv_hop;
          dc[m] = (n < FFB_HOP ? lam : 1.0f - lam) * (z[m].y * cs);
        }
        // rising half (slots 0, 1) completes block j - 1 together with the carried falling half of frame j - 1
        const int blk = j - 1;
        if (own && blk >= 0 && blk < g.F) {
          float* dst = d_x + (long)b * g.T + (long)blk * FFB_HOP;
          dst[tid] = carry[0] + dc[0];
          dst[256 + tid] = carry[1] + dc[1];
        }
        carry[0] = dc[2];
        carry[1] = dc[3];
      }
    }
    __syncthreads();                                           // the next pair's first transform writes A
  }
}

// returns 0, or -1 when the shape is outside this kernel (hop 512, even N <= 1022)
int launch_fir_fft_bwd(const float* x, int x_is_u01, const float* taps, const float* grad_out, float* d_x, float* d_taps,
                       int B, int F, int hop, int N, hipStream_t st) {
  if (hop != FFB_HOP || N < 2 || (N & 1) || N > 1022 || F < 1 || (long)F * hop >= (1L << 30)) return -1;
  if ((long)B * F == 0) return 0;
  FirFftBwdGeom g;
  g.F = F; g.N = N; g.T = F * hop;
  g.pairs = (F + 2) / 2;
  // run length: one round of resident workgroups (two per CU at this kernel's register budget), equal work; every run but an
  // utterance's first pays two of a pair's five transforms for its carry
  const long slots = DDSP_FFB_WGS * 256;
  const int Bg = t_geometry_batch > 0 ? t_geometry_batch : B;
  long per_utt = slots / (Bg > 0 ? Bg : 1);
  if (per_utt < 1) per_utt = 1;
  int run = (int)((g.pairs + per_utt - 1) / per_utt);
  if (run < 3) run = 3;
  if (const long v = knob(KNOB_FFT_RUN)) { if (v >= 1) run = (int)v; }
  if (run > g.pairs) run = g.pairs;
  g.run = run;
  g.runs_per_utt = (g.pairs + run - 1) / run;
  const long wgs = (long)B * g.runs_per_utt;
  if (wgs > 0x7fffffffL) return -1;
  // the held last row takes two atomic adds per tap (frames F - 1 and F)
  if (hipMemset2DAsync(d_taps + (long)(F - 1) * N, (size_t)F * N * sizeof(float), 0, (size_t)N * sizeof(float), (size_t)B, st) != hipSuccess)
    return -1;
  if (d_x)
    hipLaunchKernelGGL(k_fir_fft_bwd<true>, dim3((unsigned)wgs), dim3(fft::THREADS), 0, st, x, x_is_u01, taps, grad_out, d_x, d_taps, g);
  else
    hipLaunchKernelGGL(k_fir_fft_bwd<false>, dim3((unsigned)wgs), dim3(fft::THREADS), 0, st, x, x_is_u01, taps, grad_out, d_x, d_taps, g);
  return 0;
}

}  // namespace ddsp
