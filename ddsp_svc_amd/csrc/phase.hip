// HOT-1: control upsampling and the phase accumulator of Sins/CombSub
// (reference: ddsp/core.py:66-77, ddsp/vocoder.py:564-575 and :819-829).
//
// The only sequential dependency of the whole synthesiser is the cumulative sum of f0/sr over an
// utterance.  It is done as a three-level scan: each lane adds its own consecutive samples, a
// wave64 butterfly adds the 64 lanes of one frame (k_phase_frame_sums), and one small workgroup
// per utterance scans the frame totals (k_phase_frame_scan).  Consumers rebuild x[t] from the
// per-frame start value with one more wave scan (frame_phase() below) instead of reading a
// [B,T] tensor back from HBM.
#include "ddsp_common.h"
#include "kernels.h"

namespace ddsp {

// ------------------------------------------------------------------------------------------------
// core.upsample: [B,F,C] -> [B,F*hop,C]
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_upsample(const float* __restrict__ sig, int B, int F, int C, int hop,
                                                  Upsampler up, float* __restrict__ out) {
  long total = (long)B * F * hop * C;
  long stride = (long)gridDim.x * blockDim.x;
  long T = (long)F * hop;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    int c = (int)(i % C);
    long bt = i / C;
    long t = bt % T;
    long b = bt / T;
    out[i] = up.at(sig + b * (long)F * C + c, C, t);
  }
}

// core.remove_above_fmax: amps * ((pitch*k < fmax) + 1e-7), k = level_start..
__global__ void __launch_bounds__(256) k_remove_above_fmax(const float* __restrict__ amps, const float* __restrict__ pitch,
                                                           long rows, int H, float fmax, int level_start,
                                                           float* __restrict__ out) {
  long total = rows * H;
  long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    long r = i / H;
    int h = (int)(i % H);
    float p = pitch[r] * (float)(h + level_start);
    float aa = (p < fmax ? 1.0f : 0.0f) + 1e-7f;
    out[i] = amps[i] * aa;
  }
}

// ------------------------------------------------------------------------------------------------
// level 1+2: per-frame totals of f0[t]/sr.  One wave per frame, lane owns SPL consecutive samples.
// ------------------------------------------------------------------------------------------------
#ifndef DDSP_PH_FPW
#define DDSP_PH_FPW 4
#endif
constexpr int PH_FRAMES_PER_WAVE = DDSP_PH_FPW;      // consecutive frames a wave walks: amortises the launch of tiny workgroups

template <int SPL, bool POW2>
__global__ void __launch_bounds__(256) k_phase_frame_sums(const float* __restrict__ f0_frames, long n_frames, int F,
                                                          int hop, Upsampler up, PhaseCfg cfg,
                                                          double* __restrict__ sums) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // frame indices fit 32 bits (checked by the launcher): one unsigned division per wave instead of a 64-bit
  // division and remainder per frame, which were 40 % of this kernel's instructions
  const unsigned fr0 = ((unsigned)blockIdx.x * 4 + wave) * PH_FRAMES_PER_WAVE;
  unsigned b = fr0 / (unsigned)F;
  int f = (int)(fr0 - b * (unsigned)F);
#pragma unroll 1
  for (int q = 0; q < PH_FRAMES_PER_WAVE; ++q, ++f) {
    const long fr = (long)fr0 + q;
    if (fr >= n_frames) return;                     // wave-uniform
    if (f == F) { f = 0; ++b; }
    const float* row = f0_frames + (long)b * F;
    const Upsampler::Row3 rows = up.load3(row, f);
    double acc = 0.0;
#pragma unroll
    for (int r = 0; r < SPL; ++r) {
      int j = lane * SPL + r;
      if (j < hop) acc += cfg.term(POW2 ? up.at3_pow2(rows, j) : up.at3_in_frame(rows, j, hop));
    }
    acc = wave_sum(acc);
    if (lane == 0) sums[fr] = acc;
  }
}

// ------------------------------------------------------------------------------------------------
// level 3: exclusive scan of the frame totals of one utterance; also emits phase_frames
// (= 2*pi*x[:, ::hop], vocoder.py:574-575).  One 256-thread workgroup per utterance.
// ------------------------------------------------------------------------------------------------
// the scan of one utterance's frame totals s[0..F) by the 256 threads of a workgroup (both kernels below)
__device__ __forceinline__ void phase_scan_utterance(const float* __restrict__ f0_frames, const float* __restrict__ initial_phase,
                                                     int F, int hop, const Upsampler& up, const PhaseCfg& cfg, const double* s,
                                                     double* __restrict__ phase0, float* __restrict__ phase_frames, long b,
                                                     double* part) {
  const int tid = threadIdx.x;
  const int chunk = (F + 255) / 256;
  const int lo = tid * chunk;
  const int hi = (lo + chunk < F) ? lo + chunk : F;
  double local = 0.0;
  for (int f = lo; f < hi; ++f) local += s[f];
  part[tid] = local;
  __syncthreads();
  // Hillis-Steele inclusive scan over the 256 partials
  for (int d = 1; d < 256; d <<= 1) {
    double u = (tid >= d) ? part[tid - d] : 0.0;
    __syncthreads();
    part[tid] += u;
    __syncthreads();
  }
  double run = part[tid] - local;
  const float ip = cfg.has_ip ? initial_phase[b] : 0.0f;
  const float* row = f0_frames + b * F;
  for (int f = lo; f < hi; ++f) {
    phase0[b * F + f] = run;
    if (phase_frames) {
      double P = run + cfg.term(up.at(row, 1, (long)f * hop));     // inclusive sum at sample f*hop
      phase_frames[b * F + f] = kTwoPiF * cfg.wrap(P, ip);
    }
    run += s[f];
  }
}

__global__ void __launch_bounds__(256) k_phase_frame_scan(const float* __restrict__ f0_frames,
                                                          const float* __restrict__ initial_phase, int F, int hop,
                                                          Upsampler up, PhaseCfg cfg, const double* __restrict__ sums,
                                                          double* __restrict__ phase0, float* __restrict__ phase_frames) {
  __shared__ double part[256];
  const long b = blockIdx.x;
  phase_scan_utterance(f0_frames, initial_phase, F, hop, up, cfg, sums + b * F, phase0, phase_frames, b, part);
}

// Streaming shapes (B = 1, a fraction of a second per call -- gui.py:118-133): both levels in ONE launch, one workgroup per
// utterance: its four waves take the frames in turn (the frame totals exactly as k_phase_frame_sums<8, true> forms them),
// the totals wait in LDS, the same scan follows.  At such shapes a step's latency is its chain of dependent launches, about
// 9 us each, not the work; at batch shapes the per-sample float64 work wants the whole chip (the two-launch form above).
constexpr int PH_SMALL_MAX_F = 1024;

__global__ void __launch_bounds__(1024) k_phase_small(const float* __restrict__ f0_frames, const float* __restrict__ initial_phase,
                                                      int F, int hop, Upsampler up, PhaseCfg cfg, double* __restrict__ sums,
                                                      double* __restrict__ phase0, float* __restrict__ phase_frames) {
  constexpr int SPL = 8;
  __shared__ double part[1024];                      // the scan is the first 256 threads' (the others hold empty chunks)
  __shared__ double tot[PH_SMALL_MAX_F];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long b = blockIdx.x;
  const float* row = f0_frames + b * F;
#pragma unroll 1
  for (int f = wave; f < F; f += 16) {
    const Upsampler::Row3 rows = up.load3(row, f);
    double acc = 0.0;
#pragma unroll
    for (int r = 0; r < SPL; ++r) {
      const int j = lane * SPL + r;
      if (j < hop) acc += cfg.term(up.at3_pow2(rows, j));
    }
    acc = wave_sum(acc);
    if (lane == 0) { tot[f] = acc; sums[b * F + f] = acc; }
  }
  __syncthreads();
  phase_scan_utterance(f0_frames, initial_phase, F, hop, up, cfg, tot, phase0, phase_frames, b, part);
}

// ------------------------------------------------------------------------------------------------
// optional: materialise x[B,T] (API parity with the reference's intermediate; consumers do not need it)
// ------------------------------------------------------------------------------------------------
template <int SPL>
__global__ void __launch_bounds__(256) k_phase_expand(const float* __restrict__ f0_frames,
                                                      const float* __restrict__ initial_phase, long n_frames, int F,
                                                      int hop, Upsampler up, PhaseCfg cfg,
                                                      const double* __restrict__ phase0, float* __restrict__ x) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  long fr = (long)blockIdx.x * 4 + wave;
  if (fr >= n_frames) return;
  const unsigned bu = (unsigned)fr / (unsigned)F;    // 32-bit frame indices (launcher)
  long b = bu;
  int f = (int)((unsigned)fr - bu * (unsigned)F);
  const float* row = f0_frames + b * F;
  const float ip = cfg.has_ip ? initial_phase[b] : 0.0f;
  double pre[SPL];
  double acc = 0.0;
  const Upsampler::Row3 rows = up.load3(row, f);
#pragma unroll
  for (int r = 0; r < SPL; ++r) {
    int j = lane * SPL + r;
    if (j < hop) acc += cfg.term(up.at3_in_frame(rows, j, hop));
    pre[r] = acc;
  }
  double base = phase0[fr] + wave_excl_scan(acc, lane);
#pragma unroll
  for (int r = 0; r < SPL; ++r) {
    int j = lane * SPL + r;
    if (j < hop) x[(b * F + f) * (long)hop + j] = cfg.wrap(base + pre[r], ip);
  }
}

}  // namespace ddsp

// ---- host-side launchers (C++ linkage inside the library; the C ABI is in api.hip) -------------------
namespace ddsp {

static inline Upsampler make_upsampler(int F, int hop) {
  Upsampler u;
  u.F = F;
  u.scale = (float)F / (float)((long)F * hop);
  u.shift = 0;
  if (hop > 1 && (hop & (hop - 1)) == 0 && (long)F * hop <= (1L << 24)) {
    while ((1 << u.shift) < hop) ++u.shift;
  }
  return u;
}

Upsampler make_upsampler_pub(int F, int hop) { return make_upsampler(F, hop); }

int spl_for_hop(int hop) {
  int need = (hop + 63) / 64;
  return need <= 8 ? 8 : need <= 16 ? 16 : need <= 32 ? 32 : 0;
}

void launch_upsample(const float* sig, int B, int F, int C, int hop, float* out, hipStream_t st) {
  long total = (long)B * F * hop * C;
  if (total == 0) return;
  long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_upsample, dim3((unsigned)blocks), dim3(256), 0, st, sig, B, F, C, hop, make_upsampler(F, hop), out);
}

void launch_remove_above_fmax(const float* amps, const float* pitch, long rows, int H, float fmax, int level_start,
                              float* out, hipStream_t st) {
  long total = rows * H;
  if (total == 0) return;
  long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_remove_above_fmax, dim3((unsigned)blocks), dim3(256), 0, st, amps, pitch, rows, H, fmax,
                     level_start, out);
}

PhaseCfg make_phase_cfg(double sr, int infer, int has_ip) {
  PhaseCfg c;
  c.sr_d = sr;
  c.rsr_d = 1.0 / sr;
  c.sr_f = (float)sr;
  c.infer = infer;
  c.has_ip = has_ip;
  return c;
}

// returns 0 on success, -1 when hop is too large for the wave-per-frame decomposition
int launch_phase(const float* f0_frames, const float* initial_phase, int B, int F, int hop, double sr, int infer,
                 double* frame_sums, double* phase0, float* phase_frames, float* x_or_null, hipStream_t st) {
  const int spl = spl_for_hop(hop);
  if (!spl) return -1;
  const long n_frames = (long)B * F;
  if (n_frames == 0) return 0;
  if (n_frames >= (1L << 31) - 64) return -1;                     // the kernels index frames in 32 bits
  Upsampler up = make_upsampler(F, hop);
  PhaseCfg cfg = make_phase_cfg(sr, infer, initial_phase != nullptr);
  dim3 grid((unsigned)((n_frames + 3) / 4)), block(256);
  if (!x_or_null && spl == 8 && up.shift > 0 && F <= PH_SMALL_MAX_F && n_frames < kSmallRows && knob(KNOB_SMALL_PATH) != 1) {
    // (at batch shapes the one-launch form loses: B workgroups do not fill the chip -- 52 us against 10 + 5 at B = 32 x 10 s, r04_v22)
    hipLaunchKernelGGL(k_phase_small, dim3((unsigned)B), dim3(1024), 0, st, f0_frames, initial_phase, F, hop, up, cfg, frame_sums,
                       phase0, phase_frames);
    return 0;
  }
  const dim3 sgrid((unsigned)((n_frames + 4 * PH_FRAMES_PER_WAVE - 1) / (4 * PH_FRAMES_PER_WAVE)));
  if (spl == 8)
    if (up.shift > 0) hipLaunchKernelGGL((k_phase_frame_sums<8, true>), sgrid, block, 0, st, f0_frames, n_frames, F, hop, up, cfg, frame_sums);
    else hipLaunchKernelGGL((k_phase_frame_sums<8, false>), sgrid, block, 0, st, f0_frames, n_frames, F, hop, up, cfg, frame_sums);
  else if (spl == 16)
    hipLaunchKernelGGL((k_phase_frame_sums<16, false>), sgrid, block, 0, st, f0_frames, n_frames, F, hop, up, cfg, frame_sums);
  else
    hipLaunchKernelGGL((k_phase_frame_sums<32, false>), sgrid, block, 0, st, f0_frames, n_frames, F, hop, up, cfg, frame_sums);
  hipLaunchKernelGGL(k_phase_frame_scan, dim3((unsigned)B), dim3(256), 0, st, f0_frames, initial_phase, F, hop, up, cfg,
                     (const double*)frame_sums, phase0, phase_frames);
  if (x_or_null) {
    if (spl == 8)
      hipLaunchKernelGGL(k_phase_expand<8>, grid, block, 0, st, f0_frames, initial_phase, n_frames, F, hop, up, cfg,
                         (const double*)phase0, x_or_null);
    else if (spl == 16)
      hipLaunchKernelGGL(k_phase_expand<16>, grid, block, 0, st, f0_frames, initial_phase, n_frames, F, hop, up, cfg,
                         (const double*)phase0, x_or_null);
    else
      hipLaunchKernelGGL(k_phase_expand<32>, grid, block, 0, st, f0_frames, initial_phase, n_frames, F, hop, up, cfg,
                         (const double*)phase0, x_or_null);
  }
  return 0;
}

}  // namespace ddsp
