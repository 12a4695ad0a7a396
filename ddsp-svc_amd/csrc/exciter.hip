// HOT-2a: the two exciters -- band-limited combtooth (CombSub, ddsp/vocoder.py:839-840) and the
// additive sinusoid bank (Sins, ddsp/vocoder.py:580,585-594).
//
// Both rebuild the wrapped phase x[t] of their frame from the per-frame start value produced by
// k_phase_frame_scan (one wave per frame, lane owns SPL consecutive samples, one wave64 scan), so
// the [B,T] phase tensor of the reference never exists in HBM.  The sinusoid bank keeps the two
// amplitude rows it interpolates between in LDS and never materialises the reference's
// [B,T,32] temporaries.
#include "ddsp_common.h"

namespace ddsp {

template <int SPL>
struct FramePhase {
  float x[SPL];      // wrapped phase, cycles in [-0.5, 0.5]
  float f0u[SPL];    // upsampled f0 at the same samples
};

// x[t] and f0[t] for the SPL samples this lane owns in frame fr (all 64 lanes of the wave must call)
template <int SPL>
__device__ __forceinline__ void frame_phase(const float* __restrict__ f0_row, int f, int hop, const Upsampler& up,
                                            const PhaseCfg& cfg, double phase0, float ip, int lane,
                                            FramePhase<SPL>& o) {
  double pre[SPL];
  double acc = 0.0;
  const Upsampler::Row3 rows = up.load3(f0_row, f);
#pragma unroll
  for (int r = 0; r < SPL; ++r) {
    int j = lane * SPL + r;
    float v = 0.f;
    if (j < hop) {
      v = up.at3(rows, (long)f * hop + j);
      acc += cfg.term(v);
    }
    o.f0u[r] = v;
    pre[r] = acc;
  }
  double base = phase0 + wave_excl_scan(acc, lane);
#pragma unroll
  for (int r = 0; r < SPL; ++r) o.x[r] = cfg.wrap(base + pre[r], ip);
}

template <int SPL>
__device__ __forceinline__ void store_frame(float* __restrict__ dst, int hop, int lane, const float (&v)[SPL]) {
  // dst points at the first sample of the frame; lane owns [lane*SPL, lane*SPL+SPL)
  if ((hop & 3) == 0 && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
#pragma unroll
    for (int r = 0; r < SPL; r += 4) {
      int j = lane * SPL + r;
      if (j < hop) *reinterpret_cast<float4*>(dst + j) = make_float4(v[r], v[r + 1], v[r + 2], v[r + 3]);
    }
  } else {
#pragma unroll
    for (int r = 0; r < SPL; ++r) {
      int j = lane * SPL + r;
      if (j < hop) dst[j] = v[r];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// combtooth = sinc(sr * x / (f0 + 1e-3)); one wave per frame
// ------------------------------------------------------------------------------------------------
template <int SPL>
__global__ void __launch_bounds__(256) k_combtooth(const float* __restrict__ f0_frames,
                                                   const float* __restrict__ initial_phase, long n_frames, int F,
                                                   int hop, Upsampler up, PhaseCfg cfg,
                                                   const double* __restrict__ phase0, float* __restrict__ out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  long fr = (long)blockIdx.x * 4 + wave;
  if (fr >= n_frames) return;
  long b = fr / F;
  int f = (int)(fr % F);
  const float ip = cfg.has_ip ? initial_phase[b] : 0.0f;
  FramePhase<SPL> ph;
  frame_phase<SPL>(f0_frames + b * F, f, hop, up, cfg, phase0[fr], ip, lane, ph);
  float v[SPL];
#pragma unroll
  for (int r = 0; r < SPL; ++r) {
    float num = cfg.sr_f * ph.x[r];
    float den = ph.f0u[r] + 1e-3f;
    v[r] = sinc_f32(num / den);
  }
  store_frame<SPL>(out + fr * (long)hop, hop, lane, v);
}

// ------------------------------------------------------------------------------------------------
// sinusoid bank: sum_k sin(fl32(phase*k)) * lerp(A[f][k], A[f+1][k]),  A = mask * exp(c)/128
// Workgroup = 4 waves = 4 consecutive frames of one utterance; the 5 amplitude rows they touch are
// activated once and parked in LDS.
// ------------------------------------------------------------------------------------------------
template <int SPL>
__global__ void __launch_bounds__(256) k_sins_bank(const float* __restrict__ f0_frames,
                                                   const float* __restrict__ initial_phase,
                                                   const float* __restrict__ c_amp, long ld_amp, int F, int hop, int H,
                                                   Upsampler up, PhaseCfg cfg, const double* __restrict__ phase0,
                                                   float* __restrict__ out) {
  HIP_DYNAMIC_SHARED(float, rows)                   // [5][H]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int groups = (F + 3) / 4;
  const long b = blockIdx.x / groups;
  const int f_base = (int)(blockIdx.x % groups) * 4;
  const float* f0_row = f0_frames + b * F;
  const float nyq = cfg.sr_f / 2.0f;
  for (int i = threadIdx.x; i < 5 * H; i += 256) {
    int rr = i / H, k = i - rr * H;
    int f = f_base + rr;
    if (f > F - 1) f = F - 1;                       // last frame held (core.py:68)
    float a = expf(c_amp[(b * F + f) * ld_amp + k]) / 128.0f;
    float p = f0_row[f] * (float)(k + 1);
    float aa = (p < nyq ? 1.0f : 0.0f) + 1e-7f;
    rows[i] = a * aa;
  }
  __syncthreads();
  const int f = f_base + wave;
  if (f >= F) return;                               // wave-uniform, after the only barrier
  const long fr = b * F + f;
  const float ip = cfg.has_ip ? initial_phase[b] : 0.0f;
  FramePhase<SPL> ph;
  frame_phase<SPL>(f0_row, f, hop, up, cfg, phase0[fr], ip, lane, ph);
  // sum_k sin(fl32(phase*k)) * (w0*A0[k] + w1*A1[k]) is accumulated as w0*S0 + w1*S1 with S_i = sum_k sin(.)*A_i[k]:
  // two fmas per harmonic and sample instead of an interpolation plus an fma (same sum, rounding differs at 1e-7).
  // Samples are processed in pairs so that the multiplies and fmas issue as packed-f32 instructions
  // (v_pk_mul_f32 / v_pk_fma_f32): per harmonic and pair 4 packed + 2 rndne + 2 v_sin + 2 packed accumulations.
  f32x2 phase[SPL / 2], s0[SPL / 2], s1[SPL / 2];
#pragma unroll
  for (int r = 0; r < SPL / 2; ++r) {
    phase[r] = f32x2{kTwoPiF * ph.x[2 * r], kTwoPiF * ph.x[2 * r + 1]};     // vocoder.py:574
    s0[r] = f32x2{0.f, 0.f};
    s1[r] = f32x2{0.f, 0.f};
  }
  const float* ra = rows + wave * H;
  const float* rb = ra + H;
  for (int k = 0; k < H; ++k) {
    const float a0 = ra[k], a1 = rb[k];
    const float kf = (float)(k + 1);
    const f32x2 a0v = {a0, a0}, a1v = {a1, a1};
#pragma unroll
    for (int r = 0; r < SPL / 2; ++r) {
      const f32x2 s = sin_turns2(phase[r] * kf);    // the reference rounds phase*k to float32 first (vocoder.py:590)
      s0[r] = __builtin_elementwise_fma(s, a0v, s0[r]);
      s1[r] = __builtin_elementwise_fma(s, a1v, s1[r]);
    }
  }
  float acc[SPL];
#pragma unroll
  for (int r = 0; r < SPL; ++r) {
    int i0, i1;
    float w0, w1;
    up.locate((long)f * hop + lane * SPL + r, i0, i1, w0, w1);
    acc[r] = fmaf(w0, s0[r >> 1][r & 1], w1 * s1[r >> 1][r & 1]);
  }
  store_frame<SPL>(out + fr * (long)hop, hop, lane, acc);
}

// ---- launchers -----------------------------------------------------------------------------------
Upsampler make_upsampler_pub(int F, int hop);
PhaseCfg make_phase_cfg(double sr, int infer, int has_ip);
int spl_for_hop(int hop);

int launch_combtooth(const float* f0_frames, const float* initial_phase, int B, int F, int hop, double sr, int infer,
                     const double* phase0, float* out, hipStream_t st) {
  const int spl = spl_for_hop(hop);
  if (!spl) return -1;
  const long n_frames = (long)B * F;
  if (n_frames == 0) return 0;
  Upsampler up = make_upsampler_pub(F, hop);
  PhaseCfg cfg = make_phase_cfg(sr, infer, initial_phase != nullptr);
  dim3 grid((unsigned)((n_frames + 3) / 4)), block(256);
  if (spl == 8)
    hipLaunchKernelGGL(k_combtooth<8>, grid, block, 0, st, f0_frames, initial_phase, n_frames, F, hop, up, cfg, phase0, out);
  else if (spl == 16)
    hipLaunchKernelGGL(k_combtooth<16>, grid, block, 0, st, f0_frames, initial_phase, n_frames, F, hop, up, cfg, phase0, out);
  else
    hipLaunchKernelGGL(k_combtooth<32>, grid, block, 0, st, f0_frames, initial_phase, n_frames, F, hop, up, cfg, phase0, out);
  return 0;
}

int launch_sins_bank(const float* f0_frames, const float* initial_phase, const float* c_amp, long ld_amp, int B, int F,
                     int hop, int H, double sr, int infer, const double* phase0, float* out, hipStream_t st) {
  const int spl = spl_for_hop(hop);
  if (!spl) return -1;
  if ((size_t)5 * H * sizeof(float) > 60 * 1024) return -2;
  if ((long)B * F == 0) return 0;
  Upsampler up = make_upsampler_pub(F, hop);
  PhaseCfg cfg = make_phase_cfg(sr, infer, initial_phase != nullptr);
  const int groups = (F + 3) / 4;
  dim3 grid((unsigned)((long)B * groups)), block(256);
  size_t sh = (size_t)5 * H * sizeof(float);
  if (spl == 8)
    hipLaunchKernelGGL(k_sins_bank<8>, grid, block, sh, st, f0_frames, initial_phase, c_amp, ld_amp, F, hop, H, up, cfg, phase0, out);
  else if (spl == 16)
    hipLaunchKernelGGL(k_sins_bank<16>, grid, block, sh, st, f0_frames, initial_phase, c_amp, ld_amp, F, hop, H, up, cfg, phase0, out);
  else
    hipLaunchKernelGGL(k_sins_bank<32>, grid, block, sh, st, f0_frames, initial_phase, c_amp, ld_amp, F, hop, H, up, cfg, phase0, out);
  return 0;
}

}  // namespace ddsp
