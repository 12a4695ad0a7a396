// The per-frame phase rebuild and the combtooth of one frame, shared by exciter.hip (k_combtooth, the sinusoid banks) and
// ir_pfa.hip (k_front_small: exciter + tap syntheses of a streaming-shape step in one launch).
#pragma once
#include "ddsp_common.h"

namespace ddsp {

template <int SPL>
struct FramePhase {
  float x[SPL];      // wrapped phase, cycles in [-0.5, 0.5]
  float f0u[SPL];    // upsampled f0 at the same samples
};

// x[t] and f0[t] for the SPL samples this lane owns in frame fr (all 64 lanes of the wave must call)
// POW2: the caller (launcher) has checked up.shift > 0, so only the shift form of the interpolation is compiled in
template <int SPL, bool POW2 = false>
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
      v = POW2 ? up.at3_pow2(rows, j) : up.at3_in_frame(rows, j, hop);
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

// combtooth = sinc(sr * x / (f0 + 1e-3)) (vocoder.py:839-840) of frame fr, by one wave
template <int SPL, bool POW2>
__device__ __forceinline__ void combtooth_frame(const float* __restrict__ f0_frames, const float* __restrict__ initial_phase,
                                                long fr, int F, int hop, const Upsampler& up, const PhaseCfg& cfg,
                                                const double* __restrict__ phase0, float* __restrict__ out, int lane) {
  const unsigned bu = (unsigned)fr / (unsigned)F;    // 32-bit frame indices (launcher)
  const long b = bu;
  const int f = (int)((unsigned)fr - bu * (unsigned)F);
  const float ip = cfg.has_ip ? initial_phase[b] : 0.0f;
  FramePhase<SPL> ph;
  frame_phase<SPL, POW2>(f0_frames + b * F, f, hop, up, cfg, phase0[fr], ip, lane, ph);
  float v[SPL];
#pragma unroll
  for (int r = 0; r < SPL; ++r) {
    float num = cfg.sr_f * ph.x[r];
    float den = ph.f0u[r] + 1e-3f;
    v[r] = sinc_f32(div_pos(num, den));
  }
  store_frame<SPL>(out + fr * (long)hop, hop, lane, v);
}

}  // namespace ddsp
