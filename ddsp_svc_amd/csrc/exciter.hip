// HOT-2a: the two exciters -- band-limited combtooth (CombSub, ddsp/vocoder.py:839-840) and the
// additive sinusoid bank (Sins, ddsp/vocoder.py:580,585-594).
//
// Both rebuild the wrapped phase x[t] of their frame from the per-frame start value produced by
// k_phase_frame_scan (one wave per frame, lane owns SPL consecutive samples, one wave64 scan), so
// the [B,T] phase tensor of the reference never exists in HBM.  The sinusoid bank keeps the two
// amplitude rows it interpolates between in LDS and never materialises the reference's
// [B,T,32] temporaries.
#include "ddsp_common.h"
#include "frame_phase.h"
#include "kernels.h"
#include "tuning.h"
#include <stdlib.h>

namespace ddsp {

// ------------------------------------------------------------------------------------------------
// combtooth = sinc(sr * x / (f0 + 1e-3)); one wave per frame
// ------------------------------------------------------------------------------------------------
template <int SPL, bool POW2 = false>
__global__ void __launch_bounds__(256) k_combtooth(const float* __restrict__ f0_frames,
                                                   const float* __restrict__ initial_phase, long n_frames, int F,
                                                   int hop, Upsampler up, PhaseCfg cfg,
                                                   const double* __restrict__ phase0, float* __restrict__ out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long fr = (long)blockIdx.x * 4 + wave;
  if (fr >= n_frames) return;
  combtooth_frame<SPL, POW2>(f0_frames, initial_phase, fr, F, hop, up, cfg, phase0, out, lane);
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

// ------------------------------------------------------------------------------------------------
// sinusoid bank, block angle-addition form (hop = 512): one 256-thread workgroup per frame, two consecutive
// samples per thread.  Per sample the harmonics are generated 16 at a time from
//     sin((16 b + j) theta) = sin(16 b theta) cos(j theta) + cos(16 b theta) sin(j theta),   j = 1..16,
// with the table cis(j theta) built once per sample (15 rotations from an accurately evaluated cis(theta)) and the
// block seeds cis(16 b theta) evaluated accurately for every fourth block (exact float32 product split + hardware
// sine/cosine) and rotated in between.  A harmonic then costs two packed multiply-adds for the sine and two for the
// two amplitude rows (both samples at once) instead of a range reduction and a hardware sine per sample.
//
// Numerics: the reference evaluates sin(fl32(k * phase)) -- the float32 rounding of k*phase (up to 3e-5 rad at
// k = 256) is part of its result.  This form evaluates sin(k * phase) for the float32 phase without that rounding;
// on the reference-generated fixtures the two differ by 3.5e-6 (H = 256) / 7e-7 (H = 128) relative RMS of the
// exciter, inside the 1e-5 the tails are held to.  Rotation errors stay below 1e-6 (<= 15 steps for the table,
// <= 3 for the seeds).
// ------------------------------------------------------------------------------------------------
// (cos, sin) of angle a (radians, |a| up to ~1e3) evaluated for the float32 product k * theta WITHOUT rounding it:
// p_hi = fl32(k theta), p_lo = k theta - p_hi (exact, fma); revolutions = p_hi / 2pi (two-constant) + p_lo / 2pi
__device__ __forceinline__ void cis_product(float k, float theta, float& co, float& si) {
  const float inv_hi = 0.15915494f, inv_lo = 6.4206383e-9f;
  const float p_hi = k * theta;
  const float p_lo = fmaf(k, theta, -p_hi);
  const float n = rintf(p_hi * inv_hi);
  float r = fmaf(p_hi, inv_hi, -n);
  r = fmaf(p_hi, inv_lo, r);
  r = fmaf(p_lo, inv_hi, r);
  co = __builtin_amdgcn_cosf(r);
  si = __builtin_amdgcn_sinf(r);
}

// the same for the two samples of a thread at once: the multiplies and fmas packed (the same operations on each half, so the same
// bits as two calls), the rounding and the hardware sine / cosine per half
__device__ __forceinline__ void cis_product2(float k, f32x2 theta, f32x2& co, f32x2& si) {
  const f32x2 inv_hi = {0.15915494f, 0.15915494f}, inv_lo = {6.4206383e-9f, 6.4206383e-9f};
  const f32x2 kk = {k, k};
  const f32x2 p_hi = kk * theta;
  const f32x2 p_lo = __builtin_elementwise_fma(kk, theta, -p_hi);
  const f32x2 t = p_hi * inv_hi;
  const f32x2 n = {rintf(t.x), rintf(t.y)};
  f32x2 r = __builtin_elementwise_fma(p_hi, inv_hi, -n);
  r = __builtin_elementwise_fma(p_hi, inv_lo, r);
  r = __builtin_elementwise_fma(p_lo, inv_hi, r);
  co = f32x2{__builtin_amdgcn_cosf(r.x), __builtin_amdgcn_cosf(r.y)};
  si = f32x2{__builtin_amdgcn_sinf(r.x), __builtin_amdgcn_sinf(r.y)};
}

// (lam.x ad.y + ad.x, lam.y ad.y + ad.x): both halves of the result take A from the low and dA from the high half of `ad`
__device__ __forceinline__ f32x2 lerp_pair(f32x2 lam, f32x2 ad) {
#if defined(__HIP_DEVICE_COMPILE__)
  f32x2 r;
  asm("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,1,0] op_sel_hi:[1,1,0]" : "=v"(r) : "v"(lam), "v"(ad));
  return r;
#else
  return f32x2{fmaf(lam.x, ad.y, ad.x), fmaf(lam.y, ad.y, ad.x)};
#endif
}

#ifdef DDSP_AB_GENERATIONS                       // round 1 / 2's 16-harmonic blocks: A/B builds only (tools/build_variant.sh)
__global__ void __launch_bounds__(256) k_sins_bank2(const float* __restrict__ f0_frames,
                                                    const float* __restrict__ initial_phase,
                                                    const float* __restrict__ c_amp, long ld_amp, int F, int H,
                                                    Upsampler up, PhaseCfg cfg, const double* __restrict__ phase0,
                                                    float* __restrict__ out) {
  constexpr int HOP = 512;
  HIP_DYNAMIC_SHARED(float, amp)                    // [HP][2]: (A[f][k], A[f+1][k]) for k < H, zero up to HP
  __shared__ double wsum[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long fr = blockIdx.x;
  const long b = fr / F;
  const int f = (int)(fr - b * F);
  const int HP = (H + 15) & ~15;
  const float* f0_row = f0_frames + b * F;
  const float nyq = cfg.sr_f / 2.0f;
  const int f1 = f + 1 < F ? f + 1 : F - 1;           // last frame held (core.py:68)
  for (int k = tid; k < HP; k += 256) {
    float a0 = 0.f, a1 = 0.f;
    if (k < H) {
      const float e0 = expf(c_amp[(b * F + f) * ld_amp + k]) / 128.0f;       // vocoder.py:580
      const float e1 = expf(c_amp[(b * F + f1) * ld_amp + k]) / 128.0f;
      const float kk = (float)(k + 1);
      a0 = e0 * ((f0_row[f] * kk < nyq ? 1.0f : 0.0f) + 1e-7f);             // core.py:75-76
      a1 = e1 * ((f0_row[f1] * kk < nyq ? 1.0f : 0.0f) + 1e-7f);
    }
    amp[2 * k] = a0;
    amp[2 * k + 1] = a1 - a0;                       // the frame-to-frame step: upsample(A)[t] = A[f] + lambda (A[f+1] - A[f])
  }
  // wrapped phase of this thread's two samples (vocoder.py:564-572): float64 terms, block-wide exclusive scan
  const Upsampler::Row3 rows = up.load3(f0_row, f);
  const float ip = cfg.has_ip ? initial_phase[b] : 0.0f;
  const long t0 = (long)f * HOP + 2 * tid;
  const double q0 = cfg.term(up.at3_pow2(rows, 2 * tid));       // the launcher has checked up.shift > 0 (hop 512, F hop <= 2^24)
  const double q1 = cfg.term(up.at3_pow2(rows, 2 * tid + 1));
  const double mine = q0 + q1;
  const double excl = wave_excl_scan(mine, lane);
  if (lane == 63) wsum[wave] = excl + mine;
  __syncthreads();                                  // also publishes amp[]
  double base = phase0[fr] + excl;
  for (int w = 0; w < wave; ++w) base += wsum[w];
  const float xa = cfg.wrap(base + q0, ip), xb = cfg.wrap(base + q0 + q1, ip);
  const f32x2 theta = {kTwoPiF * xa, kTwoPiF * xb};                       // vocoder.py:574
  // table cis(j theta), j = 1..16, as (cos_A, cos_B) / (sin_A, sin_B) pairs
  f32x2 tc[16], ts[16];
  {
    float c0, s0, c1, s1;
    cis_product(1.0f, theta.x, c0, s0);
    cis_product(1.0f, theta.y, c1, s1);
    tc[0] = f32x2{c0, c1};
    ts[0] = f32x2{s0, s1};
#pragma unroll
    for (int j = 1; j < 16; ++j) {
      tc[j] = __builtin_elementwise_fma(tc[j - 1], tc[0], -(ts[j - 1] * ts[0]));
      ts[j] = __builtin_elementwise_fma(ts[j - 1], tc[0], tc[j - 1] * ts[0]);
    }
  }
  // interpolation weight of the thread's two samples (core.py:66-70: lambda = j / hop towards frame f + 1; exact in the
  // shift form)
  const f32x2 lam = {(float)(2 * tid) * up.scale, (float)(2 * tid + 1) * up.scale};
  // sum_k sin(k theta) A_k(t) block by block:  sin((16 b + j) theta) = Cb sin(j theta) + Sb cos(j theta), so a block
  // contributes Cb P + Sb Q with P = sum_j sin(j theta) A_j(t), Q = sum_j cos(j theta) A_j(t): three packed
  // multiply-adds per harmonic and sample pair (amplitude interpolation, P, Q)
  f32x2 S = {0.f, 0.f};
  f32x2 Cb = {1.f, 1.f}, Sb = {0.f, 0.f};            // cis(16 b theta)
  const int nblk = HP >> 4;
  for (int blk = 0; blk < nblk; ++blk) {
    if (blk > 0) {
      if ((blk & 3) == 0) {                         // accurate re-seed
        float c0, s0, c1, s1;
        cis_product((float)(16 * blk), theta.x, c0, s0);
        cis_product((float)(16 * blk), theta.y, c1, s1);
        Cb = f32x2{c0, c1};
        Sb = f32x2{s0, s1};
      } else {                                      // rotate by cis(16 theta)
        const f32x2 cn = __builtin_elementwise_fma(Cb, tc[15], -(Sb * ts[15]));
        Sb = __builtin_elementwise_fma(Sb, tc[15], Cb * ts[15]);
        Cb = cn;
      }
    }
    const float4* ap = reinterpret_cast<const float4*>(amp + 32 * blk);   // (A, dA) of two harmonics per 16-byte read
    f32x2 P = {0.f, 0.f}, Q = {0.f, 0.f};
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
      const float4 q = ap[jj];
      // amplitude of the two samples: A + lambda dA with (A, dA) taken from the halves of ONE register pair (the compiler's
      // own form of the second harmonic's interpolation copies dA into a fresh pair first: 8 v_mov per block)
      const f32x2 a = lerp_pair(lam, f32x2{q.x, q.y});
      const f32x2 b = lerp_pair(lam, f32x2{q.z, q.w});
      P = __builtin_elementwise_fma(ts[2 * jj], a, P);
      Q = __builtin_elementwise_fma(tc[2 * jj], a, Q);
      P = __builtin_elementwise_fma(ts[2 * jj + 1], b, P);
      Q = __builtin_elementwise_fma(tc[2 * jj + 1], b, Q);
    }
    S = __builtin_elementwise_fma(Cb, P, S);
    S = __builtin_elementwise_fma(Sb, Q, S);
  }
  const float r[2] = {S.x, S.y};
  float* dst = out + b * (long)F * HOP + t0;
  if ((reinterpret_cast<uintptr_t>(dst) & 7) == 0) *reinterpret_cast<float2*>(dst) = make_float2(r[0], r[1]);
  else { dst[0] = r[0]; dst[1] = r[1]; }
}
#endif  // DDSP_AB_GENERATIONS

// ------------------------------------------------------------------------------------------------
// sinusoid bank, mirrored-pair form (hop = 512; the default): blocks of 17 harmonics AROUND a centre c,
//     a+ sin((c + j) theta) + a- sin((c - j) theta)  =  sin(c theta) cos(j theta) (a+ + a-)  +  cos(c theta) sin(j theta) (a+ - a-),
// so with the frame's sums sigma_j = a(c+j) + a(c-j) and differences delta_j = a(c+j) - a(c-j) staged in LDS (they are
// linear in the amplitudes, hence interpolate between the two frames exactly as the amplitudes do), a PAIR of harmonics
// costs two interpolations and two multiply-adds: 2 packed instructions per harmonic and sample pair instead of the 3 of
// k_sins_bank2 (amplitude interpolation, P, Q), and the table cis(j theta) has 8 entries instead of 16.  Block b has centre
// c = 9 + 17 b and contributes  sin(c theta) (a_c + sum_j cos(j theta) sigma_j) + cos(c theta) sum_j sin(j theta) delta_j.
// k_sins_bank2 was measured at 100 % vector-ALU occupancy (SQ_ACTIVE_INST_VALU = kernel cycles) and power-limited clocks:
// only fewer multiply-adds make this kernel faster.  Harmonics beyond 17 * (H / 17) go into one more, zero-padded block,
// or -- one or two of them -- are evaluated on their own.  Same numerics as k_sins_bank2 (sin(k theta) for the float32
// theta without the reference's rounding of k * theta; accurate seeds every fourth block, rotations in between).
// ------------------------------------------------------------------------------------------------
constexpr int SB3_J = 8;                              // pairs per block
constexpr int SB3_W = 2 * SB3_J + 1;                  // harmonics per block: 17
constexpr int SB3_LD = 4 + 4 * SB3_J;                 // floats per block in LDS: (A_c, dA_c, 0, 0), then (sigma, dsigma, delta, ddelta) x 8

__global__ void __launch_bounds__(256) k_sins_bank3(const float* __restrict__ f0_frames,
                                                    const float* __restrict__ initial_phase,
                                                    const float* __restrict__ c_amp, long ld_amp, int F, int H, int nblk_all,
                                                    int nsingle_all, int skip_masked, Upsampler up, PhaseCfg cfg,
                                                    const double* __restrict__ phase0, float* __restrict__ out) {
  constexpr int HOP = 512;
  HIP_DYNAMIC_SHARED(float, amp)                    // [nblk][SB3_LD], then [nsingle][2]
  __shared__ double wsum[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long fr = blockIdx.x;
  const unsigned bu = (unsigned)blockIdx.x / (unsigned)F;      // 32-bit frame indices (launcher): a 64-bit division is ~160 instructions
  const long b = bu;                                           // of a workgroup that lives for one frame
  const int f = (int)((unsigned)blockIdx.x - bu * (unsigned)F);
  const float* f0_row = f0_frames + b * F;
  const float nyq = cfg.sr_f / 2.0f;
  const int f1 = f + 1 < F ? f + 1 : F - 1;           // last frame held (core.py:68)
  const float* row0 = c_amp + (b * F + f) * ld_amp;
  const float* row1 = c_amp + (b * F + f1) * ld_amp;
  const float fa = f0_row[f], fb = f0_row[f1];
  // Harmonics at or above Nyquist in BOTH frames of this hop keep 1e-7 of their amplitude (core.py:73-77: the mask is
  // (f0 k < fmax) + 1e-7).  A trailing block of 17 whose LOWEST harmonic is masked in both frames is all such harmonics
  // (f0 k grows with k), and so is every block behind it: they are left out -- at f0 = 400 Hz that is 200 of 256 harmonics.
  // What that changes is 1e-7 of the masked harmonics' amplitudes (measured against the oracle in tests/test_parity.py
  // test_sinusoid_bank; knob SINS_NOSKIP = 1 keeps them).  Workgroup-uniform; written so that a NaN f0 skips nothing.
  int nblk = nblk_all, nsingle = nsingle_all;
  if (skip_masked) {
    auto masked = [&](int k) { const float kk = (float)k; return fa * kk >= nyq && fb * kk >= nyq; };
    while (nblk > 0 && masked(1 + SB3_W * (nblk - 1))) --nblk;
    if (nblk < nblk_all) nsingle = 0;
    while (nsingle > 0 && masked(SB3_W * nblk_all + nsingle)) --nsingle;
  }
  // activated, masked amplitude of harmonic k (1-based) in the two frames; zero outside 1..H
  auto amp2 = [&](int k, float& a0, float& a1) {
    a0 = 0.f;
    a1 = 0.f;
    if (k >= 1 && k <= H) {
      const float kk = (float)k;
      a0 = (exp_hw(row0[k - 1]) / 128.0f) * ((fa * kk < nyq ? 1.0f : 0.0f) + 1e-7f);   // vocoder.py:580, core.py:75-76
      a1 = (exp_hw(row1[k - 1]) / 128.0f) * ((fb * kk < nyq ? 1.0f : 0.0f) + 1e-7f);
    }
  };
  // one staging item per (block, j), j = 0 the centre, j = 1..8 a mirrored pair; the frame-to-frame steps are kept beside
  // the values: upsample(A)[t] = A[f] + lambda (A[f+1] - A[f]) (core.py:66-70)
  for (int i = tid; i < nblk * (SB3_J + 1) + nsingle; i += 256) {
    if (i < nblk * (SB3_J + 1)) {
      const int blk = i / (SB3_J + 1), j = i - blk * (SB3_J + 1);
      const int c = SB3_J + 1 + SB3_W * blk;
      float* dst = amp + blk * SB3_LD;
      float p0, p1;
      amp2(c + j, p0, p1);
      if (j == 0) {
        dst[0] = p0; dst[1] = p1 - p0; dst[2] = 0.f; dst[3] = 0.f;
      } else {
        float m0, m1;
        amp2(c - j, m0, m1);
        const float s0 = p0 + m0, s1 = p1 + m1, d0 = p0 - m0, d1 = p1 - m1;
        dst[4 * j] = s0; dst[4 * j + 1] = s1 - s0; dst[4 * j + 2] = d0; dst[4 * j + 3] = d1 - d0;
      }
    } else {
      const int q = i - nblk * (SB3_J + 1);
      float p0, p1;
      amp2(SB3_W * nblk_all + 1 + q, p0, p1);                   // (singles survive only when no block was dropped: nblk = nblk_all)
      amp[nblk * SB3_LD + 2 * q] = p0;
      amp[nblk * SB3_LD + 2 * q + 1] = p1 - p0;
    }
  }
  // wrapped phase of this thread's two samples (vocoder.py:564-572): float64 terms, block-wide exclusive scan
  const Upsampler::Row3 rows = up.load3(f0_row, f);
  const float ip = cfg.has_ip ? initial_phase[b] : 0.0f;
  const long t0 = (long)f * HOP + 2 * tid;
  const double q0 = cfg.term(up.at3_pow2(rows, 2 * tid));       // the launcher has checked up.shift > 0 (hop 512, F hop <= 2^24)
  const double q1 = cfg.term(up.at3_pow2(rows, 2 * tid + 1));
  const double mine = q0 + q1;
  const double excl = wave_excl_scan(mine, lane);
  if (lane == 63) wsum[wave] = excl + mine;
  __syncthreads();                                  // also publishes amp[]
  double base = phase0[fr] + excl;
  for (int w = 0; w < wave; ++w) base += wsum[w];
  const float xa = cfg.wrap(base + q0, ip), xb = cfg.wrap(base + q0 + q1, ip);
  const f32x2 theta = {kTwoPiF * xa, kTwoPiF * xb};                       // vocoder.py:574
  // table cis(j theta), j = 1..8, as (cos_A, cos_B) / (sin_A, sin_B) pairs; the block-to-block rotation cis(17 theta)
  f32x2 tc[SB3_J], ts[SB3_J];
  f32x2 rc, rs;
  {
    cis_product2(1.0f, theta, tc[0], ts[0]);
#pragma unroll
    for (int j = 1; j < SB3_J; ++j) {
      tc[j] = __builtin_elementwise_fma(tc[j - 1], tc[0], -(ts[j - 1] * ts[0]));
      ts[j] = __builtin_elementwise_fma(ts[j - 1], tc[0], tc[j - 1] * ts[0]);
    }
    const f32x2 c16 = __builtin_elementwise_fma(tc[7], tc[7], -(ts[7] * ts[7]));
    const f32x2 s16 = (ts[7] * tc[7]) * f32x2{2.f, 2.f};
    rc = __builtin_elementwise_fma(c16, tc[0], -(s16 * ts[0]));
    rs = __builtin_elementwise_fma(s16, tc[0], c16 * ts[0]);
  }
  // interpolation weight of the thread's two samples (lambda = j / hop towards frame f + 1; exact in the shift form)
  const f32x2 lam = {(float)(2 * tid) * up.scale, (float)(2 * tid + 1) * up.scale};
  f32x2 S = {0.f, 0.f};
  f32x2 Cb = {1.f, 1.f}, Sb = {0.f, 0.f};            // cis(c theta) of the block
  for (int blk = 0; blk < nblk; ++blk) {
    if ((blk & 3) == 0) {                           // accurate seed
      cis_product2((float)(SB3_J + 1 + SB3_W * blk), theta, Cb, Sb);
    } else {                                        // rotate by cis(17 theta)
      const f32x2 cn = __builtin_elementwise_fma(Cb, rc, -(Sb * rs));
      Sb = __builtin_elementwise_fma(Sb, rc, Cb * rs);
      Cb = cn;
    }
    const float4* ap = reinterpret_cast<const float4*>(amp + blk * SB3_LD);
    const float4 ctr = ap[0];
    f32x2 Q = lerp_pair(lam, f32x2{ctr.x, ctr.y});  // the centre harmonic: cos(0) a_c
    f32x2 P = {0.f, 0.f};
#pragma unroll
    for (int j = 1; j <= SB3_J; ++j) {
      const float4 q = ap[j];
      const f32x2 sg = lerp_pair(lam, f32x2{q.x, q.y});
      const f32x2 dl = lerp_pair(lam, f32x2{q.z, q.w});
      Q = __builtin_elementwise_fma(tc[j - 1], sg, Q);
      P = __builtin_elementwise_fma(ts[j - 1], dl, P);
    }
    S = __builtin_elementwise_fma(Sb, Q, S);
    S = __builtin_elementwise_fma(Cb, P, S);
  }
  for (int q = 0; q < nsingle; ++q) {               // one or two harmonics beyond the last whole block
    const float k = (float)(SB3_W * nblk_all + 1 + q);
    f32x2 ck, sk;
    cis_product2(k, theta, ck, sk);
    const float2 ad = *reinterpret_cast<const float2*>(amp + nblk * SB3_LD + 2 * q);
    S = __builtin_elementwise_fma(sk, lerp_pair(lam, f32x2{ad.x, ad.y}), S);
  }
  const float r[2] = {S.x, S.y};
  float* dst = out + b * (long)F * HOP + t0;
  if ((reinterpret_cast<uintptr_t>(dst) & 7) == 0) *reinterpret_cast<float2*>(dst) = make_float2(r[0], r[1]);
  else { dst[0] = r[0]; dst[1] = r[1]; }
}

// ------------------------------------------------------------------------------------------------
// Adjoint of the sinusoid bank w.r.t. the amplitudes (hop = 512): for the frame of the workgroup,
//     R0[k] = sum_t g[t] w0[t] sin(k theta_t)     (goes to amplitude row f)
//     R1[k] = sum_t g[t] w1[t] sin(k theta_t)     (goes to amplitude row min(f+1, F-1))
// The sines are generated exactly as in k_sins_bank2 (same table, same seeds); per block of 16 harmonics every
// thread's 32 products (16 harmonics x two rows, its two samples already added) go through an LDS tile
// [32][256 (+1 pad)] and are summed column-block-wise: thread (v = tid & 31, chunk = tid >> 5) adds 32 entries, the
// two chunks of a wave meet by one cross-lane exchange, the four waves through a second small LDS array.
// k_sins_bank_bwd_combine then forms dc[f][k] = A[f][k] (R0[f][k] + R1[f-1][k] (+ R1[F-1][k] on the last row)).
// ------------------------------------------------------------------------------------------------
#ifdef DDSP_AB_GENERATIONS                       // the LDS-tile adjoint the matrix-pipe form replaced: A/B builds only
constexpr int SB_TILE_LD = 257;

__global__ void __launch_bounds__(256) k_sins_bank2_bwd(const float* __restrict__ f0_frames,
                                                        const float* __restrict__ initial_phase,
                                                        const float* __restrict__ grad_out, int F, int H, Upsampler up,
                                                        PhaseCfg cfg, const double* __restrict__ phase0,
                                                        float* __restrict__ partial /* [B*F][HP][2] */) {
  constexpr int HOP = 512;
  __shared__ float tile[32 * SB_TILE_LD];
  __shared__ float part[4][32];
  __shared__ double wsum[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long fr = blockIdx.x;
  const long b = fr / F;
  const int f = (int)(fr - b * F);
  const int HP = (H + 15) & ~15;
  const float* f0_row = f0_frames + b * F;
  const Upsampler::Row3 rows = up.load3(f0_row, f);
  const float ip = cfg.has_ip ? initial_phase[b] : 0.0f;
  const long t0 = (long)f * HOP + 2 * tid;
  const double q0 = cfg.term(up.at3_in_frame(rows, 2 * tid, HOP));
  const double q1 = cfg.term(up.at3_in_frame(rows, 2 * tid + 1, HOP));
  const double mine = q0 + q1;
  const double excl = wave_excl_scan(mine, lane);
  if (lane == 63) wsum[wave] = excl + mine;
  __syncthreads();
  double base = phase0[fr] + excl;
  for (int w = 0; w < wave; ++w) base += wsum[w];
  const float xa = cfg.wrap(base + q0, ip), xb = cfg.wrap(base + q0 + q1, ip);
  const f32x2 theta = {kTwoPiF * xa, kTwoPiF * xb};
  f32x2 tc[16], ts[16];
  {
    float c0, s0, c1, s1;
    cis_product(1.0f, theta.x, c0, s0);
    cis_product(1.0f, theta.y, c1, s1);
    tc[0] = f32x2{c0, c1};
    ts[0] = f32x2{s0, s1};
#pragma unroll
    for (int j = 1; j < 16; ++j) {
      tc[j] = __builtin_elementwise_fma(tc[j - 1], tc[0], -(ts[j - 1] * ts[0]));
      ts[j] = __builtin_elementwise_fma(ts[j - 1], tc[0], tc[j - 1] * ts[0]);
    }
  }
  // cotangent of the two samples times the interpolation weights of the two amplitude rows (core.py:66-70)
  f32x2 gw0, gw1;
  {
    const float* gp = grad_out + b * (long)F * HOP + t0;
    float w0[2], w1[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      int i0, i1;
      up.locate(t0 + q, i0, i1, w0[q], w1[q]);
    }
    gw0 = f32x2{gp[0] * w0[0], gp[1] * w0[1]};
    gw1 = f32x2{gp[0] * w1[0], gp[1] * w1[1]};
  }
  f32x2 Cb = {1.f, 1.f}, Sb = {0.f, 0.f};
  const int nblk = HP >> 4;
  const int v = tid & 31, chunk = tid >> 5;
  for (int blk = 0; blk < nblk; ++blk) {
    if (blk > 0) {
      if ((blk & 3) == 0) {
        float c0, s0, c1, s1;
        cis_product((float)(16 * blk), theta.x, c0, s0);
        cis_product((float)(16 * blk), theta.y, c1, s1);
        Cb = f32x2{c0, c1};
        Sb = f32x2{s0, s1};
      } else {
        const f32x2 cn = __builtin_elementwise_fma(Cb, tc[15], -(Sb * ts[15]));
        Sb = __builtin_elementwise_fma(Sb, tc[15], Cb * ts[15]);
        Cb = cn;
      }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const f32x2 sv = __builtin_elementwise_fma(Cb, ts[j], Sb * tc[j]);
      const f32x2 p0 = sv * gw0, p1 = sv * gw1;
      tile[(2 * j) * SB_TILE_LD + tid] = p0.x + p0.y;
      tile[(2 * j + 1) * SB_TILE_LD + tid] = p1.x + p1.y;
    }
    __syncthreads();
    float acc = 0.f;
    const float* col = tile + v * SB_TILE_LD + chunk * 32;
#pragma unroll
    for (int i = 0; i < 32; ++i) acc += col[i];
    acc += __shfl_xor(acc, 32);                                  // the wave's two chunks
    if (lane < 32) part[wave][v] = acc;
    __syncthreads();
    if (tid < 32) {
      const float tot = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
      partial[(fr * HP + 16 * blk + (tid >> 1)) * 2 + (tid & 1)] = tot;
    }
    // the next block's tile writes come after this barrier pair; part[] is rewritten only after the next first barrier
  }
}
#endif  // DDSP_AB_GENERATIONS

// ------------------------------------------------------------------------------------------------
// The same adjoint on the matrix pipe (hop = 512; the default).  R_r[k] = sum_t g[t] w_r[t] sin(k theta_t) is a reduction
// over the 512 samples of the frame -- in k_sins_bank2_bwd every thread's products went through an LDS tile and two
// barriers per 16 harmonics (0.79 ms against 0.22 ms forward).  With blocks of 33 harmonics around a centre c,
//     sin((c +- j) theta) = sin(c theta) cos(j theta) +- cos(c theta) sin(j theta),   j = 1..16,
// the sums are two small matrix products per frame,
//     E[j][n] = sum_t cos(j theta_t) alpha_n[t],   O[j][n] = sum_t sin(j theta_t) beta_n[t],
//     alpha_n = g w_r sin(c_b theta), beta_n = g w_r cos(c_b theta),   column n = (block b = n / 2, amplitude row r = n % 2),
// 16 x 512 times 512 x 16 each (8 blocks x 2 rows cover 264 harmonics), and R_r[c +- j] = E[j][n] +- O[j][n],
// R_r[c] = sum_t alpha_n[t].  One wave per frame: v_mfma_f32_16x16x4_f32 (exact float32 multiply-adds) takes four
// samples per instruction and does the reduction over t in its accumulators -- no tables, no LDS tile, no barriers in
// the loop; a lane evaluates cis((i + 1) theta) and cis(c theta) of ITS sample directly (exact float32 product split +
// hardware sine / cosine, as the forward kernel's seeds).
// ------------------------------------------------------------------------------------------------
constexpr int SBM_J = 16;                             // pairs per block
constexpr int SBM_W = 2 * SBM_J + 1;                  // harmonics per block: 33
constexpr int SBM_G = 8;                              // blocks per pass of the sample loop (16 columns)

__global__ void __launch_bounds__(64) k_sins_bank_bwd_mfma(const float* __restrict__ f0_frames,
                                                           const float* __restrict__ initial_phase,
                                                           const float* __restrict__ grad_out, int F, int H, int HP,
                                                           Upsampler up, PhaseCfg cfg, const double* __restrict__ phase0,
                                                           float* __restrict__ partial /* [B*F][HP][2] */) {
  constexpr int HOP = 512, SPL = 8;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  __shared__ float s_theta[HOP], s_gw[2][HOP];
  const int lane = threadIdx.x;
  const long fr = blockIdx.x;
  const unsigned bu = (unsigned)blockIdx.x / (unsigned)F;      // 32-bit frame indices (launcher)
  const long b = bu;
  const int f = (int)((unsigned)blockIdx.x - bu * (unsigned)F);
  const float* f0_row = f0_frames + b * F;
  {
    // wrapped phase (vocoder.py:564-574) and weighted cotangent of the lane's eight consecutive samples
    const Upsampler::Row3 rows = up.load3(f0_row, f);
    const float ip = cfg.has_ip ? initial_phase[b] : 0.0f;
    double pre[SPL];
    double acc = 0.0;
#pragma unroll
    for (int r = 0; r < SPL; ++r) {
      acc += cfg.term(up.at3_pow2(rows, lane * SPL + r));       // the launcher has checked up.shift > 0
      pre[r] = acc;
    }
    const double base = phase0[fr] + wave_excl_scan(acc, lane);
    const float* gp = grad_out + b * (long)F * HOP + (long)f * HOP + lane * SPL;
#pragma unroll
    for (int r = 0; r < SPL; ++r) {
      const int j = lane * SPL + r;
      const float lam = (float)j * up.scale;                    // interpolation weight towards frame f + 1 (core.py:66-70)
      const float gv = gp[r];
      s_theta[j] = kTwoPiF * cfg.wrap(base + pre[r], ip);
      s_gw[0][j] = gv * (1.0f - lam);
      s_gw[1][j] = gv * lam;
    }
  }
  __syncthreads();
  const int k4 = lane >> 4, i = lane & 15;
  const float jf = (float)(i + 1);                              // this lane's row of the cosine / sine matrices
  const int r = i & 1;
  const int ngroups = (H + SBM_W * SBM_G - 1) / (SBM_W * SBM_G);
  for (int grp = 0; grp < ngroups; ++grp) {
    const int c = SBM_J + 1 + SBM_W * (SBM_G * grp + (i >> 1));  // centre harmonic of this lane's column
    const float cf = (float)c;
    f32x4 accE = {0.f, 0.f, 0.f, 0.f}, accO = {0.f, 0.f, 0.f, 0.f};
    float centre = 0.f;
#pragma unroll 2
    for (int s = 0; s < HOP / 4; ++s) {
      const int t = 4 * s + k4;
      const float th = s_theta[t], w = s_gw[r][t];
      float cj, sj, cc, sc;
      cis_product(jf, th, cj, sj);
      cis_product(cf, th, cc, sc);
      const float alpha = w * sc, beta = w * cc;
      accE = __builtin_amdgcn_mfma_f32_16x16x4f32(cj, alpha, accE, 0, 0, 0);
      accO = __builtin_amdgcn_mfma_f32_16x16x4f32(sj, beta, accO, 0, 0, 0);
      centre += alpha;
    }
    centre += __shfl_xor(centre, 16);
    centre += __shfl_xor(centre, 32);
    // lane holds E / O [j = 4 k4 + e + 1][n = i]
    float* dst = partial + fr * (long)HP * 2 + r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int j = 4 * k4 + e + 1;
      const int kp = c + j, km = c - j;
      if (kp <= H) dst[2 * (kp - 1)] = accE[e] + accO[e];
      if (km <= H) dst[2 * (km - 1)] = accE[e] - accO[e];      // km >= 1 always
    }
    if (k4 == 0 && c <= H) dst[2 * (c - 1)] = centre;
  }
}

__global__ void __launch_bounds__(256) k_sins_bank_bwd_combine(const float* __restrict__ f0_frames,
                                                               const float* __restrict__ c_amp, long ld_amp,
                                                               const float* __restrict__ partial, int F, int H, int HP,
                                                               float nyq, float* __restrict__ d_c) {
  // workgroup = (frame, 256 harmonics): no per-thread index arithmetic (as a flat index over [B F H] every thread paid two
  // 64-bit divisions and two remainders, ~600 instructions for five memory accesses: 34 us per launch)
  const int k = (int)blockIdx.y * 256 + (int)threadIdx.x;
  if (k >= H) return;
  const long fr = blockIdx.x;
  const unsigned bu = (unsigned)blockIdx.x / (unsigned)F;      // 32-bit frame indices (launcher)
  const int f = (int)((unsigned)blockIdx.x - bu * (unsigned)F);
  const long b = bu;
  const long i = fr * H + k;
  float dA = partial[(fr * HP + k) * 2];
  if (f > 0) dA += partial[((fr - 1) * HP + k) * 2 + 1];
  if (f == F - 1) dA += partial[(fr * HP + k) * 2 + 1];         // the held last frame (core.py:68)
  const float a = expf(c_amp[fr * ld_amp + k]) / 128.0f;
  const float p = f0_frames[b * F + f] * (float)(k + 1);
  d_c[i] = dA * (a * ((p < nyq ? 1.0f : 0.0f) + 1e-7f));
}

// ------------------------------------------------------------------------------------------------
// The same adjoint at every other hop (the forward there is k_sins_bank): one wave per frame, each lane its SPL samples, the
// sine of every harmonic evaluated as the forward kernel does -- sin(fl32(k phase)), the reference's own rounding
// (vocoder.py:590) -- and the frame's two sums R0[k], R1[k] formed by a wave reduction per harmonic.  A fallback's speed
// (H sines per sample, as the forward one); k_sins_bank_bwd_combine finishes it as it does for the hop-512 forms.
// ------------------------------------------------------------------------------------------------
template <int SPL>
__global__ void __launch_bounds__(256) k_sins_bank_bwd_any(const float* __restrict__ f0_frames,
                                                           const float* __restrict__ initial_phase,
                                                           const float* __restrict__ grad_out, int F, int hop, int H, int HP,
                                                           Upsampler up, PhaseCfg cfg, const double* __restrict__ phase0,
                                                           float* __restrict__ partial /* [B*F][HP][2] */) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int groups = (F + 3) / 4;
  const long b = blockIdx.x / groups;
  const int f = (int)(blockIdx.x % groups) * 4 + wave;
  if (f >= F) return;                               // wave-uniform; the kernel has no barrier
  const long fr = b * F + f;
  const float ip = cfg.has_ip ? initial_phase[b] : 0.0f;
  FramePhase<SPL> ph;
  frame_phase<SPL>(f0_frames + b * F, f, hop, up, cfg, phase0[fr], ip, lane, ph);
  float phase[SPL], gw0[SPL], gw1[SPL];
#pragma unroll
  for (int r = 0; r < SPL; ++r) {
    const int j = lane * SPL + r;
    phase[r] = kTwoPiF * ph.x[r];                   // vocoder.py:574
    gw0[r] = gw1[r] = 0.f;
    if (j < hop) {
      const long t = (long)f * hop + j;
      int i0, i1;
      float w0, w1;
      up.locate(t, i0, i1, w0, w1);
      const float g = grad_out[b * (long)F * hop + t];
      gw0[r] = g * w0;
      gw1[r] = g * w1;
    }
  }
  for (int k = 0; k < H; ++k) {
    const float kf = (float)(k + 1);
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int r = 0; r < SPL; ++r) {
      const float sv = sin_turns(phase[r] * kf);
      a0 = fmaf(sv, gw0[r], a0);
      a1 = fmaf(sv, gw1[r], a1);
    }
    const float r0 = wave_sum_dpp(a0), r1 = wave_sum_dpp(a1);
    if (lane == 0) {
      partial[(fr * HP + k) * 2] = r0;
      partial[(fr * HP + k) * 2 + 1] = r1;
    }
  }
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
  if (n_frames >= (1L << 31) - 64) return -1;                     // the kernel indexes frames in 32 bits
  Upsampler up = make_upsampler_pub(F, hop);
  PhaseCfg cfg = make_phase_cfg(sr, infer, initial_phase != nullptr);
  dim3 grid((unsigned)((n_frames + 3) / 4)), block(256);
  if (spl == 8 && up.shift > 0)
    hipLaunchKernelGGL((k_combtooth<8, true>), grid, block, 0, st, f0_frames, initial_phase, n_frames, F, hop, up, cfg, phase0, out);
  else if (spl == 8)
    hipLaunchKernelGGL(k_combtooth<8>, grid, block, 0, st, f0_frames, initial_phase, n_frames, F, hop, up, cfg, phase0, out);
  else if (spl == 16)
    hipLaunchKernelGGL(k_combtooth<16>, grid, block, 0, st, f0_frames, initial_phase, n_frames, F, hop, up, cfg, phase0, out);
  else
    hipLaunchKernelGGL(k_combtooth<32>, grid, block, 0, st, f0_frames, initial_phase, n_frames, F, hop, up, cfg, phase0, out);
  return 0;
}

int make_exciter_job(const float* f0_frames, const float* initial_phase, int B, int F, int hop, double sr, int infer,
                     const double* phase0, float* out, ExciterJob* job) {
  const Upsampler up = make_upsampler_pub(F, hop);
  if (hop != 512 || up.shift <= 0 || (long)B * F >= (1L << 31) - 64) return -1;
  *job = ExciterJob{f0_frames, initial_phase, (long)B * F, F, hop, up, make_phase_cfg(sr, infer, initial_phase != nullptr), phase0, out};
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
  if (hop == 512 && up.shift > 0 && (long)B * F <= 0x7fffffffL && knob(KNOB_SINS_V1) != 1) {
    // block angle-addition forms: one workgroup per frame
#ifdef DDSP_AB_GENERATIONS
    if (knob(KNOB_SINS_V1) == 2) {                  // the 16-harmonic blocks of round 1 / 2 (same-box A/B builds)
      const size_t sh2 = (size_t)2 * ((H + 15) & ~15) * sizeof(float);
      hipLaunchKernelGGL(k_sins_bank2, dim3((unsigned)((long)B * F)), dim3(256), sh2, st, f0_frames, initial_phase, c_amp,
                         ld_amp, F, H, up, cfg, phase0, out);
      return 0;
    }
#endif
    // mirrored pairs around a centre, 17 harmonics per block; a remainder of one or two harmonics is evaluated on its own
    const int rem = H % SB3_W;
    const int nblk = H / SB3_W + (rem > 2 ? 1 : 0), nsingle = rem > 2 ? 0 : rem;
    const size_t sh3 = ((size_t)nblk * SB3_LD + 2 * (size_t)nsingle + 4) * sizeof(float);
    hipLaunchKernelGGL(k_sins_bank3, dim3((unsigned)((long)B * F)), dim3(256), sh3, st, f0_frames, initial_phase, c_amp,
                       ld_amp, F, H, nblk, nsingle, knob(KNOB_SINS_NOSKIP) == 1 ? 0 : 1, up, cfg, phase0, out);
    return 0;
  }
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

size_t sins_bank_bwd_scratch_floats(int B, int F, int H) { return (size_t)B * F * ((H + 15) & ~15) * 2; }

int launch_sins_bank_bwd(const float* f0_frames, const float* initial_phase, const float* c_amp, long ld_amp,
                         const float* grad_out, int B, int F, int hop, int H, double sr, int infer, const double* phase0,
                         float* scratch, float* d_c, hipStream_t st) {
  const int spl = spl_for_hop(hop);
  if (!spl || (long)B * F > 0x7fffffffL) return -1;
  if ((long)B * F == 0) return 0;
  Upsampler up = make_upsampler_pub(F, hop);
  PhaseCfg cfg = make_phase_cfg(sr, infer, initial_phase != nullptr);
  const int HP = (H + 15) & ~15;
  // (hop 512 with F hop > 2^24: make_upsampler has no shift form there, the forward takes the generic k_sins_bank with its
  //  float-rounded interpolation, and so must its adjoint -- the matrix-pipe kernel hard-codes the shift form)
  if (hop != 512 || up.shift <= 0) {                            // every other hop: one wave per frame, direct sines
    const dim3 grid((unsigned)((long)B * ((F + 3) / 4))), block(256);
    if (spl == 8)
      hipLaunchKernelGGL(k_sins_bank_bwd_any<8>, grid, block, 0, st, f0_frames, initial_phase, grad_out, F, hop, H, HP, up, cfg, phase0, scratch);
    else if (spl == 16)
      hipLaunchKernelGGL(k_sins_bank_bwd_any<16>, grid, block, 0, st, f0_frames, initial_phase, grad_out, F, hop, H, HP, up, cfg, phase0, scratch);
    else
      hipLaunchKernelGGL(k_sins_bank_bwd_any<32>, grid, block, 0, st, f0_frames, initial_phase, grad_out, F, hop, H, HP, up, cfg, phase0, scratch);
  }
#ifdef DDSP_AB_GENERATIONS
  else if (knob(KNOB_SINS_V1) != 0)
    hipLaunchKernelGGL(k_sins_bank2_bwd, dim3((unsigned)((long)B * F)), dim3(256), 0, st, f0_frames, initial_phase, grad_out, F,
                       H, up, cfg, phase0, scratch);
#endif
  else                                                          // hop 512 (a power of two): the matrix-pipe form, one wave per frame
    hipLaunchKernelGGL(k_sins_bank_bwd_mfma, dim3((unsigned)((long)B * F)), dim3(64), 0, st, f0_frames, initial_phase, grad_out,
                       F, H, HP, up, cfg, phase0, scratch);
  hipLaunchKernelGGL(k_sins_bank_bwd_combine, dim3((unsigned)((long)B * F), (unsigned)((H + 255) / 256)), dim3(256), 0, st, f0_frames,
                     c_amp, ld_amp, scratch, F, H, HP, (float)sr / 2.0f, d_c);
  return 0;
}

}  // namespace ddsp
