// Shared device helpers for the DDSP hot-path kernels (gfx950 / CDNA4, wave64).
// Compiled with -ffp-contract=off: every fused multiply-add below is written explicitly, because a
// few expressions must round exactly like the reference's CPU ATen kernels do (see aten_upsample_at).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DDSP_WAVE 64

namespace ddsp {

constexpr float kPiF = 3.14159265358979323846f;       // fl32(pi)  = 0x40490FDB
constexpr float kTwoPiF = 6.28318530717958647692f;    // fl32(2pi) = 0x40C90FDB
constexpr double kPiD = 3.14159265358979323846;

// What ddsp/core.py:66-70 evaluates for output position t of a control row of F frames:
// F.interpolate(cat(sig, last), size=F*hop+1, mode='linear', align_corners=True)[..., :-1].
// ATen: scale = float(F)/float(F*hop); src = scale*float(t); i0 = min(floor(src), F);
// l1 = src-i0; l0 = 1-l1; i1 = i0 + (i0 < F); out = fma(l0, in[i0], fl32(l1*in[i1]))
// (the last form probed bit-exact against torch 2.10 CPU).  Row index F is the held last frame.
struct Upsampler {
  float scale;
  int F;
  // > 0 when hop = 2^shift and F hop <= 2^24: scale = 2^-shift, (float)t and scale * t are exact, so the float
  // recipe below reduces to integer shifts -- bit-identical, about a third of the instructions
  int shift;
  __device__ __forceinline__ void locate(long t, int& i0, int& i1, float& l0, float& l1) const {
    if (shift > 0) {
      const int ti = (int)t;
      const int k = ti >> shift;
      l1 = (float)(ti - (k << shift)) * scale;
      l0 = 1.0f - l1;
      i0 = k < F - 1 ? k : F - 1;
      i1 = k + 1 < F - 1 ? k + 1 : F - 1;
      return;
    }
    float src = scale * (float)t;
    int k = (int)src;
    if (k > F) k = F;
    l1 = src - (float)k;
    l1 = l1 < 0.f ? 0.f : (l1 > 1.f ? 1.f : l1);
    l0 = 1.0f - l1;
    int k1 = k + (k < F ? 1 : 0);
    i0 = k < F - 1 ? k : F - 1;
    i1 = k1 < F - 1 ? k1 : F - 1;
  }
  // The same for the samples of ONE frame f of a stride-1 row: the three rows a sample of that frame can touch
  // (f-1 only through a rounding of scale*t at a frame boundary) are fetched once per frame and selected by index,
  // instead of two dependent loads per sample.
  struct Row3 { float a, b, c; int f; };
  __device__ __forceinline__ Row3 load3(const float* __restrict__ row, int f) const {
    Row3 r;
    r.f = f;
    r.a = row[f > 0 ? f - 1 : 0];
    r.b = row[f];
    r.c = row[f + 1 < F ? f + 1 : F - 1];
    return r;
  }
  __device__ __forceinline__ float at3(const Row3& r, long t) const {
    int i0, i1;
    float l0, l1;
    locate(t, i0, i1, l0, l1);
    const float a = i0 < r.f ? r.a : (i0 == r.f ? r.b : r.c);
    const float b = i1 < r.f ? r.a : (i1 == r.f ? r.b : r.c);
    return fmaf(l0, a, l1 * b);
  }
  // sample j of frame r.f (t = r.f * hop + j, 0 <= j < hop): in the shift form the frame index of every such sample is
  // r.f itself, so the two rows are fixed and only the weight varies
  __device__ __forceinline__ float at3_in_frame(const Row3& r, int j, int hop) const {
    if (shift > 0) {
      const float l1 = (float)j * scale;
      return fmaf(1.0f - l1, r.b, l1 * r.c);
    }
    return at3(r, (long)r.f * hop + j);
  }
  // the same when the caller knows at compile time that the shift form applies (hop a power of two, F hop <= 2^24): the
  // general path is not even compiled in -- it is most of the code of the small phase kernels, and code that is not there
  // is not fetched ("instruction cache")
  __device__ __forceinline__ float at3_pow2(const Row3& r, int j) const {
    const float l1 = (float)j * scale;
    return fmaf(1.0f - l1, r.b, l1 * r.c);
  }
  __device__ __forceinline__ float at(const float* __restrict__ row, long stride, long t) const {
    int i0, i1;
    float l0, l1;
    locate(t, i0, i1, l0, l1);
    float a = row[(long)i0 * stride];
    float b = row[(long)i1 * stride];
    return fmaf(l0, a, l1 * b);
  }
};

__device__ __forceinline__ float div_pos(float a, float b);   // a / b, b > 0 (below)

// Phase bookkeeping shared by the scan kernels and every consumer that re-derives x[t].
// infer != 0: terms and running sum in float64 (vocoder.py:566); infer == 0: float32 terms, float64
// running sum rounded to float32 per output as ATen's CPU cumsum does, then float32 wrap (vocoder.py:568).
struct PhaseCfg {
  double sr_d;
  double rsr_d;            // RN(1 / sr)
  float sr_f;
  int infer;
  int has_ip;
  // f0.double() / sr as a product with the correctly rounded reciprocal plus one fma-residual correction
  // (Markstein): q0 = a r, e = a - q0 sr (exact in fma), q = q0 + e r -- the correctly rounded quotient (checked
  // exhaustively against IEEE division over 1.5 M float32 f0 values for the common sampling rates) at 3 float64
  // operations instead of the ~15 of the hardware division sequence.
  __device__ __forceinline__ double term(float f0u) const {
    if (!infer) return (double)div_pos(f0u, sr_f);               // float32 terms (vocoder.py:568); f0u >= 0, sr_f a normal positive
    const double a = (double)f0u;
    const double q0 = a * rsr_d;
    const double e = fma(-q0, sr_d, a);
    return fma(e, rsr_d, q0);
  }
  // running (unwrapped) sum P in cycles -> wrapped float32 x, vocoder.py:569-572
  __device__ __forceinline__ float wrap(double P, float ip) const {
    if (infer) {
      double x = P;
      if (has_ip) x = x + ((double)ip / 2.0) / kPiD;
      x = x - rint(x);
      return (float)x;
    }
    float x = (float)P;
    if (has_ip) x = x + (ip / 2.0f) / kPiF;
    return x - rintf(x);
  }
};

// A float32 array seen through a buffer descriptor (V#, 128-bit, in SGPRs): the address is base + byte offset, and the
// hardware drops any access whose offset is not below the byte count -- a load returns 0, a store does nothing.  Edge
// handling (positions before / after a row, an utterance, the last block) then costs no compares, selects or exec-mask
// branches, and the per-lane address is one 32-bit VGPR plus an instruction immediate.  Rules kept by the callers: the
// descriptor is built from wave-uniform values only, and a byte offset is either non-negative (immediates may then be
// folded onto it) or the constant kOutOfRange plus an immediate below 64 KiB; a descriptor spans less than 2^30 bytes.
struct BufF32 {
  __amdgpu_buffer_rsrc_t r;
  static constexpr int kOutOfRange = 0x40000000;
  static __device__ __forceinline__ BufF32 make(const float* base, int n_floats) {
    BufF32 b;
    b.r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, n_floats > 0 ? n_floats * 4 : 0, 0x00020000);
    return b;
  }
  __device__ __forceinline__ float ld(int byte_off) const {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
  }
  __device__ __forceinline__ void st(float v, int byte_off) const {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), r, byte_off, 0, 0);
  }
  // the same with the non-temporal policy (aux bit 1): a stream this CU touches once
  __device__ __forceinline__ float ld_nt(int byte_off) const {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 2));
  }
  __device__ __forceinline__ void st_nt(float v, int byte_off) const {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), r, byte_off, 0, 2);
  }
};

// A float64 through the DPP path of the vector ALU (two 32-bit moves): CTRL / ROW_MASK as in the ISA manual; lanes that the row mask
// disables and lanes without a source lane read 0 (`old` is 0; BOUND = bound_ctrl:0 says the same for the lanes without a source).
template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ double dpp_f64(double v) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffLL), CTRL, ROW_MASK, 0xF, BOUND);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xF, BOUND);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}

// Inclusive prefix sum over the 64 lanes of a wave on the VALU: Hillis-Steele inside the rows of 16 (row_shr 1, 2, 4, 8: lanes
// without a source add 0), then the total of row 0 / 2 onto row 1 / 3 (row_bcast15) and of rows 0 - 1 onto rows 2 - 3 (row_bcast31):
// 6 additions and 12 DPP moves where the __shfl_up form went through the LDS crossbar twelve times with a compare and two
// selects per step (~80 instructions; it was 9 % of k_combtooth, a kernel that runs at the vector pipe's rate).
__device__ __forceinline__ double wave_incl_scan(double v) {
  v += dpp_f64<0x111, 0xF, true>(v);
  v += dpp_f64<0x112, 0xF, true>(v);
  v += dpp_f64<0x114, 0xF, true>(v);
  v += dpp_f64<0x118, 0xF, true>(v);
  v += dpp_f64<0x142, 0xA, false>(v);
  v += dpp_f64<0x143, 0xC, false>(v);
  return v;
}

// exclusive prefix over the 64 lanes of a wave
__device__ __forceinline__ double wave_excl_scan(double v, int /*lane*/) { return wave_incl_scan(v) - v; }

// sum over the 64 lanes, returned wave-uniform: lane 63 of the inclusive scan
__device__ __forceinline__ double wave_sum(double v) {
  const long long b = __builtin_bit_cast(long long, wave_incl_scan(v));
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}

// sum of a float over the 64 lanes of a wave, returned wave-uniform.  Four DPP butterflies (quad_perm,
// quad_perm, row_half_mirror, row_mirror) give every lane its 16-lane row sum on the VALU -- no LDS crossbar as
// __shfl_xor would use -- then four readlanes.
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));     // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));     // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false));    // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, false));    // row_mirror
  const int x = __float_as_int(v);
  return (__int_as_float(__builtin_amdgcn_readlane(x, 0)) + __int_as_float(__builtin_amdgcn_readlane(x, 16))) +
         (__int_as_float(__builtin_amdgcn_readlane(x, 32)) + __int_as_float(__builtin_amdgcn_readlane(x, 48)));
}

// maximum over the 64 lanes, the same way
__device__ __forceinline__ unsigned wave_max_dpp(unsigned v) {
  auto mx = [](unsigned a, unsigned b) { return a > b ? a : b; };
  v = mx(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false));
  v = mx(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false));
  v = mx(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false));
  v = mx(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false));
  const int x = (int)v;
  return mx(mx((unsigned)__builtin_amdgcn_readlane(x, 0), (unsigned)__builtin_amdgcn_readlane(x, 16)),
            mx((unsigned)__builtin_amdgcn_readlane(x, 32), (unsigned)__builtin_amdgcn_readlane(x, 48)));
}

// a / b in float32 for a normal b > 0 and a quotient far from the range limits (the exciters' sr x / (f0 + 1e-3), vocoder.py:839-840,
// :649): the hardware reciprocal r (v_rcp_f32, 1 ulp), q0 = a r, e = a - q0 b (exact in the fma), q = q0 + e r.  e r is off by at
// most 2^-23 of itself and |e r| <= 1.5 ulp(q0), so q is the correctly rounded quotient unless q0 + e r falls within ~2^-24 ulp of
// a rounding boundary: ~1 argument pair in 10^7 gets the neighbouring float (the IEEE sequence the compiler emits for `/`
// is v_div_scale x 2, v_rcp, four fmas, v_div_fmas, v_div_fixup: 10 instructions against 4 -- and the exciter kernels run at the
// vector pipe's rate, EXPERIMENTS 4.2).
__device__ __forceinline__ float div_pos(float a, float b) {
  const float r = __builtin_amdgcn_rcpf(b);
  const float q0 = a * r;
  const float e = fmaf(-q0, b, a);
  return fmaf(e, r, q0);
}

// tanh(x) = 1 - 2 / (exp(2 |x|) + 1) on the hardware exponential and reciprocal (1 ulp each), the sign copied back.  Absolute
// error <= 1.5e-7 (rms 4e-8; ocml's tanhf: 6e-8 / 2e-8 at ~6 x the instructions and two divergent branches).  The all-pass group
// delay pi tanh(c) (vocoder.py:581 / :834) and the NSF source's output activation (models.py:203) go through it.
__device__ __forceinline__ float tanh_hw(float x) {
  const float e = __builtin_amdgcn_exp2f(fabsf(x) * 2.88539008f);        // exp(2 |x|); inf beyond 44: the quotient is 0
  const float t = fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
  return copysignf(t, x);
}

// exp(x) on the hardware base-2 exponential: x log2(e) is split into its float32 rounding t and the residual r (two-constant
// log2(e)), exp2(t) (1 + r ln 2).  Relative error ~1e-7 for |x| <= 80 (a bare exp2(x * log2e) is off by |x| * 6e-8); 3 fma / mul +
// v_exp_f32 + 2 fma against the ~24 instructions of expf.  The control activations exp(c) of every module (vocoder.py:580, :603,
// :835-836, :661-664) go through it.
__device__ __forceinline__ float exp_hw(float x) {
  const float l2e_hi = 1.44269502f;                  // fl32(log2 e)
  const float l2e_lo = 1.92596303e-8f;               // log2 e - l2e_hi
  const float t = x * l2e_hi;
  float r = fmaf(x, l2e_hi, -t);
  r = fmaf(x, l2e_lo, r);
  const float e = __builtin_amdgcn_exp2f(t);
  const float v = fmaf(e, r * 0.693147182f, e);
  // the limits of torch.exp: beyond the finite range e is inf / 0 and the correction term would make inf - inf (x > 88.72) or
  // take r = -inf + inf at x = -inf; a NaN argument fails both compares and stays NaN
  return t >= 128.0f ? __builtin_inff() : (t < -150.0f ? 0.0f : v);
}

// torch.sinc on a float32 tensor: sin(p) / p with p = fl32(pi32 * z), 1 at z == 0 (vocoder.py:839, :649).
// |p| < 2: the Taylor series in p^2 to p^12 (truncation 1.2e-8); beyond, the hardware sine (two-constant reduction
// of the float32 p to revolutions, v_sin_f32: abs error <= 3.9e-7) times v_rcp_f32, i.e. error <= 2e-7 / |p|.
// 17 VALU + 2 transcendental instructions, no branches; against float64: max abs error 2.4e-7, rms 3e-8.
__device__ __forceinline__ float sin_turns(float a);
__device__ __forceinline__ float sinc_f32(float z) {
  const float p = kPiF * z;
  const float q = p * p;
  float s = 1.6059044e-10f;                          // 1/13!
  s = fmaf(s, q, -2.5052108e-8f);                    // -1/11!
  s = fmaf(s, q, 2.7557319e-6f);                     // 1/9!
  s = fmaf(s, q, -1.9841270e-4f);                    // -1/7!
  s = fmaf(s, q, 8.3333333e-3f);                     // 1/5!
  s = fmaf(s, q, -1.6666667e-1f);                    // -1/3!
  s = fmaf(s, q, 1.0f);
  const float big = sin_turns(p) * __builtin_amdgcn_rcpf(p);
  return fabsf(p) < 2.0f ? s : big;
}

// sin(a) for |a| up to a few thousand radians (the sinusoid bank reaches 256*pi).  The argument is brought to
// [-0.5, 0.5] revolutions with a two-constant split of 1/(2 pi) -- fma(a, hi, -n) is exact up to one rounding at
// magnitude 0.5 -- and handed to the hardware sine, which takes revolutions (v_sin_f32).  Against float64 on
// MI355X: max abs error 3.9e-7, rms 7.5e-8 over |a| <= 805 (profiles/r01_v3_sin_probe.log; ocml sinf: 7e-8 / 1.8e-8
// at 3.4x the cost).  4 VALU + 1 transcendental.
__device__ __forceinline__ float sin_turns(float a) {
  const float inv_hi = 0.15915494f;                  // fl32(1/(2 pi)) = 0x3E22F983
  const float inv_lo = 6.4206383e-9f;                // 1/(2 pi) - inv_hi
  float n = rintf(a * inv_hi);
  float r = fmaf(a, inv_hi, -n);
  r = fmaf(a, inv_lo, r);
  return __builtin_amdgcn_sinf(r);
}

// the same for two arguments at once, written on 2-vectors so the multiplies/fmas become packed-f32 instructions
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 sin_turns2(f32x2 a) {
  const f32x2 inv_hi = {0.15915494f, 0.15915494f};
  const f32x2 inv_lo = {6.4206383e-9f, 6.4206383e-9f};
  const f32x2 t = a * inv_hi;
  const f32x2 n = {rintf(t.x), rintf(t.y)};
  f32x2 r = __builtin_elementwise_fma(a, inv_hi, -n);
  r = __builtin_elementwise_fma(a, inv_lo, r);
  return f32x2{__builtin_amdgcn_sinf(r.x), __builtin_amdgcn_sinf(r.y)};
}

}  // namespace ddsp
