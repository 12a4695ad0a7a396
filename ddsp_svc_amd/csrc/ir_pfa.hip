// HOT-2b, fast form for n_mag = 256: frequency response -> per-frame FIR taps (ddsp/core.py:254-270 with the window
// helpers :185-251, fed by the activations of ddsp/vocoder.py:580-582,599 / :834-836,845) as a PRIME-FACTOR FFT.
//
// torch.fft.irfft of 256 bins is an inverse DFT of N = 510 = 2 * 3 * 5 * 17 points.  The four factors are pairwise
// coprime, so the Good-Thomas index maps split the transform into small DFTs with NO twiddle factors in between:
//     n = (30 n1 + 17 n2) mod 510,   k = (120 k1 + 391 k2) mod 510        (17 x 30;  120 = 30 * (30^-1 mod 17), 391 = 17 * (17^-1 mod 30))
//     n2 = (15 a + 10 b + 6 c) mod 30, k2 = (15 ka + 10 kb + 6 kc) mod 30  (2 x 3 x 5 inside the 30)
//     z[k] = sum_n Z[n] W^(nk)  =  DFT_30 over n2 of ( DFT_17 over n1 of Z )
// One complex transform carries TWO frames (Z = X_a + i X_b of the Hermitian-extended responses, z = ir_a + i ir_b),
// real responses (zero-phase magnitude filters) and complex ones (the all-pass) alike: 510 * (17 + 2 + 3 + 5) / 2 ~ 7 k
// complex multiply-adds per frame against 65 k (131 k for the all-pass) real ones of the dense contraction in ir.hip
// -- the tap synthesis stops being GEMM-shaped and becomes what the rest of the path is: bound by its HBM streams
// (1 KB of control in, 2 KB of taps out per frame).  The activations (exp, exp/128, all-pass pi*tanh -> cumsum ->
// cos/sin) are applied while the rows are staged, roll + window while they are stored.  ir.hip stays the path for
// every other n_mag (N = 2(n-1) has large prime factors in general: 254 = 2 * 127).
//
// Workgroup = 256 threads, 16 frames = 8 transforms per batch:
//   stage 0  coalesced loads of the 16 control rows, activation, scaled by 1/N           -> LDS rows
//   stage A  thread (t, n2), 240 of 256: gathers its 17 points, DFT-17                   -> LDS W[t][k1][n2]
//   stage B  thread (t, k1), 136 of 256: 30 contiguous points, DFT-30 in registers       -> LDS O[row][tap], rolled
//   stage C  window + fully coalesced 16-byte stores of the batch's contiguous 16 x 510 taps
#include "ddsp_common.h"
#include "fft2048.h"
#include "frame_phase.h"
#include "kernels.h"

namespace ddsp {

using fft::add_mi;
using fft::sub_mi;

#ifdef DDSP_HIP_TIMELINE
// diagnostics build only (tools/pfa_timeline.py): per-workgroup wall-clock stamps (100 MHz)
__device__ long long* g_pfa_timeline = nullptr;
__global__ void k_set_pfa_timeline(long long* p) { g_pfa_timeline = p; }
#define PFA_STAMP(slot) do { if (g_pfa_timeline && threadIdx.x == 0) g_pfa_timeline[(long)blockIdx.x * 8 + (slot)] = (long long)wall_clock64(); } while (0)
#else
#define PFA_STAMP(slot) do { } while (0)
#endif

namespace pfa {

// the 16-byte store of four taps (A/B builds: -DDDSP_PFA_NT = with the non-temporal policy)
typedef float v4f_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store4(float* p, float a, float b, float c, float d) {
#ifdef DDSP_PFA_NT
  __builtin_nontemporal_store(v4f_t{a, b, c, d}, reinterpret_cast<v4f_t*>(p));
#else
  *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
#endif
}

#ifdef DDSP_PFA_NT_LD
#define PFA_LD(p) __builtin_nontemporal_load(p)
#else
#define PFA_LD(p) (*(p))
#endif
constexpr int NB = 256, NT = 510, HALF = 255;
#ifndef DDSP_PFA_ROWS
#define DDSP_PFA_ROWS 16
#endif
// workgroups per CU the LDS of a batch allows (2 ROWS KiB each, 160 KiB per CU less the allocation granule): the register bound follows
#define DDSP_PFA_WGS (DDSP_PFA_ROWS >= 16 ? 4 : DDSP_PFA_ROWS >= 14 ? 5 : 6)
constexpr int ROWS = DDSP_PFA_ROWS, TR = ROWS / 2;   // frames and transforms per batch (even; 16: 32 KiB of LDS, four workgroups per CU)
static_assert(ROWS % 2 == 0 && ROWS <= 16 && ROWS >= 2, "batch of 2 .. 16 rows");
// W[t][k1][n2], complex: row (t, k1) = 30 words, plus one pad word every 16 rows -- the 32 rows a 32-lane LDS access of stage B
// touches then start in 32 different bank pairs (30 words = 60 banks = -4 mod 64: 16 rows a revolution, the pad shifts the
// next 16 by one pair) -- and the array fits the 32 KiB the staged rows need anyway.  (Four workgroups per CU either way: a
// fifth needs <= 31 744 B per workgroup, tools/probes/lds_occupancy.hip, and the 16 x 510 output staging alone is 32 640.)
constexpr int WSTRIDE = 30;
__device__ __forceinline__ int w_row(int row) { return row * WSTRIDE + (row >> 4); }
constexpr int KIND_REAL = 0, KIND_COMPLEX = 1, KIND_ALLPASS = 2;
enum { MODE_ROLL = 0, MODE_HANN = 1, MODE_DYNAMIC = 2 };

// cos / sin (2 pi r / 17), r = 0..8
__device__ constexpr float C17[9] = {1.0f, 0.93247222940435580f, 0.73900891722065910f, 0.44573835577653826f,
                                     0.09226835946330200f, -0.27366299007208283f, -0.60263463637925640f,
                                     -0.85021713572961420f, -0.98297309968390180f};
__device__ constexpr float S17[9] = {0.0f, 0.36124166618715290f, 0.67369564364655720f, 0.89516329135506230f,
                                     0.99573417629503450f, 0.96182564317281900f, 0.79801722728023950f,
                                     0.52643216287735580f, 0.18374951781657034f};
constexpr float C5_1 = 0.30901699437494745f, C5_2 = -0.80901699437494745f;      // cos(2 pi / 5), cos(4 pi / 5)
constexpr float S5_1 = 0.95105651629515350f, S5_2 = 0.58778525229247310f;       // sin(2 pi / 5), sin(4 pi / 5)
constexpr float S3 = 0.86602540378443860f;                                       // sin(2 pi / 3)

__device__ __forceinline__ f32x2 fma2(float c, f32x2 a, f32x2 acc) { return __builtin_elementwise_fma(f32x2{c, c}, a, acc); }

// all small transforms use the + sign (inverse DFT): X[k] = sum_n x[n] exp(+2 pi i n k / p)
__device__ __forceinline__ void dft17(f32x2 (&z)[17]) {
  f32x2 s[9], d[9];
#pragma unroll
  for (int j = 1; j <= 8; ++j) { s[j] = z[j] + z[17 - j]; d[j] = z[j] - z[17 - j]; }
  f32x2 x0 = z[0];
  f32x2 dc = x0;
#pragma unroll
  for (int j = 1; j <= 8; ++j) dc = dc + s[j];
#pragma unroll
  for (int k = 1; k <= 8; ++k) {
    f32x2 p = x0, q = f32x2{0.f, 0.f};
#pragma unroll
    for (int j = 1; j <= 8; ++j) {
      const int r = (j * k) % 17;
      const int rr = r <= 8 ? r : 17 - r;
      p = fma2(C17[rr], s[j], p);
      q = fma2(r <= 8 ? S17[rr] : -S17[rr], d[j], q);
    }
    z[k] = sub_mi(p, q);             // p + i q
    z[17 - k] = add_mi(p, q);        // p - i q
  }
  z[0] = dc;
}

__device__ __forceinline__ void dft5(f32x2& x0, f32x2& x1, f32x2& x2, f32x2& x3, f32x2& x4) {
  const f32x2 t1 = x1 + x4, t2 = x2 + x3, t3 = x1 - x4, t4 = x2 - x3;
  const f32x2 m1 = fma2(C5_2, t2, fma2(C5_1, t1, x0));
  const f32x2 m2 = fma2(C5_1, t2, fma2(C5_2, t1, x0));
  const f32x2 n1 = fma2(S5_2, t4, f32x2{S5_1, S5_1} * t3);
  const f32x2 n2 = fma2(-S5_1, t4, f32x2{S5_2, S5_2} * t3);
  x0 = x0 + t1 + t2;
  x1 = sub_mi(m1, n1);
  x4 = add_mi(m1, n1);
  x2 = sub_mi(m2, n2);
  x3 = add_mi(m2, n2);
}

__device__ __forceinline__ void dft3(f32x2& x0, f32x2& x1, f32x2& x2) {
  const f32x2 t = x1 + x2, d = x1 - x2;
  const f32x2 m = fma2(-0.5f, t, x0);
  const f32x2 n = f32x2{S3, S3} * d;
  x0 = x0 + t;
  x1 = sub_mi(m, n);
  x2 = add_mi(m, n);
}

// 30 points in the order they arrive (n2 = 0..29) -> v[k2], k2 = 0..29, through the 2 x 3 x 5 prime-factor maps;
// every index below is a compile-time constant after unrolling, so the 30 points stay in registers
__device__ __forceinline__ void dft30(f32x2 (&v)[30]) {
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const int o = 15 * a + 10 * b;
      dft5(v[o % 30], v[(o + 6) % 30], v[(o + 12) % 30], v[(o + 18) % 30], v[(o + 24) % 30]);
    }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      const int o = 15 * a + 6 * c;
      dft3(v[o % 30], v[(o + 10) % 30], v[(o + 20) % 30]);
    }
#pragma unroll
  for (int b = 0; b < 3; ++b)
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      const int o = (10 * b + 6 * c) % 30;
      const f32x2 p = v[o], q = v[(o + 15) % 30];
      v[o] = p + q;
      v[(o + 15) % 30] = p - q;
    }
  // in place: position (15 a + 10 b + 6 c) mod 30 now holds output k2 = (15 ka + 10 kb + 6 kc) mod 30 with (ka,kb,kc) = (a,b,c)
}

// cos(a) for the dynamic window, |a| < ~10 (same reduction as sin_turns, hardware cosine in revolutions; ir.hip)
__device__ __forceinline__ float cos_turns_w(float a) {
  const float inv_hi = 0.15915494f, inv_lo = 6.4206383e-9f;
  float nn = rintf(a * inv_hi);
  float r = fmaf(a, inv_hi, -nn);
  r = fmaf(a, inv_lo, r);
  return __builtin_amdgcn_cosf(r);
}

// a / b with one true division per ROW instead of one per tap: rb = RN(1 / b); q0 = a rb; e = a - q0 b (exact in fma);
// q = q0 + e rb is the correctly rounded quotient except for ~1 argument pair in 10^6 (off by one ulp).  The only place
// where an ulp matters is the window's one-sided clamp u > 1 (core.py:245: the factor jumps from 0 to 1 there), so
// quotients within a few ulps of 1 take the true division.
__device__ __forceinline__ float div_by_row(float a, float b, float rb) {
  const float q0 = a * rb;
  const float e = fmaf(-q0, b, a);
  float q = fmaf(e, rb, q0);
  if (fabsf(q - 1.0f) < 4e-7f) q = a / b;
  return q;
}

// One LDS region of 32 KiB, used three times: rows re[16][256] (+ im[16][256]) -> W[8][17][30+] complex -> O[16][510] followed
// by the 16 rows' window half widths and their reciprocals (MODE_DYNAMIC)
constexpr int W_WORDS = TR * 17 * WSTRIDE + (TR * 17 + 15) / 16;          // complex words
constexpr int U_FLOATS = 2 * ROWS * NB;                                   // 8192 floats
// the batch's window rows (MODE_DYNAMIC) live in the 32 floats the output staging O[16][510] leaves of the union region -- the
// stage that reads them reads nothing else there -- and one flag word behind it.  (Round 6 first put a larger table BEHIND the
// region: LDS reads at offsets >= 32 KiB of a workgroup's allocation turned out to be pathologically slow for the workgroups of
// a launch's first round -- stage C 10 - 15 us instead of 2.6 for half of them, tools/pfa_tail_probe.py, profiles/r06_v3_* -- so
// nothing a loop reads may live there; the flag is read once per thread.)
constexpr int HW_AT = ROWS * NT;                                          // 8160: float2 {RN(1 / hw), thr / 2} x ROWS  (or {hw, RN(1 / hw)}: below)
constexpr int HW_FLAG = U_FLOATS;                                         // != 0: some row of the batch needs the per-tap form
constexpr int LDS_FLOATS = U_FLOATS + 4;
static_assert(U_FLOATS >= 2 * W_WORDS && U_FLOATS >= HW_AT + 2 * ROWS, "union region too small");

// ---- the dynamic window of core.py:240-251, w(d) = (1 + cos(pi u)) / 2 with u = d / hw and ONLY u > 1 clamped (to u = 0, i.e.
// w = 1; core.py:245), d = j - N/2 -- per ROW: the reciprocal of hw and the clamp as a threshold on d.  u = RN(d / hw) is
// monotonic in d, so "u > 1" is "d >= thr" with thr = the smallest integer offset whose float32 quotient exceeds 1: floor(hw) + 1
// unless that quotient rounds to exactly 1 (hw one ulp below an integer), then one more -- decided with ONE true division per row,
// so the clamp is the reference's bit for bit whatever the cosine's argument rounds to.
// A row is `nice` when hw >= 0.5 and finite (every f0 below 132 kHz): only then are u / 2 revolutions inside the hardware
// cosine's range and the threshold form valid; any other row sends its whole batch to the per-tap form of rounds 2 - 5, and the
// batch's rows then hold {hw, RN(1 / hw)} instead.
__device__ __forceinline__ bool window_row_nice(float hw) { return hw >= 0.5f && hw < 3.0e38f; }
__device__ __forceinline__ float2 stage_window_row(float hw) {
  const float dc = floorf(hw) + 1.0f;
  float thr = (dc / hw > 1.0f) ? dc : dc + 1.0f;
  thr = thr < 1024.0f ? thr : 1024.0f;
  return make_float2(1.0f / hw, 0.5f * thr);
}

// two taps of one row: dh = d / 2 of both.  cos(pi u) = cos(2 pi (u / 2)), and the hardware cosine takes revolutions: u / 2 =
// dh * RN(1 / hw) -- two roundings (1.2e-7 of the angle) where the reference's float32 chain u = RN(d / hw), RN(pi u) has two of
// its own: the two windows differ by <= 2.4e-7 of the angle, i.e. <= 1.2e-6 (typically 2e-7) of the window at the ~10 rad a
// 256-bin filter reaches at f0 = 800 Hz, and NOT AT ALL in which taps are clamped.  6 packed + 2 transcendental + 4 instructions
// per pair; rounds 2 - 5 spent ~50 on a division with an exact fallback and a reduced cosine per tap.
__device__ __forceinline__ f32x2 window_pair(f32x2 dh, float2 row) {
  const f32x2 q = dh * f32x2{row.x, row.x};
  const f32x2 c = {__builtin_amdgcn_cosf(q.x), __builtin_amdgcn_cosf(q.y)};
  f32x2 w = __builtin_elementwise_fma(f32x2{0.5f, 0.5f}, c, f32x2{0.5f, 0.5f});      // (1 + cos) / 2, core.py:246
  w.x = dh.x >= row.y ? 1.0f : w.x;                        // core.py:245
  w.y = dh.y >= row.y ? 1.0f : w.y;
  return w;
}

// the window factors of a group of four consecutive positions j .. j+3 of row r (taps 0, 1 of row r + 1 at j = 508): both kernels
__device__ __forceinline__ void window_group(const float* U, int j, int r, float (&w)[4]) {
  const float2* HW2 = reinterpret_cast<const float2*>(U + HW_AT);
  const bool wrap = j + 3 >= NT;
  const float2 ha = HW2[r];
  const float2 hx = HW2[r + 1 < ROWS ? r + 1 : r];          // (a straddling group never reaches row 16: it would start past the batch)
  const float2 hb = wrap ? hx : ha;
  const float dh = 0.5f * (float)(j - HALF);
  const float dh2 = wrap ? -0.5f * (float)HALF : dh + 1.0f;
  const f32x2 wa = window_pair(f32x2{dh, dh + 0.5f}, ha);
  const f32x2 wb = window_pair(f32x2{dh2, dh2 + 0.5f}, hb);
  w[0] = wa.x; w[1] = wa.y; w[2] = wb.x; w[3] = wb.y;
}
// the same in the per-tap form (rows hold {hw, RN(1 / hw)})
__device__ __forceinline__ void window_group_per_tap(const float* U, int j, int r, float (&w)[4]) {
  const float2* HW2 = reinterpret_cast<const float2*>(U + HW_AT);
  const bool wrap = j + 3 >= NT;
  const float2 ha = HW2[r], hb = HW2[wrap && r + 1 < ROWS ? r + 1 : r];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const bool nx = wrap && e >= 2;
    const int je = nx ? e - 2 : j + e;
    float u = div_by_row((float)(je - HALF), nx ? hb.x : ha.x, nx ? hb.y : ha.y);     // core.py:244
    if (u > 1.0f) u = 0.0f;                                          // core.py:245 -- only the upper side is clamped
    w[e] = (1.0f + cos_turns_w(kPiF * u)) / 2.0f;                    // core.py:246
  }
}
// wave 0 of a workgroup stages the batch's rows from their half widths (lane = row; lanes >= ROWS pass hw = 1)
__device__ __forceinline__ void stage_window_rows(float* U, int lane, float hw) {
  const unsigned long long bad = __ballot(lane < ROWS && !window_row_nice(hw));
  const float2 row = bad ? make_float2(hw, 1.0f / hw) : stage_window_row(hw);
  if (lane < ROWS) *reinterpret_cast<float2*>(U + HW_AT + 2 * lane) = row;
  if (lane == 0) U[HW_FLAG] = bad ? 1.0f : 0.0f;
}

// (tanh_hw, the all-pass group delay's hyperbolic tangent on the hardware units: ddsp_common.h; a sum of 256 of them is off by
// 2e-6 rad rms, a fifth of what the reference's own float32 rounding of that sum -- ulp(100 rad) / 2 = 4e-6 -- does to it)

// inclusive prefix sum of a 32-bit integer over the 64 lanes of a wave (the DPP steps of wave_incl_scan, ddsp_common.h)
template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ unsigned dpp_u32(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, BOUND); }
__device__ __forceinline__ unsigned wave_incl_scan_u32(unsigned v) {
  v += dpp_u32<0x111, 0xF, true>(v);
  v += dpp_u32<0x112, 0xF, true>(v);
  v += dpp_u32<0x114, 0xF, true>(v);
  v += dpp_u32<0x118, 0xF, true>(v);
  v += dpp_u32<0x142, 0xA, false>(v);
  v += dpp_u32<0x143, 0xC, false>(v);
  return v;
}

}  // namespace pfa

// KIND / ACT / MODE are kernel ARGUMENTS, not template parameters: the three tap syntheses of a step (all-pass, dynamic
// window, Hann) are then ONE code object of ~20 KB that stays in the 64 KB instruction cache from launch to launch,
// instead of three of 11-17 KB that evict each other and the filter kernel -- on the part of the pool where an
// instruction fetch past that cache is slow, a cold launch of this kernel takes 44-62 us against 28-38 warm (DESIGN.md,
// "instruction cache").  Every switch on them is workgroup-uniform (scalar branches).
//
// A launch takes up to three JOBS (grid.y): the three tap syntheses of a CombSub / Sins step depend on the controls only, so a
// step at a streaming shape (B = 1, a fraction of a second: the chain of dependent launches IS its latency, ~9 us each)
// issues them as one launch.  The job is workgroup-uniform like everything it selects.
//
// EXC (k_front_small): one more row of workgroups (blockIdx.y = jobs.n) makes the step's exciter, the combtooth
// (frame_phase.h), 16 frames each -- it depends on the phase state only, like the tap syntheses on the controls, so a
// streaming-shape CombSub step has ONE launch in front of its filters.  A second entry point, so that the batch layout's
// kernel does not carry the exciter's code through the instruction cache.
template <bool EXC>
__device__ __forceinline__ void taps_pfa510_body(const TapsJobs& jobs, const ExciterJob& exc, float* U) {
  using namespace pfa;
  if (EXC && (int)blockIdx.y == jobs.n) {                  // workgroup-uniform; before any barrier
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll 1
    for (int q = 0; q < (ROWS + 3) / 4; ++q) {
      const long fr = (long)blockIdx.x * ROWS + q * 4 + wave;
      if (q * 4 + wave < ROWS && fr < exc.n_frames)
        combtooth_frame<8, true>(exc.f0_frames, exc.initial_phase, fr, exc.F, exc.hop, exc.up, exc.cfg, exc.phase0, exc.out, lane);
    }
    return;
  }
  const TapsJob& J = jobs.j[blockIdx.y];
  const int KIND = J.kind, ACT = J.act, MODE = J.mode;
  const float* __restrict__ a_re = J.a_re;
  const float* __restrict__ a_im = J.a_im;
  const long ld_re = J.ld_re, ld_im = J.ld_im;
  const float scale = J.scale, hw_sr = J.hw_sr;
  const float* __restrict__ hann = J.hann;
  const float* __restrict__ half_width = J.half_width;
  const long rows = J.rows;
  float* __restrict__ taps = J.taps;
  const bool half = J.half != 0;                            // rows of NB = N/2 + 1 taps (an even response under an even window: tap N - j is tap j)
  const int tid = threadIdx.x;
  const long row0 = (long)blockIdx.x * ROWS;
  float* re_s = U;
  float* im_s = U + ROWS * NB;
  f32x2* W = reinterpret_cast<f32x2*>(U);
  float* O = U;
  const float inv_n = 1.0f / (float)NT;
  PFA_STAMP(0);

  // ---- stage 0: rows -> LDS, activated and scaled by 1/N.  irfft drops Im(DC) and Im(Nyquist) (core.py:259) ----
  if (KIND == KIND_ALLPASS) {
    // exp(1j * cumsum(pi * tanh(c))) (vocoder.py:581,599 / :834,845): wave w takes rows 4w..4w+3, a lane owns 4 consecutive
    // bins.  Only the phase's fraction of a revolution matters, so the running sum is kept in REVOLUTIONS as a 32-bit fixed-point
    // number that wraps where the cosine does: g = fl32(pi tanh c) as the reference forms it, g / 2 pi * 2^32 rounded to an integer
    // in float64 (exact to 2^-33 of a revolution; the low word of the sum with 1.5 * 2^52 is that integer mod 2^32), and from there
    // integer additions -- exact and associative: the wave scan is six one-word DPP additions instead of the float64 scan of rounds
    // 2 - 5, and the sum of 256 bins is off by < 2^-25 of a revolution (2e-7 rad) whatever its size.  (The reference's float64
    // accumulation rounded to float32, vocoder.py:599 on the CPU, is off by ulp(sum) / 2: 1e-6 rad at 30 rad, 4e-6 at 100.)
    const int wave = tid >> 6, lane = tid & 63;
    const double rev_fx = 683565275.57643158978229477;      // 2^32 / (2 pi)
    const double magic = 6755399441055744.0;                // 1.5 * 2^52
    // a ROLLED loop over the wave's four rows, the next row's load in flight while a row is activated
    auto load_row = [&](int q) -> float4 {
      const long gr = row0 + wave * 4 + q;
      float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gr < rows && wave * 4 + q < ROWS) {
        const float* src = a_re + gr * ld_re + 4 * lane;
        c.x = src[0]; c.y = src[1]; c.z = src[2]; c.w = src[3];
      }
      return c;
    };
    float4 nxt = load_row(0);
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
      const float4 cv = nxt;
      if (q < 3) nxt = load_row(q + 1);
      const int r = wave * 4 + q;
      if (r >= ROWS) break;                                // wave-uniform (batches of fewer than 16 rows)
      const bool live = row0 + r < rows;
      const float g[4] = {kPiF * tanh_hw(cv.x), kPiF * tanh_hw(cv.y), kPiF * tanh_hw(cv.z), kPiF * tanh_hw(cv.w)};
      unsigned s[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const double m = fma((double)g[e], rev_fx, magic);
        const unsigned fx = (unsigned)__builtin_bit_cast(unsigned long long, m);
        s[e] = e ? s[e - 1] + fx : fx;
      }
      const unsigned before = wave_incl_scan_u32(s[3]) - s[3];
      float co[4], si[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float fr = (float)(int)(before + s[e]) * 2.3283064365386963e-10f;     // 2^-32: revolutions in [-0.5, 0.5]
        co[e] = live ? __builtin_amdgcn_cosf(fr) * inv_n : 0.f;
        si[e] = live ? __builtin_amdgcn_sinf(fr) * inv_n : 0.f;
      }
      if (lane == 0) si[0] = 0.f;                       // Im(DC)
      if (lane == 63) si[3] = 0.f;                      // Im(Nyquist)
      *reinterpret_cast<float4*>(re_s + r * NB + 4 * lane) = make_float4(co[0], co[1], co[2], co[3]);
      *reinterpret_cast<float4*>(im_s + r * NB + 4 * lane) = make_float4(si[0], si[1], si[2], si[3]);
    }
  } else {
    const int r = tid >> 4, c4 = (tid & 15) * 4;
    const long gr = row0 + r;
    const bool live = gr < rows && r < ROWS;
    const bool mine = r < ROWS;                             // batches of fewer than 16 rows: the last threads stage nothing
    const float sc = scale * inv_n;
    float v[4][4], w[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {                           // all loads first
      const int k = c4 + 64 * q;
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[q][e] = 0.f; w[q][e] = 0.f; }
      if (live) {
        const float* src = a_re + gr * ld_re + k;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[q][e] = PFA_LD(src + e);
        if (KIND == KIND_COMPLEX) {
          const float* si = a_im + gr * ld_im + k;
#pragma unroll
          for (int e = 0; e < 4; ++e) w[q][e] = si[e];
        }
      }
    }
    if (ACT == 1) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[q][e] = exp_hw(v[q][e]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = c4 + 64 * q;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[q][e] = live ? v[q][e] * sc : 0.f;
        w[q][e] = live ? w[q][e] * sc : 0.f;
      }
      if (KIND == KIND_COMPLEX && mine) {
        if (k == 0) w[q][0] = 0.f;
        if (k + 3 == NB - 1) w[q][3] = 0.f;
        *reinterpret_cast<float4*>(im_s + r * NB + k) = make_float4(w[q][0], w[q][1], w[q][2], w[q][3]);
      }
      if (mine) *reinterpret_cast<float4*>(re_s + r * NB + k) = make_float4(v[q][0], v[q][1], v[q][2], v[q][3]);
    }
  }
  __syncthreads();
  PFA_STAMP(1);

  // ---- stage A: DFT-17 over n1 of Z[(30 n1 + 17 n2) mod 510], Z = X_a + i X_b, X[510 - k] = conj X[k] ----
  // A REAL response (zero-phase magnitude filter) is EVEN after its Hermitian extension, Z[-n] = Z[n], and so is its transform:
  // column 30 - n2 of this stage is column n2 with k1 reversed, W[k1][30 - n2] = W[17 - k1][n2], so 16 of the 30 columns are
  // computed (128 threads: two full waves instead of four), and stage B computes rows k1 = 0 .. 8 only -- row 17 - k1 is the same
  // numbers at the mirrored taps -- reading the missing columns from the mirrored row and writing every result twice.  (Round 2
  // built this when a launch's length was set by its latencies and it did not pay; in the fused front launch, which runs at the
  // vector pipe's rate, it does: EXPERIMENTS 6.8.  -DDDSP_PFA_NOSYM: the full form, for A/Bs.)
#ifdef DDSP_PFA_NOSYM
  const bool sym = false;
#else
  const bool sym = KIND == KIND_REAL;                       // workgroup-uniform
#endif
  {
    const bool act = sym ? tid < TR * 16 : tid < TR * 30;
    const int t = !act ? 0 : (sym ? tid >> 4 : tid / 30), n2 = !act ? 0 : (sym ? tid & 15 : tid - 30 * t);
    f32x2 z[17];
    {
      const float* ra = re_s + (2 * t) * NB;
      const float* rb = ra + NB;
      const float* ia = im_s + (2 * t) * NB;
      const float* ib = ia + NB;
      int k = 17 * n2;                                     // < 510
      if (KIND == KIND_REAL) {
#pragma unroll
        for (int n1 = 0; n1 < 17; ++n1) {
          const int kk = k > HALF ? NT - k : k;
          z[n1] = f32x2{ra[kk], rb[kk]};
          k += 30;
          if (k >= NT) k -= NT;
        }
      } else {
#pragma unroll
        for (int n1 = 0; n1 < 17; ++n1) {
          const bool mir = k > HALF;
          const int kk = mir ? NT - k : k;
          const float sg = mir ? -1.0f : 1.0f;
          z[n1] = f32x2{fmaf(-sg, ib[kk], ra[kk]), fmaf(sg, ia[kk], rb[kk])};
          k += 30;
          if (k >= NT) k -= NT;
        }
      }
    }
    __syncthreads();                                       // every thread holds its inputs: the rows may be overwritten
    if (act) {
      dft17(z);
#pragma unroll
      for (int k1 = 0; k1 < 17; ++k1) W[w_row(t * 17 + k1) + n2] = z[k1];
    }
  }
  __syncthreads();
  PFA_STAMP(2);

  // ---- stage B: DFT-30 over n2; output index m = (120 k1 + 391 k2) mod 510; roll by N/2: tap j = (m + 255) mod 510 ----
  {
    const bool act = sym ? tid < TR * 9 : tid < TR * 17;
    const int t = !act ? 0 : (sym ? tid / 9 : tid / 17), k1 = !act ? 0 : (sym ? tid - 9 * t : tid - 17 * t);
    f32x2 v[30];
    {
      const f32x2* src = W + w_row(t * 17 + k1);
      if (sym) {                                            // columns 16 .. 29 live in the mirrored row (row 0 mirrors itself)
        const f32x2* msrc = W + w_row(t * 17 + (k1 ? 17 - k1 : 0));
#pragma unroll
        for (int n2 = 0; n2 < 30; ++n2) v[n2] = n2 <= 15 ? src[n2] : msrc[30 - n2];
      } else {
#pragma unroll
        for (int n2 = 0; n2 < 30; ++n2) v[n2] = src[n2];
      }
    }
    float hw_row = 1.f;
    if (MODE == MODE_DYNAMIC && tid < ROWS) {              // the batch's 16 half widths: in flight across the barrier
      const long gr = row0 + tid;
      if (gr < rows) hw_row = half_width[gr];
    }
    __syncthreads();                                       // W is in registers: the region becomes the output staging
    if (MODE == MODE_DYNAMIC && tid < 64) {                // wave 0: the rows' window constants (behind the output staging)
      float x = hw_row;
      if (hw_sr > 0.f) x = (1.5f * hw_sr) / (x + 1e-3f);   // vocoder.py:851, same float32 operations
      stage_window_rows(U, tid, x);
    }
    if (act) {
      dft30(v);
      // an even response (sym): only the results at taps j <= N/2 are used -- tap N - j is the same number.  (Row k1 = 0 holds
      // both j and N - j of its own; they agree to rounding only, so the one at j <= N/2 serves both and a response comes out
      // EXACTLY even whichever layout stores it.)  half: the staging rows hold the NB taps j <= N/2.
      const int rl = half ? NB : NT;
      float* oa = O + (2 * t) * rl;
      float* ob = oa + rl;
      const int base = (120 * k1 + HALF) % NT;
#pragma unroll
      for (int k2 = 0; k2 < 30; ++k2) {
        int j = base + (391 * k2) % NT;                     // the second term is a compile-time constant
        if (j >= NT) j -= NT;
        if (!sym) {
          oa[j] = v[k2].x;
          ob[j] = v[k2].y;
        } else {
          const int jm = j ? NT - j : 0;                    // z[-m] = z[m]: tap 510 - j
          const int jl = j <= HALF ? j : jm;                // the one of the two at or below N/2
          if (k1 != 0 || j <= HALF) {
            oa[jl] = v[k2].x;
            ob[jl] = v[k2].y;
            if (!half) {
              const int jh = NT - jl;                       // its mirror image (jl = 0 has none; jl = N/2 is its own)
              if (jl != 0) { oa[jh] = v[k2].x; ob[jh] = v[k2].y; }
            }
          }
        }
      }
    }
  }
  __syncthreads();
  PFA_STAMP(3);

  // ---- stage C: window and store; the batch's 16 x 510 taps are one contiguous, 16-byte aligned stretch.  A thread's
  // groups of four are 1024 floats apart: row + 2, tap + 4 (mod 510).  Eight groups per thread, unrolled: the window
  // values of all eight are fetched first (MODE_HANN) or come from LDS (MODE_DYNAMIC), so a memory latency is paid once
  // and not once per group (tools/pfa_timeline.py: stage C 3.5 -> 2.2 us and 4.9 -> 4.0 us) ----
  if (half) {
    // rows of NB = 256 taps: a thread's groups of four are four rows apart at the same taps j .. j + 3 -- one window fetch serves
    // all of them (MODE_HANN), half the stores and window factors of the full rows
    static_assert(NB == 256, "the half-row store: 64 threads per row");
    float* dst = taps + row0 * NB;
    const long left = rows - row0;                         // rows left in the tensor from this batch on
    const int j = 4 * (tid & 63), rb = tid >> 6;
    float2 wa = make_float2(1.f, 1.f), wb = wa;
    if (MODE == MODE_HANN) {
      wa = *reinterpret_cast<const float2*>(hann + j);
      wb = *reinterpret_cast<const float2*>(hann + j + 2);
    }
#pragma unroll
    for (int it = 0; it < (ROWS + 3) / 4; ++it) {
      const int r = rb + 4 * it;
      if (ROWS % 4 != 0 && r >= ROWS) break;
      const float4 o = *reinterpret_cast<const float4*>(O + r * NB + j);
      const float w[4] = {wa.x, wa.y, wb.x, wb.y};          // (never MODE_DYNAMIC: that window is not even, launch_taps_pfa510)
      if (r < left) store4(dst + r * NB + j, o.x * w[0], o.y * w[1], o.z * w[2], o.w * w[3]);
    }
    PFA_STAMP(4);
    return;
  }
  float* dst = taps + row0 * NT;
  const long total = rows * (long)NT - row0 * NT;         // floats left in the tensor from this batch on
  constexpr int GROUPS = (ROWS * NT / 4 + 255) / 256;     // 8; the last one is partial
  const int r0 = (4 * tid) / NT, j0 = 4 * tid - r0 * NT;
  int rr[GROUPS], jj[GROUPS];
#pragma unroll
  for (int it = 0; it < GROUPS; ++it) {
    const int j = j0 + 4 * it;
    const bool c = j >= NT;
    jj[it] = c ? j - NT : j;
    rr[it] = r0 + 2 * it + (c ? 1 : 0);
  }
  auto fetch = [&](int it, float (&ov)[4]) -> bool {       // false: this thread has no group `it` (only the last one is partial)
    const int i4 = tid + 256 * it;
    if (it == GROUPS - 1 && i4 >= ROWS * NT / 4) return false;
    const float4 o = *reinterpret_cast<const float4*>(O + 4 * i4);
    ov[0] = o.x; ov[1] = o.y; ov[2] = o.z; ov[3] = o.w;
    return true;
  };
  auto put = [&](int it, const float (&ov)[4]) {
    const int i = 4 * (tid + 256 * it);
    if (i + 3 < total) {
      store4(dst + i, ov[0], ov[1], ov[2], ov[3]);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (i + e < total) dst[i + e] = ov[e];
    }
  };
  // one loop per window mode (the mode is workgroup-uniform: one scalar branch, not one per group)
  if (MODE == MODE_HANN) {
    float2 wa[GROUPS], wb[GROUPS];
#pragma unroll
    for (int it = 0; it < GROUPS; ++it) {
      const int j = jj[it];
      const int j2 = j + 3 >= NT ? 0 : j + 2;              // j is even: a group straddles a row end only at j = 508
      wa[it] = *reinterpret_cast<const float2*>(hann + j);
      wb[it] = *reinterpret_cast<const float2*>(hann + j2);
    }
#pragma unroll
    for (int it = 0; it < GROUPS; ++it) {
      float ov[4];
      if (!fetch(it, ov)) break;
      ov[0] *= wa[it].x; ov[1] *= wa[it].y; ov[2] *= wb[it].x; ov[3] *= wb[it].y;
      put(it, ov);
    }
  } else if (MODE == MODE_DYNAMIC) {
    // (everything the loops read is in LDS)
    int j = j0, r = r0;
    if (U[HW_FLAG] == 0.0f) {                               // workgroup-uniform
#pragma unroll 2
      for (int it = 0; it < GROUPS; ++it) {
        float ov[4], w[4];
        if (!fetch(it, ov)) break;
        window_group(U, j, r, w);
        ov[0] *= w[0]; ov[1] *= w[1]; ov[2] *= w[2]; ov[3] *= w[3];
        put(it, ov);
        j += 4;                                             // the next group: 1024 floats on = two rows and four taps
        r += 2;
        if (j >= NT) { j -= NT; r += 1; }
      }
    } else {                                                // half widths below 0.5, negative, infinite, NaN
#pragma unroll 1
      for (int it = 0; it < GROUPS; ++it) {
        float ov[4], w[4];
        if (!fetch(it, ov)) break;
        window_group_per_tap(U, j, r, w);
        ov[0] *= w[0]; ov[1] *= w[1]; ov[2] *= w[2]; ov[3] *= w[3];
        put(it, ov);
        j += 4;
        r += 2;
        if (j >= NT) { j -= NT; r += 1; }
      }
    }
  } else {
#pragma unroll 1
    for (int it = 0; it < GROUPS; ++it) {
      float ov[4];
      if (!fetch(it, ov)) break;
      put(it, ov);
    }
  }
  PFA_STAMP(4);
}

__global__ void __launch_bounds__(256, DDSP_PFA_WGS) k_taps_pfa510(TapsJobs jobs) {
  __shared__ __attribute__((aligned(16))) float U[pfa::LDS_FLOATS];
  taps_pfa510_body<false>(jobs, ExciterJob{}, U);
}

__global__ void __launch_bounds__(256, DDSP_PFA_WGS) k_front_small(TapsJobs jobs, ExciterJob exc) {
  __shared__ __attribute__((aligned(16))) float U[pfa::LDS_FLOATS];
  taps_pfa510_body<true>(jobs, exc, U);
}

// ------------------------------------------------------------------------------------------------
// Adjoint of the tap synthesis at 256 bins (what autograd returns for core.py:254-270 + the window helpers): d_taps
// [rows, 510] -> gradient of the one-sided response (or of the raw control through the exp activation).  The forward is
// window . roll . irfft, so the adjoint is a FORWARD real DFT of the windowed, un-rolled tap gradients,
//     D[k] = sum_m dz[m] exp(-2 pi i k m / 510),   dz[m] = w[j] d_taps[j],  j = (m + 255) mod 510,
//     d re_k = (c_k / N) Re D[k],   d im_k = (2 / N) Im D[k]  (0 at DC and Nyquist),
// through the same prime-factor maps and the same small transforms as k_taps_pfa510: two rows ride in one complex
// transform as conj(dz_a + i dz_b), Out = IDFT+(that) = conj(D_a + i D_b), and the two spectra are separated with
// Out[510 - k].  Replaces the dense MFMA contraction k_ir_gemm_bwd at n_mag = 256 (0.10-0.13 ms per launch there).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256, 4) k_taps_pfa510_bwd(TapsBwdJobs jobs, long rows) {
  using namespace pfa;
  const TapsBwdJob& J = jobs.j[blockIdx.y];                 // workgroup-uniform, like everything it selects
  const int ACT = J.act, HAS_IM = J.has_im, MODE = J.mode;
  const float* __restrict__ d_taps = J.d_taps;
  const float* __restrict__ ctrl = J.ctrl;
  const long ld_ctrl = J.ld_ctrl;
  const float scale = J.scale, hw_sr = J.hw_sr;
  const float* __restrict__ hann = J.hann;
  const float* __restrict__ half_width = J.half_width;
  float* __restrict__ d_re = J.d_re;
  float* __restrict__ d_im = J.d_im;
  __shared__ __attribute__((aligned(16))) float U[LDS_FLOATS];
  const int tid = threadIdx.x;
  const long row0 = (long)blockIdx.x * ROWS;
  float* Zf = U;                                            // stage 0 -> A: Z[t][m] complex, (row 2t, -row 2t+1)
  f32x2* W = reinterpret_cast<f32x2*>(U);                  // A -> B
  f32x2* Out = reinterpret_cast<f32x2*>(U);                // B -> C: Out[t][k] complex
  if (MODE == MODE_DYNAMIC && tid < 64) {                  // the batch's window rows (stage_window_rows), as the forward kernel's
    const long gr = row0 + tid;
    float x = (tid < ROWS && gr < rows) ? half_width[gr] : 1.0f;
    if (hw_sr > 0.f && tid < ROWS && gr < rows) x = (1.5f * hw_sr) / (x + 1e-3f);   // vocoder.py:851, the forward kernel's operations
    stage_window_rows(U, tid, x);
  }
  if (MODE == MODE_DYNAMIC) __syncthreads();
  const bool fast_window = MODE == MODE_DYNAMIC && U[HW_FLAG] == 0.0f;      // workgroup-uniform
  // the exp activation's derivative wants the raw control of this thread's bin in all 16 rows at the very end: fetched now, so
  // that stage C does not pay a memory latency per row pair (it was a third of this kernel's time: 46 us against 29 without)
  float cv[ROWS];
  if (ACT == 1) {
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const long gr = row0 + r;
      cv[r] = gr < rows ? ctrl[gr * ld_ctrl + tid] : 0.0f;
    }
  }

  // ---- stage 0: windowed tap gradients -> Z, un-rolled: m = (j + 255) mod 510.  The batch's 16 x 510 gradients are one
  // contiguous, 16-byte aligned stretch; a thread's groups of four are 1024 floats apart.  The window is the forward kernel's ----
  {
    const float* src = d_taps + row0 * NT;
    const long total = rows * (long)NT - row0 * NT;
    constexpr int GROUPS = (ROWS * NT / 4 + 255) / 256;     // 8; the last one is partial
    int r = (4 * tid) / NT, j = 4 * tid - r * NT;
#pragma unroll 2
    for (int it = 0; it < GROUPS; ++it) {
      const int i = 4 * (tid + 256 * it);
      if (i >= ROWS * NT) break;
      float4 gv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i + 3 < total) gv = *reinterpret_cast<const float4*>(src + i);
      else {
        if (i < total) gv.x = src[i];
        if (i + 1 < total) gv.y = src[i + 1];
        if (i + 2 < total) gv.z = src[i + 2];
      }
      const float ge[4] = {gv.x, gv.y, gv.z, gv.w};
      float we[4] = {1.0f, 1.0f, 1.0f, 1.0f};
      if (fast_window) window_group(U, j, r, we);
      else if (MODE == MODE_DYNAMIC) window_group_per_tap(U, j, r, we);
      else if (MODE == MODE_HANN) {                             // two 8-byte loads per group (j is even; a group straddles a row end only at j = 508)
        const float2 wa = *reinterpret_cast<const float2*>(hann + j);
        const float2 wb = *reinterpret_cast<const float2*>(hann + (j + 3 >= NT ? 0 : j + 2));
        we[0] = wa.x; we[1] = wa.y; we[2] = wb.x; we[3] = wb.y;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        int je = j + e, re_ = r;
        if (je >= NT) { je -= NT; re_ += 1; }
        const float w = we[e];
        int m = je + HALF;
        if (m >= NT) m -= NT;
        const float v = w * ge[e];
        Zf[2 * ((re_ >> 1) * NT + m) + (re_ & 1)] = (re_ & 1) ? -v : v;
      }
      j += 4;
      r += 2;
      if (j >= NT) { j -= NT; r += 1; }
    }
  }
  __syncthreads();

  // ---- stage A: DFT-17 over n1 of Z[(30 n1 + 17 n2) mod 510] ----
  {
    const bool act = tid < TR * 30;
    const int t = act ? tid / 30 : 0, n2 = act ? tid - 30 * t : 0;
    f32x2 z[17];
    {
      const f32x2* zr = reinterpret_cast<const f32x2*>(Zf) + t * NT;
      int k = 17 * n2;
#pragma unroll
      for (int n1 = 0; n1 < 17; ++n1) {
        z[n1] = zr[k];
        k += 30;
        if (k >= NT) k -= NT;
      }
    }
    __syncthreads();                                       // every thread holds its inputs: the region may be overwritten
    if (act) {
      dft17(z);
#pragma unroll
      for (int k1 = 0; k1 < 17; ++k1) W[w_row(t * 17 + k1) + n2] = z[k1];
    }
  }
  __syncthreads();

  // ---- stage B: DFT-30 over n2; output bin k = (120 k1 + 391 k2) mod 510 ----
  {
    const bool act = tid < TR * 17;
    const int t = act ? tid / 17 : 0, k1 = act ? tid - 17 * t : 0;
    f32x2 v[30];
    {
      const f32x2* src = W + w_row(t * 17 + k1);
#pragma unroll
      for (int n2 = 0; n2 < 30; ++n2) v[n2] = src[n2];
    }
    __syncthreads();                                       // W is in registers: the region becomes Out
    if (act) {
      dft30(v);
      f32x2* o = Out + t * NT;
      const int base = (120 * k1) % NT;
#pragma unroll
      for (int k2 = 0; k2 < 30; ++k2) {
        int k = base + (391 * k2) % NT;                     // the second term is a compile-time constant
        if (k >= NT) k -= NT;
        o[k] = v[k2];
      }
    }
  }
  __syncthreads();

  // ---- stage C with the all-pass activation's adjoint behind it (d_ap): the batch's (d re, d im) go to LDS row by row instead of
  // to memory, then wave w takes rows 4w .. 4w+3, a lane owns four consecutive bins (the forward kernel's stage 0 backwards:
  // theta_k = sum_{j<=k} pi tanh c_j in fixed-point revolutions, d theta_k = -sin theta_k d re_k + cos theta_k d im_k, and
  // d c_j = pi (1 - tanh^2 c_j) sum_{k>=j} d theta_k, the suffix sum in float64 as k_allpass_backward_256 forms it) ----
  if (J.d_ap) {                                             // workgroup-uniform
    const int k = tid;
    const int kn = k == 0 ? 0 : NT - k;
    const float inv_n = 1.0f / (float)NT;
    const float ce = (k == 0 || k == NB - 1) ? 0.5f * inv_n : inv_n;
    const float ci = (k == 0 || k == NB - 1) ? 0.0f : inv_n;
    float gre[ROWS], gim[ROWS];
#pragma unroll
    for (int t = 0; t < TR; ++t) {
      const f32x2 a = Out[t * NT + k], b = Out[t * NT + kn];
      gre[2 * t] = ce * (a.x + b.x); gre[2 * t + 1] = ce * (-a.y - b.y);
      gim[2 * t] = ci * (b.y - a.y); gim[2 * t + 1] = ci * (b.x - a.x);
    }
    const int wave = tid >> 6, lane = tid & 63;
    const float* __restrict__ apc = J.ap_ctrl;
    const long ld_ap = J.ld_ap;
    float4 cq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {                           // the wave's four control rows: in flight across the two barriers
      const long gr = row0 + wave * 4 + q;
      cq[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gr < rows && wave * 4 + q < ROWS) {
        const float* src = apc + gr * ld_ap + 4 * lane;
        cq[q] = make_float4(src[0], src[1], src[2], src[3]);
      }
    }
    __syncthreads();                                       // every thread holds its bins: Out becomes d re [16][256], d im [16][256]
#pragma unroll
    for (int r = 0; r < ROWS; ++r) { U[r * NB + k] = gre[r]; U[(ROWS + r) * NB + k] = gim[r]; }
    __syncthreads();
    const double rev_fx = 683565275.57643158978229477;      // 2^32 / (2 pi): the forward kernel's phase, bit for bit
    const double magic = 6755399441055744.0;                // 1.5 * 2^52
#pragma unroll
    for (int q = 0; q < 4; ++q) {                           // (unrolled: cq[q] stays in registers)
      const int r = wave * 4 + q;
      const long gr = row0 + r;
      if (r >= ROWS || gr >= rows) break;                  // wave-uniform
      const float4 rv = *reinterpret_cast<const float4*>(U + r * NB + 4 * lane);
      const float4 iv = *reinterpret_cast<const float4*>(U + (ROWS + r) * NB + 4 * lane);
      const float dr[4] = {rv.x, rv.y, rv.z, rv.w}, di[4] = {iv.x, iv.y, iv.z, iv.w};
      const float th[4] = {tanh_hw(cq[q].x), tanh_hw(cq[q].y), tanh_hw(cq[q].z), tanh_hw(cq[q].w)};
      unsigned s[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const double m = fma((double)(kPiF * th[e]), rev_fx, magic);
        const unsigned fx = (unsigned)__builtin_bit_cast(unsigned long long, m);
        s[e] = e ? s[e - 1] + fx : fx;
      }
      const unsigned before = wave_incl_scan_u32(s[3]) - s[3];
      float dth[4];
      double dth_sum = 0.0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float fr = (float)(int)(before + s[e]) * 2.3283064365386963e-10f;
        const float co = __builtin_amdgcn_cosf(fr), si = __builtin_amdgcn_sinf(fr);
        dth[e] = fmaf(-si, dr[e], co * di[e]);
        dth_sum += (double)dth[e];
      }
      const double incl_before = wave_excl_scan(dth_sum, lane);
      const double total = wave_sum(dth_sum);
      double suffix = total - incl_before - dth_sum;
      float o[4];
#pragma unroll
      for (int e = 3; e >= 0; --e) {
        suffix += (double)dth[e];
        o[e] = (float)suffix * (kPiF * (1.0f - th[e] * th[e]));
      }
      *reinterpret_cast<float4*>(J.d_ap + gr * NB + 4 * lane) = make_float4(o[0], o[1], o[2], o[3]);
    }
    return;
  }

  // ---- stage C: separate the two rows of a transform, scale, activation derivative, coalesced stores (thread = bin) ----
  {
    const int k = tid;                                     // 0..255
    const int kn = k == 0 ? 0 : NT - k;
    const float inv_n = 1.0f / (float)NT;
    const float ce = (k == 0 || k == NB - 1) ? 0.5f * inv_n : inv_n;     // c_k / 2N
    const float ci = (k == 0 || k == NB - 1) ? 0.0f : inv_n;             // 2 / 2N, Im(DC) = Im(Nyquist) = 0
#pragma unroll
    for (int t = 0; t < TR; ++t) {
      const f32x2 a = Out[t * NT + k], b = Out[t * NT + kn];
      // D_a = (conj a + b) / 2, D_b = (conj a - b) / 2i
      float gre[2] = {ce * (a.x + b.x), ce * (-a.y - b.y)};
      const float gim[2] = {ci * (b.y - a.y), ci * (b.x - a.x)};
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const long gr = row0 + 2 * t + s2;
        if (gr >= rows) continue;
        float g = gre[s2];
        if (ACT == 1) g = g * (scale * exp_hw(cv[2 * t + s2]));            // d exp(c) = exp(c): the forward kernel's exponential
        d_re[gr * NB + k] = g;
        if (HAS_IM) d_im[gr * NB + k] = gim[s2];
      }
    }
  }
}

// 0 = taken, -1 = not this kernel's shape (the caller uses the dense adjoint)
int launch_taps_pfa510_bwd(const float* d_taps, const float* ctrl, long ld_ctrl, int act, float scale, const float* table,
                           int mode, const float* half_width, long rows, int n, int has_im, float* d_re, float* d_im,
                           hipStream_t st) {
  if (n != pfa::NB || rows <= 0) return -1;
  if ((reinterpret_cast<uintptr_t>(d_taps) & 15) != 0) return -1;
  const long KP = ((long)n + 15) / 16 * 16, NP = ((long)n + 255) / 256 * 256;
  const float* hann = table + 2 * KP * NP;
  const int m = mode == pfa::MODE_HANN ? pfa::MODE_HANN : (mode == pfa::MODE_DYNAMIC ? pfa::MODE_DYNAMIC : pfa::MODE_ROLL);
  TapsBwdJobs jobs;
  jobs.n = 1;
  jobs.j[0] = TapsBwdJob{act == 1 ? 1 : 0, has_im ? 1 : 0, m, d_taps, ctrl, ld_ctrl, scale, hann, half_width, 0.f, d_re, d_im, nullptr, 0, nullptr};
  jobs.j[1] = jobs.j[2] = jobs.j[0];
  hipLaunchKernelGGL(k_taps_pfa510_bwd, dim3((unsigned)((rows + pfa::ROWS - 1) / pfa::ROWS)), dim3(256), 0, st, jobs, rows);
  return 0;
}

int launch_taps_pfa510_bwd_jobs(const TapsBwdJobs& in, const float* table, long rows, hipStream_t st) {
  if (in.n < 1 || in.n > 3 || rows <= 0) return -1;
  const long KP = ((long)pfa::NB + 15) / 16 * 16, NP = ((long)pfa::NB + 255) / 256 * 256;
  TapsBwdJobs jobs = in;
  for (int i = 0; i < jobs.n; ++i) {
    if ((reinterpret_cast<uintptr_t>(jobs.j[i].d_taps) & 15) != 0) return -1;
    if (jobs.j[i].d_ap && (!jobs.j[i].has_im || !jobs.j[i].ap_ctrl || (reinterpret_cast<uintptr_t>(jobs.j[i].d_ap) & 15) != 0)) return -1;
    jobs.j[i].hann = table + 2 * KP * NP;                   // the periodic Hann of the basis table (k_ir_table, ir.hip)
  }
  for (int i = jobs.n; i < 3; ++i) jobs.j[i] = jobs.j[0];
  hipLaunchKernelGGL(k_taps_pfa510_bwd, dim3((unsigned)((rows + pfa::ROWS - 1) / pfa::ROWS), (unsigned)jobs.n), dim3(256), 0, st, jobs, rows);
  return 0;
}

// returns 0 when the fast form took the call, -1 when the shape is not its (the caller then uses the dense contraction)
int launch_taps_pfa510(const float* a_re, long ld_re, const float* a_im, long ld_im, int allpass_from_control, int act,
                       float scale, const float* table, int mode, const float* half_width, long rows, int n, float* taps,
                       hipStream_t st, float hw_from_f0_sr, TapsJobs* batch, int half_rows) {
  if (n != pfa::NB || rows <= 0) return -1;                 // (knob TAPS_GEMM is the caller's decision: read once per API call)
  if ((reinterpret_cast<uintptr_t>(taps) & 15) != 0) return -1;
  const long KP = ((long)n + 15) / 16 * 16, NP = ((long)n + 255) / 256 * 256;
  const float* hann = table + 2 * KP * NP;                 // the periodic Hann of the basis table (k_ir_table, ir.hip)
  dim3 grid((unsigned)((rows + pfa::ROWS - 1) / pfa::ROWS)), block(256);
  int kind = pfa::KIND_REAL;
  if (allpass_from_control) {
    kind = pfa::KIND_ALLPASS;
    act = 0;
  } else if (a_im) {
    if (act != 0) return -1;
    kind = pfa::KIND_COMPLEX;
  }
  const int m = mode == pfa::MODE_HANN ? pfa::MODE_HANN : (mode == pfa::MODE_DYNAMIC ? pfa::MODE_DYNAMIC : pfa::MODE_ROLL);
#ifdef DDSP_PFA_NOSYM
  if (half_rows) return -1;
#endif
  if (half_rows && (kind != pfa::KIND_REAL || m == pfa::MODE_DYNAMIC)) return -1;   // a zero-phase response under an even window (the dynamic one clamps one side only, core.py:245)
  TapsJob job{kind, act == 1 ? 1 : 0, m, a_re, ld_re, a_im, ld_im, scale, hann, half_width, hw_from_f0_sr, rows, taps, half_rows ? 1 : 0};
  if (batch) {                                              // collected, launched by launch_taps_pfa510_batch (all jobs: the same row count)
    if (batch->n >= 3 || (batch->n > 0 && batch->j[0].rows != rows)) return -1;
    batch->j[batch->n++] = job;
    return 0;
  }
  TapsJobs jobs;
  jobs.j[0] = job;
  jobs.n = 1;
  hipLaunchKernelGGL(k_taps_pfa510, grid, block, 0, st, jobs);
  return 0;
}

int launch_taps_pfa510_batch(const TapsJobs& jobs, hipStream_t st, const ExciterJob* exciter) {
  if (jobs.n < 1) return 0;
  const long rows = jobs.j[0].rows;
  const unsigned gx = (unsigned)((rows + pfa::ROWS - 1) / pfa::ROWS);
  if (exciter) {
    if (exciter->n_frames != rows || exciter->hop != 512 || exciter->up.shift <= 0) return -1;
    hipLaunchKernelGGL(k_front_small, dim3(gx, (unsigned)jobs.n + 1u), dim3(256), 0, st, jobs, *exciter);
  } else {
    hipLaunchKernelGGL(k_taps_pfa510, dim3(gx, (unsigned)jobs.n), dim3(256), 0, st, jobs);
  }
  return 0;
}

}  // namespace ddsp

#ifdef DDSP_HIP_TIMELINE
extern "C" int ddsp_hip_debug_set_pfa_timeline(long long* p, void* stream) {
  hipLaunchKernelGGL(ddsp::k_set_pfa_timeline, dim3(1), dim3(1), 0, (hipStream_t)stream, p);
  return (int)hipGetLastError();
}
#endif
