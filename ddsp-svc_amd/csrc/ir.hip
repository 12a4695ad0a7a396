// HOT-2b: frequency response -> per-frame FIR taps  (reference: ddsp/core.py:254-270 with the
// window helpers :185-251, fed by the activations of ddsp/vocoder.py:580-582,599 / :834-836,845).
//
// torch.fft.irfft of a one-sided response with n bins is a fixed linear map R^n (x R^n) -> R^N,
// N = 2(n-1): the Hermitian synthesis sum.  Here it is a dense contraction of the [B*F, n]
// control matrix with a precomputed [n, n] cosine (and sine) basis -- genuinely GEMM-shaped, so
// it runs on the f32 MFMA pipe (v_mfma_f32_32x32x2_f32: exact f32 fmaf chains at the vector
// rate), with the activation (exp / exp/128 / all-pass cos,sin) fused into the A-operand staging
// and the roll + window + mirror (real responses give symmetric taps, so only m = 0..N/2 is
// contracted) fused into the epilogue.
#include "ddsp_common.h"

namespace ddsp {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { IR_MODE_ROLL = 0, IR_MODE_HANN = 1, IR_MODE_DYNAMIC = 2 };
enum { IR_ACT_NONE = 0, IR_ACT_EXP = 1 };

// ------------------------------------------------------------------------------------------------
// basis table, built once per n_mag (caller caches it):
//   TE[k][m] =  (c_k/N) cos(2 pi k m / N)   c_0 = c_{n-1} = 1, else 2           k,m in [0,n)
//   TO[k][m] = -(2/N)   sin(2 pi k m / N)   rows 0 and n-1 are zero: irfft ignores Im(DC), Im(Nyquist)
//   HANN[j]  = 0.5 - 0.5 cos(2 pi j / N)    periodic Hann, j in [0,N)
// layout: TE (n*n floats) | TO (n*n floats) | HANN (N floats)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_ir_table(int n, float* __restrict__ table) {
  const long N = 2L * (n - 1);
  const long nn = (long)n * n;
  const long total = 2 * nn + N;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    float v;
    if (i < 2 * nn) {
      const bool odd = i >= nn;
      const long e = odd ? i - nn : i;
      const long k = e / n, m = e % n;
      const double frac = 2.0 * (double)((k * m) % N) / (double)N;       // angle / pi, reduced exactly
      const bool edge = (k == 0) || (k == n - 1);
      if (!odd) v = (float)((edge ? 1.0 : 2.0) / (double)N * cospi(frac));
      else v = edge ? 0.0f : (float)(-2.0 / (double)N * sinpi(frac));
    } else {
      const long j = i - 2 * nn;
      v = (float)(0.5 - 0.5 * cospi(2.0 * (double)j / (double)N));
    }
    table[i] = v;
  }
}

// ------------------------------------------------------------------------------------------------
// all-pass response from the raw group-delay control: theta = cumsum(pi*tanh(c)) over bins,
// (cos theta, sin theta).  One wave per frame; float64 scan, reduced mod 2pi before the float sincos.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_allpass_response(const float* __restrict__ c, long ld, long rows, int n,
                                                          float* __restrict__ re, float* __restrict__ im) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * 4 + wave;
  if (r >= rows) return;
  const float* cr = c + r * ld;
  const int per = (n + 63) / 64;
  const int k0 = lane * per;
  double local = 0.0;
  for (int q = 0; q < per; ++q) {
    int k = k0 + q;
    if (k < n) local += (double)(kPiF * tanhf(cr[k]));
  }
  double run = wave_excl_scan(local, lane);
  for (int q = 0; q < per; ++q) {
    int k = k0 + q;
    if (k < n) {
      run += (double)(kPiF * tanhf(cr[k]));
      double red = run - (2.0 * kPiD) * rint(run / (2.0 * kPiD));
      float s, co;
      sincosf((float)red, &s, &co);
      re[r * n + k] = co;
      im[r * n + k] = s;
    }
  }
}

// vocoder.py:851  half_width_frames = 1.5 * sr / (f0_frames + 1e-3)
__global__ void __launch_bounds__(256) k_half_width(const float* __restrict__ f0_frames, long rows, float sr,
                                                    float* __restrict__ hw) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < rows) hw[i] = (1.5f * sr) / (f0_frames[i] + 1e-3f);
}

// ------------------------------------------------------------------------------------------------
// taps = window( roll( X * T ) ):  64x64 output tile per workgroup, 4 waves of one 32x32 MFMA tile
// each, K streamed through LDS in chunks of 32.
//   a_re / a_im : [rows, n] with row stride ld_* (raw control if ACT_EXP, else the response itself)
//   E = sum_k re_k TE[k][m],  O = sum_k im_k TO[k][m];  zero-phase taps: z[m] = E+O, z[N-m] = E-O
//   causal form: taps[N/2 + m] = z[m] (m < N/2), taps[N/2 - m] = z[N-m] (m >= 1)   (roll by N/2)
// ------------------------------------------------------------------------------------------------
constexpr int GM = 64, GN = 64, GK = 32;
constexpr int LDA = GM + 1;    // A is stored k-major ([k][row]); +1 keeps the transposing stores conflict-free
constexpr int LDB = GN;

template <int ACT>
__device__ __forceinline__ float ir_activate(float v, float scale) {
  if (ACT == IR_ACT_EXP) return expf(v) * scale;
  return v * scale;
}

template <int ACT, bool HAS_IM>
__global__ void __launch_bounds__(256) k_ir_gemm(const float* __restrict__ a_re, long ld_re,
                                                 const float* __restrict__ a_im, long ld_im, float scale,
                                                 const float* __restrict__ table, int mode,
                                                 const float* __restrict__ half_width, long rows, int n,
                                                 float* __restrict__ taps) {
  __shared__ float As[GK * LDA];
  __shared__ float Bs[GK * LDB];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, l = tid & 63;
  const int wr = wave >> 1, wc = wave & 1;
  const long row0 = (long)blockIdx.x * GM;
  const int col0 = blockIdx.y * GN;
  const int N = 2 * (n - 1);
  const long nn = (long)n * n;

  f32x16 accE, accO;
#pragma unroll
  for (int i = 0; i < 16; ++i) { accE[i] = 0.f; accO[i] = 0.f; }

  const int a_kk = tid & 31, a_r0 = tid >> 5;          // A staging: 32 consecutive k per row, 8 rows per sweep
  const int b_mm = tid & 63, b_k0 = tid >> 6;          // B staging: 64 consecutive m per k, 4 k per sweep

  for (int part = 0; part < (HAS_IM ? 2 : 1); ++part) {
    const float* A = part ? a_im : a_re;
    const long ld = part ? ld_im : ld_re;
    const float* Tb = table + (part ? nn : 0);
    for (int k0 = 0; k0 < n; k0 += GK) {
      __syncthreads();                                  // previous chunk fully consumed
#pragma unroll
      for (int q = 0; q < GM / 8; ++q) {
        const int rr = a_r0 + 8 * q;
        const long r = row0 + rr;
        const int k = k0 + a_kk;
        float v = 0.f;
        if (r < rows && k < n) v = ir_activate<ACT>(A[r * ld + k], scale);
        As[a_kk * LDA + rr] = v;
      }
#pragma unroll
      for (int q = 0; q < GK / 4; ++q) {
        const int kk = b_k0 + 4 * q;
        const int k = k0 + kk;
        const int m = col0 + b_mm;
        Bs[kk * LDB + b_mm] = (k < n && m < n) ? Tb[(long)k * n + m] : 0.f;
      }
      __syncthreads();
#pragma unroll
      for (int ks = 0; ks < GK; ks += 2) {
        const float av = As[(ks + (l >> 5)) * LDA + wr * 32 + (l & 31)];
        const float bv = Bs[(ks + (l >> 5)) * LDB + wc * 32 + (l & 31)];
        if (part == 0) accE = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, accE, 0, 0, 0);
        else accO = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, accO, 0, 0, 0);
      }
    }
  }

  // epilogue: roll + mirror + window.  C/D layout: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5)
  const int m = col0 + wc * 32 + (l & 31);
  const int half = N / 2;
  const float* hann = table + 2 * nn;
  if (m > half) return;                                 // also covers m >= n
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const long r = row0 + wr * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5);
    if (r >= rows) continue;
    const float E = accE[reg];
    const float O = HAS_IM ? accO[reg] : 0.f;
    const float hw = (mode == IR_MODE_DYNAMIC) ? half_width[r] : 1.f;
    float* dst = taps + r * (long)N;
#pragma unroll
    for (int side = 0; side < 2; ++side) {
      if (side == 0 && m >= half) continue;             // z[N/2] lands only at taps[0]
      if (side == 1 && m == 0) continue;                // z[0] lands only at taps[N/2]
      const int j = side == 0 ? half + m : half - m;
      const float z = side == 0 ? E + O : E - O;
      float w = 1.f;
      if (mode == IR_MODE_HANN) {
        w = hann[j];
      } else if (mode == IR_MODE_DYNAMIC) {
        float u = (float)(j - half) / hw;               // core.py:244
        if (u > 1.0f) u = 0.0f;                         // core.py:245 -- only the upper side is clamped
        w = (1.0f + cosf(kPiF * u)) / 2.0f;             // core.py:246
      }
      dst[j] = z * w;
    }
  }
}

// ---- launchers -----------------------------------------------------------------------------------
size_t ir_table_floats(int n) { return (size_t)2 * n * n + 2 * (size_t)(n - 1); }

void launch_ir_table(int n, float* table, hipStream_t st) {
  long total = (long)ir_table_floats(n);
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_ir_table, dim3((unsigned)blocks), dim3(256), 0, st, n, table);
}

void launch_allpass_response(const float* c, long ld, long rows, int n, float* re, float* im, hipStream_t st) {
  if (rows == 0) return;
  hipLaunchKernelGGL(k_allpass_response, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, c, ld, rows, n, re, im);
}

void launch_half_width(const float* f0_frames, long rows, float sr, float* hw, hipStream_t st) {
  if (rows == 0) return;
  hipLaunchKernelGGL(k_half_width, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, f0_frames, rows, sr, hw);
}

void launch_ir_gemm(const float* a_re, long ld_re, const float* a_im, long ld_im, int act, float scale,
                    const float* table, int mode, const float* half_width, long rows, int n, float* taps,
                    hipStream_t st) {
  if (rows == 0) return;
  dim3 grid((unsigned)((rows + GM - 1) / GM), (unsigned)((n + GN - 1) / GN)), block(256);
  if (a_im) {
    if (act == IR_ACT_EXP)
      hipLaunchKernelGGL((k_ir_gemm<IR_ACT_EXP, true>), grid, block, 0, st, a_re, ld_re, a_im, ld_im, scale, table, mode, half_width, rows, n, taps);
    else
      hipLaunchKernelGGL((k_ir_gemm<IR_ACT_NONE, true>), grid, block, 0, st, a_re, ld_re, a_im, ld_im, scale, table, mode, half_width, rows, n, taps);
  } else {
    if (act == IR_ACT_EXP)
      hipLaunchKernelGGL((k_ir_gemm<IR_ACT_EXP, false>), grid, block, 0, st, a_re, ld_re, a_im, ld_im, scale, table, mode, half_width, rows, n, taps);
    else
      hipLaunchKernelGGL((k_ir_gemm<IR_ACT_NONE, false>), grid, block, 0, st, a_re, ld_re, a_im, ld_im, scale, table, mode, half_width, rows, n, taps);
  }
}

}  // namespace ddsp
