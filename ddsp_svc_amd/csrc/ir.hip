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
// layout: TE | TO | HANN; TE and TO are stored [KP][NP] with KP = n rounded up to the k-chunk (16) and
// NP = n rounded up to the column tile (256), zero outside [0,n) x [0,n), so the contraction kernel
// streams them with unguarded 16-byte loads.
// ------------------------------------------------------------------------------------------------
constexpr int GM = 64, GN = 256, KC = 16;
__host__ __device__ inline long ir_kp(int n) { return ((long)n + KC - 1) / KC * KC; }
__host__ __device__ inline long ir_np(int n) { return ((long)n + GN - 1) / GN * GN; }

__global__ void __launch_bounds__(256) k_ir_table(int n, float* __restrict__ table) {
  const long N = 2L * (n - 1);
  const long KP = ir_kp(n), NP = ir_np(n);
  const long plane = KP * NP;
  const long total = 2 * plane + N;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    float v = 0.f;
    if (i < 2 * plane) {
      const bool odd = i >= plane;
      const long e = odd ? i - plane : i;
      const long k = e / NP, m = e % NP;
      if (k < n && m < n) {
        const double frac = 2.0 * (double)((k * m) % N) / (double)N;     // angle / pi, reduced exactly
        const bool edge = (k == 0) || (k == n - 1);
        if (!odd) v = (float)((edge ? 1.0 : 2.0) / (double)N * cospi(frac));
        else v = edge ? 0.0f : (float)(-2.0 / (double)N * sinpi(frac));
      }
    } else {
      const long j = i - 2 * plane;
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
                                                          float* __restrict__ re, float* __restrict__ im, int vec4) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * 4 + wave;
  if (r >= rows) return;
  const float* cr = c + r * ld;
  const double inv_2pi = 0.15915494309189533577;
  // exp(1j * theta): theta reduced to revolutions in float64, then the hardware sine / cosine (abs error <= 4e-7
  // on a unit-magnitude response)
  auto cis = [&](double run, float& co, float& si) {
    const double rev = run * inv_2pi;
    const float fr = (float)(rev - rint(rev));
    co = __builtin_amdgcn_cosf(fr);
    si = __builtin_amdgcn_sinf(fr);
  };
  if (vec4) {
    // n == 256 with 16-byte aligned rows: a lane owns 4 consecutive bins -- one 16-byte load, two 16-byte stores
    const float4 cv = *reinterpret_cast<const float4*>(cr + 4 * lane);
    const float g0 = kPiF * tanhf(cv.x), g1 = kPiF * tanhf(cv.y), g2 = kPiF * tanhf(cv.z), g3 = kPiF * tanhf(cv.w);
    const double local = (((double)g0 + (double)g1) + (double)g2) + (double)g3;   // same order as the scalar path
    double run = wave_excl_scan(local, lane);
    float4 co, si;
    run += (double)g0; cis(run, co.x, si.x);
    run += (double)g1; cis(run, co.y, si.y);
    run += (double)g2; cis(run, co.z, si.z);
    run += (double)g3; cis(run, co.w, si.w);
    *reinterpret_cast<float4*>(re + r * n + 4 * lane) = co;
    *reinterpret_cast<float4*>(im + r * n + 4 * lane) = si;
    return;
  }
  const int per = (n + 63) / 64;
  const int k0 = lane * per;
  double local = 0.0;
  for (int q = 0; q < per; ++q) {
    const int k = k0 + q;
    if (k < n) local += (double)(kPiF * tanhf(cr[k]));          // vocoder.py:581 / :834
  }
  double run = wave_excl_scan(local, lane);
  for (int q = 0; q < per; ++q) {
    const int k = k0 + q;
    if (k < n) {
      run += (double)(kPiF * tanhf(cr[k]));
      float co, si;
      cis(run, co, si);
      re[r * n + k] = co;
      im[r * n + k] = si;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// taps = window( roll( X * T ) ).  Workgroup tile = 64 frames x 256 taps-columns (all of m = 0..N/2 for
// n_mag <= 256, so every control row is read from HBM exactly once), 4 waves of 64x64 = 2x2 MFMA
// 32x32x2 accumulators each; K is streamed in chunks of 16 through a double-buffered LDS stage: the
// global loads of chunk c+1 are in flight (registers) while chunk c is contracted, one barrier per chunk.
//   a_re / a_im : [rows, n] with row stride ld_* (raw control if ACT_EXP, else the response itself)
//   E = sum_k re_k TE[k][m],  O = sum_k im_k TO[k][m];  zero-phase taps: z[m] = E+O, z[N-m] = E-O
//   causal form: taps[N/2 + m] = z[m] (m < N/2), taps[N/2 - m] = z[N-m] (m >= 1)   (roll by N/2)
// Within a chunk the MFMA k index is permuted (step s of lane-half h takes k = 8h + s) so a lane's eight
// A values are two contiguous 16-byte LDS reads.
// ------------------------------------------------------------------------------------------------
constexpr int LDA = 20;          // floats per A row in LDS (16 + pad; rows stay 16-byte aligned)
constexpr int LDB = GN + 4;      // floats per B row in LDS: rows 8 apart land 32 banks apart
constexpr int IR_STAGE = GM * LDA + KC * LDB;     // floats per pipeline stage

template <int ACT>
__device__ __forceinline__ float ir_activate(float v, float scale) {
  if (ACT == IR_ACT_EXP) return expf(v) * scale;
  return v * scale;
}

// cos(a) for the dynamic window, |a| < ~10: same reduction as sin_turns, hardware cosine (revolutions)
__device__ __forceinline__ float cos_turns(float a) {
  const float inv_hi = 0.15915494f, inv_lo = 6.4206383e-9f;
  float nn = rintf(a * inv_hi);
  float r = fmaf(a, inv_hi, -nn);
  r = fmaf(a, inv_lo, r);
  return __builtin_amdgcn_cosf(r);
}

template <int ACT, bool HAS_IM, int MODE>
__global__ void __launch_bounds__(256, 2) k_ir_gemm(const float* __restrict__ a_re, long ld_re,
                                                 const float* __restrict__ a_im, long ld_im, float scale,
                                                 const float* __restrict__ table,
                                                 const float* __restrict__ half_width, long rows, int n,
                                                 float* __restrict__ taps, int a_vec_ok, float hw_sr) {
  __shared__ __attribute__((aligned(16))) float stage[2 * IR_STAGE];
  const int tid = threadIdx.x;
  const int wc = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave = 64-column group
  const int l = tid & 63;
  const int li = l & 31, h = l >> 5;
  const long row0 = (long)blockIdx.x * GM;
  const int col0 = blockIdx.y * GN;
  const int N = 2 * (n - 1);
  const int KP = (int)ir_kp(n);
  const long NP = ir_np(n);
  const long plane = (long)KP * NP;
  const int nch = KP / KC;
  const int total_ch = HAS_IM ? 2 * nch : nch;

  f32x16 accE[2][2], accO[2][2];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int i = 0; i < 16; ++i) { accE[rb][cb][i] = 0.f; accO[rb][cb][i] = 0.f; }

  // staging roles: A -- thread (row = tid/4, 4 consecutive k); B -- thread (k = tid/64 + 4j, 4 consecutive columns)
  const int a_row = tid >> 2, a_kq = (tid & 3) * 4;
  const int b_k = tid >> 6, b_c4 = (tid & 63) * 4;
  struct Chunk { float4 a, b0, b1, b2, b3; };

  auto fetch = [&](int cc) -> Chunk {
    const int part = (HAS_IM && cc >= nch) ? 1 : 0;
    const int k0 = (cc - part * nch) * KC;
    const float* A = part ? a_im : a_re;
    const long ld = part ? ld_im : ld_re;
    const long r = row0 + a_row;
    const int k = k0 + a_kq;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < rows) {
      const float* src = A + r * ld + k;
      if (a_vec_ok && k + 3 < n) {
        v = *reinterpret_cast<const float4*>(src);
      } else {
        if (k < n) v.x = src[0];
        if (k + 1 < n) v.y = src[1];
        if (k + 2 < n) v.z = src[2];
        if (k + 3 < n) v.w = src[3];
      }
    }
    Chunk c;
    c.a = v;
    const float* Tb = table + (part ? plane : 0) + (long)(k0 + b_k) * NP + col0 + b_c4;
    c.b0 = *reinterpret_cast<const float4*>(Tb);
    c.b1 = *reinterpret_cast<const float4*>(Tb + 4 * NP);
    c.b2 = *reinterpret_cast<const float4*>(Tb + 8 * NP);
    c.b3 = *reinterpret_cast<const float4*>(Tb + 12 * NP);
    return c;
  };
  // the activation is applied here, after the MFMAs of the previous chunk, so the contraction never waits on
  // the global load it overlaps.  Out-of-range k stay exactly zero (exp(garbage)*0 must not become NaN).
  auto park = [&](int buf, int cc, const Chunk& c) {
    float* As = stage + buf * IR_STAGE;
    float* Bs = As + GM * LDA;
    const int part = (HAS_IM && cc >= nch) ? 1 : 0;
    const int k = (cc - part * nch) * KC + a_kq;
    const bool live = row0 + a_row < rows;
    float4 v;
    v.x = (live && k < n) ? ir_activate<ACT>(c.a.x, scale) : 0.f;
    v.y = (live && k + 1 < n) ? ir_activate<ACT>(c.a.y, scale) : 0.f;
    v.z = (live && k + 2 < n) ? ir_activate<ACT>(c.a.z, scale) : 0.f;
    v.w = (live && k + 3 < n) ? ir_activate<ACT>(c.a.w, scale) : 0.f;
    *reinterpret_cast<float4*>(As + a_row * LDA + a_kq) = v;
    *reinterpret_cast<float4*>(Bs + (b_k + 0) * LDB + b_c4) = c.b0;
    *reinterpret_cast<float4*>(Bs + (b_k + 4) * LDB + b_c4) = c.b1;
    *reinterpret_cast<float4*>(Bs + (b_k + 8) * LDB + b_c4) = c.b2;
    *reinterpret_cast<float4*>(Bs + (b_k + 12) * LDB + b_c4) = c.b3;
  };
  auto contract = [&](int buf, f32x16 (&acc)[2][2]) {
    const float* As = stage + buf * IR_STAGE;
    const float* Bs = As + GM * LDA;
    float av[2][8];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const float4 lo = *reinterpret_cast<const float4*>(As + (rb * 32 + li) * LDA + 8 * h);
      const float4 hi = *reinterpret_cast<const float4*>(As + (rb * 32 + li) * LDA + 8 * h + 4);
      av[rb][0] = lo.x; av[rb][1] = lo.y; av[rb][2] = lo.z; av[rb][3] = lo.w;
      av[rb][4] = hi.x; av[rb][5] = hi.y; av[rb][6] = hi.z; av[rb][7] = hi.w;
    }
    // all B operands of the chunk are read before the first MFMA (16 registers): with one read per step the
    // compiler re-uses two registers and every group of four MFMAs waits for its LDS read
    const float* bp = Bs + (8 * h) * LDB + wc * 64 + li;
    float bv[8][2];
#pragma unroll
    for (int sidx = 0; sidx < 8; ++sidx) {
      bv[sidx][0] = bp[sidx * LDB];
      bv[sidx][1] = bp[sidx * LDB + 32];
    }
    __builtin_amdgcn_sched_barrier(0);                // keep the reads above the MFMAs (the scheduler sinks them otherwise)
#pragma unroll
    for (int sidx = 0; sidx < 8; ++sidx) {
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0][sidx], bv[sidx][0], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0][sidx], bv[sidx][1], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1][sidx], bv[sidx][0], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1][sidx], bv[sidx][1], acc[1][1], 0, 0, 0);
    }
  };

  Chunk nxt = fetch(0);
  park(0, 0, nxt);
  __syncthreads();
  for (int cc = 0; cc < total_ch; ++cc) {
    const bool more = cc + 1 < total_ch;
    if (more) nxt = fetch(cc + 1);
    if (!HAS_IM || cc < nch) contract(cc & 1, accE);
    else contract(cc & 1, accO);
    if (more) park((cc + 1) & 1, cc + 1, nxt);
    __syncthreads();
  }

  // epilogue: roll + mirror + window.  C/D layout: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5).
  //   z[m] lands at taps[N/2 + m] (side 0, m < N/2) and z[N - m] at taps[N/2 - m] (side 1, m >= 1); z[N/2] only at taps[0]
  // Everything a store needs besides the accumulator is fetched up front -- the window value of the lane's
  // columns (periodic Hann) or the half width of the lane's rows (dynamic window) -- so the stores stream out
  // back to back: a load between two stores would wait for every store before it (one counter tracks both).
  const int half = N / 2;
  const float* hann = table + 2 * plane;
  // full tiles take a branch-free path: the two columns without a partner (m = 0 has no side 1, m = N/2 no side 0)
  // store their one value twice instead of being masked
  const bool full = row0 + GM <= rows && col0 + GN - 1 <= half;                  // workgroup-uniform
  int jcol[2][2];                                     // [cb][side] output column; -1 = nothing to store (masked path)
  bool flip[2][2];                                    // store the other side's value (full path duplicates)
  float wcol[2][2];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    const int m = col0 + wc * 64 + cb * 32 + li;
#pragma unroll
    for (int side = 0; side < 2; ++side) {
      const bool ok = m <= half && (side == 0 ? m < half : m >= 1);
      int j = side == 0 ? half + m : half - m;
      flip[cb][side] = false;
      if (!ok) {
        if (full) { j = side == 0 ? half - m : half + m; flip[cb][side] = true; }   // m = N/2 -> taps[0]; m = 0 -> taps[N/2]
        else j = -1;
      }
      jcol[cb][side] = j;
      wcol[cb][side] = (MODE == IR_MODE_HANN && j >= 0) ? hann[j] : 1.f;
    }
  }
  auto emit = [&](float* dst, float E, float O, float hw, int cb, int side) {
    const int j = jcol[cb][side];
    const bool minus = (side == 1) != flip[cb][side];
    const float z = minus ? E - O : E + O;
    float w = wcol[cb][side];
    if (MODE == IR_MODE_DYNAMIC) {
      float u = (float)(j - half) / hw;               // core.py:244
      if (u > 1.0f) u = 0.0f;                         // core.py:245 -- only the upper side is clamped
      w = (1.0f + cos_turns(kPiF * u)) / 2.0f;        // core.py:246
    }
    dst[j] = z * w;
  };
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    float hwr[16];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const long r = row0 + rb * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
      float hv = (MODE == IR_MODE_DYNAMIC && r < rows) ? half_width[r] : 1.f;
      // hw_sr > 0: `half_width` holds f0 and the width is formed here, vocoder.py:851 (same float32 operations as a separate pass would do)
      if (MODE == IR_MODE_DYNAMIC && hw_sr > 0.f) hv = (1.5f * hw_sr) / (hv + 1e-3f);
      hwr[reg] = hv;
    }
    if (full) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        float* dst = taps + (row0 + rb * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h) * (long)N;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int side = 0; side < 2; ++side)
            emit(dst, accE[rb][cb][reg], HAS_IM ? accO[rb][cb][reg] : 0.f, hwr[reg], cb, side);
      }
    } else {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const long r = row0 + rb * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        if (r >= rows) continue;
        float* dst = taps + r * (long)N;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int side = 0; side < 2; ++side)
            if (jcol[cb][side] >= 0) emit(dst, accE[rb][cb][reg], HAS_IM ? accO[rb][cb][reg] : 0.f, hwr[reg], cb, side);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// apply_window_to_impulse_response / apply_dynamic_window_to_impulse_response (core.py:185-251) on taps that are
// already in the time domain: zero-phase form in, windowed causal form out,
//     out[r][j] = in[r][(j - N/2) mod N] * w_r(j),
// w = periodic Hann of N (MODE_HANN), the f0-dependent raised cosine with its one-sided clamp (MODE_DYNAMIC,
// core.py:244-246; N may be odd there: positions run from -(N/2) to (N+1)/2 - 1), or 1 (MODE_ROLL).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_window_taps(const float* __restrict__ in, int mode,
                                                     const float* __restrict__ half_width, long rows, int N,
                                                     float* __restrict__ out) {
  const long total = rows * (long)N;
  const long stride = (long)gridDim.x * blockDim.x;
  const int half = N / 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long r = i / N;
    const int j = (int)(i - r * N);
    int src = j - half;
    if (src < 0) src += N;
    float w = 1.0f;
    if (mode == IR_MODE_HANN) {
      // the reference rolls the window by N/2, multiplies, and rolls the product by N/2 again (core.py:209-235): the tap that
      // lands at j carries hann[(j - 2 (N/2)) mod N] -- hann[j] for even N, hann[(j + 1) mod N] for odd N
      int wi = j - 2 * half;
      if (wi < 0) wi += N;
      w = (float)(0.5 - 0.5 * cospi(2.0 * (double)wi / (double)N));
    } else if (mode == IR_MODE_DYNAMIC) {
      float u = (float)(j - half) / half_width[r];
      if (u > 1.0f) u = 0.0f;
      w = (1.0f + cos_turns(kPiF * u)) / 2.0f;
    }
    out[i] = in[r * N + src] * w;
  }
}

// ------------------------------------------------------------------------------------------------
// Adjoint of the tap synthesis: d_taps [rows, N] -> gradient of the one-sided response (or of the raw control
// through the exp activation).  With dE[m] = w+ dt[N/2 + m] + w- dt[N/2 - m] and dO[m] = w+ dt[N/2 + m] - w- dt[N/2 - m]
// (window factors of the two taps a bin pair feeds; m = 0 and m = N/2 have one tap only)
//     d re_k = sum_m dE[m] TE[k][m] = c_k sum_m (dE[m] / c_m) TE[m][k],      d im_k = sum_m dO[m] TO[m][k]
// (TE[k][m] / c_k is symmetric, TO is symmetric where it is not zero), i.e. the SAME contraction against the SAME
// basis table as the forward kernel with the roles of bin and tap index swapped: same tiling, staging and MFMA loop;
// only the A-operand staging (built from two mirrored tap gradients and their window factors) and the epilogue
// (scale by c_k, activation derivative, [rows, n] stores) differ.
// ------------------------------------------------------------------------------------------------
template <int ACT, bool HAS_IM, int MODE>
__global__ void __launch_bounds__(256, 2) k_ir_gemm_bwd(const float* __restrict__ d_taps, const float* __restrict__ ctrl,
                                                     long ld_ctrl, float scale, const float* __restrict__ table,
                                                     const float* __restrict__ half_width, long rows, int n,
                                                     float* __restrict__ d_re, float* __restrict__ d_im) {
  __shared__ __attribute__((aligned(16))) float stage[2 * IR_STAGE];
  const int tid = threadIdx.x;
  const int wc = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63;
  const int li = l & 31, h = l >> 5;
  const long row0 = (long)blockIdx.x * GM;
  const int col0 = blockIdx.y * GN;
  const int N = 2 * (n - 1);
  const int half = N / 2;
  const int KP = (int)ir_kp(n);
  const long NP = ir_np(n);
  const long plane = (long)KP * NP;
  const int nch = KP / KC;
  const int total_ch = HAS_IM ? 2 * nch : nch;
  const float* hann = table + 2 * plane;

  f32x16 accE[2][2], accO[2][2];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int i = 0; i < 16; ++i) { accE[rb][cb][i] = 0.f; accO[rb][cb][i] = 0.f; }

  const int a_row = tid >> 2, a_kq = (tid & 3) * 4;
  const int b_k = tid >> 6, b_c4 = (tid & 63) * 4;
  const long ar = row0 + a_row;
  const bool a_live = ar < rows;
  const float* dt = d_taps + (a_live ? ar : 0) * (long)N;
  const float hw = (MODE == IR_MODE_DYNAMIC && a_live) ? half_width[ar] : 1.f;
  struct Chunk { float up[4], dn[4]; float4 b0, b1, b2, b3; };

  // tap gradients of the four contraction indices m = k0 + a_kq .. + 3 (clamped addresses, masked in park)
  auto fetch = [&](int cc) -> Chunk {
    const int part = (HAS_IM && cc >= nch) ? 1 : 0;
    const int k0 = (cc - part * nch) * KC;
    Chunk c;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      int m = k0 + a_kq + q;
      m = m > half ? half : m;
      c.up[q] = dt[half + (m < half ? m : 0)];
      c.dn[q] = dt[half - m];
    }
    const float* Tb = table + (part ? plane : 0) + (long)(k0 + b_k) * NP + col0 + b_c4;
    c.b0 = *reinterpret_cast<const float4*>(Tb);
    c.b1 = *reinterpret_cast<const float4*>(Tb + 4 * NP);
    c.b2 = *reinterpret_cast<const float4*>(Tb + 8 * NP);
    c.b3 = *reinterpret_cast<const float4*>(Tb + 12 * NP);
    return c;
  };
  auto park = [&](int buf, int cc, const Chunk& c) {
    float* As = stage + buf * IR_STAGE;
    float* Bs = As + GM * LDA;
    const int part = (HAS_IM && cc >= nch) ? 1 : 0;
    const int k0 = (cc - part * nch) * KC;
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int m = k0 + a_kq + q;
      float wp = 1.f, wm = 1.f;
      if (MODE == IR_MODE_HANN) {
        const int mm = m > half ? half : m;
        wp = hann[half + (mm < half ? mm : 0)];
        wm = hann[half - mm];
      } else if (MODE == IR_MODE_DYNAMIC) {
        float u = (float)m / hw;                                // core.py:244, tap N/2 + m
        const float un = -u;                                    // tap N/2 - m: never above 1, never clamped
        if (u > 1.0f) u = 0.0f;                                 // core.py:245
        wp = (1.0f + cos_turns(kPiF * u)) / 2.0f;
        wm = (1.0f + cos_turns(kPiF * un)) / 2.0f;
      }
      const float up = (a_live && m < half) ? wp * c.up[q] : 0.f;        // taps[N/2 + m] exists for m < N/2
      const float dn = (a_live && m >= 1 && m <= half) ? wm * c.dn[q] : 0.f;   // taps[N/2 - m] for 1 <= m <= N/2
      const float ce = (m == 0 || m == half) ? 1.0f : 0.5f;      // 1 / c_m
      v[q] = part ? (up - dn) : ce * (up + dn);
    }
    *reinterpret_cast<float4*>(As + a_row * LDA + a_kq) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(Bs + (b_k + 0) * LDB + b_c4) = c.b0;
    *reinterpret_cast<float4*>(Bs + (b_k + 4) * LDB + b_c4) = c.b1;
    *reinterpret_cast<float4*>(Bs + (b_k + 8) * LDB + b_c4) = c.b2;
    *reinterpret_cast<float4*>(Bs + (b_k + 12) * LDB + b_c4) = c.b3;
  };
  auto contract = [&](int buf, f32x16 (&acc)[2][2]) {
    const float* As = stage + buf * IR_STAGE;
    const float* Bs = As + GM * LDA;
    float av[2][8];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const float4 lo = *reinterpret_cast<const float4*>(As + (rb * 32 + li) * LDA + 8 * h);
      const float4 hi = *reinterpret_cast<const float4*>(As + (rb * 32 + li) * LDA + 8 * h + 4);
      av[rb][0] = lo.x; av[rb][1] = lo.y; av[rb][2] = lo.z; av[rb][3] = lo.w;
      av[rb][4] = hi.x; av[rb][5] = hi.y; av[rb][6] = hi.z; av[rb][7] = hi.w;
    }
    const float* bp = Bs + (8 * h) * LDB + wc * 64 + li;
#pragma unroll
    for (int sidx = 0; sidx < 8; ++sidx) {
      const float b0 = bp[sidx * LDB], b1 = bp[sidx * LDB + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0][sidx], b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0][sidx], b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1][sidx], b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1][sidx], b1, acc[1][1], 0, 0, 0);
    }
  };

  Chunk nxt = fetch(0);
  park(0, 0, nxt);
  __syncthreads();
  for (int cc = 0; cc < total_ch; ++cc) {
    const bool more = cc + 1 < total_ch;
    if (more) nxt = fetch(cc + 1);
    if (!HAS_IM || cc < nch) contract(cc & 1, accE);
    else contract(cc & 1, accO);
    if (more) park((cc + 1) & 1, cc + 1, nxt);
    __syncthreads();
  }

  // epilogue: column = bin k; d re_k = c_k accE, d im_k = accO; through the exp activation d c = d re * scale * exp(c)
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    const int k = col0 + wc * 64 + cb * 32 + li;
    if (k >= n) continue;
    const float ck = (k == 0 || k == n - 1) ? 1.0f : 2.0f;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const long r = row0 + rb * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        if (r >= rows) continue;
        float g = ck * accE[rb][cb][reg];
        if (ACT == IR_ACT_EXP) g = g * (scale * expf(ctrl[r * ld_ctrl + k]));
        d_re[r * n + k] = g;
        if (HAS_IM) d_im[r * n + k] = accO[rb][cb][reg];
      }
    }
  }
}

// adjoint of k_allpass_response: (d_re, d_im) [rows, n] -> gradient of the raw group-delay control.  One wave per row:
// theta rebuilt as in the forward kernel, d theta = -sin d_re + cos d_im, suffix sums over the bins, times
// pi (1 - tanh^2 c).
__global__ void __launch_bounds__(256) k_allpass_backward(const float* __restrict__ c, long ld, long rows, int n,
                                                          const float* __restrict__ d_re, const float* __restrict__ d_im,
                                                          float* __restrict__ d_c) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * 4 + wave;
  if (r >= rows) return;
  const float* cr = c + r * ld;
  const int per = (n + 63) / 64;
  const int k0 = lane * per;
  const double inv_2pi = 0.15915494309189533577;
  double local = 0.0;
  for (int q = 0; q < per; ++q) {
    const int k = k0 + q;
    if (k < n) local += (double)(kPiF * tanhf(cr[k]));
  }
  double run = wave_excl_scan(local, lane);
  // first pass: d theta of the lane's bins and their sum
  double dth_sum = 0.0;
  for (int q = 0; q < per; ++q) {
    const int k = k0 + q;
    if (k < n) {
      run += (double)(kPiF * tanhf(cr[k]));
      const double rev = run * inv_2pi;
      const float fr = (float)(rev - rint(rev));
      const float co = __builtin_amdgcn_cosf(fr), si = __builtin_amdgcn_sinf(fr);
      const float dth = fmaf(-si, d_re[r * n + k], co * d_im[r * n + k]);
      d_c[r * n + k] = dth;                                     // parked; turned into the suffix sum below
      dth_sum += (double)dth;
    }
  }
  // suffix sums: total of the lanes after this one, then a backward walk over the lane's own bins
  const double incl_before = wave_excl_scan(dth_sum, lane);    // sum over lanes < this one
  const double total = wave_sum(dth_sum);
  double suffix = total - incl_before - dth_sum;                // sum over lanes > this one
  for (int q = per - 1; q >= 0; --q) {
    const int k = k0 + q;
    if (k < n) {
      suffix += (double)d_c[r * n + k];
      const float th = tanhf(cr[k]);
      d_c[r * n + k] = (float)suffix * (kPiF * (1.0f - th * th));
    }
  }
}

// The same at 256 bins (every shipped configuration): a lane owns four consecutive bins, fetched as one 16-byte load per
// array, tanh evaluated once per bin and the d theta values kept in registers instead of parked in d_c -- the same
// operations in the same order as the general kernel above, so the same bits (113 MB of traffic per launch at
// B = 32 x 10 s; the general form spent 64 us on it: three tanhf per bin and twelve 4-byte accesses per lane).
__global__ void __launch_bounds__(256) k_allpass_backward_256(const float* __restrict__ c, long ld, long rows,
                                                              const float* __restrict__ d_re, const float* __restrict__ d_im,
                                                              float* __restrict__ d_c) {
  constexpr int n = 256;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * 4 + wave;
  if (r >= rows) return;
  const float4 cv = *reinterpret_cast<const float4*>(c + r * ld + 4 * lane);
  const float4 rv = *reinterpret_cast<const float4*>(d_re + r * n + 4 * lane);
  const float4 iv = *reinterpret_cast<const float4*>(d_im + r * n + 4 * lane);
  const double inv_2pi = 0.15915494309189533577;
  const float th[4] = {tanhf(cv.x), tanhf(cv.y), tanhf(cv.z), tanhf(cv.w)};
  const float dr[4] = {rv.x, rv.y, rv.z, rv.w}, di[4] = {iv.x, iv.y, iv.z, iv.w};
  double local = 0.0;
#pragma unroll
  for (int q = 0; q < 4; ++q) local += (double)(kPiF * th[q]);
  double run = wave_excl_scan(local, lane);
  float dth[4];
  double dth_sum = 0.0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    run += (double)(kPiF * th[q]);
    const double rev = run * inv_2pi;
    const float fr = (float)(rev - rint(rev));
    const float co = __builtin_amdgcn_cosf(fr), si = __builtin_amdgcn_sinf(fr);
    dth[q] = fmaf(-si, dr[q], co * di[q]);
    dth_sum += (double)dth[q];
  }
  const double incl_before = wave_excl_scan(dth_sum, lane);
  const double total = wave_sum(dth_sum);
  double suffix = total - incl_before - dth_sum;
  float o[4];
#pragma unroll
  for (int q = 3; q >= 0; --q) {
    suffix += (double)dth[q];
    o[q] = (float)suffix * (kPiF * (1.0f - th[q] * th[q]));
  }
  *reinterpret_cast<float4*>(d_c + r * n + 4 * lane) = make_float4(o[0], o[1], o[2], o[3]);
}

// ---- launchers -----------------------------------------------------------------------------------
size_t ir_table_floats(int n) { return (size_t)(2 * ir_kp(n) * ir_np(n)) + 2 * (size_t)(n - 1); }

void launch_ir_table(int n, float* table, hipStream_t st) {
  long total = (long)ir_table_floats(n);
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_ir_table, dim3((unsigned)blocks), dim3(256), 0, st, n, table);
}

void launch_allpass_response(const float* c, long ld, long rows, int n, float* re, float* im, hipStream_t st) {
  if (rows == 0) return;
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  const int vec4 = (n == 256 && (ld & 3) == 0 && al16(c) && al16(re) && al16(im)) ? 1 : 0;
  hipLaunchKernelGGL(k_allpass_response, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, c, ld, rows, n, re, im, vec4);
}

void launch_ir_gemm(const float* a_re, long ld_re, const float* a_im, long ld_im, int act, float scale,
                    const float* table, int mode, const float* half_width, long rows, int n, float* taps,
                    hipStream_t st, float hw_from_f0_sr) {
  if (rows == 0) return;
  dim3 grid((unsigned)((rows + GM - 1) / GM), (unsigned)(ir_np(n) / GN)), block(256);
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  const int vec = (al16(a_re) && (ld_re & 3) == 0 && (!a_im || (al16(a_im) && (ld_im & 3) == 0))) ? 1 : 0;
#define DDSP_IR_LAUNCH(ACT_, IM_, MODE_)                                                                          \
  hipLaunchKernelGGL((k_ir_gemm<ACT_, IM_, MODE_>), grid, block, 0, st, a_re, ld_re, a_im, ld_im, scale, table, \
                     half_width, rows, n, taps, vec, hw_from_f0_sr)
#define DDSP_IR_MODES(ACT_, IM_)                                              \
  do {                                                                        \
    if (mode == IR_MODE_HANN) DDSP_IR_LAUNCH(ACT_, IM_, IR_MODE_HANN);        \
    else if (mode == IR_MODE_DYNAMIC) DDSP_IR_LAUNCH(ACT_, IM_, IR_MODE_DYNAMIC); \
    else DDSP_IR_LAUNCH(ACT_, IM_, IR_MODE_ROLL);                             \
  } while (0)
  if (a_im) {
    if (act == IR_ACT_EXP) DDSP_IR_MODES(IR_ACT_EXP, true);
    else DDSP_IR_MODES(IR_ACT_NONE, true);
  } else {
    if (act == IR_ACT_EXP) DDSP_IR_MODES(IR_ACT_EXP, false);
    else DDSP_IR_MODES(IR_ACT_NONE, false);
  }
#undef DDSP_IR_MODES
#undef DDSP_IR_LAUNCH
}

void launch_window_taps(const float* in, int mode, const float* half_width, long rows, int N, float* out, hipStream_t st) {
  if (rows == 0) return;
  long blocks = (rows * (long)N + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_window_taps, dim3((unsigned)blocks), dim3(256), 0, st, in, mode, half_width, rows, N, out);
}

void launch_ir_gemm_bwd(const float* d_taps, const float* ctrl, long ld_ctrl, int act, float scale, const float* table,
                        int mode, const float* half_width, long rows, int n, int has_im, float* d_re, float* d_im,
                        hipStream_t st) {
  if (rows == 0) return;
  dim3 grid((unsigned)((rows + GM - 1) / GM), (unsigned)(ir_np(n) / GN)), block(256);
#define DDSP_IRB_LAUNCH(ACT_, IM_, MODE_)                                                                        \
  hipLaunchKernelGGL((k_ir_gemm_bwd<ACT_, IM_, MODE_>), grid, block, 0, st, d_taps, ctrl, ld_ctrl, scale, table, \
                     half_width, rows, n, d_re, d_im)
  if (has_im) {                                                 // complex response (all-pass: no window; torch.complex(mag, 0): windowed)
    if (mode == IR_MODE_HANN) DDSP_IRB_LAUNCH(IR_ACT_NONE, true, IR_MODE_HANN);
    else if (mode == IR_MODE_DYNAMIC) DDSP_IRB_LAUNCH(IR_ACT_NONE, true, IR_MODE_DYNAMIC);
    else DDSP_IRB_LAUNCH(IR_ACT_NONE, true, IR_MODE_ROLL);
  } else if (act == IR_ACT_EXP) {
    if (mode == IR_MODE_HANN) DDSP_IRB_LAUNCH(IR_ACT_EXP, false, IR_MODE_HANN);
    else if (mode == IR_MODE_DYNAMIC) DDSP_IRB_LAUNCH(IR_ACT_EXP, false, IR_MODE_DYNAMIC);
    else DDSP_IRB_LAUNCH(IR_ACT_EXP, false, IR_MODE_ROLL);
  } else {
    if (mode == IR_MODE_HANN) DDSP_IRB_LAUNCH(IR_ACT_NONE, false, IR_MODE_HANN);
    else if (mode == IR_MODE_DYNAMIC) DDSP_IRB_LAUNCH(IR_ACT_NONE, false, IR_MODE_DYNAMIC);
    else DDSP_IRB_LAUNCH(IR_ACT_NONE, false, IR_MODE_ROLL);
  }
#undef DDSP_IRB_LAUNCH
}

void launch_allpass_backward(const float* c, long ld, long rows, int n, const float* d_re, const float* d_im, float* d_c,
                             hipStream_t st) {
  if (rows == 0) return;
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (n == 256 && (ld & 3) == 0 && al16(c) && al16(d_re) && al16(d_im) && al16(d_c)) {
    hipLaunchKernelGGL(k_allpass_backward_256, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, c, ld, rows, d_re, d_im, d_c);
    return;
  }
  hipLaunchKernelGGL(k_allpass_backward, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, c, ld, rows, n, d_re, d_im, d_c);
}

}  // namespace ddsp
