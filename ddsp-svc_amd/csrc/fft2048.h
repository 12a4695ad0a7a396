// 2048-point complex FFT for one 128-thread workgroup (2 waves), 16 points per thread, data in
// registers, two LDS exchanges.  Building block of the FFT-domain time-varying FIR (fir_fft.hip).
//
// Decimation in frequency with N = 16 * 16 * 8:  n = 128 n1 + 8 n2 + n3,  k = k1 + 16 k2 + 256 k3
//   pass 1  thread p = 8 n2 + n3 holds z[128 n1 + p]: 16-point DFT over n1, times W_2048^(p k1)
//   pass 2  thread q = 8 k1 + n3 gathers n2 = 0..15:   16-point DFT over n2, times W_128^(n3 k2)
//   pass 3  thread r = (k1 = r & 15, k2 = (r >> 4) + 8 s), s = 0,1: two 8-point DFTs over n3
// so thread r ends with Z[r + 128 s + 256 k3] = Z[128 m + r], m = s + 2 k3: the same "slot m, lane r"
// layout the input had.  Natural-order LDS traffic (k = 128 m + r) is therefore conflict-free, and the
// inverse transform (conjugate, forward, conjugate) chains without any reordering.
// Complex values are f32x2 so adds / multiplies become packed-f32 instructions.
#pragma once
#include "ddsp_common.h"

namespace ddsp {
namespace fft {

constexpr int N = 2048;
constexpr int THREADS = 128;
constexpr int ROW = 136;                 // exchange row stride in complex words (128 + 8: rows 8 banks-pairs apart)
constexpr int EX_WORDS = 16 * ROW;       // complex words in the exchange buffer (>= N)

__device__ __forceinline__ f32x2 cmul(f32x2 a, f32x2 b) {
  const f32x2 t = f32x2{a.x, a.x} * b;
  return __builtin_elementwise_fma(f32x2{a.y, a.y}, f32x2{-b.y, b.x}, t);
}
__device__ __forceinline__ f32x2 mul_mi(f32x2 a) { return f32x2{a.y, -a.x}; }     // a * (-i)
__device__ __forceinline__ f32x2 cconj(f32x2 a) { return f32x2{a.x, -a.y}; }

// forward 4-point DFT (W4 = -i), in place: (a0,a1,a2,a3) -> (X0,X1,X2,X3)
__device__ __forceinline__ void dft4(f32x2& a0, f32x2& a1, f32x2& a2, f32x2& a3) {
  const f32x2 t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, t3 = mul_mi(a1 - a3);
  a0 = t0 + t2;
  a2 = t0 - t2;
  a1 = t1 + t3;
  a3 = t1 - t3;
}

// forward 16-point DFT in place: v[n] -> v[k]
__device__ __forceinline__ void dft16(f32x2 (&v)[16]) {
  const float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, H = 0.70710678118654752f;
#pragma unroll
  for (int b = 0; b < 4; ++b) dft4(v[b], v[4 + b], v[8 + b], v[12 + b]);        // v[4 k' + b] = T[k'][b]
  // T[k'][b] *= W16^(b k')
  v[4 + 1] = cmul(v[4 + 1], f32x2{C1, -S1});
  v[4 + 2] = cmul(v[4 + 2], f32x2{H, -H});
  v[4 + 3] = cmul(v[4 + 3], f32x2{S1, -C1});
  v[8 + 1] = cmul(v[8 + 1], f32x2{H, -H});
  v[8 + 2] = mul_mi(v[8 + 2]);
  v[8 + 3] = cmul(v[8 + 3], f32x2{-H, -H});
  v[12 + 1] = cmul(v[12 + 1], f32x2{S1, -C1});
  v[12 + 2] = cmul(v[12 + 2], f32x2{-H, -H});
  v[12 + 3] = cmul(v[12 + 3], f32x2{-C1, S1});
#pragma unroll
  for (int k = 0; k < 4; ++k) dft4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);   // -> X[k + 4 j] at v[4k + j]
  // transpose the 4x4 index so that v[k] = X[k]
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int j = k + 1; j < 4; ++j) {
      const f32x2 t = v[4 * k + j];
      v[4 * k + j] = v[4 * j + k];
      v[4 * j + k] = t;
    }
}

// forward 8-point DFT in place on v[0..7]
__device__ __forceinline__ void dft8(f32x2* v) {
  const float H = 0.70710678118654752f;
  f32x2 e0 = v[0] + v[4], e1 = v[1] + v[5], e2 = v[2] + v[6], e3 = v[3] + v[7];     // k' = 0
  f32x2 o0 = v[0] - v[4], o1 = v[1] - v[5], o2 = v[2] - v[6], o3 = v[3] - v[7];     // k' = 1, then * W8^b
  o1 = cmul(o1, f32x2{H, -H});
  o2 = mul_mi(o2);
  o3 = cmul(o3, f32x2{-H, -H});
  dft4(e0, e1, e2, e3);                    // X[0], X[2], X[4], X[6]
  dft4(o0, o1, o2, o3);                    // X[1], X[3], X[5], X[7]
  v[0] = e0; v[2] = e1; v[4] = e2; v[6] = e3;
  v[1] = o0; v[3] = o1; v[5] = o2; v[7] = o3;
}

// per-thread twiddles, computed once per workgroup lifetime
struct Twiddles {
  f32x2 w1[16];      // W_2048^(p k1),      p = tid            (pass 1)
  f32x2 w2[16];      // W_128^(n3 k2),      n3 = tid & 7       (pass 2)
  __device__ __forceinline__ void init(int tid) {
    const int n3 = tid & 7;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      float s, c;
      sincospif(-(float)((tid * k) & 2047) / 1024.0f, &s, &c);       // exact argument: m/1024, m < 2048
      w1[k] = f32x2{c, s};
      sincospif(-(float)((n3 * k) & 127) / 64.0f, &s, &c);
      w2[k] = f32x2{c, s};
    }
  }
};

// v[n1] = z[128 n1 + tid]  ->  v[m] = Z[128 m + tid].  ex: EX_WORDS complex words of LDS, free on entry
// (every thread past its last read of it) and free again on return.
__device__ __forceinline__ void forward(f32x2 (&v)[16], const Twiddles& tw, f32x2* ex, int tid) {
  dft16(v);
#pragma unroll
  for (int k = 1; k < 16; ++k) v[k] = cmul(v[k], tw.w1[k]);
#pragma unroll
  for (int k = 0; k < 16; ++k) ex[k * ROW + tid] = v[k];                 // [k1][p]
  __syncthreads();
  const int k1 = tid >> 3, n3 = tid & 7;
#pragma unroll
  for (int n2 = 0; n2 < 16; ++n2) v[n2] = ex[k1 * ROW + n2 * 8 + n3];
  __syncthreads();                                                       // rows are rewritten in place below
  dft16(v);
#pragma unroll
  for (int k = 1; k < 16; ++k) v[k] = cmul(v[k], tw.w2[k]);
#pragma unroll
  for (int k2 = 0; k2 < 16; ++k2) ex[k1 * ROW + k2 * 8 + n3] = v[k2];    // [k1][k2][n3]
  __syncthreads();
  const int r1 = tid & 15, r2 = tid >> 4;
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) v[8 * s + j] = ex[r1 * ROW + (r2 + 8 * s) * 8 + j];
  __syncthreads();                                                       // exchange buffer free again
  dft8(&v[0]);
  dft8(&v[8]);
  // v[8 s + k3] = Z[tid + 128 s + 256 k3]  ->  slot m = s + 2 k3
  f32x2 t[16];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int k3 = 0; k3 < 8; ++k3) t[s + 2 * k3] = v[8 * s + k3];
#pragma unroll
  for (int m = 0; m < 16; ++m) v[m] = t[m];
}

}  // namespace fft
}  // namespace ddsp
