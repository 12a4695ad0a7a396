// 2048-point complex FFT for one 256-thread workgroup (4 waves), 8 points per thread, data in registers,
// three LDS exchanges through two ping-pong buffers.  Building block of the FFT-domain time-varying FIR.
//
// Decimation in frequency with N = 8 * 8 * 8 * 4:
//      n = 256 n1 + 32 n2 + 4 n3 + n4,      k = k1 + 8 k2 + 64 k3 + 512 k4
//   pass 1  thread p = tid holds z[256 n1 + p]:            DFT8 over n1, times W_2048^(p k1)       -> A[k1][p]
//   pass 2  thread (k1 = tid >> 5, c = tid & 31 = 4 n3 + n4): DFT8 over n2, times W_256^(c k2)       -> B[n3][k2][k1^n3][n4]
//   pass 3  thread tid = n4 + 4 k1 + 32 k2:                  DFT8 over n3, times W_32^(n4 k3)        -> A[k3][k2][k1][n4]
//   pass 4  thread r = k1 + 8 k2 + 64 (k3 & 3), s = k3 >> 2 = 0,1: two DFT4 over n4
// so thread r ends with Z[r + 256 s + 512 k4] = Z[256 m + r], m = s + 2 k4: the same "slot m, lane r" layout
// the input had.  Natural-order LDS traffic (k = 256 m + r) is conflict-free and the inverse transform
// (conjugate, forward, conjugate) chains without reordering.  Every exchange is written and read with lane-
// consecutive addresses except the pass-2 stores, whose rows are XOR-swizzled by n3 (at most 2-way conflicts).
// Complex values are f32x2 so adds / multiplies become packed-f32 instructions (v_pk_add/mul/fma_f32).
#pragma once
#include "ddsp_common.h"

namespace ddsp {
namespace fft {

constexpr int N = 2048;
constexpr int THREADS = 256;
constexpr int SLOTS = N / THREADS;       // complex points per thread
constexpr int EX_WORDS = N;              // complex words per exchange buffer

// Complex helpers.  A complex number is one 64-bit VGPR pair; rotations by +-i, conjugation and the cross terms
// of a complex product are operand swizzles (op_sel / op_sel_hi) and sign bits (neg_lo / neg_hi) of the packed
// instructions, written out as inline assembly in the device pass because the compiler otherwise materialises
// them with v_mov / v_xor.  The host pass (and the CPU emulator of the test-suite) sees the plain C++ form.
#if defined(__HIP_DEVICE_COMPILE__)
#define DDSP_PK2(res, text, a, b) asm(text : "=v"(res) : "v"(a), "v"(b))
#define DDSP_PK3(res, text, a, b, c) asm(text : "=v"(res) : "v"(a), "v"(b), "v"(c))
#endif

// a * b
__device__ __forceinline__ f32x2 cmul(f32x2 a, f32x2 b) {
#if defined(__HIP_DEVICE_COMPILE__)
  f32x2 t, r;
  DDSP_PK2(t, "v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]", a, b);                                     // (ax bx, ax by)
  DDSP_PK3(r, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]", a, b, t);  // + (-ay by, ay bx)
  return r;
#else
  const f32x2 t = f32x2{a.x, a.x} * b;
  return __builtin_elementwise_fma(f32x2{a.y, a.y}, f32x2{-b.y, b.x}, t);
#endif
}
// a * k for a wave-uniform factor k (the W_8 constants of the radix-8 butterflies): k travels as an SGPR pair, so no v_mov
// builds it in vector registers -- 28 of them per pass of k_fir_blk6 otherwise (the compiler rematerialises constants rather
// than keep four registers for them)
__device__ __forceinline__ f32x2 cmul_uniform(f32x2 a, f32x2 k) {
#if defined(__HIP_DEVICE_COMPILE__)
  f32x2 t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "s"(k));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "s"(k), "v"(t));
  return r;
#else
  const f32x2 t = f32x2{a.x, a.x} * k;
  return __builtin_elementwise_fma(f32x2{a.y, a.y}, f32x2{-k.y, k.x}, t);
#endif
}
// the two halves of a * b apart, so that independent products can be issued between them (an instruction that consumes its
// predecessor's result waits ~3 cycles for it, and behind a packed producer the compiler adds an s_nop on top -- about the
// issue time of one more instruction each, tools/probes/valu_probe2.hip)
__device__ __forceinline__ f32x2 cmul_lo(f32x2 a, f32x2 b) {
#if defined(__HIP_DEVICE_COMPILE__)
  f32x2 t;
  DDSP_PK2(t, "v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]", a, b);                                     // (ax bx, ax by)
  return t;
#else
  return f32x2{a.x, a.x} * b;
#endif
}
__device__ __forceinline__ f32x2 cmul_hi(f32x2 a, f32x2 b, f32x2 t) {
#if defined(__HIP_DEVICE_COMPILE__)
  f32x2 r;
  DDSP_PK3(r, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]", a, b, t);  // + (-ay by, ay bx)
  return r;
#else
  return __builtin_elementwise_fma(f32x2{a.y, a.y}, f32x2{-b.y, b.x}, t);
#endif
}
// conj(a) * b in the same two halves: (ax bx, ax by) + ay (by, -bx)
__device__ __forceinline__ f32x2 cmulc_hi(f32x2 a, f32x2 b, f32x2 t) {
#if defined(__HIP_DEVICE_COMPILE__)
  f32x2 r;
  DDSP_PK3(r, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[0,1,0]", a, b, t);  // + (ay by, -ay bx)
  return r;
#else
  return __builtin_elementwise_fma(f32x2{a.y, a.y}, f32x2{b.y, -b.x}, t);
#endif
}
__device__ __forceinline__ f32x2 cmulc(f32x2 a, f32x2 b) { return cmulc_hi(a, b, cmul_lo(a, b)); }
// v[k] *= w[k], k = 1..7, for one array / for two arrays against the same factors: up to four products in flight
__device__ __forceinline__ void twiddle7(f32x2* v, const f32x2* w) {
  f32x2 t1 = cmul_lo(v[1], w[1]), t2 = cmul_lo(v[2], w[2]), t3 = cmul_lo(v[3], w[3]), t4 = cmul_lo(v[4], w[4]);
  v[1] = cmul_hi(v[1], w[1], t1); v[2] = cmul_hi(v[2], w[2], t2); v[3] = cmul_hi(v[3], w[3], t3); v[4] = cmul_hi(v[4], w[4], t4);
  t1 = cmul_lo(v[5], w[5]); t2 = cmul_lo(v[6], w[6]); t3 = cmul_lo(v[7], w[7]);
  v[5] = cmul_hi(v[5], w[5], t1); v[6] = cmul_hi(v[6], w[6], t2); v[7] = cmul_hi(v[7], w[7], t3);
}
__device__ __forceinline__ void twiddle7x2(f32x2* u, const f32x2* wu, f32x2* v, const f32x2* wv) {
#pragma unroll
  for (int k = 1; k < 7; k += 2) {
    const f32x2 t1 = cmul_lo(u[k], wu[k]), t2 = cmul_lo(v[k], wv[k]), t3 = cmul_lo(u[k + 1], wu[k + 1]), t4 = cmul_lo(v[k + 1], wv[k + 1]);
    u[k] = cmul_hi(u[k], wu[k], t1); v[k] = cmul_hi(v[k], wv[k], t2);
    u[k + 1] = cmul_hi(u[k + 1], wu[k + 1], t3); v[k + 1] = cmul_hi(v[k + 1], wv[k + 1], t4);
  }
  const f32x2 t1 = cmul_lo(u[7], wu[7]), t2 = cmul_lo(v[7], wv[7]);
  u[7] = cmul_hi(u[7], wu[7], t1); v[7] = cmul_hi(v[7], wv[7], t2);
}
// t + (-i) d
__device__ __forceinline__ f32x2 add_mi(f32x2 t, f32x2 d) {
#if defined(__HIP_DEVICE_COMPILE__)
  f32x2 r;
  DDSP_PK2(r, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]", t, d);
  return r;
#else
  return t + f32x2{d.y, -d.x};
#endif
}
// t - (-i) d
__device__ __forceinline__ f32x2 sub_mi(f32x2 t, f32x2 d) {
#if defined(__HIP_DEVICE_COMPILE__)
  f32x2 r;
  DDSP_PK2(r, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]", t, d);
  return r;
#else
  return t - f32x2{d.y, -d.x};
#endif
}
// a + conj(z)  and  a - conj(z)
__device__ __forceinline__ f32x2 add_conj(f32x2 a, f32x2 z) {
#if defined(__HIP_DEVICE_COMPILE__)
  f32x2 r;
  DDSP_PK2(r, "v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]", a, z);
  return r;
#else
  return f32x2{a.x + z.x, a.y - z.y};
#endif
}
__device__ __forceinline__ f32x2 sub_conj(f32x2 a, f32x2 z) {
#if defined(__HIP_DEVICE_COMPILE__)
  f32x2 r;
  DDSP_PK2(r, "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]", a, z);
  return r;
#else
  return f32x2{a.x - z.x, a.y + z.y};
#endif
}
// (p.y * q.x, p.x * q.y): swap the halves of p, scale per half
__device__ __forceinline__ f32x2 swap_scale(f32x2 p, f32x2 q) {
#if defined(__HIP_DEVICE_COMPILE__)
  f32x2 r;
  DDSP_PK2(r, "v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]", p, q);
  return r;
#else
  return f32x2{p.y * q.x, p.x * q.y};
#endif
}
// (p.y * q.x + a.x, p.x * q.y + a.y): the same plus an addend
__device__ __forceinline__ f32x2 swap_scale_add(f32x2 p, f32x2 q, f32x2 a) {
#if defined(__HIP_DEVICE_COMPILE__)
  f32x2 r;
  DDSP_PK3(r, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1]", p, q, a);
  return r;
#else
  return f32x2{fmaf(p.y, q.x, a.x), fmaf(p.x, q.y, a.y)};
#endif
}
// (v.x - g.y, -v.y - g.x) = conj(v) - i conj(g)
__device__ __forceinline__ f32x2 conj_minus_i_conj(f32x2 v, f32x2 g) {
#if defined(__HIP_DEVICE_COMPILE__)
  f32x2 r;
  DDSP_PK2(r, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[1,1]", v, g);
  return r;
#else
  return f32x2{v.x - g.y, -v.y - g.x};
#endif
}
__device__ __forceinline__ f32x2 cconj(f32x2 a) { return f32x2{a.x, -a.y}; }

// forward 4-point DFT (W4 = -i), in place: (a0,a1,a2,a3) -> (X0,X1,X2,X3)
__device__ __forceinline__ void dft4(f32x2& a0, f32x2& a1, f32x2& a2, f32x2& a3) {
  const f32x2 t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, d = a1 - a3;
  a0 = t0 + t2;
  a2 = t0 - t2;
  a1 = add_mi(t1, d);
  a3 = sub_mi(t1, d);
}
// the same with input a2 still to be multiplied by -i (folded into the first butterfly)
__device__ __forceinline__ void dft4_rot2(f32x2& a0, f32x2& a1, f32x2& a2, f32x2& a3) {
  const f32x2 t0 = add_mi(a0, a2), t1 = sub_mi(a0, a2), t2 = a1 + a3, d = a1 - a3;
  a0 = t0 + t2;
  a2 = t0 - t2;
  a1 = add_mi(t1, d);
  a3 = sub_mi(t1, d);
}

// forward 8-point DFT in place on v[0..7]  (n = 4 a + b, k = a' + 2 b')
__device__ __forceinline__ void dft8(f32x2* v) {
  const float H = 0.70710678118654752f;
  f32x2 e0 = v[0] + v[4], e1 = v[1] + v[5], e2 = v[2] + v[6], e3 = v[3] + v[7];     // a' = 0
  f32x2 o0 = v[0] - v[4], o1 = v[1] - v[5], o2 = v[2] - v[6], o3 = v[3] - v[7];     // a' = 1, then * W8^b
  o1 = cmul_uniform(o1, f32x2{H, -H});
  o3 = cmul_uniform(o3, f32x2{-H, -H});
  dft4(e0, e1, e2, e3);                    // X[0], X[2], X[4], X[6]
  dft4_rot2(o0, o1, o2, o3);               // X[1], X[3], X[5], X[7]   (o2 * W8^2 = o2 * -i inside)
  v[0] = e0; v[2] = e1; v[4] = e2; v[6] = e3;
  v[1] = o0; v[3] = o1; v[5] = o2; v[7] = o3;
}

// the same when v[4..7] are known to be zero (a transform whose input is zero-padded to twice its length: the first
// pass of a linear convolution): the first butterfly column degenerates to copies
__device__ __forceinline__ void dft8_lo4(f32x2* v) {
  const float H = 0.70710678118654752f;
  f32x2 e0 = v[0], e1 = v[1], e2 = v[2], e3 = v[3];
  f32x2 o0 = v[0], o1 = cmul_uniform(v[1], f32x2{H, -H}), o2 = v[2], o3 = cmul_uniform(v[3], f32x2{-H, -H});
  dft4(e0, e1, e2, e3);
  dft4_rot2(o0, o1, o2, o3);
  v[0] = e0; v[2] = e1; v[4] = e2; v[6] = e3;
  v[1] = o0; v[3] = o1; v[5] = o2; v[7] = o3;
}

// exp(-i pi x): the twiddles' one call site of sincospif -- NOT inlined, so that a kernel's 24 twiddles cost one copy of its
// ~40 instructions instead of 24 (a third of the code of the short-time kernels; code that is not there is not fetched,
// DESIGN.md "instruction cache")
__device__ __attribute__((noinline)) inline f32x2 cis_mpi(float x) {
  float s, c;
  sincospif(-x, &s, &c);
  return f32x2{c, s};
}

// per-thread twiddles, computed once per workgroup lifetime (exact arguments: multiples of 2^-10)
struct Twiddles {
  f32x2 w1[8];       // W_2048^(p k),   p = tid
  f32x2 w2[8];       // W_256^(c k),    c = tid & 31
  f32x2 w3[8];       // W_32^(n4 k),    n4 = tid & 3
  __device__ __forceinline__ void init(int tid) {
    const int c = tid & 31, n4 = tid & 3;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      w1[k] = cis_mpi((float)((tid * k) & 2047) / 1024.0f);
      w2[k] = cis_mpi((float)((c * k) & 255) / 128.0f);
      w3[k] = cis_mpi((float)((n4 * k) & 31) / 16.0f);
    }
  }
};

// v[n1] = z[256 n1 + tid]  ->  v[m] = Z[256 m + tid].
// Buffers: A must be free of readers on entry; on return A may still be read by slower waves (pass 4) and B
// is free.  A caller that next WRITES B and synchronises before touching A again needs no extra barrier.
__device__ __forceinline__ void forward(f32x2 (&v)[8], const Twiddles& tw, f32x2* A, f32x2* B, int tid) {
  dft8(v);
#pragma unroll
  for (int k = 1; k < 8; ++k) v[k] = cmul(v[k], tw.w1[k]);
#pragma unroll
  for (int k = 0; k < 8; ++k) A[k * 256 + tid] = v[k];                          // [k1][p]
  __syncthreads();
  {
    const int k1 = tid >> 5, c = tid & 31;
#pragma unroll
    for (int n2 = 0; n2 < 8; ++n2) v[n2] = A[k1 * 256 + n2 * 32 + c];
    dft8(v);
#pragma unroll
    for (int k = 1; k < 8; ++k) v[k] = cmul(v[k], tw.w2[k]);
    const int n3 = c >> 2, n4 = c & 3;
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) B[n3 * 256 + ((k2 * 32 + k1 * 4 + n4) ^ (n3 * 4))] = v[k2];   // [n3][k2][k1 ^ n3][n4]
  }
  __syncthreads();
  {
#pragma unroll
    for (int n3 = 0; n3 < 8; ++n3) v[n3] = B[n3 * 256 + (tid ^ (n3 * 4))];      // tid = n4 + 4 k1 + 32 k2
    dft8(v);
#pragma unroll
    for (int k = 1; k < 8; ++k) v[k] = cmul(v[k], tw.w3[k]);
#pragma unroll
    for (int k3 = 0; k3 < 8; ++k3) A[k3 * 256 + tid] = v[k3];                   // [k3][k2][k1][n4], k1 = (tid>>2)&7, k2 = tid>>5
  }
  __syncthreads();
  {
    // r = k1' + 8 k2' + 64 k3lo reads (k1', k2', k3 = k3lo + 4 s): word k3 * 256 + (k2' * 32 + k1' * 4) + n4
    const int k1 = tid & 7, k2 = (tid >> 3) & 7, k3lo = tid >> 6;
    const f32x2* src = A + k2 * 32 + k1 * 4;
    f32x2 t[8];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      f32x2 a0 = src[(k3lo + 4 * s) * 256 + 0], a1 = src[(k3lo + 4 * s) * 256 + 1];
      f32x2 a2 = src[(k3lo + 4 * s) * 256 + 2], a3 = src[(k3lo + 4 * s) * 256 + 3];
      dft4(a0, a1, a2, a3);                                                       // k4 = 0..3 -> slot m = s + 2 k4
      t[s] = a0; t[s + 2] = a1; t[s + 4] = a2; t[s + 6] = a3;
    }
#pragma unroll
    for (int m = 0; m < 8; ++m) v[m] = t[m];
  }
}

}  // namespace fft
}  // namespace ddsp
