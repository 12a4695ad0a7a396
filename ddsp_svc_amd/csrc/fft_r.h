// Complex FFT of N = 512 R points (R = 1: 512, R = 2: 1024, R = 4: 2048, R = 8: 4096) for one workgroup of P = 64 R threads, 8 points per
// thread in registers, three LDS exchanges through two ping-pong buffers -- the generalisation of fft2048.h
// (whose complex helpers and radix-4/8 butterflies it re-uses) that the short-time spectral filters of
// CombSubFast (window 1024) and CombSubSuperFast (window 2048) are built on.
//
// Decimation in frequency with N = 8 * 8 * 8 * R:
//      n = P n1 + 8R n2 + R n3 + n4,          k = k1 + 8 k2 + 64 k3 + 512 k4
//   pass 1  thread p = tid holds z[P n1 + p]:                DFT8 over n1, times W_N^(p k1)      -> A[k1][p]
//   pass 2  thread (k1 = tid / 8R, c = tid % 8R = R n3 + n4): DFT8 over n2, times W_P^(c k2)      -> B[n3][k2][k1^n3][n4]
//   pass 3  thread tid = n4 + R k1 + 8R k2:                   DFT8 over n3, times W_8R^(n4 k3)    -> A[k3][k2][k1][n4]
//   pass 4  thread r = k1 + 8 k2 + 64 (k3 % R), s = k3 / R:   8/R DFT_R over n4
// so thread r ends with Z[r + P s + 512 k4] = Z[P m + r], m = s + (8/R) k4: the "slot m, lane r" layout the
// input had; the inverse transform (conjugate, forward, conjugate) chains without reordering.
#pragma once
#include "fft2048.h"

namespace ddsp {
namespace fft {

template <int R>
struct Plan {
  static_assert(R == 1 || R == 2 || R == 4 || R == 8, "N = 512, 1024, 2048 or 4096");
  static constexpr int N = 512 * R;
  static constexpr int P = 64 * R;          // threads
  static constexpr int SLOTS = 8;           // complex points per thread: k = P m + tid
  static constexpr int C = 8 * R;           // extent of the (n3, n4) index

  struct Tw {
    f32x2 w1[8];       // W_N^(tid k)
    f32x2 w2[8];       // W_P^(c k),   c = tid % 8R
    f32x2 w3[8];       // W_8R^(n4 k), n4 = tid % R
    __device__ __forceinline__ void init(int tid) {
      const int c = tid & (C - 1), n4 = tid & (R - 1);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        w1[k] = cis_mpi((float)((tid * k) & (N - 1)) / (float)(N / 2));
        w2[k] = cis_mpi((float)((c * k) & (P - 1)) / (float)(P / 2));
        w3[k] = cis_mpi((float)((n4 * k) & (C - 1)) / (float)(C / 2));
      }
    }
  };

  // R = 2: the third set W_16^(n4 k) depends on the lane parity only -- 1 on even lanes, a constant on odd ones -- so it needs
  // no registers: apply_w3 multiplies the odd lanes (exec mask) by constants held in scalar registers.  14 VGPRs less for a
  // kernel that counts them (k_fir_blk6); the same 14 packed instructions.
  struct TwLite {
    f32x2 w1[8];       // W_N^(tid k)
    f32x2 w2[8];       // W_P^(c k),   c = tid % 8R
    __device__ __forceinline__ void init(int tid) {
      static_assert(R == 2, "the scalar third set is implemented for the 1024-point plan");
      const int c = tid & (C - 1);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        w1[k] = cis_mpi((float)((tid * k) & (N - 1)) / (float)(N / 2));
        w2[k] = cis_mpi((float)((c * k) & (P - 1)) / (float)(P / 2));
      }
    }
  };
  static __device__ __forceinline__ void apply_w3(f32x2 (&v)[8], const Tw& tw, int) { twiddle7(v, tw.w3); }
  static __device__ __forceinline__ void apply_w3(f32x2 (&v)[8], const TwLite&, int tid) {
    // W_16^k = cos(pi k / 8) - i sin(pi k / 8), k = 1..7, correctly rounded
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, H = 0.70710678118654752f;
    const f32x2 c1 = {C1, -S1}, c2 = {H, -H}, c3 = {S1, -C1}, c4 = {0.f, -1.f}, c5 = {-S1, -C1}, c6 = {-H, -H}, c7 = {-C1, -S1};
#if defined(__HIP_DEVICE_COMPILE__)
    // one block: exec narrowed to the odd lanes, seven products (v_pk_mul + v_pk_fma each, in place, four in flight), exec
    // restored.  The factors are SGPR pairs; the even lanes' values are not touched.
    f32x2 t0, t1, t2, t3;
    unsigned long long saved;
#define DDSP_W3M(t, a, w) "v_pk_mul_f32 %[" #t "], %[" #a "], %[" #w "] op_sel_hi:[0,1]\n"
#define DDSP_W3F(t, a, w) "v_pk_fma_f32 %[" #a "], %[" #a "], %[" #w "], %[" #t "] op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n"
    asm("s_mov_b64 %[sv], exec\n"
        "s_mov_b32 exec_lo, 0xaaaaaaaa\n"
        "s_mov_b32 exec_hi, 0xaaaaaaaa\n"
        DDSP_W3M(t0, a1, w1) DDSP_W3M(t1, a2, w2) DDSP_W3M(t2, a3, w3) DDSP_W3M(t3, a4, w4)
        DDSP_W3F(t0, a1, w1) DDSP_W3F(t1, a2, w2) DDSP_W3F(t2, a3, w3) DDSP_W3F(t3, a4, w4)
        DDSP_W3M(t0, a5, w5) DDSP_W3M(t1, a6, w6) DDSP_W3M(t2, a7, w7)
        DDSP_W3F(t0, a5, w5) DDSP_W3F(t1, a6, w6) DDSP_W3F(t2, a7, w7)
        "s_mov_b64 exec, %[sv]\n"
        : [a1] "+v"(v[1]), [a2] "+v"(v[2]), [a3] "+v"(v[3]), [a4] "+v"(v[4]), [a5] "+v"(v[5]), [a6] "+v"(v[6]), [a7] "+v"(v[7]),
          [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [sv] "=&s"(saved)
        : [w1] "s"(c1), [w2] "s"(c2), [w3] "s"(c3), [w4] "s"(c4), [w5] "s"(c5), [w6] "s"(c6), [w7] "s"(c7));
#undef DDSP_W3M
#undef DDSP_W3F
#else
    if (tid & 1) {
      v[1] = cmul(v[1], c1); v[2] = cmul(v[2], c2); v[3] = cmul(v[3], c3); v[4] = cmul(v[4], c4);
      v[5] = cmul(v[5], c5); v[6] = cmul(v[6], c6); v[7] = cmul(v[7], c7);
    }
#endif
  }

  // v[n1] = z[P n1 + tid]  ->  v[m] = Z[P m + tid].  A and B hold N complex words each.
  // A must be free of readers on entry; on return A may still be read by slower waves (pass 4), B is free.
  // HI_ZERO: v[4..7] are zero on entry (an input zero-padded to twice its length) and need not be set.
  // The four passes, each between two barriers of forward():
  template <bool HI_ZERO>
  static __device__ __forceinline__ void pass1(f32x2 (&v)[8], const Tw& tw, f32x2* A, int tid) {
    if (HI_ZERO) dft8_lo4(v);
    else dft8(v);
    twiddle7(v, tw.w1);
    // R = 2: a row of the second pass is 16 words, so the two rows k1, k1 + 1 that a 32-lane read touches would start in
    // the same banks; odd rows are stored with their 16-blocks swapped in pairs (R >= 4: rows of 32 words or more, nothing to do)
    constexpr int SW = R == 2 ? C : 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) A[k * P + (tid ^ ((k & 1) * SW))] = v[k];            // [k1][p]
  }
  static __device__ __forceinline__ void pass2(f32x2 (&v)[8], const Tw& tw, const f32x2* A, f32x2* B, int tid) {
    const int k1 = tid / C, c = tid & (C - 1);
#pragma unroll
    for (int n2 = 0; n2 < 8; ++n2) v[n2] = A[k1 * P + (R == 2 ? (n2 ^ (k1 & 1)) : n2) * C + c];
    dft8(v);
    twiddle7(v, tw.w2);
    const int n3 = c / R, n4 = c & (R - 1);
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) B[n3 * P + ((k2 * C + k1 * R + n4) ^ (n3 * R))] = v[k2];   // [n3][k2][k1 ^ n3][n4]
  }
  static __device__ __forceinline__ void pass3(f32x2 (&v)[8], const Tw& tw, const f32x2* B, f32x2* A, int tid) {
#pragma unroll
    for (int n3 = 0; n3 < 8; ++n3) v[n3] = B[n3 * P + (tid ^ (n3 * R))];            // tid = n4 + R k1 + 8R k2
    dft8(v);
    twiddle7(v, tw.w3);
    // R >= 4: the last pass reads R consecutive words per thread, 16 bytes at a time.  Such reads are served in groups of
    // 16 lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 -- over 64 banks of 4 bytes, and with thread
    // (k1, k2) at k2 * 64 R + k1 * 8 R bytes the lanes of a group meet in the same banks (R = 4: k2 = 0 | 3 and 1 | 2, two-way;
    // R = 8: also k1 | k1 + 4, four-way).  So the 16-byte pairs of a thread's run are stored permuted -- [n4 ^ 2 (k2 bit 1)]
    // for R = 4, [n4 ^ 2 (k1 bit 2) ^ 4 (k2 bit 1)] for R = 8 -- and pass4 reads them back the same way (R = 4: as explicit
    // 16-byte loads; written as four 8-byte elements the compiler split them and the conflicts tripled):
    // SQ_LDS_BANK_CONFLICT 8 % of the LDS cycles -> 0 on both plans (profiles/r03_v34_loss_pmc.txt).
    const int at = R == 8 ? tid ^ (((tid >> 5) & 1) << 1) ^ (((tid >> 7) & 1) << 2)
                          : (R == 4 ? tid ^ (((tid >> 6) & 1) << 1) : tid);
#pragma unroll
    for (int k3 = 0; k3 < 8; ++k3) A[k3 * P + at] = v[k3];                          // [k3][k2][k1][n4]
  }
  static __device__ __forceinline__ void pass4(f32x2 (&v)[8], const f32x2* A, int tid) {
    const int k1 = tid & 7, k2 = (tid >> 3) & 7, k3lo = tid >> 6;
    const f32x2* src = A + k2 * C + k1 * R;
    f32x2 t[8];
    if constexpr (R == 8) {
      const int sw = (((k1 >> 2) & 1) << 1) ^ (((k2 >> 1) & 1) << 2);               // see pass3
#pragma unroll
      for (int n4 = 0; n4 < 8; ++n4) t[n4] = src[k3lo * P + (n4 ^ sw)];
      dft8(t);                                                                      // k4 = 0..7 -> slot m = k4
    } else if constexpr (R == 1) {
#pragma unroll
      for (int s = 0; s < 8; ++s) t[s] = src[s * P];                                // nothing left to transform: slot m = k3
    } else if constexpr (R == 4) {
      typedef float f32x4 __attribute__((ext_vector_type(4)));
      const int sw = ((k2 >> 1) & 1) << 1;                                          // see pass3: the run's 16-byte pairs swapped
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(src + (k3lo + 4 * s) * P + sw);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(src + (k3lo + 4 * s) * P + (2 ^ sw));
        f32x2 a0{lo.x, lo.y}, a1{lo.z, lo.w}, a2{hi.x, hi.y}, a3{hi.z, hi.w};
        dft4(a0, a1, a2, a3);                                                       // k4 = 0..3 -> slot m = s + 2 k4
        t[s] = a0; t[s + 2] = a1; t[s + 4] = a2; t[s + 6] = a3;
      }
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const f32x2 a0 = src[(k3lo + 2 * s) * P + 0], a1 = src[(k3lo + 2 * s) * P + 1];
        t[s] = a0 + a1;                                                             // k4 = 0, 1 -> slot m = s + 4 k4
        t[s + 4] = a0 - a1;
      }
    }
#pragma unroll
    for (int m = 0; m < 8; ++m) v[m] = t[m];
  }
  template <bool HI_ZERO = false>
  static __device__ __forceinline__ void forward(f32x2 (&v)[8], const Tw& tw, f32x2* A, f32x2* B, int tid) {
    pass1<HI_ZERO>(v, tw, A, tid);
    __syncthreads();
    pass2(v, tw, A, B, tid);
    __syncthreads();
    pass3(v, tw, B, A, tid);
    __syncthreads();
    pass4(v, A, tid);
  }
  // Two independent transforms in lockstep (v through A, B; u through A2, B2): the same arithmetic as two forward()
  // calls, behind shared barriers -- twice the work between two barriers, and one stream's exchange latency under the other's
  // butterflies.
  template <bool HI_ZERO = false>
  static __device__ __forceinline__ void forward2(f32x2 (&v)[8], f32x2 (&u)[8], const Tw& tw, f32x2* A, f32x2* B, f32x2* A2,
                                                  f32x2* B2, int tid) {
    pass1<HI_ZERO>(v, tw, A, tid);
    pass1<HI_ZERO>(u, tw, A2, tid);
    __syncthreads();
    pass2(v, tw, A, B, tid);
    pass2(u, tw, A2, B2, tid);
    __syncthreads();
    pass3(v, tw, B, A, tid);
    pass3(u, tw, B2, A2, tid);
    __syncthreads();
    pass4(v, A, tid);
    pass4(u, A2, tid);
  }

  // ---- the pair without the last exchange (R = 2) -------------------------------------------------------------
  // forward_s stops after pass 3 and takes the final DFT_2 over n4 -- the low lane bit -- across lane pairs (DPP), so
  // the spectrum stays in the "scrambled" layout S
  //     thread tid = k4 + 2 k1 + 16 k2, slot k3  <->  Z[k1 + 8 k2 + 64 k3 + 512 k4],
  // two LDS exchanges instead of three.  Pointwise spectral work does not care about the layout; a mirrored bin
  // Z[-k] is reached through s_index().  transposed() runs the same factorisation backwards (the DFT matrix is
  // symmetric, so the transposed algorithm is the DFT itself): it takes layout S and returns natural order
  // v[m] = X[P m + tid], again with two exchanges and with the same twiddle registers.
  static __device__ __forceinline__ int s_index(int tid, int slot) {
    static_assert(R == 2, "layout S is implemented for the 1024-point plan");
    return ((tid >> 1) & 7) + 8 * (tid >> 4) + 64 * slot + 512 * (tid & 1);
  }
  // Where bin k is parked for the mirrored read Z[-k]: lane pairs hold k and k + 512 -- the same banks -- so the upper
  // half is stored with bits 3 and 4 flipped (conflict-free 16-lane stores and, but for two lanes, 32-lane loads).
  static __device__ __forceinline__ int parked(int k) { return k ^ (((k >> 9) & 1) * 24); }
  // DFT_2 over the low lane bit: even lane a0 + a1, odd lane a0 - a1.
  // FLIP: one instruction per value instead of two -- "mine + sigma * neighbour's" with the DPP operand as the multiplicand
  // of a v_fmac_f32 (sigma = +1 on even, -1 on odd lanes): even lanes a0 + a1, odd lanes a1 - a0, i.e. the odd lanes hold
  // the NEGATED result.  That is layout S-: bin k of thread tid carries the factor sigma(tid) = (-1)^k4.  A pointwise
  // product of two S- spectra is a plain layout-S spectrum (sigma^2 = 1), and the transposed factorisation never mixes
  // the two lane parities again (n4 = tid & 1 stays a passive time-index digit), so transposed<FLIP> returns natural
  // time order with the samples of odd threads negated -- a sign the caller folds into its next multiply-add.
  template <bool FLIP = false>
  static __device__ __forceinline__ void lane_pair_dft2(f32x2 (&v)[8], int tid) {
    const float sgn = (tid & 1) ? -1.0f : 1.0f;
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (FLIP) {
      // sixteen v_fmac_f32 with a DPP multiplicand in one block (VOP2: the compiler's own form of this expression is two
      // v_mov_b32_dpp and one packed multiply-add per complex value).  The block opens with the two wait states a DPP
      // read needs behind the vector instruction that wrote its source; inside it no instruction reads another's result.
      float r[16];
#pragma unroll
      for (int k = 0; k < 8; ++k) { r[2 * k] = v[k].x; r[2 * k + 1] = v[k].y; }
#define DDSP_FD(i) "v_fmac_f32_dpp %" #i ", %" #i ", %16 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
      asm("s_nop 1\n" DDSP_FD(0) DDSP_FD(1) DDSP_FD(2) DDSP_FD(3) DDSP_FD(4) DDSP_FD(5) DDSP_FD(6) DDSP_FD(7)
          DDSP_FD(8) DDSP_FD(9) DDSP_FD(10) DDSP_FD(11) DDSP_FD(12) DDSP_FD(13) DDSP_FD(14) DDSP_FD(15)
          : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
            "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
          : "v"(sgn));
#undef DDSP_FD
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = f32x2{r[2 * k], r[2 * k + 1]};
      return;
    }
#endif
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      // mov_dpp (no "old" operand: every lane of a quad_perm has a valid source, so the destination needs no initial value)
      const float px = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v[k].x), 0xB1, 0xF, 0xF, true));
      const float py = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v[k].y), 0xB1, 0xF, 0xF, true));
      if (FLIP) v[k] = f32x2{fmaf(px, sgn, v[k].x), fmaf(py, sgn, v[k].y)};
      else v[k] = f32x2{fmaf(sgn, v[k].x, px), fmaf(sgn, v[k].y, py)};  // quad_perm [1,0,3,2]: the neighbour's value
    }
  }
  // v[n1] = z[P n1 + tid] -> v[k3] = Z[s_index(tid, k3)].  X must be free of readers on entry; on return X is free
  // and Y may still be read by slower waves.
  // HI_ZERO: v[4..7] are zero on entry (input zero-padded from 512 to 1024 points)
  // FLIP: layout S- (lane_pair_dft2 above)
  template <bool HI_ZERO = false, bool FLIP = false, class TW = Tw>
  static __device__ __forceinline__ void forward_s(f32x2 (&v)[8], const TW& tw, f32x2* X, f32x2* Y, int tid) {
    static_assert(R == 2, "layout S is implemented for the 1024-point plan");
    if (HI_ZERO) dft8_lo4(v);
    else dft8(v);
    twiddle7(v, tw.w1);
#pragma unroll
    for (int k = 0; k < 8; ++k) X[k * P + (tid ^ ((k & 1) * C))] = v[k];             // odd rows: 16-blocks swapped in pairs
    __syncthreads();
    {
      const int k1 = tid / C, c = tid & (C - 1);
#pragma unroll
      for (int n2 = 0; n2 < 8; ++n2) v[n2] = X[k1 * P + (n2 ^ (k1 & 1)) * C + c];    // rows k1, k1 + 1 of a 32-lane read: other banks
      dft8(v);
      twiddle7(v, tw.w2);
      const int n3 = c / R, n4 = c & (R - 1);
#pragma unroll
      for (int k2 = 0; k2 < 8; ++k2) Y[n3 * P + ((k2 * C + k1 * R + n4) ^ (n3 * R))] = v[k2];
    }
    __syncthreads();
#pragma unroll
    for (int n3 = 0; n3 < 8; ++n3) v[n3] = Y[n3 * P + (tid ^ (n3 * R))];
    dft8(v);
    apply_w3(v, tw, tid);
    lane_pair_dft2<FLIP>(v, tid);
  }
  // Two independent transforms in lockstep (u through X0 / Y0, v through X1 / Y1): the same passes as forward_s, but
  // each barrier serves both, and between two barriers a wave has the other transform's arithmetic to issue while one
  // transform's LDS round trip is in flight -- half the barriers, twice the independent work per interval.
  template <bool HI_ZERO = false, bool FLIP = false>
  static __device__ __forceinline__ void forward_s2(f32x2 (&u)[8], f32x2 (&v)[8], const Tw& tw, f32x2* X0, f32x2* Y0,
                                                    f32x2* X1, f32x2* Y1, int tid) {
    static_assert(R == 2, "layout S is implemented for the 1024-point plan");
    if (HI_ZERO) { dft8_lo4(u); dft8_lo4(v); }
    else { dft8(u); dft8(v); }
    twiddle7x2(u, tw.w1, v, tw.w1);
#pragma unroll
    for (int k = 0; k < 8; ++k) { X0[k * P + (tid ^ ((k & 1) * C))] = u[k]; X1[k * P + (tid ^ ((k & 1) * C))] = v[k]; }
    __syncthreads();
    {
      const int k1 = tid / C, c = tid & (C - 1);
#pragma unroll
      for (int n2 = 0; n2 < 8; ++n2) {
        u[n2] = X0[k1 * P + (n2 ^ (k1 & 1)) * C + c];
        v[n2] = X1[k1 * P + (n2 ^ (k1 & 1)) * C + c];
      }
      dft8(u);
      dft8(v);
      twiddle7x2(u, tw.w2, v, tw.w2);
      const int n3 = c / R, n4 = c & (R - 1);
#pragma unroll
      for (int k2 = 0; k2 < 8; ++k2) {
        const int a = n3 * P + ((k2 * C + k1 * R + n4) ^ (n3 * R));
        Y0[a] = u[k2];
        Y1[a] = v[k2];
      }
    }
    __syncthreads();
#pragma unroll
    for (int n3 = 0; n3 < 8; ++n3) { u[n3] = Y0[n3 * P + (tid ^ (n3 * R))]; v[n3] = Y1[n3 * P + (tid ^ (n3 * R))]; }
    dft8(u);
    dft8(v);
    twiddle7x2(u, tw.w3, v, tw.w3);
    lane_pair_dft2<FLIP>(u, tid);
    lane_pair_dft2<FLIP>(v, tid);
  }
  // v[k3] = Z[s_index(tid, k3)] -> v[m] = sum_k Z[k] W_N^(k (P m + tid)).  Y must be free of readers on entry; X
  // becomes free at the first barrier (its last readers are whoever used it before this call); on return Y is
  // free and X may still be read by slower waves.
  template <bool FLIP = false>
  static __device__ __forceinline__ void transposed(f32x2 (&v)[8], const Tw& tw, f32x2* Y, f32x2* X, int tid) {
    static_assert(R == 2, "layout S is implemented for the 1024-point plan");
    lane_pair_dft2<FLIP>(v, tid);
    twiddle7(v, tw.w3);
    dft8(v);
    const int ts = tid ^ (((tid / R) & 1) * C);                  // tid = n4 + R k1 + 8R k2: odd k1 sit in the neighbouring 16-block
#pragma unroll
    for (int n3 = 0; n3 < 8; ++n3) Y[n3 * P + (ts ^ (n3 * R))] = v[n3];
    __syncthreads();
    {
      const int k1 = tid / C, c = tid & (C - 1);
      const int n3 = c / R, n4 = c & (R - 1);
#pragma unroll
      for (int k2 = 0; k2 < 8; ++k2) v[k2] = Y[n3 * P + (((k2 ^ (k1 & 1)) * C + k1 * R + n4) ^ (n3 * R))];   // rows k1, k1 + 1: other banks
      twiddle7(v, tw.w2);
      dft8(v);
#pragma unroll
      for (int n2 = 0; n2 < 8; ++n2) X[k1 * P + n2 * C + c] = v[n2];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = X[k * P + tid];
    twiddle7(v, tw.w1);
    dft8(v);
  }
  // transposed(v) and forward_s<true>(u) in lockstep: an inverse transform (layout S -> natural order) beside the forward
  // transform of an unrelated zero-padded input (natural order -> layout S) -- the same two exchanges and three butterfly
  // columns each, in opposite order, so every barrier serves both and each wave has the other transform's arithmetic to
  // issue while one's LDS round trip is in flight.  Four distinct buffers: Yv and Xu must be free of readers on entry,
  // Xv and Yu become free at the first barrier; on return Yv and Xu are free, Xv and Yu may still be read by slower waves.
  // U_PRUNED: u[4..7] are zero on entry (a zero-padded input, as forward_s<true>); false: a full 1024-point input
  template <bool FLIP = false, bool U_PRUNED = true>
  static __device__ __forceinline__ void transposed_and_forward_s(f32x2 (&v)[8], f32x2 (&u)[8], const Tw& tw, f32x2* Yv,
                                                                  f32x2* Xv, f32x2* Xu, f32x2* Yu, int tid) {
    static_assert(R == 2, "layout S is implemented for the 1024-point plan");
    lane_pair_dft2<FLIP>(v, tid);
    if (U_PRUNED) dft8_lo4(u);
    else dft8(u);
    twiddle7x2(v, tw.w3, u, tw.w1);
    dft8(v);
    const int ts = tid ^ (((tid / R) & 1) * C);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      Yv[k * P + (ts ^ (k * R))] = v[k];
      Xu[k * P + (tid ^ ((k & 1) * C))] = u[k];
    }
    __syncthreads();
    {
      const int k1 = tid / C, c = tid & (C - 1);
      const int n3 = c / R, n4 = c & (R - 1);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        v[k] = Yv[n3 * P + (((k ^ (k1 & 1)) * C + k1 * R + n4) ^ (n3 * R))];
        u[k] = Xu[k1 * P + (k ^ (k1 & 1)) * C + c];
      }
      dft8(u);
      twiddle7x2(v, tw.w2, u, tw.w2);
      dft8(v);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        Xv[k1 * P + k * C + c] = v[k];
        Yu[n3 * P + ((k * C + k1 * R + n4) ^ (n3 * R))] = u[k];
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) { v[k] = Xv[k * P + tid]; u[k] = Yu[k * P + (tid ^ (k * R))]; }
    dft8(u);
    twiddle7x2(v, tw.w1, u, tw.w3);
    dft8(v);
    lane_pair_dft2<FLIP>(u, tid);
  }
  // The same two transforms on THREE buffers: u runs one exchange behind v, so that v's first buffer is free again when u
  // needs its second one.  Three barriers instead of the lockstep form's two, but 24 KB of exchange space instead of 32 --
  // what a kernel that wants six 128-thread workgroups per CU (three waves per SIMD) can afford.
  //     interval 0   v: first column -> Y
  //     interval 1   v: Y -> middle column -> X          u: first column -> Q
  //     interval 2   v: X -> last column (done)          u: Q -> middle column -> Y
  //     interval 3                                       u: Y -> last column, lane-pair step (done)
  // Y and Q must be free of readers on entry (Q: by the first barrier), X becomes free at the first barrier; on return X and
  // Q are free and Y may still be read by slower waves.
  template <bool FLIP = false, bool U_PRUNED = true, class TW = Tw>
  static __device__ __forceinline__ void transposed_then_forward_s(f32x2 (&v)[8], f32x2 (&u)[8], const TW& tw, f32x2* Y,
                                                                   f32x2* X, f32x2* Q, int tid) {
    static_assert(R == 2, "layout S is implemented for the 1024-point plan");
    const int k1 = tid / C, c = tid & (C - 1);
    const int n3 = c / R, n4 = c & (R - 1);
    lane_pair_dft2<FLIP>(v, tid);
    apply_w3(v, tw, tid);
    dft8(v);
    const int ts = tid ^ (((tid / R) & 1) * C);
#pragma unroll
    for (int k = 0; k < 8; ++k) Y[k * P + (ts ^ (k * R))] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = Y[n3 * P + (((k ^ (k1 & 1)) * C + k1 * R + n4) ^ (n3 * R))];
    if (U_PRUNED) dft8_lo4(u);
    else dft8(u);
    twiddle7x2(v, tw.w2, u, tw.w1);
    dft8(v);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      X[k1 * P + k * C + c] = v[k];
      Q[k * P + (tid ^ ((k & 1) * C))] = u[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) { v[k] = X[k * P + tid]; u[k] = Q[k1 * P + (k ^ (k1 & 1)) * C + c]; }
    dft8(u);
    twiddle7x2(v, tw.w1, u, tw.w2);
    dft8(v);
#pragma unroll
    for (int k = 0; k < 8; ++k) Y[n3 * P + ((k * C + k1 * R + n4) ^ (n3 * R))] = u[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) u[k] = Y[k * P + (tid ^ (k * R))];
    dft8(u);
    apply_w3(u, tw, tid);
    lane_pair_dft2<FLIP>(u, tid);
  }
};

}  // namespace fft
}  // namespace ddsp
