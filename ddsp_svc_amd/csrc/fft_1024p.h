// The 1024-point plan of fft_r.h (8 x 8 x 8 x 2, 128 threads, 8 points per thread, layout S / S-) on PADDED exchange rows, for
// the kernel that has to live on 168 registers (k_fir_blk6, three waves per SIMD).
//
// fft_r.h keeps its exchanges conflict-free with XOR swizzles (tid ^ 2 n3, ...): eight distinct addresses per exchange that
// the compiler either keeps in registers for the whole loop (k_fir_blk: ~40 of its 226) or recomputes every pass
// (measured: +146 vector instructions per pass, which ate all that the third wave per SIMD had bought).  Here every exchange
// is ONE base register plus instruction immediates:
//     first exchange     [k1][p], rows of 128 words; odd rows with their 16-blocks swapped in pairs (two bases, as fft_r.h)
//     second exchange    [n3][k2 k1 n4], rows of 130 words: the 16 lanes of a store group (k1 fixed; n3, n4 vary) fall into
//                        16 different bank pairs (130 = 2 mod 16), the reads are lane-consecutive
//     transposed first   [n3][k2 k1 n4] again, rows of 132 words: the 32 lanes of a read group (two k1; n3, n4 vary) cover
//                        all 64 banks (132 = 4 mod 32), the stores are lane-consecutive
//     transposed second  [k1][n2 c], rows of 128 words, lane-consecutive both ways
// and the mirrored read Z[-k] of a parked spectrum is one base too (mirror_base below).  A buffer is 8 x 132 = 1056 words.
#pragma once
#include "fft_r.h"

namespace ddsp {
namespace fft {

// Timing experiments only (tools/gpu_r04.sh; results are WRONG under any of them): DDSP_ABL_NOLDS drops the exchange traffic,
// DDSP_ABL_NOMATH the butterflies and twiddles, DDSP_ABL_NOBAR the barriers -- what each part of a pass costs beside the others.
#if defined(DDSP_ABL_NOLDS)
#define DDSP_P_ST(dst, val) do { } while (0)
#define DDSP_P_LD(dst, src) do { } while (0)
#else
#define DDSP_P_ST(dst, val) (dst) = (val)
#define DDSP_P_LD(dst, src) (dst) = rd(src)
#endif
#if defined(DDSP_ABL_NOMATH)
#define DDSP_P_MATH(x) do { } while (0)
#else
#define DDSP_P_MATH(x) x
#endif
#if defined(DDSP_ABL_NOBAR)
#define DDSP_P_SYNC() do { } while (0)
#else
#define DDSP_P_SYNC() __syncthreads()
#endif

struct Plan1024P {
  using Base = Plan<2>;
  using Tw = Base::TwLite;
  static constexpr int N = 1024, P = 128, C = 16, R = 2;
  static constexpr int S2 = 130, S3 = 132;
  static constexpr int WORDS = 8 * S3;                            // complex words per exchange buffer (8448 bytes)

  // the loop-invariant word offsets of a thread (eight registers; everything else is an immediate)
  struct Ix {
    int t;          // tid: second-exchange reads, transposed stores and last reads
    int w1a, w1b;   // first-exchange stores of even / odd rows: tid, tid ^ 16
    int r1e, r1o;   // first-exchange reads of even / odd n2
    int w2;         // second-exchange stores: n3 * 130 + 2 k1 + n4
    int r3;         // transposed first-exchange reads: n3 * 132 + 2 k1 + n4
    int w4;         // transposed second-exchange stores: 128 k1 + c
    __device__ __forceinline__ void init(int tid) {
      const int k1 = tid / C, c = tid & (C - 1), n3 = c / R, n4 = c & (R - 1);
      t = tid;
      w1a = tid;
      w1b = tid ^ C;
      r1e = k1 * P + c + (k1 & 1) * C;
      r1o = k1 * P + c - (k1 & 1) * C;
      w2 = n3 * S2 + k1 * R + n4;
      r3 = n3 * S3 + k1 * R + n4;
      w4 = k1 * P + c;
    }
  };

  // An exchange READ, as a volatile access: that keeps the compiler from pairing two of them into one ds_read2_b64 -- a form
  // served in four 16-lane groups per half (8 LDS cycles for 16 bytes per lane) where two ds_read_b64 take 2 cycles each
  // (MI355X_MICROARCH.md, LDS table), and these layouts are conflict-free for the 32-lane form.  Measured [MI355X]
  // (profiles/r04_v2_*): SQ_LDS_IDX_ACTIVE 21.7 M -> 16.3 M cycles per launch, kernel alone 79.7 -> 76.1 us.
  // (-DDDSP_P_PAIRED_READS: the compiler's pairing, for A/B runs)
  static __device__ __forceinline__ f32x2 rd(const f32x2* p) {
#if !defined(DDSP_P_PAIRED_READS) && defined(__HIP_DEVICE_COMPILE__)
    return *(const volatile __attribute__((address_space(3))) f32x2*)p;
#else
    return *p;
#endif
  }

  // a read of a PARKED spectrum (own slot / mirror image): paired by the compiler unless DDSP_P_SINGLE_PARKED
  static __device__ __forceinline__ f32x2 rd_parked(const f32x2* p) {
#if defined(DDSP_P_SINGLE_PARKED) && defined(__HIP_DEVICE_COMPILE__)
    return *(const volatile __attribute__((address_space(3))) f32x2*)p;
#else
    return *p;
#endif
  }
  static __device__ __forceinline__ int s_index(int tid, int slot) { return Base::s_index(tid, slot); }
  static __device__ __forceinline__ int parked(int k) { return Base::parked(k); }
  // Where thread tid finds the mirror image of its slot m, parked(-(s_index(tid, m)) mod 1024), as base - 64 m: with
  // k = a + 64 m + 512 k4 (a = k1 + 8 k2 < 64, k4 = tid & 1) the mirror image is (64 - a) + 64 (15 - m - 8 k4), whose bit 9
  // is 1 - k4 for every m, so the bank swizzle of parked() is a per-thread constant.  Threads 0 and 1 (a = 0) hold the two
  // bins that are their own mirror images in slot 0: for them base - 64 m is right for m >= 1 and points at an unrelated
  // word INSIDE the buffer for m = 0 -- the caller takes the thread's own value there.
  static __device__ __forceinline__ int mirror_base(int tid) {
    const int a = ((tid >> 1) & 7) + 8 * (tid >> 4), k4 = tid & 1;
    return ((64 - a) ^ (24 * (1 - k4))) + 64 * (15 - 8 * k4);
  }

  // v[n1] = z[128 n1 + tid] -> v[k3] = Z[s_index(tid, k3)] (layout S, FLIP: S-).  X must be free of readers on entry; on
  // return X is free and Y may still be read by slower waves.  HI_ZERO: v[4..7] are zero on entry.
  template <bool HI_ZERO = false, bool FLIP = false>
  static __device__ __forceinline__ void forward_s(f32x2 (&v)[8], const Tw& tw, f32x2* X, f32x2* Y, const Ix& ix) {
    DDSP_P_MATH({
    if (HI_ZERO) dft8_lo4(v);
    else dft8(v);
    twiddle7(v, tw.w1);
    });
#pragma unroll
    for (int k = 0; k < 8; ++k) DDSP_P_ST(X[k * P + ((k & 1) ? ix.w1b : ix.w1a)], v[k]);
    DDSP_P_SYNC();
#pragma unroll
    for (int n2 = 0; n2 < 8; ++n2) DDSP_P_LD(v[n2], X + ((n2 & 1) ? ix.r1o : ix.r1e) + n2 * C);
    DDSP_P_MATH({
    dft8(v);
    twiddle7(v, tw.w2);
    });
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) DDSP_P_ST(Y[ix.w2 + k2 * C], v[k2]);
    DDSP_P_SYNC();
#pragma unroll
    for (int n3 = 0; n3 < 8; ++n3) DDSP_P_LD(v[n3], Y + n3 * S2 + ix.t);
    DDSP_P_MATH({
    dft8(v);
    Base::apply_w3(v, tw, ix.t);
    Base::template lane_pair_dft2<FLIP>(v, ix.t);
    });
  }

  // The inverse (layout S -> natural order, v) and one exchange behind it the forward transform of a zero-padded input
  // (natural order -> layout S, u) on three buffers -- fft_r.h, transposed_then_forward_s:
  //     interval 0   v: first column -> Y
  //     interval 1   v: Y -> middle column -> X          u: first column -> Q
  //     interval 2   v: X -> last column (done)          u: Q -> middle column -> Y
  //     interval 3                                       u: Y -> last column, lane-pair step (done)
  // Y and Q must be free of readers on entry (Q: by the first barrier), X becomes free at the first barrier; on return X and
  // Q are free and Y may still be read by slower waves.
  // U_FULL: u is a full 1024-point input (k_fir_blk_bwd6's cotangent pair) instead of one zero-padded to twice its length
  template <bool FLIP = false, bool U_FULL = false>
  static __device__ __forceinline__ void transposed_then_forward_s(f32x2 (&v)[8], f32x2 (&u)[8], const Tw& tw, f32x2* Y, f32x2* X,
                                                                   f32x2* Q, const Ix& ix) {
    DDSP_P_MATH({
    Base::template lane_pair_dft2<FLIP>(v, ix.t);
    Base::apply_w3(v, tw, ix.t);
    dft8(v);
    });
#pragma unroll
    for (int k = 0; k < 8; ++k) DDSP_P_ST(Y[k * S3 + ix.t], v[k]);
    DDSP_P_SYNC();
#pragma unroll
    for (int k = 0; k < 8; ++k) DDSP_P_LD(v[k], Y + ix.r3 + k * C);
    DDSP_P_MATH({
    if (U_FULL) dft8(u);
    else dft8_lo4(u);
    twiddle7x2(v, tw.w2, u, tw.w1);
    dft8(v);
    });
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      DDSP_P_ST(X[ix.w4 + k * C], v[k]);
      DDSP_P_ST(Q[k * P + ((k & 1) ? ix.w1b : ix.w1a)], u[k]);
    }
    DDSP_P_SYNC();
#pragma unroll
    for (int k = 0; k < 8; ++k) { DDSP_P_LD(v[k], X + k * P + ix.t); DDSP_P_LD(u[k], Q + ((k & 1) ? ix.r1o : ix.r1e) + k * C); }
    DDSP_P_MATH({
    dft8(u);
    twiddle7x2(v, tw.w1, u, tw.w2);
    dft8(v);
    });
#pragma unroll
    for (int k = 0; k < 8; ++k) DDSP_P_ST(Y[ix.w2 + k * C], u[k]);
    DDSP_P_SYNC();
#pragma unroll
    for (int k = 0; k < 8; ++k) DDSP_P_LD(u[k], Y + k * S2 + ix.t);
    DDSP_P_MATH({
    dft8(u);
    Base::apply_w3(u, tw, ix.t);
    Base::template lane_pair_dft2<FLIP>(u, ix.t);
    });
  }
};

}  // namespace fft
}  // namespace ddsp
