// Unit-test kernels for the in-register / LDS FFT plans of ddsp_svc_amd/csrc/fft_r.h, run under the CPU emulator
// (tests/test_fft_plans.py compares them with numpy.fft).  TEST INFRASTRUCTURE ONLY.
#include "fft_1024p.h"

namespace {
using ddsp::f32x2;
using ddsp::fft::Plan;

// mode 0: Plan<R>::forward                 natural order in  -> natural order out
// mode 1: Plan<2>::forward_s<false>        natural in        -> out[s_index(tid, slot)]  (so out is in natural bin order)
// mode 2: Plan<2>::forward_s<true>         (upper half of the input must be zero)
// mode 3: Plan<2>::transposed              in[k] is read into layout S, out natural = DFT of in
// mode 4: Plan<2>::forward_s2<true>        two inputs (in, in2) -> (out, out2), both as mode 2
// mode 5: Plan<2>::transposed_and_forward_s   in -> out as mode 3, in2 -> out2 as mode 2
// modes 10..12 (every R): the lockstep pair forward2 (full / zero-padded inputs) and the zero-padded forward
// modes 13, 14: Plan<2>::transposed_then_forward_s (the three-buffer, staggered form of mode 5), plain and in layout S-
// modes 15, 16: the padded-row plan of fft_1024p.h (k_fir_blk6): forward_s<true, true> as mode 6, transposed_then_forward_s<true> as
//                mode 14; mode 17: its mirror_base against parked(-k) for every slot but the two self-mirrored ones (out = count of mismatches)
// modes 6..9: modes 2..5 in the sign-carrying layout S- (FLIP = true): what odd threads hold is written out negated
//             again, so the expected values are those of modes 2..5
template <int R>
__global__ void k_plan(int mode, const f32x2* in, const f32x2* in2, f32x2* out, f32x2* out2) {
  using PL = Plan<R>;
  constexpr int N = PL::N, P = PL::P;
  __shared__ __attribute__((aligned(16))) f32x2 ex[4][N];
  const int tid = threadIdx.x;
  typename PL::Tw tw;
  tw.init(tid);
  f32x2 v[8], u[8];
  if (mode == 0) {
    for (int m = 0; m < 8; ++m) v[m] = in[P * m + tid];
    PL::forward(v, tw, ex[0], ex[1], tid);
    for (int m = 0; m < 8; ++m) out[P * m + tid] = v[m];
  }
  if (mode == 10 || mode == 11) {                            // 10: forward2, 11: forward2<true> (upper halves of the inputs zero)
    for (int m = 0; m < 8; ++m) { v[m] = in[P * m + tid]; u[m] = in2[P * m + tid]; }
    if (mode == 10) PL::forward2(v, u, tw, ex[0], ex[1], ex[2], ex[3], tid);
    else PL::template forward2<true>(v, u, tw, ex[0], ex[1], ex[2], ex[3], tid);
    for (int m = 0; m < 8; ++m) { out[P * m + tid] = v[m]; out2[P * m + tid] = u[m]; }
  }
  if (mode == 12) {                                          // forward<true>: the pruned first pass alone
    for (int m = 0; m < 4; ++m) v[m] = in[P * m + tid];
    PL::template forward<true>(v, tw, ex[0], ex[1], tid);
    for (int m = 0; m < 8; ++m) out[P * m + tid] = v[m];
  }
  if constexpr (R == 2) {
    if (mode == 1 || mode == 2) {
      for (int m = 0; m < 8; ++m) v[m] = in[P * m + tid];
      if (mode == 1) PL::template forward_s<false>(v, tw, ex[0], ex[1], tid);
      else PL::template forward_s<true>(v, tw, ex[0], ex[1], tid);
      for (int m = 0; m < 8; ++m) out[PL::s_index(tid, m)] = v[m];
    } else if (mode == 3) {
      for (int m = 0; m < 8; ++m) v[m] = in[PL::s_index(tid, m)];
      PL::transposed(v, tw, ex[0], ex[1], tid);
      for (int m = 0; m < 8; ++m) out[P * m + tid] = v[m];
    } else if (mode == 4) {
      for (int m = 0; m < 8; ++m) { v[m] = in[P * m + tid]; u[m] = in2[P * m + tid]; }
      PL::template forward_s2<true>(v, u, tw, ex[0], ex[1], ex[2], ex[3], tid);
      for (int m = 0; m < 8; ++m) { out[PL::s_index(tid, m)] = v[m]; out2[PL::s_index(tid, m)] = u[m]; }
    } else if (mode == 5) {
      for (int m = 0; m < 8; ++m) { v[m] = in[PL::s_index(tid, m)]; u[m] = in2[P * m + tid]; }
      PL::transposed_and_forward_s(v, u, tw, ex[0], ex[1], ex[2], ex[3], tid);
      for (int m = 0; m < 8; ++m) { out[P * m + tid] = v[m]; out2[PL::s_index(tid, m)] = u[m]; }
    }
    const float sg = (tid & 1) ? -1.0f : 1.0f;
    if (mode == 13 || mode == 14) {
      for (int m = 0; m < 8; ++m) { v[m] = in[PL::s_index(tid, m)]; u[m] = in2[P * m + tid]; }
      if (mode == 13) PL::transposed_then_forward_s(v, u, tw, ex[0], ex[1], ex[2], tid);
      else PL::template transposed_then_forward_s<true>(v, u, tw, ex[0], ex[1], ex[2], tid);
      const float s2 = mode == 14 ? sg : 1.0f;
      for (int m = 0; m < 8; ++m) { out[P * m + tid] = v[m] * s2; out2[PL::s_index(tid, m)] = u[m] * s2; }
    }
    if (mode >= 15 && mode <= 17) {
      using PP = ddsp::fft::Plan1024P;
      __shared__ __attribute__((aligned(16))) f32x2 exp_[3][PP::WORDS];
      typename PP::Tw twl;
      twl.init(tid);
      typename PP::Ix ix;
      ix.init(tid);
      if (mode == 15) {
        for (int m = 0; m < 8; ++m) v[m] = in[P * m + tid];
        PP::template forward_s<true, true>(v, twl, exp_[0], exp_[1], ix);
        for (int m = 0; m < 8; ++m) out[PL::s_index(tid, m)] = v[m] * sg;
      } else if (mode == 16) {
        for (int m = 0; m < 8; ++m) { v[m] = in[PL::s_index(tid, m)]; u[m] = in2[P * m + tid]; }
        PP::template transposed_then_forward_s<true>(v, u, twl, exp_[0], exp_[1], exp_[2], ix);
        for (int m = 0; m < 8; ++m) { out[P * m + tid] = v[m] * sg; out2[PL::s_index(tid, m)] = u[m] * sg; }
      } else {
        int bad = 0;
        const int mb = PP::mirror_base(tid);
        for (int m = 0; m < 8; ++m) {
          const int want = PP::parked((1024 - PL::s_index(tid, m)) & 1023);
          const bool self = tid < 2 && m == 0;
          if (!self && mb - 64 * m != want) ++bad;
          if (mb - 64 * m < 0 || mb - 64 * m >= PP::WORDS) ++bad;
        }
        out[tid] = f32x2{(float)bad, 0.f};
      }
    }
    if (mode == 6) {
      for (int m = 0; m < 8; ++m) v[m] = in[P * m + tid];
      PL::template forward_s<true, true>(v, tw, ex[0], ex[1], tid);
      for (int m = 0; m < 8; ++m) out[PL::s_index(tid, m)] = v[m] * sg;
    } else if (mode == 7) {
      for (int m = 0; m < 8; ++m) v[m] = in[PL::s_index(tid, m)];
      PL::template transposed<true>(v, tw, ex[0], ex[1], tid);
      for (int m = 0; m < 8; ++m) out[P * m + tid] = v[m] * sg;
    } else if (mode == 8) {
      for (int m = 0; m < 8; ++m) { v[m] = in[P * m + tid]; u[m] = in2[P * m + tid]; }
      PL::template forward_s2<true, true>(v, u, tw, ex[0], ex[1], ex[2], ex[3], tid);
      for (int m = 0; m < 8; ++m) { out[PL::s_index(tid, m)] = v[m] * sg; out2[PL::s_index(tid, m)] = u[m] * sg; }
    } else if (mode == 9) {
      for (int m = 0; m < 8; ++m) { v[m] = in[PL::s_index(tid, m)]; u[m] = in2[P * m + tid]; }
      PL::template transposed_and_forward_s<true>(v, u, tw, ex[0], ex[1], ex[2], ex[3], tid);
      for (int m = 0; m < 8; ++m) { out[P * m + tid] = v[m] * sg; out2[PL::s_index(tid, m)] = u[m] * sg; }
    }
  }
}
}  // namespace

// complex arrays as interleaved float pairs; returns 0
extern "C" int emu_fft_plan(int R, int mode, const float* in, const float* in2, float* out, float* out2) {
  const f32x2* a = reinterpret_cast<const f32x2*>(in);
  const f32x2* b = reinterpret_cast<const f32x2*>(in2);
  f32x2* c = reinterpret_cast<f32x2*>(out);
  f32x2* d = reinterpret_cast<f32x2*>(out2);
  if (R == 1) hipLaunchKernelGGL(k_plan<1>, dim3(1), dim3(64), 0, nullptr, mode, a, b, c, d);
  else if (R == 2) hipLaunchKernelGGL(k_plan<2>, dim3(1), dim3(128), 0, nullptr, mode, a, b, c, d);
  else if (R == 4) hipLaunchKernelGGL(k_plan<4>, dim3(1), dim3(256), 0, nullptr, mode, a, b, c, d);
  else if (R == 8) hipLaunchKernelGGL(k_plan<8>, dim3(1), dim3(512), 0, nullptr, mode, a, b, c, d);
  else return -1;
  return 0;
}
