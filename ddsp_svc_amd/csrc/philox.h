// Counter-based uniform noise for the in-kernel draw of the noise branch (opt-in; ddsp/vocoder.py:603,854 draw
// torch.rand_like(harmonic) from torch's global generator -- THIS IS A DIFFERENT STREAM: reproducible from (seed, offset),
// independent of launch geometry, but not the numbers torch.rand would produce for the same seed).
//
// Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11; the generator behind curand /
// torch on GPUs): counter = (128 * block + lane, utterance, offset_lo, offset_hi), key = (seed_lo, seed_hi), where
// `block` is the hop-512 block of the sample and `lane` its position modulo 128; the four 32-bit outputs are the
// samples 512 block + 128 m + lane, m = 0..3 (the four samples one thread of the filter kernel owns in that block), each
// mapped to u = (x >> 8) * 2^-24 in [0, 1).  oracle/ddsp_oracle.py restates it in numpy.
#pragma once
#include <stdint.h>

namespace ddsp {

struct NoiseGen {
  unsigned long long seed, offset;
  int on;
  unsigned utt0;      // utterance number of the launch's first row (a sub-batch of a larger call draws ITS utterances' numbers)
};

struct Quad { float u[4]; };

__device__ __forceinline__ Quad philox_uniform4(const NoiseGen& g, unsigned utterance, unsigned block, unsigned lane) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
  uint32_t c0 = 128u * block + lane, c1 = utterance + g.utt0, c2 = (uint32_t)g.offset, c3 = (uint32_t)(g.offset >> 32);
  uint32_t k0 = (uint32_t)g.seed, k1 = (uint32_t)(g.seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
    const uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += W0; k1 += W1;
  }
  Quad q;
  q.u[0] = (float)(c0 >> 8) * 5.9604644775390625e-8f;      // 2^-24
  q.u[1] = (float)(c1 >> 8) * 5.9604644775390625e-8f;
  q.u[2] = (float)(c2 >> 8) * 5.9604644775390625e-8f;
  q.u[3] = (float)(c3 >> 8) * 5.9604644775390625e-8f;
  return q;
}

}  // namespace ddsp
