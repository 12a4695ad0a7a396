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

// The raw generator: four 32-bit words of Philox4x32-10 for one counter / key.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t (&x)[4]) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
    const uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += W0; k1 += W1;
  }
  x[0] = c0; x[1] = c1; x[2] = c2; x[3] = c3;
}

// Standard-normal draw for the harmonic source of NSF-HiFiGAN (nsf_hifigan/models.py:168 draws torch.randn_like(sine_waves):
// [B, T, dim] values -- THIS IS A DIFFERENT STREAM, as the uniform draw above).  Four normals per counter by Box-Muller:
// counter = (t, utterance, offset_lo, 4 * offset_hi + j), key = (seed_lo, seed_hi ^ 'NORM'), j < 4; the words (x0, x1) and (x2, x3) give
//     u1 = ((x >> 8) + 1) 2^-24 in (0, 1],  u2 = (x' >> 8) 2^-24 in [0, 1),  r = sqrt(-2 ln u1),  z = r cos(2 pi u2), r sin(2 pi u2)
// and harmonic h of sample t takes normal h % 4 of call j = h / 4 (|z| <= 5.77).  The hardware log2 / sine / cosine (the latter
// two take revolutions: u2 itself) are within ~1e-6 of the float64 restatement in oracle/ddsp_oracle.py.
struct Normal4 { float z[4]; };

__device__ __forceinline__ Normal4 philox_normal4(const NoiseGen& g, unsigned utterance, unsigned t, unsigned j) {
  uint32_t x[4];
  // the key's high word carries a domain tag ("NORM"): with the plain key the words of (seed, offset) were, for offset_hi = 0 and
  // j = 0, the very words the UNIFORM draw above maps to the noise branch's samples -- two draws a caller may seed alike
  philox4x32_10(t, utterance + g.utt0, (uint32_t)g.offset, 4u * (uint32_t)(g.offset >> 32) + j, (uint32_t)g.seed,
                (uint32_t)(g.seed >> 32) ^ 0x4E4F524Du, x);
  Normal4 n;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const float u1 = (float)((x[2 * p] >> 8) + 1u) * 5.9604644775390625e-8f;
    const float u2 = (float)(x[2 * p + 1] >> 8) * 5.9604644775390625e-8f;
    const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));     // -2 ln 2 * log2(u1); hardware root (1 ulp) behind the hardware log
    n.z[2 * p] = r * __builtin_amdgcn_cosf(u2);
    n.z[2 * p + 1] = r * __builtin_amdgcn_sinf(u2);
  }
  return n;
}

}  // namespace ddsp
