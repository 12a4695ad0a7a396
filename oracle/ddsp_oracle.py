"""CPU oracle for the DDSP harmonic-plus-noise hot path (TEST INFRASTRUCTURE ONLY).

This file is the *checker*, never the product: only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it.  Nothing under ``ddsp_svc_amd/``
imports it, and the product path raises if the HIP library is missing.

It restates, in numpy, what ``/root/reference/ddsp/core.py`` and the DSP tails of
``ddsp/vocoder.py`` ``Sins.forward`` / ``CombSub.forward`` compute.  Every function cites the
reference lines it follows.  Arithmetic is float64 *except* at the handful of places where
the reference's float32 rounding changes results at the 1e-7 level or above (per-sample f0
interpolation before the phase scan, ``phase * k``, ``pi * z`` inside sinc, window/mask
threshold compares) -- those are reproduced in float32 on purpose.

Parity pinning: the reference ships no golden vectors, so this oracle is
pinned against outputs of the reference itself, generated in the build container by
``tests/golden/make_golden.py`` (which imports ``/root/reference``) and committed as
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks every fixture.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
F64 = np.float64
PI32 = F32(np.pi)
TWO_PI32 = F32(2.0 * np.pi)

MODE_ROLL = 0      # hann_window=False                      (core.py:269)
MODE_HANN = 1      # hann_window=True, half_width=None      (core.py:263-264 -> :185-237)
MODE_DYNAMIC = 2   # hann_window=True, half_width given     (core.py:265-266 -> :240-251)


# --------------------------------------------------------------------------------------
# a1  control-rate -> sample-rate linear interpolation                      core.py:66-70
# --------------------------------------------------------------------------------------
def _lerp_weights(hop: int):
    j = np.arange(hop, dtype=F32)
    w1 = (j / F32(hop)).astype(F32)          # exact for power-of-two hops, as in ATen
    w0 = (F32(1.0) - w1).astype(F32)
    return w0, w1


def upsample(sig: np.ndarray, hop: int) -> np.ndarray:
    """``[B,F,C] -> [B,F*hop,C]`` float32, last frame held.

    core.py:66-70 appends a copy of the last frame and calls
    ``interpolate(mode='linear', align_corners=True)`` to ``F*hop+1`` points, so the source
    coordinate of output ``t`` is exactly ``t/hop``.  ATen's CPU kernel evaluates the two-tap
    blend as ``fma(w0, a, fl32(w1*b))`` (probed bit-exact on 2.10 / AVX2); the same expression
    is used here and in the HIP kernels so the per-sample f0 fed to the phase scan is
    identical.
    """
    sig = np.asarray(sig, dtype=F32)
    B, Fr, C = sig.shape
    nxt = np.concatenate([sig[:, 1:], sig[:, -1:]], axis=1)
    w0, w1 = _lerp_weights(hop)
    a = sig[:, :, None, :]
    b = nxt[:, :, None, :]
    p1 = (w1[None, None, :, None] * b).astype(F32)                      # fl32(w1*b)
    out = (w0[None, None, :, None].astype(F64) * a.astype(F64) + p1.astype(F64)).astype(F32)
    return out.reshape(B, Fr * hop, C)


# --------------------------------------------------------------------------------------
# a2  phase accumulation                     vocoder.py:564-575 (Sins), :819-829 (CombSub)
# --------------------------------------------------------------------------------------
def wrapped_phase(f0_frames: np.ndarray, sr: float, hop: int,
                  initial_phase: np.ndarray | None = None, infer: bool = True):
    """Return ``(x, phase_frames)``: ``x [B,T]`` float32 in [-0.5, 0.5] cycles and
    ``phase_frames [B,F]`` float32 radians (``2*pi*x[:, ::hop]``).

    infer=True : ``cumsum(f0.double()/sr)``   -- float64 terms, float64 running sum (:566)
    infer=False: ``cumsum(f0/sr)`` on float32 -- ATen's CPU cumsum keeps a float64 running
                 sum and rounds every *output* to float32 (probed), then wraps in float32 (:568)
    ``initial_phase`` (radians, one per utterance) is added as ``ip/2/pi`` in x's dtype (:569-570).
    ``x - round(x)`` uses round-half-to-even (:571).
    """
    f0f = np.asarray(f0_frames, dtype=F32)
    if f0f.ndim == 2:
        f0f = f0f[:, :, None]
    f0 = upsample(f0f, hop)[:, :, 0]                                    # [B,T] float32
    if infer:
        x = np.cumsum(f0.astype(F64) / F64(sr), axis=1)
        if initial_phase is not None:
            x = x + (np.asarray(initial_phase, dtype=F32).reshape(-1, 1).astype(F64) / 2.0 / np.pi)
        x = x - np.rint(x)
        x = x.astype(F32)
    else:
        terms = (f0 / F32(sr)).astype(F32)
        x = np.cumsum(terms.astype(F64), axis=1).astype(F32)
        if initial_phase is not None:
            ip = np.asarray(initial_phase, dtype=F32).reshape(-1, 1)
            x = (x + ((ip / F32(2.0)).astype(F32) / PI32).astype(F32)).astype(F32)
        x = (x - np.rint(x)).astype(F32)
    phase_frames = (TWO_PI32 * x[:, ::hop]).astype(F32)
    return x, phase_frames


# --------------------------------------------------------------------------------------
# a3  Nyquist mask at frame rate                                            core.py:73-77
# --------------------------------------------------------------------------------------
def remove_above_fmax(amplitudes: np.ndarray, pitch: np.ndarray, fmax: float,
                      level_start: int = 1) -> np.ndarray:
    """``amps * ((pitch*k < fmax) + 1e-7)`` for k = level_start.. ; the product ``pitch*k`` and
    the compare are float32 as in the reference (core.py:75-76)."""
    amplitudes = np.asarray(amplitudes, dtype=F32)
    pitch = np.asarray(pitch, dtype=F32)
    if pitch.ndim == 2:
        pitch = pitch[:, :, None]
    H = amplitudes.shape[-1]
    k = np.arange(level_start, H + level_start, dtype=F32)
    aa = ((pitch * k).astype(F32) < F32(fmax)).astype(F32) + F32(1e-7)
    return (amplitudes * aa).astype(F32)


# --------------------------------------------------------------------------------------
# a4  additive sinusoid bank                                     vocoder.py:580,585-594
# --------------------------------------------------------------------------------------
def sinusoid_bank(x: np.ndarray, f0_frames: np.ndarray, amp_ctrl: np.ndarray,
                  sr: float, hop: int) -> np.ndarray:
    """``sum_k sin(fl32(phase*k)) * upsample(A)[t,k]`` with ``A = mask(exp(c)/128)``.

    The argument ``phase*k`` is rounded to float32 exactly as the reference does
    (vocoder.py:592) -- at k=256 that rounding is worth 3e-5 rad; the sine itself and the
    harmonic sum are float64 here.
    """
    amp_ctrl = np.asarray(amp_ctrl, dtype=F32)
    B, Fr, H = amp_ctrl.shape
    A = (np.exp(amp_ctrl.astype(F64)) / 128.0).astype(F32)               # :580
    A = remove_above_fmax(A, f0_frames, F32(sr) / F32(2.0), 1)           # :585
    phase = (TWO_PI32 * np.asarray(x, dtype=F32)).astype(F32)            # :574
    out = np.zeros(phase.shape, dtype=F64)
    ks = np.arange(1, H + 1, dtype=F32)
    step = 16
    for h0 in range(0, H, step):
        kk = ks[h0:h0 + step]
        arg = (phase[:, :, None] * kk[None, None, :]).astype(F32)        # fl32(phase*k)
        amp = upsample(A[:, :, h0:h0 + step], hop)
        out += (np.sin(arg.astype(F64)) * amp.astype(F64)).sum(-1)
    return out


# --------------------------------------------------------------------------------------
# a5  combtooth exciter                                               vocoder.py:839-840
# --------------------------------------------------------------------------------------
def combtooth(x: np.ndarray, f0_frames: np.ndarray, sr: float, hop: int) -> np.ndarray:
    """``sinc(sr*x/(f0+1e-3))``: z is formed in float32 (int-tensor * float32, float32 add and
    divide), and ``torch.sinc`` evaluates ``sin(fl32(pi32*z))/fl32(pi32*z)`` in float32, 1 at 0.
    The float32 product pi*z (|z| up to ~340) is reproduced; sine/divide are float64."""
    f0f = np.asarray(f0_frames, dtype=F32)
    if f0f.ndim == 2:
        f0f = f0f[:, :, None]
    f0 = upsample(f0f, hop)[:, :, 0]
    num = (F32(sr) * np.asarray(x, dtype=F32)).astype(F32)
    den = (f0 + F32(1e-3)).astype(F32)
    z = (num / den).astype(F32)
    p = (PI32 * z).astype(F32).astype(F64)
    safe = np.where(p == 0.0, 1.0, p)
    return np.where(p == 0.0, 1.0, np.sin(safe) / safe)


# --------------------------------------------------------------------------------------
# a6/a7  frequency response -> impulse response            core.py:254-270, :185-251
# --------------------------------------------------------------------------------------
def allpass_response(gd_ctrl: np.ndarray):
    """vocoder.py:581,599 / :834,845: ``exp(1j*cumsum(pi*tanh(c), -1))`` -> (re, im)."""
    gd = np.pi * np.tanh(np.asarray(gd_ctrl, dtype=F32).astype(F64))
    th = np.cumsum(gd, axis=-1)
    return np.cos(th), np.sin(th)


def impulse_response(resp_re: np.ndarray, resp_im: np.ndarray | None, mode: int,
                     half_width: np.ndarray | None = None) -> np.ndarray:
    """``[B,F,n]`` one-sided response -> ``[B,F,N]`` causal-form taps, ``N = 2(n-1)``, float64.

    irfft (core.py:259) is written out as the explicit Hermitian synthesis sum: only the real
    part of bins 0 and n-1 contributes (the all-pass response's imaginary DC/Nyquist parts are
    dropped -- quirk Q2).  Then:
      MODE_ROLL    : roll by N/2                                           (core.py:269)
      MODE_HANN    : (ir * roll(hann_periodic_N, N/2)) rolled by N/2       (core.py:209-235, padding==0)
      MODE_DYNAMIC : roll by N/2, times (1+cos(pi*w))/2 with
                     w = arange(-N/2, N/2)/half_width and ONLY w>1 clamped to 0 (so the
                     window is 1 there, and w<-1 keeps oscillating -- quirk Q1)  (core.py:244-249)
    """
    re = np.asarray(resp_re, dtype=F64)
    n = re.shape[-1]
    N = 2 * (n - 1)
    m = np.arange(N)
    k = np.arange(1, n - 1)
    ang = 2.0 * np.pi * ((k[:, None] * m[None, :]) % N) / N              # [n-2, N]
    ir = re[..., 0:1] + re[..., n - 1:n] * np.where(m % 2 == 0, 1.0, -1.0)
    ir = ir + 2.0 * (re[..., 1:n - 1] @ np.cos(ang))
    if resp_im is not None:
        im = np.asarray(resp_im, dtype=F64)
        ir = ir - 2.0 * (im[..., 1:n - 1] @ np.sin(ang))
    ir = ir / N
    ir = np.roll(ir, N // 2, axis=-1)
    if mode == MODE_ROLL:
        return ir
    if mode == MODE_HANN:
        hann = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(N) / N)        # torch.hann_window: periodic
        return ir * hann
    if mode == MODE_DYNAMIC:
        hw = np.asarray(half_width, dtype=F32)
        if hw.ndim == 2:
            hw = hw[:, :, None]
        w = (np.arange(-(N // 2), (N + 1) // 2, dtype=F32)[None, None, :] / hw).astype(F32)
        w = np.where(w > F32(1.0), F32(0.0), w).astype(F32)
        arg = (PI32 * w).astype(F32)                                     # np.pi * fp32 tensor
        return ir * ((1.0 + np.cos(arg.astype(F64))) / 2.0)
    raise ValueError(f"unknown window mode {mode}")


def combsub_half_width(f0_frames: np.ndarray, sr: float) -> np.ndarray:
    """vocoder.py:851: ``1.5 * sr / (f0_frames + 1e-3)`` in float32."""
    f0f = np.asarray(f0_frames, dtype=F32)
    return ((F32(1.5) * F32(sr)) / (f0f + F32(1e-3)).astype(F32)).astype(F32)


# --------------------------------------------------------------------------------------
# a8  time-varying FIR                                                   core.py:120-182
# --------------------------------------------------------------------------------------
def ltv_fir_blockfft(audio: np.ndarray, ir: np.ndarray) -> np.ndarray:
    """The reference's own algorithm (core.py:155-182): pad by hop, 50%-overlap frames of
    2*hop, periodic Bartlett window, zero-padded FFT product with the per-frame taps (last
    tap frame repeated, :167), overlap-add, drop the first hop, crop ``[N/2 : N/2+T]``.
    float64 numpy; this is the leg ``bench.py`` times as the CPU baseline."""
    audio = np.asarray(audio, dtype=F64)
    ir = np.asarray(ir, dtype=F64)
    B, T = audio.shape
    Fr, N = ir.shape[1], ir.shape[2]
    hop = T // Fr
    fs = 2 * hop
    padded = np.pad(audio, ((0, 0), (hop, hop)))
    idx = (np.arange(Fr + 1) * hop)[:, None] + np.arange(fs)[None, :]
    frames = padded[:, idx]                                              # [B,F+1,2hop]
    i = np.arange(fs)
    bart = 1.0 - np.abs(2.0 * i / fs - 1.0)                              # torch.bartlett_window: periodic
    frames = frames * bart
    L = N + fs - 1
    nfft = 1 << int(np.ceil(np.log2(L)))                                 # linear conv: any size >= L is equivalent (Q3)
    taps = np.concatenate([ir, ir[:, -1:, :]], axis=1)
    seg = np.fft.irfft(np.fft.rfft(frames, nfft) * np.fft.rfft(taps, nfft), nfft)[..., :L]
    total = Fr * hop + L
    ola = np.zeros((B, total), dtype=F64)
    for j in range(Fr + 1):
        ola[:, j * hop:j * hop + L] += seg[:, j]
    ola = ola[:, hop:]
    return ola[:, N // 2:N // 2 + T]


def ltv_fir_direct(audio: np.ndarray, ir: np.ndarray) -> np.ndarray:
    """Definition form of the same operator:
    ``y[t] = sum_m h_s[m] x[s]``, ``s = t + N/2 - m``, ``h_s`` = taps linearly interpolated
    between frames ``floor(s/hop)`` and the next one (last held), indexed by the INPUT sample.
    O(T*N) per-sample loop -- for small cases only; cross-checks ``ltv_fir_blockfft``."""
    audio = np.asarray(audio, dtype=F64)
    ir = np.asarray(ir, dtype=F64)
    B, T = audio.shape
    Fr, N = ir.shape[1], ir.shape[2]
    hop = T // Fr
    y = np.zeros((B, T), dtype=F64)
    D = N // 2
    for s in range(T):
        k = s // hop
        lam = (s % hop) / hop
        k1 = min(k + 1, Fr - 1)
        h = (1.0 - lam) * ir[:, k, :] + lam * ir[:, k1, :]               # [B,N]
        t0 = s - D                                                        # t = s - D + m
        lo = max(0, -t0)
        hi = min(N, T - t0)
        if hi > lo:
            y[:, t0 + lo:t0 + hi] += h[:, lo:hi] * audio[:, s:s + 1]
    return y


def frequency_filter(audio, resp_re, resp_im=None, mode=MODE_HANN, half_width=None,
                     fir=ltv_fir_blockfft) -> np.ndarray:
    """core.py:273-280."""
    return fir(audio, impulse_response(resp_re, resp_im, mode, half_width))


# --------------------------------------------------------------------------------------
# a9-a11  the two DSP tails
# --------------------------------------------------------------------------------------
def sins_dsp(f0_frames, c_amp, c_gd, c_noise, noise, sr=44100, hop=512,
             initial_phase=None, infer=True, fir=ltv_fir_blockfft):
    """DSP tail of Sins.forward (vocoder.py:564-611) from raw controls and a supplied
    uniform(-1,1) ``noise [B,T]``.  Returns dict(signal, harmonic, noise, x, phase_frames)."""
    x, pf = wrapped_phase(f0_frames, sr, hop, initial_phase, infer)
    sinus = sinusoid_bank(x, f0_frames, c_amp, sr, hop)
    are, aim = allpass_response(c_gd)
    harmonic = frequency_filter(sinus, are, aim, MODE_ROLL, fir=fir)       # :597-600
    nz_mag = np.exp(np.asarray(c_noise, dtype=F32).astype(F64)) / 128.0    # :582
    nz = frequency_filter(np.asarray(noise, dtype=F64), nz_mag, None, MODE_HANN, fir=fir)  # :604-607
    return dict(signal=harmonic + nz, harmonic=harmonic, noise=nz, x=x, phase_frames=pf,
                exciter=sinus)


def combsub_dsp(f0_frames, c_gd, c_harm, c_noise, noise, sr=44100, hop=512,
                initial_phase=None, infer=True, fir=ltv_fir_blockfft):
    """DSP tail of CombSub.forward (vocoder.py:819-862)."""
    x, pf = wrapped_phase(f0_frames, sr, hop, initial_phase, infer)
    comb = combtooth(x, f0_frames, sr, hop)
    are, aim = allpass_response(c_gd)
    h1 = frequency_filter(comb, are, aim, MODE_ROLL, fir=fir)              # :843-846
    src = np.exp(np.asarray(c_harm, dtype=F32).astype(F64))                # :835
    hw = combsub_half_width(f0_frames, sr)
    harmonic = frequency_filter(h1, src, None, MODE_DYNAMIC, hw, fir=fir)  # :847-851
    nz_mag = np.exp(np.asarray(c_noise, dtype=F32).astype(F64)) / 128.0    # :836
    nz = frequency_filter(np.asarray(noise, dtype=F64), nz_mag, None, MODE_HANN, fir=fir)  # :855-858
    return dict(signal=harmonic + nz, harmonic=harmonic, noise=nz, x=x, phase_frames=pf,
                exciter=comb)


# --------------------------------------------------------------------------------------
# synthetic inputs shared by tests, smoke() and bench.py
# --------------------------------------------------------------------------------------
def synth_f0(B: int, Fr: int, sr: float = 44100.0, hop: int = 512, seed: int = 1234) -> np.ndarray:
    """Vibrato + random-walk drift f0 curves in [65, 800] Hz, ``[B,F,1]`` float32."""
    rng = np.random.default_rng(seed)
    base = rng.uniform(100.0, 400.0, size=(B, 1))
    vib = rng.uniform(0.0, 1.0, size=(B, 1))
    ph = rng.uniform(0.0, 2.0 * np.pi, size=(B, 1))
    drift = np.cumsum(rng.normal(0.0, 0.05, size=(B, Fr)), axis=1)
    t = np.arange(Fr)[None, :] * hop / sr
    semis = vib * np.sin(2.0 * np.pi * 5.5 * t + ph) + 0.5 * drift
    f0 = np.clip(base * 2.0 ** (semis / 12.0), 65.0, 800.0)
    return f0.astype(F32)[:, :, None]


def synth_controls(B: int, Fr: int, sizes, seed: int = 4321, scale: float = 1.0):
    """Raw control streams ~ N(0, scale) per split, list of ``[B,F,n_i]`` float32."""
    rng = np.random.default_rng(seed)
    return [(scale * rng.standard_normal((B, Fr, n))).astype(F32) for n in sizes]


def synth_noise(B: int, T: int, seed: int = 99) -> np.ndarray:
    """``2*U[0,1) - 1`` as the reference draws it (vocoder.py:603,854), float32."""
    rng = np.random.default_rng(seed)
    return (rng.random((B, T), dtype=F32) * F32(2.0) - F32(1.0)).astype(F32)


# --------------------------------------------------------------------------------------
# opt-in in-kernel noise draw (include/ddsp_hip.h, ddsp_hip_uniform_noise): Philox4x32-10
# (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11 -- the published algorithm; pinned
# below in tests/test_noise_rng.py against the known-answer vectors of the Random123 distribution)
# --------------------------------------------------------------------------------------
def philox4x32_10(counter, key):
    """``counter [..., 4]``, ``key [..., 2]`` uint32 -> ``[..., 4]`` uint32"""
    c = np.array(counter, dtype=np.uint64) & np.uint64(0xFFFFFFFF)
    k = np.array(key, dtype=np.uint64) & np.uint64(0xFFFFFFFF)
    c0, c1, c2, c3 = (c[..., i].copy() for i in range(4))
    k0, k1 = k[..., 0].copy(), k[..., 1].copy()
    M0, M1, W0, W1, MASK = (np.uint64(v) for v in (0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85, 0xFFFFFFFF))
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2                      # 32 x 32 -> 64 bit products
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & MASK, p1 >> np.uint64(32), p1 & MASK
        c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    return np.stack([c0, c1, c2, c3], -1).astype(np.uint32)


def uniform_noise(B: int, T: int, seed: int, offset: int) -> np.ndarray:
    """``u [B,T]`` float32 in [0,1): counter = (128 * (t // 512) + t % 128, b, offset lo, offset hi), key = (seed lo, hi),
    output word (t % 512) // 128, ``u = (x >> 8) * 2**-24``."""
    t = np.arange(T, dtype=np.uint64)
    blocks = (T + 511) // 512
    lane, word, blk = t % np.uint64(128), (t % np.uint64(512)) // np.uint64(128), t // np.uint64(512)
    ctr = np.zeros((B, blocks * 128, 4), dtype=np.uint64)
    idx = np.arange(blocks * 128, dtype=np.uint64)
    ctr[..., 0] = idx[None, :]
    ctr[..., 1] = np.arange(B, dtype=np.uint64)[:, None]
    ctr[..., 2] = np.uint64(offset & 0xFFFFFFFF)
    ctr[..., 3] = np.uint64((offset >> 32) & 0xFFFFFFFF)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint64)
    x = philox4x32_10(ctr, np.broadcast_to(key, ctr.shape[:-1] + (2,)))
    sel = x[:, (blk * np.uint64(128) + lane).astype(np.int64), word.astype(np.int64)]
    return ((sel >> np.uint32(8)).astype(np.float32) * F32(2.0 ** -24)).astype(F32)


def normal_noise(B: int, T: int, dim: int, seed: int, offset: int) -> np.ndarray:
    """``z [B,T,dim]`` float64, the opt-in in-kernel standard-normal draw of the NSF harmonic source (csrc/philox.h,
    ``ddsp_hip_normal_noise``; it stands where nsf_hifigan/models.py:168 draws ``torch.randn_like``): counter = (t, b, offset lo,
    4 * offset hi + j), key = (seed lo, hi ^ 'NORM'); words (x0, x1), (x2, x3) -> Box-Muller pairs with u1 = ((x >> 8) + 1) 2^-24,
    u2 = (x' >> 8) 2^-24; harmonic h takes normal h % 4 of call j = h // 4."""
    groups = (dim + 3) // 4
    ctr = np.zeros((B, T, groups, 4), dtype=np.uint64)
    ctr[..., 0] = np.arange(T, dtype=np.uint64)[None, :, None]
    ctr[..., 1] = np.arange(B, dtype=np.uint64)[:, None, None]
    ctr[..., 2] = np.uint64(offset & 0xFFFFFFFF)
    ctr[..., 3] = (np.uint64(4 * ((offset >> 32) & 0x3FFFFFFF)) + np.arange(groups, dtype=np.uint64))[None, None, :]
    if dim > 16:
        raise ValueError("dim <= 16: the counter's last word holds four calls per offset")
    key = np.array([seed & 0xFFFFFFFF, ((seed >> 32) & 0xFFFFFFFF) ^ 0x4E4F524D], dtype=np.uint64)   # domain tag "NORM" (philox.h)
    x = philox4x32_10(ctr, np.broadcast_to(key, ctr.shape[:-1] + (2,))).astype(np.uint64)
    z = np.empty((B, T, groups, 4), dtype=np.float64)
    for p in range(2):
        u1 = ((x[..., 2 * p] >> np.uint64(8)) + np.uint64(1)).astype(np.float64) * 2.0 ** -24
        u2 = (x[..., 2 * p + 1] >> np.uint64(8)).astype(np.float64) * 2.0 ** -24
        r = np.sqrt(-2.0 * np.log(u1))
        z[..., 2 * p] = r * np.cos(2.0 * np.pi * u2)
        z[..., 2 * p + 1] = r * np.sin(2.0 * np.pi * u2)
    return z.reshape(B, T, groups * 4)[..., :dim]


# --------------------------------------------------------------------------------------
# 8-f #1  CombSubFast / CombSubSuperFast: short-time spectral filtering     vocoder.py:613-786
# --------------------------------------------------------------------------------------
def fast_source_gen(f0_frames: np.ndarray, sr: float, hop: int):
    """``CombSubSuperFast.fast_source_gen`` (vocoder.py:639-651): closed-form per-frame phase.

    Everything is float32, op for op, in the reference's order (the buffers ``sampling_rate`` and
    ``block_size`` are integer 0-dim tensors, so every division is a float32 division by the exactly
    representable 44100 / 512); the only wide step is ``cumsum`` over frames, for which ATen's CPU
    kernel keeps a float64 running sum and rounds each output to float32 (the same behaviour
    ``wrapped_phase(infer=False)`` reproduces).  Returns ``(combtooth [B,T] f64, phase_frames [B,F]
    f32, rad_acc [B,F] f32)``; the float32 argument of the final ``sinc`` is exact, its sine and
    divide are evaluated in float64 here.
    """
    f0f = np.asarray(f0_frames, dtype=F32)
    if f0f.ndim == 2:
        f0f = f0f[:, :, None]
    B, Fr, _ = f0f.shape
    n = np.arange(hop, dtype=F32)[None, None, :]
    n1 = (n + F32(1.0)).astype(F32)
    s0 = (f0f / F32(sr)).astype(F32)                                               # :641
    ds0 = np.concatenate([s0[:, 1:] - s0[:, :-1], np.zeros((B, 1, 1), F32)], axis=1).astype(F32)   # :642
    a = (s0 * n1).astype(F32)
    h = (F32(0.5) * ds0).astype(F32)
    b = (((h * n).astype(F32) * n1).astype(F32) / F32(hop)).astype(F32)
    rad = (a + b).astype(F32)                                                      # :643
    s0n = (s0 + ((ds0 * n).astype(F32) / F32(hop)).astype(F32)).astype(F32)        # :644
    last = rad[..., -1:]
    rad2 = (np.fmod((last + F32(0.5)).astype(F32), F32(1.0)).astype(F32) - F32(0.5)).astype(F32)   # :645
    acc = np.cumsum(rad2.astype(F64), axis=1).astype(F32)                          # :646 cumsum (f64 running sum)
    rad_acc = np.fmod(acc, F32(1.0)).astype(F32)
    shifted = np.concatenate([np.zeros((B, 1, 1), F32), rad_acc[:, :-1]], axis=1)
    rad = (rad + shifted).astype(F32)                                              # :647
    rad = (rad - np.rint(rad)).astype(F32)                                         # :648
    z = (rad / (s0n + F32(1e-5)).astype(F32)).astype(F32)                          # :649
    p = (PI32 * z).astype(F32).astype(F64)
    safe = np.where(p == 0.0, 1.0, p)
    comb = np.where(p == 0.0, 1.0, np.sin(safe) / safe).reshape(B, Fr * hop)
    phase_frames = (TWO_PI32 * rad[:, :, 0]).astype(F32)                           # :650
    return comb, phase_frames, rad_acc[:, :, 0]


def spectral_filters(c_mag, c_phase, scale: float = 1.0):
    """vocoder.py:661-664 / :758-761: ``exp(mag + 1j*pi*phase) * scale`` with the last frame appended
    once more -> complex ``[B,F+1,n]``.  ``c_phase=None`` is the zero-phase noise filter of CombSubFast."""
    mag = np.exp(np.asarray(c_mag, dtype=F32).astype(F64)) * scale
    if c_phase is None:
        H = mag.astype(np.complex128)
    else:
        ang = (PI32 * np.asarray(c_phase, dtype=F32)).astype(F32).astype(F64)      # 1j*np.pi*float32 tensor
        H = mag * (np.cos(ang) + 1j * np.sin(ang))
    return np.concatenate([H, H[:, -1:]], axis=1)


def _frames(sig, win: int, hop: int, pad_mode: str):
    """centered framing shared by torch.stft(center=True) and CombSubFast's pad+unfold: ``[B,T] ->
    [B, T//hop + 1, win]`` with win/2 samples of padding on both sides."""
    sig = np.asarray(sig, dtype=F64)
    half = win // 2
    padded = np.pad(sig, ((0, 0), (half, half)), mode=pad_mode)
    nfr = (padded.shape[1] - win) // hop + 1
    idx = (np.arange(nfr) * hop)[:, None] + np.arange(win)[None, :]
    return padded[:, idx]


def combsubfast_dsp(f0_frames, c_hmag, c_hphase, c_nmag, noise, sr=44100, hop=512,
                    initial_phase=None, infer=True):
    """DSP tail of ``CombSubFast.forward`` (vocoder.py:743-786).  ``noise [B,T]`` is the already scaled
    uniform(-1,1) draw (:771).  Frames of ``2*hop`` with zero padding (:766), ``sqrt(hann)`` analysis and
    synthesis window (:726,:767,:780), circular per-frame filtering in the rfft domain (:777),
    overlap-add and crop (:783-784)."""
    x, pf = wrapped_phase(f0_frames, sr, hop, initial_phase, infer)
    comb = combtooth(x, f0_frames, sr, hop)                                         # :764 (same formula as CombSub)
    win = 2 * hop
    w = np.sqrt((0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(win) / win)).astype(F32).astype(F64))
    w = w.astype(F32).astype(F64)                                                   # the buffer is float32
    Hs = spectral_filters(c_hmag, c_hphase)                                         # :758-759
    Hn = spectral_filters(c_nmag, None, 1.0 / 128.0)                                # :760-761
    cf = np.fft.rfft(_frames(comb, win, hop, "constant") * w, win)                  # :766-768
    nf = np.fft.rfft(_frames(noise, win, hop, "constant") * w, win)                 # :772-774
    out = np.fft.irfft(cf * Hs + nf * Hn, win) * w                                  # :777-780
    B, nfr, _ = out.shape
    ola = np.zeros((B, (nfr + 1) * hop), dtype=F64)
    for j in range(nfr):
        ola[:, j * hop:j * hop + win] += out[:, j]
    return dict(signal=ola[:, hop:-hop], x=x, phase_frames=pf, exciter=comb)


def combsubsuperfast_dsp(f0_frames, c_hmag, c_hphase, c_nmag, c_nphase, noise, sr=44100, hop=512,
                         win=2048, window=None):
    """DSP tail of ``CombSubSuperFast.forward`` (vocoder.py:653-710).  ``noise [B,T]`` is the
    ``randn_like`` draw (:687).  torch.stft (center=True, reflect padding unless the signal is not longer
    than win/2, periodic Hann) of exciter and noise, complex filters per frame, torch.istft
    (overlap-add of windowed inverse frames divided by the summed squared window, centre crop)."""
    comb, pf, _ = fast_source_gen(f0_frames, sr, hop)
    if window is None:
        window = (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(win) / win)).astype(F32)   # torch.hann_window
    w = np.asarray(window, dtype=F32).astype(F64)
    T = comb.shape[1]
    mode = "reflect" if T > win // 2 else "constant"                                # :667-670
    Hs = spectral_filters(c_hmag, c_hphase)                                         # :661-662
    Hn = spectral_filters(c_nmag, c_nphase, 1.0 / 128.0)                            # :663-664
    cf = np.fft.rfft(_frames(comb, win, hop, mode) * w, win)                        # :671-679
    nf = np.fft.rfft(_frames(noise, win, hop, mode) * w, win)                       # :683-691
    spec = cf * Hs + nf * Hn                                                        # :694
    out = np.fft.irfft(spec, win) * w                                               # :697-702
    B, nfr, _ = out.shape
    total = win + hop * (nfr - 1)
    ola = np.zeros((B, total), dtype=F64)
    env = np.zeros(total, dtype=F64)
    for j in range(nfr):
        ola[:, j * hop:j * hop + win] += out[:, j]
        env[j * hop:j * hop + win] += w * w
    half = win // 2
    sl = slice(half, half + hop * (nfr - 1))
    return dict(signal=ola[:, sl] / env[sl], phase_frames=pf, exciter=comb)


def synth_gauss(B: int, T: int, seed: int = 77) -> np.ndarray:
    """standard normal noise as ``randn_like`` draws it (vocoder.py:687), float32."""
    rng = np.random.default_rng(seed)
    return rng.standard_normal((B, T)).astype(F32)


# --------------------------------------------------------------------------------------
# 8-f #2  waveform -> log-mel front-end of the cascade           nsf_hifigan/nvSTFT.py:73-117
# --------------------------------------------------------------------------------------
def mel_filterbank_slaney(sr: float, n_fft: int, n_mels: int, fmin: float, fmax: float) -> np.ndarray:
    """Restatement of ``librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax)`` with its defaults (``htk=False``,
    ``norm='slaney'``), which nvSTFT.py:90 calls: Slaney's auditory-toolbox scale (linear below 1 kHz, 200/3 Hz
    per mel; logarithmic above with step log(6.4)/27), triangular filters between consecutive band edges,
    each scaled by ``2 / (f[m+2] - f[m])``.  librosa is not installed in this image (pinned version unknown:
    requirements.txt lists it without one), so this function is checked against librosa's published algorithm
    only -- PARITY UNPINNED for the filterbank itself; ``get_mel`` is pinned with this basis injected into
    the reference (tests/golden/make_golden.py).  float32 result like librosa's."""
    f_sp = 200.0 / 3.0
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp

    def hz_to_mel(f):
        f = np.asarray(f, dtype=F64)
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, f / f_sp)

    def mel_to_hz(m):
        m = np.asarray(m, dtype=F64)
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)

    fftfreqs = np.linspace(0.0, sr / 2.0, n_fft // 2 + 1)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    weights = np.maximum(0.0, np.minimum(lower, upper))
    weights *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return weights.astype(F32)


def get_mel(y: np.ndarray, mel_basis: np.ndarray, n_fft: int = 2048, win_size: int = 2048, hop: int = 512,
            clip_val: float = 1e-5, window=None, keyshift: float = 0, speed: float = 1, center: bool = False) -> np.ndarray:
    """``STFT.get_mel(y, keyshift, speed, center)`` (nvSTFT.py:73-117): transform / window stretched to
    ``round(n 2^(keyshift/12))`` points and the hop to ``round(hop speed)`` (:82-85); manual padding of ``(win'-hop')//2``
    left and ``max((win'-hop'+1)//2, win'-len-pad_left)`` right (reflect when the right pad is shorter than the signal, else
    zeros, :97-103); ``torch.stft`` with the periodic Hann window of ``win'`` points centred in the transform (:106;
    ``center=True`` adds its own reflect padding of ``n'//2``); ``sqrt(re^2 + im^2 + 1e-9)`` (:108); when ``keyshift != 0``
    the spectrum is cut / zero-filled to ``n_fft//2 + 1`` bins and scaled by ``win / win'`` (:109-114); mel matmul (:115),
    ``log(clamp(., clip_val))`` (:116) -> ``[B, n_mels, frames]``.  float64 arithmetic.  ``window``: the float32 Hann
    window of ``win'`` points (default: built here)."""
    y = np.asarray(y, dtype=F64)
    B, T = y.shape
    factor = 2 ** (keyshift / 12)
    n_new = int(np.round(n_fft * factor))
    win_new = int(np.round(win_size * factor))
    hop_new = int(np.round(hop * speed))
    if window is None:
        window = (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(win_new) / win_new)).astype(F32)
    w = np.asarray(window, dtype=F32).astype(F64)
    if w.shape[0] != win_new or win_new > n_new:
        raise ValueError("window length")
    left = (n_new - win_new) // 2                                               # torch.stft centres a short window
    w = np.pad(w, (left, n_new - win_new - left))
    pad_left = (win_new - hop_new) // 2
    pad_right = max((win_new - hop_new + 1) // 2, win_new - T - pad_left)
    mode = "reflect" if pad_right < T else "constant"
    yp = np.pad(y, ((0, 0), (pad_left, pad_right)), mode=mode)
    if center:
        if n_new // 2 >= yp.shape[1]:
            raise ValueError("reflection longer than the signal")
        yp = np.pad(yp, ((0, 0), (n_new // 2, n_new // 2)), mode="reflect")
    if yp.shape[1] < n_new:
        raise ValueError("transform longer than the padded signal")
    nfr = (yp.shape[1] - n_new) // hop_new + 1
    idx = (np.arange(nfr) * hop_new)[:, None] + np.arange(n_new)[None, :]
    spec = np.fft.fft(yp[:, idx] * w, axis=-1)[..., :n_new // 2 + 1]            # [B, frames, bins]; any length, odd ones too
    mag = np.sqrt(spec.real ** 2 + spec.imag ** 2 + 1e-9)
    if keyshift != 0:
        size = n_fft // 2 + 1
        if mag.shape[-1] < size:
            mag = np.pad(mag, ((0, 0), (0, 0), (0, size - mag.shape[-1])))
        mag = mag[..., :size] * win_size / win_new
    mel = np.einsum("mk,bfk->bmf", np.asarray(mel_basis, dtype=F32).astype(F64), mag)
    return np.log(np.maximum(mel, clip_val))


# --------------------------------------------------------------------------------------
# 8-f #3 (first part)  gradient of the short-time spectral tail w.r.t. the controls
# --------------------------------------------------------------------------------------
def stft_filter_backward(grad_out, exciter, noise, c_hmag, c_hphase, c_nmag, c_nphase, window, hop=512,
                         pad_mode="reflect", normalize=True, noise_scale=1.0 / 128.0):
    """What autograd returns for the controls of ``CombSubSuperFast.forward`` (vocoder.py:661-708; with
    ``pad_mode='constant'``, ``normalize=False``, ``c_nphase=None`` and the sqrt-Hann window: ``CombSubFast``,
    :758-784) given ``grad_out = dL/dsignal [B,T]``, written out analytically in float64:

      gamma_j = w * (grad_out / env)[frame j]                  (adjoint of crop + envelope division + window + overlap-add)
      G_j[k]  = c_k / N * rfft(gamma_j)[k]                     (adjoint of irfft; c_0 = c_N/2 = 1 and imaginary part 0, else 2)
      dL/dmag   = Re(conj(G) E H),   dL/dphase = -pi Im(conj(G) E H)        per filter, E = rfft of the windowed
      exciter (noise) frame, H the filter; the last control row also receives the repeated last frame (:662,:664).

    Returns ``(d_hmag, d_hphase, d_nmag, d_nphase|None)``, each ``[B,F,n]``."""
    w = np.asarray(window, dtype=F32).astype(F64)
    win = w.shape[0]
    g = np.asarray(grad_out, dtype=F64)
    B, T = g.shape
    Fr = T // hop
    Hs = spectral_filters(c_hmag, c_hphase)
    Hn = spectral_filters(c_nmag, c_nphase, noise_scale)
    E = np.fft.rfft(_frames(exciter, win, hop, pad_mode) * w, win)
    U = np.fft.rfft(_frames(noise, win, hop, pad_mode) * w, win)
    nfr = E.shape[1]
    half = win // 2
    if normalize:
        env = np.zeros(win + hop * (nfr - 1))
        for j in range(nfr):
            env[j * hop:j * hop + win] += w * w
        g = g / env[half:half + T]
    gp = np.pad(g, ((0, 0), (half, half)))                      # zero gradient outside the cropped range
    idx = (np.arange(nfr) * hop)[:, None] + np.arange(win)[None, :]
    Gam = np.fft.rfft(gp[:, idx] * w, win)
    ck = np.full(half + 1, 2.0)
    ck[0] = ck[half] = 1.0
    G = Gam * ck / win
    G[..., 0] = G[..., 0].real
    G[..., half] = G[..., half].real

    def fold(x):                                                # frame F shares the last control row
        out = x[:, :Fr].copy()
        out[:, Fr - 1] += x[:, Fr]
        return out
    ps = np.conj(G) * E * Hs
    pn = np.conj(G) * U * Hn
    return (fold(ps.real), fold(-np.pi * ps.imag), fold(pn.real),
            None if c_nphase is None else fold(-np.pi * pn.imag))


# --------------------------------------------------------------------------------------
# 8-f #4  harmonic source of NSF-HiFiGAN                       nsf_hifigan/models.py:101-204
# --------------------------------------------------------------------------------------
def sine_source(f0, upp: int, sr: float, weight, bias, rand_ini, noise, sine_amp: float = 0.1,
                noise_std: float = 0.003, voiced_threshold: float = 0.0):
    """``SourceModuleHnNSF.forward(f0, upp)`` (models.py:198-204) = ``tanh(Linear(SineGen(f0, upp)))`` with the two
    random draws of ``SineGen`` supplied: ``rand_ini [dim]`` (random initial phase per harmonic, first entry 0,
    models.py:150-152) and ``noise [B, L*upp, dim]`` (``randn_like``, models.py:168).

    ``_f02sine`` (models.py:140-154) in the reference's float32 operation order: ``rad = f0/sr * (i+1)`` inside a
    frame, frame totals wrapped by ``fmod(.+0.5, 1)-0.5``, ``cumsum`` over frames (float64 running sum, float32
    outputs) ``.fmod(1)`` shifted by one frame, times the harmonic number, plus ``rand_ini``; the sine of the float32
    product ``2*pi*rad`` is taken in float64 here.  ``forward`` (models.py:156-171): ``* sine_amp``, voiced mask
    ``f0 > threshold`` held over the frame (nearest upsampling), noise amplitude ``uv*noise_std + (1-uv)*sine_amp/3``.
    Returns ``[B, L*upp]`` float64."""
    f0 = np.asarray(f0, dtype=F32)
    B, L = f0.shape
    dim = np.asarray(weight).reshape(-1).shape[0]
    i1 = np.arange(1, upp + 1, dtype=F32)[None, None, :]
    s = (f0 / F32(sr)).astype(F32)[:, :, None]
    rad = (s * i1).astype(F32)                                                  # :141
    rad2 = (np.fmod((rad[..., -1:] + F32(0.5)).astype(F32), F32(1.0)).astype(F32) - F32(0.5)).astype(F32)   # :142
    acc = np.fmod(np.cumsum(rad2.astype(F64), axis=1).astype(F32), F32(1.0)).astype(F32)                    # :143
    shifted = np.concatenate([np.zeros((B, 1, 1), F32), acc[:, :-1]], axis=1)                               # :144
    rad = (rad + shifted).astype(F32).reshape(B, L * upp, 1)
    h = np.arange(1, dim + 1, dtype=F32)[None, None, :]
    rad = ((rad * h).astype(F32) + np.asarray(rand_ini, dtype=F32).reshape(1, 1, dim)).astype(F32)          # :146-149
    arg = (TWO_PI32 * rad).astype(F32).astype(F64)
    sines = np.sin(arg) * F64(F32(sine_amp))                                    # :150, :162
    uv = np.repeat((f0 > F32(voiced_threshold)).astype(F64), upp, axis=1)[:, :, None]                       # :163-164
    noise_amp = uv * F64(F32(noise_std)) + (1.0 - uv) * F64(F32(sine_amp) / F32(3.0))                       # :165
    waves = sines * uv + noise_amp * np.asarray(noise, dtype=F32).astype(F64)                               # :166-167
    merged = waves @ np.asarray(weight, dtype=F32).astype(F64).reshape(dim) + F64(np.asarray(bias, dtype=F32).reshape(()))
    return np.tanh(merged)


# --------------------------------------------------------------------------------------
# 8-f #3 (second part)  adjoints of the time-varying FIR                  core.py:120-182
# --------------------------------------------------------------------------------------
def ltv_fir_backward(grad_out, audio, ir):
    """What autograd returns for ``fft_convolve(audio, impulse_response)`` (core.py:120-182) given
    ``grad_out = dL/dout [B,T]``: ``(d_audio [B,T], d_ir [B,F,N])``, written out from the hop-block form of the
    operator (``ltv_fir_direct``): block ``b`` contributes ``conv(x_b (1-lambda), ir_b) + conv(x_b lambda,
    ir_min(b+1,F-1))`` at output offset ``b hop - N/2``, so with ``seg_b[n] = grad_out[b hop - N/2 + n]``

      d_ir[b]            += sum_s x_b[s] (1 - lambda_s) seg_b[s + m]
      d_ir[min(b+1,F-1)] += sum_s x_b[s] lambda_s       seg_b[s + m]
      d_audio[b hop + s]  = (1 - lambda_s) sum_m seg_b[s + m] ir_b[m] + lambda_s sum_m seg_b[s + m] ir_b+1[m].

    float64; correlations through numpy FFTs of a size that cannot alias."""
    g = np.asarray(grad_out, dtype=F64)
    x = np.asarray(audio, dtype=F64)
    ir = np.asarray(ir, dtype=F64)
    B, T = x.shape
    Fr, N = ir.shape[1], ir.shape[2]
    hop = T // Fr
    D = N // 2
    L = hop + N - 1
    nfft = 1 << int(np.ceil(np.log2(L + max(hop, N))))
    lam = np.arange(hop) / hop
    d_x = np.zeros_like(x)
    d_ir = np.zeros_like(ir)
    gp = np.pad(g, ((0, 0), (D, L)))                            # gp[:, t + D] = g[:, t]
    for b in range(Fr):
        b1 = min(b + 1, Fr - 1)
        seg = gp[:, b * hop:b * hop + L]                        # seg[n] = g[b hop - D + n]
        S = np.fft.rfft(seg, nfft)
        xb = x[:, b * hop:(b + 1) * hop]
        for rows, wgt in ((b, 1.0 - lam), (b1, lam)):
            X = np.fft.rfft(xb * wgt, nfft)
            d_ir[:, rows] += np.fft.irfft(np.conj(X) * S, nfft)[:, :N]          # sum_s x[s] seg[s + m]
            H = np.fft.rfft(ir[:, rows], nfft)
            d_x[:, b * hop:(b + 1) * hop] += wgt * np.fft.irfft(np.conj(H) * S, nfft)[:, :hop]
    return d_x, d_ir


def impulse_response_backward(d_taps, mode: int, half_width=None):
    """Adjoint of ``impulse_response`` w.r.t. the one-sided response: ``d_taps [B,F,N] -> (d_re, d_im) [B,F,n]``
    (what autograd returns for the real and imaginary part of ``magnitudes`` in core.py:254-270; the window of
    core.py:185-251 depends on f0 only and is a constant factor here).  float64."""
    dt = np.asarray(d_taps, dtype=F64)
    N = dt.shape[-1]
    n = N // 2 + 1
    if mode == MODE_HANN:
        dt = dt * (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(N) / N))
    elif mode == MODE_DYNAMIC:
        hw = np.asarray(half_width, dtype=F32)
        if hw.ndim == 2:
            hw = hw[:, :, None]
        w = (np.arange(-(N // 2), (N + 1) // 2, dtype=F32)[None, None, :] / hw).astype(F32)
        w = np.where(w > F32(1.0), F32(0.0), w).astype(F32)
        dt = dt * ((1.0 + np.cos((PI32 * w).astype(F32).astype(F64))) / 2.0)
    d_ir = np.roll(dt, -(N // 2), axis=-1)                      # adjoint of roll(ir, N/2)
    m = np.arange(N)
    k = np.arange(n)
    ang = 2.0 * np.pi * ((k[:, None] * m[None, :]) % N) / N      # [n, N]
    ck = np.full(n, 2.0)
    ck[0] = ck[n - 1] = 1.0
    d_re = (d_ir @ np.cos(ang).T) * ck / N
    d_im = -(d_ir @ np.sin(ang).T) * 2.0 / N
    d_im[..., 0] = 0.0
    d_im[..., n - 1] = 0.0
    return d_re, d_im


def combsub_dsp_backward(grad_out, f0_frames, c_gd, c_harm, c_noise, noise, sr=44100, hop=512, initial_phase=None, infer=True):
    """What autograd returns for the three raw controls of ``combsub_dsp`` (the DSP tail of CombSub.forward,
    vocoder.py:834-862) given ``grad_out = dL/dsignal [B,T]``: the adjoints above chained backwards through the cascade
    all-pass filter -> dynamic-window filter, and through the noise branch.  float64; returns
    ``dict(group_delay, harmonic_magnitude, noise_magnitude)``, each ``[B,F,n]``.  Pinned to the reference's own autograd by
    tests/test_oracle_golden.py (fixture combsub_grad.npz)."""
    g = np.asarray(grad_out, dtype=F64)
    x, _ = wrapped_phase(f0_frames, sr, hop, initial_phase, infer)
    comb = combtooth(x, f0_frames, sr, hop)
    are, aim = allpass_response(c_gd)
    taps_ap = impulse_response(are, aim, MODE_ROLL)
    h1 = ltv_fir_blockfft(comb, taps_ap)
    src = np.exp(np.asarray(c_harm, dtype=F32).astype(F64))
    hw = combsub_half_width(f0_frames, sr)
    taps_h = impulse_response(src, None, MODE_DYNAMIC, hw)
    nz_mag = np.exp(np.asarray(c_noise, dtype=F32).astype(F64)) / 128.0
    taps_nz = impulse_response(nz_mag, None, MODE_HANN)
    d_h1, d_taps_h = ltv_fir_backward(g, h1, taps_h)
    d_src, _ = impulse_response_backward(d_taps_h, MODE_DYNAMIC, hw)
    _, d_taps_ap = ltv_fir_backward(d_h1, comb, taps_ap)
    d_re, d_im = impulse_response_backward(d_taps_ap, MODE_ROLL)
    _, d_taps_nz = ltv_fir_backward(g, np.asarray(noise, dtype=F64), taps_nz)
    d_nz, _ = impulse_response_backward(d_taps_nz, MODE_HANN)
    return dict(group_delay=allpass_backward(c_gd, d_re, d_im), harmonic_magnitude=d_src * src,       # d exp(c) = exp(c)
                noise_magnitude=d_nz * nz_mag)


def allpass_backward(gd_ctrl, d_re, d_im):
    """Adjoint of ``allpass_response``: gradients of (cos theta, sin theta), ``theta = cumsum(pi tanh(c))``, back to
    the raw group-delay control ``c`` (vocoder.py:581,599 / :834,845)."""
    c = np.asarray(gd_ctrl, dtype=F32).astype(F64)
    th = np.cumsum(np.pi * np.tanh(c), axis=-1)
    d_th = -np.sin(th) * np.asarray(d_re, dtype=F64) + np.cos(th) * np.asarray(d_im, dtype=F64)
    d_gd = np.flip(np.cumsum(np.flip(d_th, axis=-1), axis=-1), axis=-1)          # suffix sums
    return d_gd * np.pi * (1.0 - np.tanh(c) ** 2)


def sinusoid_bank_backward(grad_out, x, f0_frames, amp_ctrl, sr: float, hop: int):
    """Adjoint of ``sinusoid_bank`` w.r.t. the raw amplitude control (what autograd returns through
    vocoder.py:580,585-594): with ``A = mask * exp(c)/128`` and ``w0/w1`` the interpolation weights of core.py:66-70,
    ``dA[f,k] = sum_{t in frame f} g[t] w0[t] sin(k phase[t]) + sum_{t in frame f-1} g[t] w1[t] sin(k phase[t])``
    (the held last frame also takes its own w1 part) and ``dc = dA * A``.  float64; the sine argument is rounded to
    float32 exactly as in the forward oracle."""
    amp_ctrl = np.asarray(amp_ctrl, dtype=F32)
    B, Fr, H = amp_ctrl.shape
    A = (np.exp(amp_ctrl.astype(F64)) / 128.0).astype(F32)
    A = remove_above_fmax(A, f0_frames, F32(sr) / F32(2.0), 1).astype(F64)
    phase = (TWO_PI32 * np.asarray(x, dtype=F32)).astype(F32)
    g = np.asarray(grad_out, dtype=F64)
    w0, w1 = _lerp_weights(hop)
    G0 = (g.reshape(B, Fr, hop) * w0.astype(F64))
    G1 = (g.reshape(B, Fr, hop) * w1.astype(F64))
    dA = np.zeros((B, Fr, H), dtype=F64)
    ks = np.arange(1, H + 1, dtype=F32)
    for h0 in range(0, H, 16):
        kk = ks[h0:h0 + 16]
        S = np.sin((phase[:, :, None] * kk[None, None, :]).astype(F32).astype(F64)).reshape(B, Fr, hop, -1)
        R0 = np.einsum("bft,bftk->bfk", G0, S)
        R1 = np.einsum("bft,bftk->bfk", G1, S)
        dA[:, :, h0:h0 + 16] += R0
        dA[:, 1:, h0:h0 + 16] += R1[:, :-1]
        dA[:, -1, h0:h0 + 16] += R1[:, -1]
    return dA * A


# --------------------------------------------------------------------------------------
# spectral loss of the training loop: ddsp/loss.py:9-54
# --------------------------------------------------------------------------------------
def _sss_spectra(x, n_fft: int, hop: int):
    """torchaudio.transforms.Spectrogram(n_fft, hop_length=hop, power=None, center=False) as loss.py:20 configures it
    (torchaudio 0.x/2.x functional.spectrogram: periodic Hann window of n_fft, frames every hop from sample 0, no
    padding, one-sided rfft).  torchaudio is absent from this container: its published semantics are restated here
    (and in the stand-in tests/golden/make_golden.py installs to run the reference's SSSLoss) -- parity unpinned
    for that one library call."""
    x = np.asarray(x, F64)
    B, T = x.shape
    n_frames = 1 + (T - n_fft) // hop
    idx = np.arange(n_frames)[:, None] * hop + np.arange(n_fft)[None, :]
    window = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n_fft) / n_fft)
    X = np.fft.rfft(x[:, idx] * window, axis=-1)                   # [B, frames, bins]
    return X, window, idx


def sss_loss(x_true, x_pred, n_fft: int, alpha: float = 1.0, overlap: float = 0.0, eps: float = 1e-7) -> float:
    """SSSLoss.forward (loss.py:22-31)."""
    hop = int(n_fft * (1 - overlap))                                # loss.py:19
    Xt, window, _ = _sss_spectra(x_true, n_fft, hop)
    Xp, _, _ = _sss_spectra(x_pred, n_fft, hop)
    wn = np.sqrt((window ** 2).sum())                              # normalized=True -> "window" normalisation
    St, Sp = np.abs(Xt) / wn + eps, np.abs(Xp) / wn + eps           # loss.py:23-24
    conv = np.mean(np.sqrt(((St - Sp) ** 2).sum((1, 2))) / np.sqrt(((St + Sp) ** 2).sum((1, 2))))   # :26
    log_term = np.mean(np.abs(np.log(St) - np.log(Sp)))             # :28
    return float(conv + alpha * log_term)                           # :30


def sss_loss_backward(x_true, x_pred, n_fft: int, alpha: float = 1.0, overlap: float = 0.0, eps: float = 1e-7):
    """d SSSLoss / d x_pred, analytically (what autograd returns for loss.py:22-31)."""
    hop = int(n_fft * (1 - overlap))
    Xt, window, idx = _sss_spectra(x_true, n_fft, hop)
    Xp, _, _ = _sss_spectra(x_pred, n_fft, hop)
    wn = np.sqrt((window ** 2).sum())
    At, Ap = np.abs(Xt), np.abs(Xp)
    St, Sp = At / wn + eps, Ap / wn + eps
    B = St.shape[0]
    d, s = St - Sp, St + Sp
    nd = np.sqrt((d ** 2).sum((1, 2)))[:, None, None]
    ns = np.sqrt((s ** 2).sum((1, 2)))[:, None, None]
    g = (-d / (nd * ns) - nd * s / ns ** 3) / B - alpha * np.sign(np.log(St) - np.log(Sp)) / (Sp * St.size)
    G = np.where(Ap > 0, g / wn * Xp / np.where(Ap > 0, Ap, 1.0), 0.0)          # dRe + i dIm
    full = np.zeros(G.shape[:-1] + (n_fft,), np.complex128)
    full[..., :G.shape[-1]] = G
    dframes = (np.fft.ifft(full, axis=-1) * n_fft).real * window   # x_n enters X_k through exp(-i 2 pi k n / n_fft)
    dx = np.zeros(np.asarray(x_pred).shape, F64)
    for b in range(B):
        np.add.at(dx[b], idx, dframes[b])
    return dx
