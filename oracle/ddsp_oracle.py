"""CPU oracle for the DDSP harmonic-plus-noise hot path (TEST INFRASTRUCTURE ONLY).

This file is the *checker*, never the product: only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it.  Nothing under ``ddsp-svc_amd/``
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
