"""Parity of the CombSubFast / CombSubSuperFast HIP path against the CPU oracle and the
reference-generated fixtures, on both backends of tests/backends.py (``emu`` here, ``gpu`` on the MI355X).

Tolerances (float32 path vs float64 oracle):
  fast_source_gen   rad_acc / phase_frames bit exact (the float32 recipe is reproduced op for op);
                    combtooth <= 3e-7 absolute (float32 sine and divide inside sinc)
  spectral filter   RMS <= 2e-6 relative
  full tails        RMS <= 1e-5 relative to the signal against the reference's own outputs
"""
import os

import numpy as np
import pytest
import torch

from oracle import ddsp_oracle as O
from tests.backends import BACKENDS, dev  # noqa: F401

SR, HOP = 44100, 512


def rms(a):
    return float(np.sqrt(np.mean(np.square(np.asarray(a, dtype=np.float64)))))


def T_(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def N_(t):
    return t.detach().cpu().numpy()


def hann(win):
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(win) / win)).astype(np.float32)


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
def test_fast_source_golden(dev, golden_dir):
    from ddsp_svc_amd import synth
    g = np.load(os.path.join(golden_dir, "fastsrc.npz"))
    st = synth.fast_source(T_(g["f0_frames"], dev), SR, HOP, want_combtooth=True)
    _, pf, ra = O.fast_source_gen(g["f0_frames"], SR, HOP)
    assert np.array_equal(N_(st.rad_acc), ra)
    assert np.array_equal(N_(st.phase_frames)[..., 0], g["phase_frames"])
    assert np.abs(N_(st.combtooth) - g["combtooth"]).max() <= 3e-7


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
@pytest.mark.parametrize("B,F,hop", [(1, 1, HOP), (2, 300, HOP), (1, 257, HOP), (2, 37, 440), (1, 21, 441), (3, 9, 64)])
def test_fast_source_shapes(dev, B, F, hop):
    """hop 512 / 64: the four-samples-per-thread exciter kernel with a shift for the frame index and exact reciprocal
    multiplications for the recipe's divisions by the hop; 440: the same kernel with true divisions; 441: the general form"""
    from ddsp_svc_amd import synth
    f0 = O.synth_f0(B, F, SR, hop, seed=F)
    st = synth.fast_source(T_(f0, dev), SR, hop, want_combtooth=True)
    comb, pf, ra = O.fast_source_gen(f0, SR, hop)
    assert np.array_equal(N_(st.rad_acc), ra)
    assert np.array_equal(N_(st.phase_frames)[..., 0], pf)
    assert np.abs(N_(st.combtooth) - comb).max() <= 3e-7


def _filter_case(B, F, win, seed, noise_phase=True):
    rng = np.random.default_rng(seed)
    n = win // 2 + 1
    T = F * HOP
    exc = rng.standard_normal((B, T)).astype(np.float32)
    nz = rng.standard_normal((B, T)).astype(np.float32)
    hm, hp, nm, nph = [(s * rng.standard_normal((B, F, n))).astype(np.float32) for s in (1.0, 1.5, 1.0, 1.5)]
    return exc, nz, hm, hp, nm, (nph if noise_phase else None)


def _oracle_filter(exc, nz, hm, hp, nm, nph, win, w, reflect, normalize):
    Hs = O.spectral_filters(hm, hp)
    Hn = O.spectral_filters(nm, nph, 1.0 / 128.0)
    mode = "reflect" if reflect else "constant"
    w64 = w.astype(np.float64)
    spec = np.fft.rfft(O._frames(exc, win, HOP, mode) * w64, win) * Hs + \
        np.fft.rfft(O._frames(nz, win, HOP, mode) * w64, win) * Hn
    fr = np.fft.irfft(spec, win) * w64
    B, nfr, _ = fr.shape
    total = win + HOP * (nfr - 1)
    ola = np.zeros((B, total))
    env = np.zeros(total)
    for j in range(nfr):
        ola[:, j * HOP:j * HOP + win] += fr[:, j]
        env[j * HOP:j * HOP + win] += w64 * w64
    sl = slice(win // 2, win // 2 + HOP * (nfr - 1))
    return ola[:, sl] / env[sl] if normalize else ola[:, sl]


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
@pytest.mark.parametrize("win,B,F,reflect,normalize,nphase", [
    (2048, 1, 3, True, True, True),          # shortest reflect case, every frame touches a boundary
    (2048, 2, 11, True, True, True),         # odd frame count: the last pair has one live frame
    (2048, 1, 12, False, True, True),        # zero padding with the istft envelope
    (2048, 1, 2, False, True, True),         # T <= win/2 (the reference then pads with zeros)
    (2048, 1, 1, False, True, True),
    (1024, 2, 10, False, False, False),      # CombSubFast geometry
    (1024, 1, 7, False, False, False),
    (1024, 1, 1, False, False, False),
    (1024, 1, 9, True, True, True),          # the generic entry point takes every combination
])
def test_stft_filter(dev, win, B, F, reflect, normalize, nphase):
    from ddsp_svc_amd import synth
    exc, nz, hm, hp, nm, nph = _filter_case(B, F, win, 100 * win + F, nphase)
    w = hann(win) if normalize else np.sqrt(hann(win)).astype(np.float32)
    out = N_(synth.stft_filter(T_(exc, dev), T_(nz, dev), T_(hm, dev), T_(hp, dev), T_(nm, dev),
                               None if nph is None else T_(nph, dev), T_(w, dev), HOP,
                               pad_reflect=reflect, normalize=normalize))
    ref = _oracle_filter(exc, nz, hm, hp, nm, nph, win, w, reflect, normalize)
    assert out.shape == ref.shape
    assert rms(out - ref) <= 2e-6 * rms(ref), (rms(out - ref), rms(ref))


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
def test_stft_filter_run_split(dev, knobs):
    """the split of an utterance into runs of frame pairs (with warm-up) must not change a single bit"""
    from ddsp_svc_amd import synth
    win, B, F = 2048, 1, 37
    exc, nz, hm, hp, nm, nph = _filter_case(B, F, win, 5)
    args = [T_(a, dev) for a in (exc, nz, hm, hp, nm, nph, hann(win))]
    outs = []
    for run in ("1", "3", "8", "1000"):
        knobs("STFT_RUN", int(run))
        outs.append(N_(synth.stft_filter(*args, HOP)))
    for o in outs[1:]:
        assert np.array_equal(o, outs[0])


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
def test_stft_filter_strided_controls(dev):
    """controls arrive as torch.split views of Unit2Control's output (row stride = sum of the splits)"""
    from ddsp_svc_amd import synth
    win, B, F = 2048, 2, 6
    exc, nz, hm, hp, nm, nph = _filter_case(B, F, win, 9)
    packed = T_(np.concatenate([hm, hp, nm, nph], axis=-1), dev)
    views = torch.split(packed, [1025] * 4, dim=-1)
    a = synth.stft_filter(T_(exc, dev), T_(nz, dev), *views, T_(hann(win), dev), HOP)
    b = synth.stft_filter(T_(exc, dev), T_(nz, dev), T_(hm, dev), T_(hp, dev), T_(nm, dev), T_(nph, dev),
                          T_(hann(win), dev), HOP)
    assert torch.equal(a, b)


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
@pytest.mark.parametrize("name,infer", [("csfast_a.npz", True), ("csfast_train.npz", False)])
def test_combsubfast_golden(dev, golden_dir, name, infer):
    from ddsp_svc_amd import synth
    g = np.load(os.path.join(golden_dir, name))
    f0 = T_(g["f0_frames"], dev)
    st = synth.phase(f0, SR, HOP, None, infer)
    assert np.abs(N_(st.phase_frames)[..., 0] - g["phase_frames"]).max() <= 4e-7
    sig = synth.combsubfast_synth(f0, st, T_(g["ctrl_harmonic_magnitude"], dev), T_(g["ctrl_harmonic_phase"], dev),
                                  T_(g["ctrl_noise_magnitude"], dev), T_(g["noise"], dev), T_(g["window"], dev),
                                  SR, HOP)
    err = rms(N_(sig) - g["signal"])
    assert err <= 1e-5 * rms(g["signal"]), (err, rms(g["signal"]))
    # uniform draw handed over raw: 2u - 1 applied on load
    u01 = ((g["noise"].astype(np.float64) + 1.0) / 2.0).astype(np.float32)
    sig2 = synth.combsubfast_synth(f0, st, T_(g["ctrl_harmonic_magnitude"], dev), T_(g["ctrl_harmonic_phase"], dev),
                                   T_(g["ctrl_noise_magnitude"], dev), T_(u01, dev), T_(g["window"], dev),
                                   SR, HOP, noise_is_u01=True)
    assert rms(N_(sig2) - g["signal"]) <= 1e-5 * rms(g["signal"])


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
@pytest.mark.parametrize("name", ["cssuper_a.npz", "cssuper_short.npz", "cssuper_f3.npz"])
def test_combsubsuperfast_golden(dev, golden_dir, name):
    from ddsp_svc_amd import synth
    g = np.load(os.path.join(golden_dir, name))
    f0 = T_(g["f0_frames"], dev)
    st = synth.fast_source(f0, SR, HOP)
    assert np.array_equal(N_(st.phase_frames)[..., 0], g["phase_frames"])
    sig = synth.combsubsuperfast_synth(f0, st, T_(g["ctrl_harmonic_magnitude"], dev),
                                       T_(g["ctrl_harmonic_phase"], dev), T_(g["ctrl_noise_magnitude"], dev),
                                       T_(g["ctrl_noise_phase"], dev), T_(g["noise"], dev), T_(g["window"], dev),
                                       SR, HOP)
    err = rms(N_(sig) - g["signal"])
    assert err <= 1e-5 * rms(g["signal"]), (err, rms(g["signal"]))


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
def test_combsubsuperfast_vs_oracle(dev):
    from ddsp_svc_amd import synth
    B, F = 3, 21
    f0 = O.synth_f0(B, F, SR, HOP, seed=2)
    f0[1] = np.clip(f0[1] * 2.3, 65, 800)
    hm, hp, nm, nph = O.synth_controls(B, F, [1025] * 4, seed=3)
    nz = O.synth_gauss(B, F * HOP, seed=4)
    w = hann(2048)
    st = synth.fast_source(T_(f0, dev), SR, HOP)
    sig = synth.combsubsuperfast_synth(T_(f0, dev), st, T_(hm, dev), T_(hp, dev), T_(nm, dev), T_(nph, dev),
                                       T_(nz, dev), T_(w, dev), SR, HOP)
    ref = O.combsubsuperfast_dsp(f0, hm, hp, nm, nph, nz, SR, HOP, 2048, w)
    assert rms(N_(sig) - ref["signal"]) <= 2e-6 * rms(ref["signal"])


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
def test_fast_errors(dev):
    from ddsp_svc_amd import synth
    exc, nz, hm, hp, nm, nph = _filter_case(1, 4, 2048, 1)
    args = [T_(a, dev) for a in (exc, nz, hm, hp, nm, nph)]
    with pytest.raises(RuntimeError):          # unsupported window length
        synth.stft_filter(*args[:2], *[a[..., :257].contiguous() for a in args[2:]], T_(hann(512), dev), HOP)
    with pytest.raises(RuntimeError):          # unsupported hop
        synth.stft_filter(*args, T_(hann(2048), dev), 256)
    with pytest.raises(RuntimeError):          # reflect padding needs T > win/2
        e2, n2, a, b_, c, d = _filter_case(1, 2, 2048, 1)
        synth.stft_filter(*[T_(x, dev) for x in (e2, n2, a, b_, c, d)], T_(hann(2048), dev), HOP, pad_reflect=True)
