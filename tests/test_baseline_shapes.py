"""Reference-pinned parity at the BASELINE.json shapes ("missing" #2 and #7):

  cfg 1 exactly     B = 1, 5 s, F = 431: CombSub(256, 128, 256) and Sins(128, 256, 256) (= configs/sins.yaml:20-22),
                    infer True / False -- outputs of the UNMODIFIED reference modules with drawn controls
  10 s phase scan   F = 862: the 441 344-term cumulative sum of vocoder.py:564-575, float64 and float32 variants
  n_mag 128/257/512 ``frequency_impulse_response`` / ``frequency_filter`` in the three window modes, and module tails
                    with mixed bin counts (N = 512 takes the hop-block FFT filter's largest size, N = 1022 the direct form)

The fixtures come from tests/golden/make_golden.py --baseline-shapes (which imports /root/reference).  Large inputs are
regenerated here from seeds and identified by a few stored numbers; large outputs are stored decimated (every 97th
sample, per-frame RMS, two contiguous stretches).  Each case runs three ways: the numpy oracle (CPU), the HIP sources
under the CPU emulator, and the product path on the MI355X (``-m gpu``).
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


def wrapdiff(a, b, period=1.0):
    d = np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)
    return d - period * np.rint(d / period)


def input_checks(a):
    a = np.asarray(a, np.float64).reshape(-1)
    return np.array([a.sum(), np.square(a).sum(), a[0], a[1], a[a.size // 2], a[-1]], np.float64)


def cfg1_inputs(g, kind):
    """regenerate the seeded inputs of a cfg1_* fixture and make sure they are the arrays the reference saw"""
    sizes = [int(s) for s in g["sizes"]]
    F = g["f0_frames"].shape[1]
    ctrls = O.synth_controls(1, F, sizes, seed=int(g["seeds"][0]), scale=float(g["ctrl_scale"]))
    noise = O.synth_noise(1, F * HOP, seed=int(g["seeds"][1]))
    names = ("amplitudes", "group_delay", "noise_magnitude") if kind == "sins" else \
        ("group_delay", "harmonic_magnitude", "noise_magnitude")
    for k, c in zip(names, ctrls):
        assert np.array_equal(input_checks(c), g["check_" + k]), "regenerated control stream differs from the fixture's"
    assert np.array_equal(input_checks(noise), g["noise_check"]), "regenerated noise differs from the fixture's"
    return g["f0_frames"], ctrls, noise


def check_summary(got, g, key, rel, abs_tol=1e-4):
    """compare a [B,T] waveform with the decimated view of the reference's output stored under ``key``"""
    got = np.asarray(got)
    dec = int(g["decim"])
    ref = g[key + "_dec"]
    err = rms(got[:, ::dec] - ref)
    assert err <= rel * rms(ref) and err <= abs_tol, (key, "decimated", err, rms(ref))
    B, T = got.shape
    frms = np.sqrt(np.mean(np.square(got.astype(np.float64)).reshape(B, T // HOP, HOP), -1))
    assert np.abs(frms - g[key + "_frame_rms"]).max() <= max(10 * rel, 1e-5) * g[key + "_frame_rms"].max(), (key, "frame rms")
    for i in range(2):
        a, b = (int(v) for v in g[f"{key}_win{i}_range"])
        w = g[f"{key}_win{i}"]
        e = rms(got[:, a:b] - w)
        assert e <= rel * max(rms(w), rms(ref)) and e <= abs_tol, (key, "window", i, e, rms(w))


CFG1 = [("combsub", True), ("combsub", False), ("sins", True), ("sins", False)]


def _cfg1_tol(kind, infer):
    # infer=False: the float32 running sum of vocoder.py:567-568 sits at |x| ~ 1e3 where one float32 ulp is 6e-5 cycles; a
    # rounding flip there would move the exciter by up to 1e-4.  At this shape the kernel's float64-accumulated scan
    # reproduces ATen's float32 cumsum bit for bit (test_cfg1_hip counts the flips: none), so the train-mode tails meet
    # nearly the infer-mode bar; the short fixtures of tests/test_parity.py keep the looser one for shapes where a flip
    # does occur.
    return 1e-5 if infer else 2e-5


@pytest.mark.parametrize("kind,infer", CFG1)
def test_cfg1_oracle(golden_dir, kind, infer):
    g = np.load(os.path.join(golden_dir, f"cfg1_{kind}_infer{int(infer)}.npz"))
    f0, c, noise = cfg1_inputs(g, kind)
    assert f0.shape == (1, 431, 1)
    x, pf = O.wrapped_phase(f0, SR, HOP, None, infer)
    assert np.array_equal(pf, g["phase_frames"])
    fn = O.sins_dsp if kind == "sins" else O.combsub_dsp
    r = fn(f0, c[0], c[1], c[2], noise, SR, HOP, infer=infer)
    for key in ("signal", "harmonic", "noise_out"):
        check_summary(r["noise" if key == "noise_out" else key], g, key, 5e-6 if infer else _cfg1_tol(kind, infer))


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
@pytest.mark.parametrize("kind,infer", CFG1)
def test_cfg1_hip(dev, golden_dir, kind, infer):
    """BASELINE cfg 1 through the drop-in boundary: HOT-1 + the fused tail, controls handed over as torch.split views"""
    from ddsp_svc_amd import synth
    g = np.load(os.path.join(golden_dir, f"cfg1_{kind}_infer{int(infer)}.npz"))
    f0n, c, noise = cfg1_inputs(g, kind)
    f0 = T_(f0n, dev)
    st = synth.phase(f0, SR, HOP, None, infer)
    tol_x = 6e-8 if infer else 1.3e-4
    assert np.abs(wrapdiff(N_(st.phase_frames)[..., 0], g["phase_frames"], 2 * np.pi)).max() <= 2 * np.pi * tol_x * 1.01
    # how many phase samples differ from the reference's recipe at all (the oracle reproduces the reference bit for bit,
    # test_cfg1_oracle): a bound on the rounding flips behind the tail tolerances
    x_hip = N_(synth.phase(f0, SR, HOP, None, infer, want_x=True).x)
    x_ref, _ = O.wrapped_phase(f0n, SR, HOP, None, infer)
    flips = float((x_hip != x_ref).mean())
    assert flips <= (2e-3 if infer else 1e-4), flips
    cat = torch.cat([T_(a, dev) for a in c], -1)
    c0, c1, c2 = torch.split(cat, [int(s) for s in g["sizes"]], dim=-1)
    fn = synth.sins_synth if kind == "sins" else synth.combsub_synth
    sig, harm, nz = fn(f0, st, c0, c1, c2, T_(noise, dev), SR, HOP)
    tol = _cfg1_tol(kind, infer)
    for got, key in ((sig, "signal"), (harm, "harmonic"), (nz, "noise_out")):
        check_summary(N_(got), g, key, tol)
        assert float((np.abs(N_(got)[:, ::int(g["decim"])] - g[key + "_dec"]) > 1e-4).mean()) == 0.0     # no sample off by the flip size
    # the signal-only call the benchmark times gives the same waveform
    sig2 = fn(f0, st, c0, c1, c2, T_(noise, dev), SR, HOP, want_components=False)[0]
    assert rms(N_(sig2) - N_(sig)) <= 1e-6 * rms(N_(sig))


@pytest.mark.parametrize("infer", [True, False])
def test_phase_10s_oracle(golden_dir, infer):
    g = np.load(os.path.join(golden_dir, "phase_10s.npz"))
    x, pf = O.wrapped_phase(g["f0_frames"], SR, HOP, None, infer)
    dec = int(g["decim"])
    assert np.array_equal(x[:, ::dec], g[f"x_dec_infer{int(infer)}"])
    assert np.array_equal(x[:, -2048:], g[f"x_tail_infer{int(infer)}"])
    assert np.array_equal(pf, g[f"phase_frames_infer{int(infer)}"])


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
@pytest.mark.parametrize("infer", [True, False])
def test_phase_10s_hip(dev, golden_dir, infer):
    """F = 862: the 441 344-term scan against the reference's own torch.cumsum (float64 for infer, ATen's
    float64-accumulated float32 cumsum otherwise), incl. an utterance that accumulates 8 000 cycles"""
    from ddsp_svc_amd import synth
    g = np.load(os.path.join(golden_dir, "phase_10s.npz"))
    st = synth.phase(T_(g["f0_frames"], dev), SR, HOP, None, infer, want_x=True)
    x, pf = N_(st.x), N_(st.phase_frames)[..., 0]
    dec = int(g["decim"])
    # infer: re-associated float64 scan -> at most a float32 rounding flip of the wrapped value (<= 1 ulp at 0.5).
    # train: the float32 outputs are roundings of a float64 running sum at |x| up to 8e3 (ulp 4.9e-4 cycles)
    tol = 6e-8 if infer else float(np.spacing(np.float32(8e3)))     # one float32 spacing at the largest running sum (4.9e-4)
    xd, xr = x[:, ::dec], g[f"x_dec_infer{int(infer)}"]
    assert np.abs(wrapdiff(xd, xr)).max() <= tol
    assert np.abs(wrapdiff(x[:, -2048:], g[f"x_tail_infer{int(infer)}"])).max() <= tol
    assert np.abs(wrapdiff(pf, g[f"phase_frames_infer{int(infer)}"], 2 * np.pi)).max() <= 2 * np.pi * tol * 1.01
    # How many samples are the reference's bit for bit: the reference's SEQUENTIAL float64 cumsum drifts from the exact
    # sum by up to 4e-9 cycles over 441 344 terms (measured: 3e-11 / 1.8e-9 / 3.7e-9 for the three utterances), which
    # flips the float32 rounding of the wrapped value for up to 14 % of the samples of an utterance; the tree-shaped scan
    # here stays within 1e-12 of the exact sum.  So: every difference is one rounding flip (tol above), the count is
    # bounded, and against the exactly accumulated phase (long double) the kernel's float32 output is the correctly
    # rounded value almost everywhere -- more often than the reference's own.
    assert (xd != xr).mean() < (0.2 if infer else 1e-3)
    if infer:
        f0u = O.upsample(g["f0_frames"], HOP)[..., 0]
        exact = np.cumsum((f0u.astype(np.float64) / float(SR)).astype(np.longdouble), axis=1)
        exact = (exact - np.rint(exact)).astype(np.float64)
        hip_ok = (np.abs(wrapdiff(x, exact.astype(np.float32))) == 0).mean()
        ref_ok = (np.abs(wrapdiff(xr, exact[:, ::dec].astype(np.float32))) == 0).mean()
        assert hip_ok > 0.995 and hip_ok >= ref_ok, (hip_ok, ref_ok)


def _responses(g):
    """the reference-side response tensors, formed on the CPU with the very operations make_golden.py used"""
    c = torch.from_numpy(g["ctrl"])
    gd = np.pi * torch.tanh(c)
    ap = torch.exp(1.j * torch.cumsum(gd, axis=-1))
    return ap, torch.exp(c)


@pytest.mark.parametrize("n_mag", [128, 257, 512])
def test_filters_large_oracle(golden_dir, n_mag):
    g = np.load(os.path.join(golden_dir, f"filter_n{n_mag}.npz"))
    ap, mag = _responses(g)
    for mode, key_ir, key_y, re, im, hw in ((O.MODE_ROLL, "ir_roll", "y_roll", ap.real.numpy(), ap.imag.numpy(), None),
                                            (O.MODE_HANN, "ir_hann", "y_hann", mag.numpy(), None, None),
                                            (O.MODE_DYNAMIC, "ir_dyn", "y_dyn", mag.numpy(), None, g["half_width"])):
        ir = O.impulse_response(re, im, mode, hw)
        assert rms(ir - g[key_ir]) <= 2e-6 * max(rms(g[key_ir]), 1e-3), key_ir
        y = O.ltv_fir_blockfft(g["audio"], g[key_ir])
        assert rms(y - g[key_y]) <= 2e-6 * rms(g[key_y]), key_y


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
@pytest.mark.parametrize("n_mag", [128, 257, 512])
def test_filters_large_hip(dev, golden_dir, n_mag):
    """tap synthesis (GEMM column tiles beyond 256, odd N / 2), the filter dispatch for N = 254 / 512 / 1022 and the
    one-call frequency_filter, against the reference's outputs"""
    from ddsp_svc_amd import core
    g = np.load(os.path.join(golden_dir, f"filter_n{n_mag}.npz"))
    ap, mag = _responses(g)
    ap, mag = ap.to(dev), mag.to(dev)
    audio = T_(g["audio"], dev)
    hw = T_(g["half_width"], dev).unsqueeze(-1)
    for key_ir, key_y, m, kw in (("ir_roll", "y_roll", ap, dict(hann_window=False)), ("ir_hann", "y_hann", mag, dict()),
                                 ("ir_dyn", "y_dyn", mag, dict(half_width_frames=hw))):
        taps = core.frequency_impulse_response(m, **kw)
        assert rms(N_(taps) - g[key_ir]) <= 2e-6 * max(rms(g[key_ir]), 1e-3), key_ir
        y = core.fft_convolve(audio, T_(g[key_ir], dev))
        assert rms(N_(y) - g[key_y]) <= 2e-6 * rms(g[key_y]), key_y
        y = core.frequency_filter(audio, m, **kw)
        assert rms(N_(y) - g[key_y]) <= 2e-6 * rms(g[key_y]), ("frequency_filter", key_y)


def _tail_keys(name):
    return ("ctrl_amplitudes", "ctrl_group_delay", "ctrl_noise_magnitude") if name.startswith("sins") else \
        ("ctrl_group_delay", "ctrl_harmonic_magnitude", "ctrl_noise_magnitude")


@pytest.mark.parametrize("name", ["combsub_mixed.npz", "sins_mixed.npz"])
def test_mixed_tails_oracle(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name))
    fn = O.sins_dsp if name.startswith("sins") else O.combsub_dsp
    r = fn(g["f0_frames"], *(g[k] for k in _tail_keys(name)), g["noise"], SR, HOP)
    for key in ("signal", "harmonic", "noise"):
        ref = g["noise_out" if key == "noise" else key]
        assert rms(r[key] - ref) <= 5e-6 * rms(ref), key


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
@pytest.mark.parametrize("name", ["combsub_mixed.npz", "sins_mixed.npz"])
def test_mixed_tails_hip(dev, golden_dir, name):
    """CombSub(257, 128, 512) / Sins(64, 512, 128): every filter of the tail takes a different kernel"""
    from ddsp_svc_amd import synth
    g = np.load(os.path.join(golden_dir, name))
    f0 = T_(g["f0_frames"], dev)
    st = synth.phase(f0, SR, HOP)
    fn = synth.sins_synth if name.startswith("sins") else synth.combsub_synth
    out = fn(f0, st, *(T_(g[k], dev) for k in _tail_keys(name)), T_(g["noise"], dev), SR, HOP)
    for got, key in zip(out, ("signal", "harmonic", "noise_out")):
        err = rms(N_(got) - g[key])
        assert err <= 1e-5 * rms(g[key]) and err <= 1e-4, (key, err, rms(g[key]))
