"""Pin the numpy oracle against outputs of the reference itself (tests/golden/*.npz, made by
tests/golden/make_golden.py in the build container).  CPU only."""
import os

import numpy as np
import pytest

from oracle import ddsp_oracle as O


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def rms(a):
    return float(np.sqrt(np.mean(np.square(np.asarray(a, dtype=np.float64)))))


@pytest.mark.parametrize("name", ["upsample.npz", "upsample_hop64.npz"])
def test_upsample_bit_exact(golden_dir, name):
    g = _load(golden_dir, name)
    out = O.upsample(g["sig"], int(g["hop"]))
    assert out.dtype == np.float32
    assert np.array_equal(out, g["out"])          # fma(w0,a,fl32(w1*b)) is ATen's exact blend


@pytest.mark.parametrize("infer", [True, False])
@pytest.mark.parametrize("use_ip", [False, True])
def test_phase(golden_dir, infer, use_ip):
    g = _load(golden_dir, "phase.npz")
    tag = f"infer{int(infer)}_ip{int(use_ip)}"
    x, pf = O.wrapped_phase(g["f0_frames"], 44100, 512, g["initial_phase"] if use_ip else None, infer)
    # the per-sample fp32 f0, the fp64 (or fp64-accumulated fp32) scan and the wrap are all
    # reproduced bit for bit
    assert np.array_equal(x, g["x_" + tag])
    assert np.array_equal(pf, g["phase_frames_" + tag])


@pytest.mark.parametrize("n_mag", [65, 129, 256])
def test_impulse_response_modes(golden_dir, n_mag):
    g = _load(golden_dir, f"filter_n{n_mag}.npz")
    for mode, key, re, im, hw in ((O.MODE_ROLL, "ir_roll", g["resp_re"], g["resp_im"], None),
                                  (O.MODE_HANN, "ir_hann", g["mag"], None, None),
                                  (O.MODE_DYNAMIC, "ir_dyn", g["mag"], None, g["half_width"])):
        ir = O.impulse_response(re, im, mode, hw)
        ref = g[key]
        assert rms(ir - ref) <= 2e-6 * max(rms(ref), 1e-3), (mode, rms(ir - ref), rms(ref))


@pytest.mark.parametrize("n_mag", [65, 129, 256])
def test_frequency_filter_modes(golden_dir, n_mag):
    g = _load(golden_dir, f"filter_n{n_mag}.npz")
    for mode, key, re, im, hw in ((O.MODE_ROLL, "y_roll", g["resp_re"], g["resp_im"], None),
                                  (O.MODE_HANN, "y_hann", g["mag"], None, None),
                                  (O.MODE_DYNAMIC, "y_dyn", g["mag"], None, g["half_width"])):
        y = O.frequency_filter(g["audio"], re, im, mode, hw)
        ref = g[key]
        assert rms(y - ref) <= 2e-6 * rms(ref), (mode, rms(y - ref), rms(ref))


def test_fir_definition_matches_block_fft(golden_dir):
    g = _load(golden_dir, "filter_n65.npz")
    ir = O.impulse_response(g["mag"], None, O.MODE_HANN)
    a = O.ltv_fir_blockfft(g["audio"], ir)
    b = O.ltv_fir_direct(g["audio"], ir)
    assert np.abs(a - b).max() <= 1e-11 * max(1.0, np.abs(a).max())


@pytest.mark.parametrize("name,infer", [("sins_h256.npz", True), ("sins_h128.npz", True),
                                        ("sins_h40_train.npz", False)])
def test_sins_tail(golden_dir, name, infer):
    g = _load(golden_dir, name)
    r = O.sins_dsp(g["f0_frames"], g["ctrl_amplitudes"], g["ctrl_group_delay"],
                   g["ctrl_noise_magnitude"], g["noise"], infer=infer)
    # the reference's own fp32 pipeline sits ~1.5e-6 (relative) from the float64 oracle
    for k, gk in (("signal", "signal"), ("harmonic", "harmonic"), ("noise", "noise_out")):
        assert rms(r[k] - g[gk]) <= 5e-6 * rms(g[gk]), (k, rms(r[k] - g[gk]), rms(g[gk]))


@pytest.mark.parametrize("name,infer", [("combsub_256.npz", True), ("combsub_128.npz", True),
                                        ("combsub_small_train.npz", False)])
def test_combsub_tail(golden_dir, name, infer):
    g = _load(golden_dir, name)
    r = O.combsub_dsp(g["f0_frames"], g["ctrl_group_delay"], g["ctrl_harmonic_magnitude"],
                      g["ctrl_noise_magnitude"], g["noise"], infer=infer)
    for k, gk in (("signal", "signal"), ("harmonic", "harmonic"), ("noise", "noise_out")):
        assert rms(r[k] - g[gk]) <= 5e-6 * rms(g[gk]), (k, rms(r[k] - g[gk]), rms(g[gk]))


def test_combsub_tail_adjoint(golden_dir):
    """the composed adjoint (what bench.py's training row is gated on) against the gradients the REFERENCE's autograd
    produced through ddsp.vocoder.CombSub's own DSP tail (make_golden.py, combsub_grad.npz)"""
    g = _load(golden_dir, "combsub_grad.npz")
    got = O.combsub_dsp_backward(g["cotangent"], g["f0_frames"], g["ctrl_group_delay"], g["ctrl_harmonic_magnitude"],
                                 g["ctrl_noise_magnitude"], g["noise"])
    for k in ("group_delay", "harmonic_magnitude", "noise_magnitude"):
        ref = g["grad_" + k]
        assert rms(got[k] - ref) <= 2e-5 * rms(ref), (k, rms(got[k] - ref), rms(ref))


# ---- CombSubFast / CombSubSuperFast -------------------------------------------
def test_fast_source_gen(golden_dir):
    g = _load(golden_dir, "fastsrc.npz")
    comb, pf, _ = O.fast_source_gen(g["f0_frames"], 44100, 512)
    assert np.array_equal(pf, g["phase_frames"])                 # the float32 recipe is reproduced bit for bit
    assert np.abs(comb - g["combtooth"]).max() <= 2e-7           # float64 sine/divide vs torch's float32 sinc


@pytest.mark.parametrize("name,infer", [("csfast_a.npz", True), ("csfast_train.npz", False)])
def test_combsubfast_tail(golden_dir, name, infer):
    g = _load(golden_dir, name)
    r = O.combsubfast_dsp(g["f0_frames"], g["ctrl_harmonic_magnitude"], g["ctrl_harmonic_phase"],
                          g["ctrl_noise_magnitude"], g["noise"], infer=infer)
    assert np.array_equal(r["phase_frames"], g["phase_frames"])
    assert rms(r["signal"] - g["signal"]) <= 2e-6 * rms(g["signal"])


@pytest.mark.parametrize("name", ["cssuper_a.npz", "cssuper_short.npz", "cssuper_f3.npz"])
def test_combsubsuperfast_tail(golden_dir, name):
    g = _load(golden_dir, name)
    r = O.combsubsuperfast_dsp(g["f0_frames"], g["ctrl_harmonic_magnitude"], g["ctrl_harmonic_phase"],
                               g["ctrl_noise_magnitude"], g["ctrl_noise_phase"], g["noise"], window=g["window"])
    assert np.array_equal(r["phase_frames"], g["phase_frames"])
    assert np.abs(r["exciter"] - g["combtooth"]).max() <= 2e-7
    assert rms(r["signal"] - g["signal"]) <= 2e-6 * rms(g["signal"])
    # the default window of the oracle is the module's buffer
    r2 = O.combsubsuperfast_dsp(g["f0_frames"], g["ctrl_harmonic_magnitude"], g["ctrl_harmonic_phase"],
                                g["ctrl_noise_magnitude"], g["ctrl_noise_phase"], g["noise"])
    assert rms(r2["signal"] - r["signal"]) <= 1e-6 * rms(g["signal"])
