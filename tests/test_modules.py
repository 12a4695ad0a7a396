"""Module-level drop-in tests for ``ddsp_svc_amd.vocoder`` (reference: ddsp/vocoder.py:532-611, 788-862).

Two layers:
  * with a small stand-in for Unit2Control (runs everywhere, both backends): constructor / buffers /
    state_dict keys / forward signature and return structure, and the DSP result against the oracle on
    the controls the stand-in produced;
  * against the REFERENCE modules themselves (only where /root/reference is importable, i.e. the build
    container): same weights, same inputs, same injected noise -> same waveform.
"""
import os
import sys
from unittest import mock
from unittest.mock import MagicMock

import numpy as np
import pytest
import torch

from oracle import ddsp_oracle as O
from tests.backends import BACKENDS, dev  # noqa: F401  (fixture)

SR, HOP = 44100, 512
REF = os.environ.get("DDSP_REFERENCE_PATH", "/root/reference")


def rms(a):
    return float(np.sqrt(np.mean(np.square(np.asarray(a, dtype=np.float64)))))


class TinyUnit2Control(torch.nn.Module):
    """Stand-in with Unit2Control's interface (ddsp/unit2control.py:26-109): forward(units, f0, phase,
    volume, spk_id, spk_mix_dict) -> (dict of [B,F,n] views of one tensor, hidden [B,F,256])."""

    def __init__(self, n_unit, n_spk, output_splits, **kwargs):
        super().__init__()
        self.kwargs = kwargs
        self.output_splits = output_splits
        self.proj = torch.nn.Linear(n_unit + 3, 256)
        self.dense_out = torch.nn.Linear(256, sum(output_splits.values()))

    def forward(self, units, f0, phase, volume, spk_id=None, spk_mix_dict=None, aug_shift=None):
        x = torch.tanh(self.proj(torch.cat([units, (1 + f0 / 700).log(), phase / np.pi, volume], -1)))
        e = self.dense_out(x)
        return dict(zip(self.output_splits, torch.split(e, list(self.output_splits.values()), dim=-1))), x


def _inputs(B, F, n_unit, device, seed=0):
    g = torch.Generator().manual_seed(seed)
    f0 = torch.from_numpy(O.synth_f0(B, F, SR, HOP, seed=seed + 3))
    f0[0] = torch.clamp(f0[0] * 2.2, 65, 800)                   # dynamic-window quirk region (f0 > 259 Hz)
    units = torch.randn(B, F, n_unit, generator=g)
    vol = torch.rand(B, F, 1, generator=g) * 0.1
    u = torch.rand(B, F * HOP, generator=g)
    return units.to(device), f0.to(device), vol.to(device), u.to(device)


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
@pytest.mark.parametrize("kind", ["combsub", "sins"])
def test_dropin_module_structure_and_dsp(dev, kind, monkeypatch):
    from ddsp_svc_amd import vocoder as V
    torch.manual_seed(0)
    B, F, n_unit = 2, 7, 12
    if kind == "combsub":
        m = V.CombSub(SR, HOP, 65, 33, 17, n_unit=n_unit, n_spk=1, unit2ctrl_factory=TinyUnit2Control)
        keys = ("group_delay", "harmonic_magnitude", "noise_magnitude")
    else:
        m = V.Sins(SR, HOP, 24, 65, 17, n_unit=n_unit, n_spk=1, unit2ctrl_factory=TinyUnit2Control)
        keys = ("amplitudes", "group_delay", "noise_magnitude")
    m = m.to(dev).eval()
    sd = m.state_dict()
    assert "sampling_rate" in sd and "block_size" in sd and sd["sampling_rate"].dim() == 0      # vocoder.py:546-547
    assert any(k.startswith("unit2ctrl.") for k in sd)
    assert tuple(m.unit2ctrl.output_splits) == keys                                             # vocoder.py:549-554 / 804-808
    units, f0, vol, u = _inputs(B, F, n_unit, dev)
    captured = {}
    m.unit2ctrl.register_forward_hook(lambda mod, i, o: captured.update(ctrls=o[0], phase=i[2]))
    monkeypatch.setattr(torch, "rand", lambda *a, **k: u)                                       # the rand_like draw
    with torch.no_grad():
        signal, hidden, (harmonic, noise) = m(units, f0, vol, spk_id=None, spk_mix_dict=None, initial_phase=None,
                                              infer=True)
    assert signal.shape == (B, F * HOP) and hidden.shape == (B, F, 256)
    assert harmonic.shape == signal.shape and noise.shape == signal.shape
    # phase_frames handed to Unit2Control == 2*pi*x[:, ::hop] of the oracle
    f0n = f0.cpu().numpy()
    _, pf = O.wrapped_phase(f0n, SR, HOP)
    d = captured["phase"].cpu().numpy()[..., 0] - pf
    d = d - 2 * np.pi * np.round(d / (2 * np.pi))
    assert np.abs(d).max() <= 5e-7
    c = [captured["ctrls"][k].detach().cpu().numpy() for k in keys]
    nz = (u.cpu().numpy() * np.float32(2) - np.float32(1)).astype(np.float32)
    ref = O.combsub_dsp(f0n, c[0], c[1], c[2], nz, SR, HOP) if kind == "combsub" else O.sins_dsp(f0n, c[0], c[1], c[2], nz, SR, HOP)
    for got, key in ((signal, "signal"), (harmonic, "harmonic"), (noise, "noise")):
        e = rms(got.cpu().numpy() - ref[key])
        assert e <= 1e-5 * rms(ref[key]) and e <= 1e-4, (key, e)
    # return_components=False skips the tuple (callers discard it: solver.py:37, main.py:259)
    m.return_components = False
    with torch.no_grad():
        s2, _, (h2, n2) = m(units, f0, vol)
    assert h2 is None and n2 is None
    assert rms(s2.cpu().numpy() - signal.cpu().numpy()) <= 1e-7
    # with gradients enabled the tail runs through the differentiable primitives (values: tests/test_backward_fir.py)
    m.return_components = True
    s3, _, (h3, n3) = m(units, f0, vol)
    assert s3.requires_grad and rms(s3.detach().cpu().numpy() - ref["signal"]) <= 1e-5 * rms(ref["signal"])
    s3.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.unit2ctrl.parameters())


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
@pytest.mark.parametrize("kind", ["fast", "superfast"])
def test_dropin_fast_modules(dev, kind, monkeypatch):
    """CombSubFast / CombSubSuperFast: constructor, buffers, Unit2Control arguments, forward contract and the DSP
    result against the oracle on the controls the stand-in produced (vocoder.py:613-786)."""
    from ddsp_svc_amd import vocoder as V
    torch.manual_seed(0)
    B, F, n_unit = 2, 7, 12
    if kind == "fast":
        m = V.CombSubFast(SR, HOP, n_unit=n_unit, n_spk=1, pcmer_norm=True, unit2ctrl_factory=TinyUnit2Control)
        keys = {"harmonic_magnitude": 513, "harmonic_phase": 513, "noise_magnitude": 513}
        assert m.unit2ctrl.kwargs == {"use_pitch_aug": False, "pcmer_norm": True}                 # vocoder.py:733
    else:
        m = V.CombSubSuperFast(SR, HOP, 2048, n_unit=n_unit, n_spk=1, unit2ctrl_factory=TinyUnit2Control)
        keys = {"harmonic_magnitude": 1025, "harmonic_phase": 1025, "noise_magnitude": 1025, "noise_phase": 1025}
        assert m.unit2ctrl.kwargs == {"use_pitch_aug": False, "use_naive_v2": True, "use_conv_stack": True}   # :637
    m = m.to(dev).eval()
    sd = m.state_dict()
    assert sd["sampling_rate"].dim() == 0 and sd["block_size"].dim() == 0
    if kind == "fast":
        assert torch.equal(sd["window"].cpu(), torch.sqrt(torch.hann_window(1024)))                # :726
    else:
        assert int(sd["win_length"]) == 2048 and torch.equal(sd["window"].cpu(), torch.hann_window(2048))   # :628-629
    assert dict(m.unit2ctrl.output_splits) == keys
    units, f0, vol, u = _inputs(B, F, n_unit, dev)
    gz = torch.randn(B, F * HOP, generator=torch.Generator().manual_seed(5)).to(dev)
    captured = {}
    m.unit2ctrl.register_forward_hook(lambda mod, i, o: captured.update(ctrls=o[0], phase=i[2]))
    monkeypatch.setattr(torch, "rand", lambda *a, **k: u)
    monkeypatch.setattr(torch, "randn", lambda *a, **k: gz)
    with torch.no_grad():
        signal, hidden, (h2, n2) = m(units, f0, vol, spk_id=None, spk_mix_dict=None, aug_shift=None,
                                     initial_phase=None, infer=True)
    assert signal.shape == (B, F * HOP) and hidden.shape == (B, F, 256)
    assert h2 is signal and n2 is signal                                                         # :710, :786
    f0n = f0.cpu().numpy()
    c = {k: v.detach().cpu().numpy() for k, v in captured["ctrls"].items()}
    if kind == "fast":
        nz = (u.cpu().numpy() * np.float32(2) - np.float32(1)).astype(np.float32)
        ref = O.combsubfast_dsp(f0n, c["harmonic_magnitude"], c["harmonic_phase"], c["noise_magnitude"], nz, SR, HOP)
        d = captured["phase"].cpu().numpy()[..., 0] - ref["phase_frames"]
        assert np.abs(d - 2 * np.pi * np.round(d / (2 * np.pi))).max() <= 5e-7
    else:
        ref = O.combsubsuperfast_dsp(f0n, c["harmonic_magnitude"], c["harmonic_phase"], c["noise_magnitude"],
                                     c["noise_phase"], gz.cpu().numpy(), SR, HOP, 2048)
        assert np.array_equal(captured["phase"].cpu().numpy()[..., 0], ref["phase_frames"])
        comb, pf = m.fast_source_gen(f0)                                                         # :639-651
        assert comb.shape == (B, F * HOP) and pf.shape == (B, F, 1)
        assert np.abs(comb.cpu().numpy() - ref["exciter"]).max() <= 3e-7
    e = rms(signal.cpu().numpy() - ref["signal"])
    assert e <= 1e-5 * rms(ref["signal"]), (e, rms(ref["signal"]))
    # with gradients enabled the spectral tail runs through autograd (tests/test_backward_fast.py checks the values)
    s2, _, _ = m(units, f0, vol)
    assert s2.requires_grad and rms(s2.detach().cpu().numpy() - ref["signal"]) <= 1e-5 * rms(ref["signal"])
    s2.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.unit2ctrl.parameters())


def _import_reference():
    if not os.path.isdir(os.path.join(REF, "ddsp")):
        pytest.skip("reference checkout not present (only in the build container)")
    if REF not in sys.path:
        sys.path.insert(0, REF)
    for name in ["transformers", "pyworld", "parselmouth", "torchcrepe", "resampy", "fairseq", "torchaudio",
                 "torchaudio.transforms", "gin", "local_attention", "librosa", "librosa.sequence", "librosa.util",
                 "librosa.filters", "librosa.core", "soundfile"]:
        sys.modules.setdefault(name, MagicMock())
    import ddsp.core as rcore
    import ddsp.vocoder as rvoc
    return rcore, rvoc


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
@pytest.mark.parametrize("kind", ["combsub", "sins"])
def test_against_reference_module(dev, kind):
    """Same weights (strict state_dict load), inputs and noise -> the reference's waveform.  The reference module runs its
    own CPU PyTorch path (the north star's yardstick); the drop-in runs on ``dev`` -- the emulator, or the MI355X when a
    reference checkout travels with the snapshot (DDSP_REFERENCE_PATH, tools/with_reference.sh)."""
    rcore, rvoc = _import_reference()
    ref_cls = {"combsub": getattr(rvoc, "_reference_CombSub", rvoc.CombSub),
               "sins": getattr(rvoc, "_reference_Sins", rvoc.Sins)}[kind]
    from ddsp_svc_amd import vocoder as V
    torch.manual_seed(1)
    B, F, n_unit = 2, 9, 16
    args = (SR, HOP, 65, 33, 65) if kind == "combsub" else (SR, HOP, 40, 65, 33)
    ref = ref_cls(*args, n_unit=n_unit, n_spk=1).eval()
    ours = (V.CombSub if kind == "combsub" else V.Sins)(*args, n_unit=n_unit, n_spk=1).eval()   # real Unit2Control
    ours.load_state_dict(ref.state_dict(), strict=True)
    ours = ours.to(dev)
    units, f0, vol, u = _inputs(B, F, n_unit, torch.device("cpu"), seed=5)
    ud = u.to(dev)
    with torch.no_grad():
        with mock.patch("torch.rand_like", side_effect=lambda t: u.reshape(t.shape)):
            r_sig, r_hid, (r_h, r_n) = ref(units, f0, vol, infer=True)
        with mock.patch("torch.rand", side_effect=lambda *a, **k: ud):
            o_sig, o_hid, (o_h, o_n) = ours(units.to(dev), f0.to(dev), vol.to(dev), infer=True)
    o_sig, o_hid, o_h, o_n = (t.cpu() for t in (o_sig, o_hid, o_h, o_n))
    # Unit2Control itself runs as PyTorch on either device: on the GPU its float32 GEMMs round differently from the CPU's
    hid_tol = 1e-6 if dev.type == "cpu" else 2e-4
    assert rms((o_hid - r_hid).numpy()) <= hid_tol * max(rms(r_hid.numpy()), 1e-12) + 1e-7
    sig_tol = 2e-5 if dev.type == "cpu" else 5e-5      # measured on the MI355X: 3.5e-6 (profiles/r04_v11_reference_on_gpu.log)
    for got, want, name in ((o_sig, r_sig, "signal"), (o_h, r_h, "harmonic"), (o_n, r_n, "noise")):
        e = rms((got - want).numpy())
        print("%s module on %s against the reference's CPU path: %s rms error %.2e (rms %.2e)" % (kind, dev, name, e, rms(want.numpy())))
        assert e <= sig_tol * rms(want.numpy()) and e <= 1e-4, (name, e, rms(want.numpy()))
    if dev.type != "cpu":
        # the DSP alone, on the controls the REFERENCE's Unit2Control produced on the CPU: the north star's bar proper
        # (identical f0 / amplitude / noise inputs -> within 1e-4 RMS of the reference CPU PyTorch path)
        captured = {}
        h = ref.unit2ctrl.register_forward_hook(lambda mod, i, o: captured.update(ctrls=o[0]))
        with torch.no_grad(), mock.patch("torch.rand_like", side_effect=lambda t: u.reshape(t.shape)):
            ref(units, f0, vol, infer=True)
        h.remove()
        from ddsp_svc_amd import synth
        c = [v.to(dev) for v in captured["ctrls"].values()]
        st = synth.phase(f0.to(dev), SR, HOP)
        tail = synth.combsub_synth if kind == "combsub" else synth.sins_synth
        d_sig, d_h, d_n = tail(f0.to(dev), st, c[0], c[1], c[2], ud, SR, HOP, noise_is_u01=True)
        for got, want, name in ((d_sig, r_sig, "signal"), (d_h, r_h, "harmonic"), (d_n, r_n, "noise")):
            e = rms((got.cpu() - want).numpy())
            print("%s DSP on %s, controls from the reference's Unit2Control: %s rms error %.2e (rms %.2e)" % (kind, dev, name, e, rms(want.numpy())))
            assert e <= 1e-5 * rms(want.numpy()) and e <= 1e-4, ("dsp on reference controls", name, e, rms(want.numpy()))


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
@pytest.mark.parametrize("kind", ["fast", "superfast"])
def test_fast_against_reference_module(dev, kind):
    """CombSubFast / CombSubSuperFast with the reference's own Unit2Control: same weights (strict state_dict load,
    window buffers included), inputs and noise -> the reference's waveform (reference on its CPU path, drop-in on ``dev``)."""
    rcore, rvoc = _import_reference()
    from ddsp_svc_amd import vocoder as V
    name = {"fast": "CombSubFast", "superfast": "CombSubSuperFast"}[kind]
    ref_cls = getattr(rvoc, "_reference_" + name, getattr(rvoc, name))
    torch.manual_seed(2)
    B, F, n_unit = 2, 9, 16
    args = (SR, HOP) if kind == "fast" else (SR, HOP, 2048)
    ref = ref_cls(*args, n_unit=n_unit, n_spk=1).eval()
    ours = getattr(V, name)(*args, n_unit=n_unit, n_spk=1).eval()
    ours.load_state_dict(ref.state_dict(), strict=True)
    ours = ours.to(dev)
    units, f0, vol, u = _inputs(B, F, n_unit, torch.device("cpu"), seed=6)
    gz = torch.randn(B, F * HOP, generator=torch.Generator().manual_seed(7))
    ud, gd = u.to(dev), gz.to(dev)
    with torch.no_grad():
        with mock.patch("torch.rand_like", side_effect=lambda t: u.reshape(t.shape)), \
                mock.patch("torch.randn_like", side_effect=lambda t: gz.reshape(t.shape)):
            r_sig, r_hid, _ = ref(units, f0, vol, infer=True)
        with mock.patch("torch.rand", side_effect=lambda *a, **k: ud), \
                mock.patch("torch.randn", side_effect=lambda *a, **k: gd):
            o_sig, o_hid, _ = ours(units.to(dev), f0.to(dev), vol.to(dev), infer=True)
    o_sig, o_hid = o_sig.cpu(), o_hid.cpu()
    on_cpu = dev.type == "cpu"                          # on the GPU Unit2Control's own GEMMs round differently (see above)
    assert rms((o_hid - r_hid).numpy()) <= (1e-6 if on_cpu else 2e-4) * max(rms(r_hid.numpy()), 1e-12) + 1e-7
    e = rms((o_sig - r_sig).numpy())
    print("%s module on %s against the reference's CPU path: signal rms error %.2e (rms %.2e)" % (name, dev, e, rms(r_sig.numpy())))
    assert e <= (2e-5 if on_cpu else 5e-5) * rms(r_sig.numpy()) and e <= 1e-4, (e, rms(r_sig.numpy()))   # measured on the MI355X: 9e-7


def test_patch_reference_swaps_classes_and_keeps_cpu_core():
    """patch_reference(): classes and ddsp.core names rebound (incl. the two window helpers, core.py:185,240, and the
    names ddsp.vocoder bound at import, vocoder.py:16); host tensors keep the reference's code -- functions AND module
    forward (load_model's default device is 'cpu', vocoder.py:506) -- with the reference's autograd intact;
    unpatch_reference() restores everything."""
    rcore, rvoc = _import_reference()
    from ddsp_svc_amd import vocoder as V
    originals = {n: getattr(rcore, n) for n in V.PATCHED_CORE_FUNCTIONS}
    ref_combsub = rvoc.CombSub
    try:
        V.patch_reference()
        assert rvoc.Sins is V.Sins and rvoc.CombSub is V.CombSub
        assert rvoc.CombSubFast is V.CombSubFast and rvoc.CombSubSuperFast is V.CombSubSuperFast
        for n in V.PATCHED_CORE_FUNCTIONS:
            assert getattr(rcore, n) is not originals[n] and getattr(rcore, "_reference_" + n) is originals[n]
        assert rvoc.upsample is rcore.upsample and rvoc.frequency_filter is rcore.frequency_filter
        sig = torch.rand(1, 4, 2, requires_grad=True)
        out = rcore.upsample(sig, 8)                         # CPU tensors keep the reference implementation ...
        assert out.shape == (1, 32, 2)
        np.testing.assert_array_equal(out.detach().numpy(), O.upsample(sig.detach().numpy(), 8))
        out.sum().backward()                                 # ... and its autograd
        assert sig.grad is not None and float(sig.grad.sum()) == pytest.approx(64.0)
        ir = torch.randn(1, 3, 30)
        np.testing.assert_array_equal(rcore.apply_window_to_impulse_response(ir).numpy(),
                                      originals["apply_window_to_impulse_response"](ir).numpy())
        # a drop-in module fed host tensors runs the reference's forward (no HIP library involved)
        torch.manual_seed(1)
        m = rvoc.CombSub(SR, HOP, 17, 9, 9, n_unit=8, n_spk=1).eval()
        assert type(m) is V.CombSub and V.CombSub._reference_cls is ref_combsub
        units, f0, vol, u = _inputs(1, 4, 8, torch.device("cpu"), seed=2)
        with torch.no_grad(), mock.patch("torch.rand_like", side_effect=lambda t: u.reshape(t.shape)):
            sig_cpu, _, _ = m(units, f0, vol)
            ref = ref_combsub(SR, HOP, 17, 9, 9, n_unit=8, n_spk=1).eval()
            ref.load_state_dict(m.state_dict(), strict=True)
            sig_ref, _, _ = ref(units, f0, vol)
        assert torch.equal(sig_cpu, sig_ref)
    finally:
        V.unpatch_reference()
    assert rvoc.CombSub is ref_combsub and V.CombSub._reference_cls is None
    for n in V.PATCHED_CORE_FUNCTIONS:
        assert getattr(rcore, n) is originals[n] and not hasattr(rcore, "_reference_" + n)
    assert rvoc.upsample is originals["upsample"]


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
def test_patch_reference_reaches_the_cascades(dev):
    """The diffusion / reflow cascades bind CombSubFast / CombSubSuperFast by name when they are imported
    (diffusion/vocoder.py:13, reflow/vocoder.py:12) and build them inside Unit2Wav / Unit2WavFast
    (diffusion/vocoder.py:234,282; reflow/vocoder.py:164).  After patch_reference() those constructors must yield
    the HIP-backed classes, take the reference's checkpoints unchanged (strict) and produce the reference's DDSP
    waveform -- BASELINE.json cfg 5's integration seam, on the emulated device here."""
    rcore, rvoc = _import_reference()
    import importlib
    dvoc = importlib.import_module("diffusion.vocoder")
    fvoc = importlib.import_module("reflow.vocoder")
    from ddsp_svc_amd import vocoder as V
    saved = {m: {n: getattr(m, n) for n in ("CombSubFast", "CombSubSuperFast") if hasattr(m, n)} for m in (dvoc, fvoc)}
    ref_names = {n: getattr(rvoc, "_reference_" + n, getattr(rvoc, n)) for n in ("Sins", "CombSub", "CombSubFast",
                                                                                 "CombSubSuperFast")}
    n_unit = 16
    torch.manual_seed(4)
    ref_fast = dvoc.Unit2WavFast(SR, HOP, 2048, n_unit, 1, n_layers=1, n_chans=32).eval()
    ref_slow = dvoc.Unit2Wav(SR, HOP, n_unit, 1, n_layers=1, n_chans=32).eval()
    ref_flow = fvoc.Unit2Wav(SR, HOP, 2048, n_unit, 1, n_layers=1, n_chans=32).eval()
    try:
        V.patch_reference()
        assert dvoc.CombSubFast is V.CombSubFast and dvoc.CombSubSuperFast is V.CombSubSuperFast
        assert fvoc.CombSubSuperFast is V.CombSubSuperFast
        ours_fast = dvoc.Unit2WavFast(SR, HOP, 2048, n_unit, 1, n_layers=1, n_chans=32).eval()
        ours_slow = dvoc.Unit2Wav(SR, HOP, n_unit, 1, n_layers=1, n_chans=32).eval()
        ours_flow = fvoc.Unit2Wav(SR, HOP, 2048, n_unit, 1, n_layers=1, n_chans=32).eval()
        assert type(ours_fast.ddsp_model) is V.CombSubSuperFast and type(ours_slow.ddsp_model) is V.CombSubFast
        assert type(ours_flow.ddsp_model) is V.CombSubSuperFast
        units, f0, vol, u = _inputs(1, 6, n_unit, torch.device("cpu"), seed=8)
        gz = torch.randn(1, 6 * HOP, generator=torch.Generator().manual_seed(9))
        ud, gd = u.to(dev), gz.to(dev)
        for ours, ref in ((ours_fast, ref_fast), (ours_slow, ref_slow), (ours_flow, ref_flow)):
            ours.load_state_dict(ref.state_dict(), strict=True)
            ours = ours.to(dev)                          # the whole cascade object, as main_diff.py moves it (diffusion/vocoder.py:49)
            with torch.no_grad():
                with mock.patch("torch.rand_like", side_effect=lambda t: u.reshape(t.shape)), \
                        mock.patch("torch.randn_like", side_effect=lambda t: gz.reshape(t.shape)):
                    r_wav, r_hid, _ = ref.ddsp_model(units, f0, vol, infer=True)
                with mock.patch("torch.rand", side_effect=lambda *a, **k: ud), \
                        mock.patch("torch.randn", side_effect=lambda *a, **k: gd):
                    o_wav, o_hid, _ = ours.ddsp_model(units.to(dev), f0.to(dev), vol.to(dev), infer=True)
            e = rms((o_wav.cpu() - r_wav).numpy())
            print("%s.ddsp_model on %s against the reference's CPU path: rms error %.2e (rms %.2e)" % (type(ref).__name__, dev, e, rms(r_wav.numpy())))
            tol = 2e-5 if dev.type == "cpu" else 5e-5    # Unit2Control's float32 GEMMs on the GPU against the CPU's (measured: 9e-7)
            assert e <= tol * rms(r_wav.numpy()) and e <= 1e-4, (type(ref).__name__, e, rms(r_wav.numpy()))
    finally:
        V.unpatch_reference()
        for m, names in saved.items():
            for n, c in names.items():
                setattr(m, n, c)


class _SpectrogramStandIn(torch.nn.Module):
    """torchaudio.transforms.Spectrogram as ddsp/loss.py:20 configures it (torchaudio is not installed here; the same
    stand-in as tests/golden/make_golden.py): periodic Hann, center=False, power=1, normalized by the window's L2 norm."""

    def __init__(self, n_fft, win_length=None, hop_length=None, power=1, normalized=True, center=False, **kw):
        super().__init__()
        self.n_fft, self.hop = n_fft, hop_length
        self.register_buffer("window", torch.hann_window(n_fft))

    def forward(self, x):
        z = torch.stft(x, self.n_fft, hop_length=self.hop, win_length=self.n_fft, window=self.window, center=False,
                       normalized=False, onesided=True, return_complex=True)
        return z.abs() / self.window.pow(2).sum().sqrt()


@pytest.mark.parametrize("dev", ["emu"], indirect=True)
def test_patch_reference_loss(dev, monkeypatch):
    """patch_reference_loss(): ``ddsp.loss.SSSLoss`` / ``RSSLoss`` (and the name train.py imported) become the drop-ins;
    the reference's own classes, run as they are over a stand-in for torchaudio's Spectrogram, give the same value and the
    same gradient for the same draw of transform sizes (loss.py:47)."""
    _import_reference()
    sys.modules["torchaudio"].transforms.Spectrogram = _SpectrogramStandIn
    sys.modules.pop("ddsp.loss", None)
    from ddsp_svc_amd import loss as L
    dl = L.patch_reference_loss()
    try:
        assert dl.SSSLoss is L.SSSLoss and dl.RSSLoss is L.RSSLoss
        ref_cls = dl._reference_RSSLoss
        g = torch.Generator().manual_seed(4)
        xt = torch.randn(2, 9000, generator=g) * 0.1
        xp = (xt * 0.8 + 0.05 * torch.randn(2, 9000, generator=g))
        sizes = torch.tensor([397, 1024, 263])
        monkeypatch.setattr(torch, "randint", lambda *a, **k: sizes)
        ref = ref_cls(256, 1100, 3, eps=1e-5, device="cpu")
        a = xp.clone().requires_grad_(True)
        want = ref(a, xt)
        want.backward()
        ours = dl.RSSLoss(256, 1100, 3, eps=1e-5, device="cpu")
        b = xp.clone().requires_grad_(True)
        got = ours(b, xt)
        got.backward()
        assert "RandomScaleWaveLoss" in type(got.grad_fn).__name__
        assert abs(float(got.detach()) - float(want.detach())) <= 2e-5 * float(want.detach())
        assert rms(b.grad.numpy() - a.grad.numpy()) <= 2e-4 * rms(a.grad.numpy())
    finally:
        dl.SSSLoss, dl.RSSLoss = dl._reference_SSSLoss, dl._reference_RSSLoss


# ---- mixed-precision callers (configs/diffusion-new-fp16.yaml:38, diffusion/solver_new.py:132: the cascade's DDSP stage runs under
# torch.autocast(fp16 | bf16); its Unit2Control then emits HALF-precision controls) ------------------------------------------------
def _autocast(dev, dtype):
    return torch.autocast("cuda" if dev.type == "cuda" else "cpu", dtype=dtype)


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("kind", ["fast", "superfast", "combsub", "sins"])
def test_dropin_modules_under_autocast(dev, kind, dtype, monkeypatch):
    """Under autocast the stand-in network's linears emit ``dtype`` controls.  What the reference does with them (probed in the build
    container, CPU: CombSubFast / SuperFast under bf16 -> float32 waveform from the UPCAST controls, ``hidden`` in bf16; Sins /
    CombSub raise -- complex bf16 / ComplexHalf exp do not exist there) fixes the contract of the drop-ins: float32 waveform whatever
    the controls' dtype, computed from the controls' values as they are (upcast, never re-rounded), ``hidden`` passed through, and a
    backward pass that hands every parameter a finite gradient through the ``dtype`` controls."""
    from ddsp_svc_amd import vocoder as V
    if dev.type == "cpu" and dtype == torch.float16:
        pytest.skip("torch's CPU linear under float16 autocast is not what the cascade runs (the GPU leg covers float16)")
    torch.manual_seed(0)
    B, F, n_unit = 2, 6, 12
    if kind == "fast":
        m = V.CombSubFast(SR, HOP, n_unit=n_unit, n_spk=1, unit2ctrl_factory=TinyUnit2Control)
    elif kind == "superfast":
        m = V.CombSubSuperFast(SR, HOP, 2048, n_unit=n_unit, n_spk=1, unit2ctrl_factory=TinyUnit2Control)
    elif kind == "combsub":
        m = V.CombSub(SR, HOP, 65, 33, 17, n_unit=n_unit, n_spk=1, unit2ctrl_factory=TinyUnit2Control)
    else:
        m = V.Sins(SR, HOP, 24, 65, 17, n_unit=n_unit, n_spk=1, unit2ctrl_factory=TinyUnit2Control)
    m = m.to(dev).eval()
    units, f0, vol, u = _inputs(B, F, n_unit, dev)
    gz = torch.randn(B, F * HOP, generator=torch.Generator().manual_seed(5)).to(dev)
    captured = {}
    m.unit2ctrl.register_forward_hook(lambda mod, i, o: captured.update(ctrls=o[0]))
    monkeypatch.setattr(torch, "rand", lambda *a, **k: u)
    monkeypatch.setattr(torch, "randn", lambda *a, **k: gz)
    with torch.no_grad(), _autocast(dev, dtype):
        signal, hidden, _ = m(units, f0, vol, infer=True)
    assert all(v.dtype == dtype for v in captured["ctrls"].values())          # the network really ran in reduced precision
    assert signal.dtype == torch.float32 and hidden.dtype == dtype and torch.isfinite(signal).all()
    # the oracle on the controls' values (upcast to float32 exactly): the waveform is the reference DSP's for THOSE controls
    c = {k: v.detach().float().cpu().numpy() for k, v in captured["ctrls"].items()}
    f0n, un = f0.cpu().numpy(), u.cpu().numpy()
    nz = (un * np.float32(2) - np.float32(1)).astype(np.float32)
    if kind == "fast":
        ref = O.combsubfast_dsp(f0n, c["harmonic_magnitude"], c["harmonic_phase"], c["noise_magnitude"], nz, SR, HOP)
    elif kind == "superfast":
        ref = O.combsubsuperfast_dsp(f0n, c["harmonic_magnitude"], c["harmonic_phase"], c["noise_magnitude"], c["noise_phase"],
                                     gz.cpu().numpy(), SR, HOP, 2048)
    elif kind == "combsub":
        ref = O.combsub_dsp(f0n, c["group_delay"], c["harmonic_magnitude"], c["noise_magnitude"], nz, SR, HOP)
    else:
        ref = O.sins_dsp(f0n, c["amplitudes"], c["group_delay"], c["noise_magnitude"], nz, SR, HOP)
    e = rms(signal.cpu().numpy() - ref["signal"])
    assert e <= 1e-5 * rms(ref["signal"]) and e <= 1e-4, (e, rms(ref["signal"]))
    # training under autocast (solver_new.py:132-147): gradients reach the float32 parameters through the `dtype` controls
    with _autocast(dev, dtype):
        s2, _, _ = m(units, f0, vol, infer=False)
        loss = (s2.float() ** 2).mean()
    assert s2.dtype == torch.float32 and s2.requires_grad
    loss.backward()
    grads = [p.grad for p in m.unit2ctrl.parameters()]
    assert all(g is not None and g.dtype == torch.float32 and torch.isfinite(g).all() for g in grads)
    assert any(float(g.abs().max()) > 0 for g in grads)


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
@pytest.mark.parametrize("kind", ["fast", "superfast"])
def test_fast_modules_under_autocast_match_reference(dev, kind):
    """... and against the REFERENCE's own classes under the same context (bf16, where the reference runs on the CPU): same
    weights, inputs and noise -> same dtypes, same waveform (the reference on its CPU path under autocast('cpu'), the drop-in on
    ``dev`` under its device's autocast)."""
    rcore, rvoc = _import_reference()
    from ddsp_svc_amd import vocoder as V
    name = {"fast": "CombSubFast", "superfast": "CombSubSuperFast"}[kind]
    ref_cls = getattr(rvoc, "_reference_" + name, getattr(rvoc, name))
    torch.manual_seed(2)
    B, F, n_unit = 2, 9, 16
    args = (SR, HOP) if kind == "fast" else (SR, HOP, 2048)
    ref = ref_cls(*args, n_unit=n_unit, n_spk=1).eval()
    ours = getattr(V, name)(*args, n_unit=n_unit, n_spk=1).eval()
    ours.load_state_dict(ref.state_dict(), strict=True)
    ours = ours.to(dev)
    units, f0, vol, u = _inputs(B, F, n_unit, torch.device("cpu"), seed=6)
    gz = torch.randn(B, F * HOP, generator=torch.Generator().manual_seed(7))
    ud, gd = u.to(dev), gz.to(dev)
    with torch.no_grad():
        with torch.autocast("cpu", dtype=torch.bfloat16), mock.patch("torch.rand_like", side_effect=lambda t: u.reshape(t.shape)), \
                mock.patch("torch.randn_like", side_effect=lambda t: gz.reshape(t.shape)):
            r_sig, r_hid, _ = ref(units, f0, vol, infer=True)
        with _autocast(dev, torch.bfloat16), mock.patch("torch.rand", side_effect=lambda *a, **k: ud), \
                mock.patch("torch.randn", side_effect=lambda *a, **k: gd):
            o_sig, o_hid, _ = ours(units.to(dev), f0.to(dev), vol.to(dev), infer=True)
    # (`hidden` is the network's own output, passed through: bf16 under the CPU's autocast policy, float32 under the GPU's, where
    # the reference's Unit2Control ends in a layer norm that autocast keeps in float32)
    assert o_sig.dtype == r_sig.dtype == torch.float32 and r_hid.dtype == torch.bfloat16
    assert o_hid.dtype == (torch.bfloat16 if dev.type == "cpu" else o_hid.dtype) and o_hid.dtype in (torch.bfloat16, torch.float32)
    e = rms((o_sig.cpu() - r_sig).numpy())
    print("%s under bf16 autocast on %s against the reference under bf16 autocast on the CPU: rms error %.2e (rms %.2e)"
          % (name, dev, e, rms(r_sig.numpy())))
    # on the CPU both networks round alike; what is left (measured: 7e-5 relative) is the reference's own tail under autocast --
    # the ops its policy leaves in bf16 round intermediates that the kernels keep in float32 (the drop-in is the more accurate
    # of the two: test above).  On the GPU the bf16 network's GEMMs round differently from the CPU's and the controls differ
    # by bf16 ulps (0.4 %): the waveform follows them
    assert e <= (1e-3 if dev.type == "cpu" else 2e-2) * rms(r_sig.numpy()), (e, rms(r_sig.numpy()))
