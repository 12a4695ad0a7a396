"""BASELINE cfg 5's seam (main_diff.py:356-359,378: DDSP synthesiser -> log-mel -> sampler -> NSF-HiFiGAN) around the
HIP kernels, with stand-in networks (tools/standins.py: the reference's networks, third-party dependencies and checkpoints
cannot travel to the GPU box):

  * the chain run through the drop-in MODULES equals the same chain run op by op through the functional entry points
    (same seeds, same draw order) -- small on the emulator, B = 64 x 10 s on the MI355X;
  * ``patch_reference()`` against a FAKE reference package (module objects that bind names the way the reference's
    modules do: ``ddsp.vocoder`` imports the ddsp.core functions by name, ``diffusion.vocoder`` binds CombSubSuperFast
    at import): the rebinding logic and the dispatch rules run on the GPU box, where the real checkout is absent.
    The REAL reference goes through the same code in tests/test_modules.py (build container, emulator).
"""
import sys
import types

import numpy as np
import pytest
import torch

from tests.backends import BACKENDS, dev  # noqa: F401

SR, HOP = 44100, 512


def _bench_f0(B, F, seed):
    rng = np.random.default_rng(seed)
    base = rng.uniform(100.0, 400.0, size=(B, 1))
    t = np.arange(F)[None, :] * HOP / SR
    f0 = base * 2.0 ** (0.5 * np.sin(2 * np.pi * 5.5 * t + rng.uniform(0, 6.28, size=(B, 1))) / 12.0)
    return torch.from_numpy(f0.astype(np.float32))[:, :, None]


def _run_seam(device, B, F, n_unit):
    from tools.standins import CascadeSeam
    torch.manual_seed(0)
    seam = CascadeSeam(SR, HOP, n_unit=n_unit).to(device).eval()
    g = torch.Generator().manual_seed(1)
    units = torch.randn(B, F, n_unit, generator=g).to(device)
    vol = (torch.rand(B, F, 1, generator=g) * 0.1).to(device)
    f0 = _bench_f0(B, F, 2).to(device)
    f0[0, F // 3:F // 2] = 0.0                                   # an unvoiced stretch for the NSF source's uv gate (f0 > 0 elsewhere)
    f0c = torch.where(f0 > 0, f0, torch.full_like(f0, 80.0))     # the DDSP stage always gets interpolated, positive f0
    torch.manual_seed(77)
    a = seam(units, f0c, vol)
    torch.manual_seed(77)
    b = seam.op_by_op(units, f0c, vol)
    T = F * HOP
    assert a[0].shape == (B, T) and a[1].shape == (B, T) and a[2].shape == (B, F, 128)
    for x, y in zip(a, b):
        assert torch.isfinite(x).all()
        assert torch.equal(x, y)
    assert float(a[1].abs().max()) > 1e-4 and float(a[0].abs().max()) > 1e-6
    return a


@pytest.mark.parametrize("dev", ["emu"], indirect=True)
def test_seam_small_emulated(dev):
    _run_seam(dev, 2, 9, 16)


@pytest.mark.gpu
def test_seam_cfg5_gpu():
    """cfg 5 at its BASELINE size: B = 64 utterances x 10 s on one MI355X, one stream"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    wav, ddsp_wav, ddsp_mel = _run_seam(torch.device("cuda:0"), 64, 862, 768)
    assert wav.shape == (64, 862 * HOP)


# ---- patch_reference() against a fake reference package ---------------------------------------------------------------
def _fake_reference():
    """module objects shaped like the reference's import graph (only names and binding behaviour; the bodies are torch
    one-liners of this test, the real code stays in /root/reference)"""
    from tools.standins import StandInUnit2Control
    import torch.nn.functional as Fnn
    mods = {}

    def mod(name):
        m = types.ModuleType(name)
        mods[name] = m
        return m
    ddsp, core, voc, u2c = mod("ddsp"), mod("ddsp.core"), mod("ddsp.vocoder"), mod("ddsp.unit2control")
    dif, dvoc = mod("diffusion"), mod("diffusion.vocoder")
    ddsp.__path__, dif.__path__ = [], []

    def upsample(signal, factor):
        s = signal.permute(0, 2, 1)
        s = Fnn.interpolate(torch.cat((s, s[:, :, -1:]), 2), size=s.shape[-1] * factor + 1, mode="linear", align_corners=True)
        return s[:, :, :-1].permute(0, 2, 1)
    core.upsample = upsample
    for n in ("remove_above_fmax", "frequency_filter", "fft_convolve", "frequency_impulse_response",
              "apply_window_to_impulse_response", "apply_dynamic_window_to_impulse_response"):
        setattr(core, n, (lambda nn: (lambda *a, **k: ("reference", nn)))(n))
    u2c.Unit2Control = StandInUnit2Control

    class RefModel(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

        def forward(self, *a, **k):
            return "reference forward"
    for n in ("Sins", "CombSub", "CombSubFast", "CombSubSuperFast"):
        setattr(voc, n, type(n, (RefModel,), {}))
    voc.upsample, voc.remove_above_fmax, voc.frequency_filter = core.upsample, core.remove_above_fmax, core.frequency_filter  # vocoder.py:16
    dvoc.CombSubFast, dvoc.CombSubSuperFast = voc.CombSubFast, voc.CombSubSuperFast                                       # diffusion/vocoder.py:13
    ddsp.core, ddsp.vocoder, ddsp.unit2control, dif.vocoder = core, voc, u2c, dvoc
    # the scripts that bind `upsample` by name (flask_api.py:12, flask_api_diff.py:12, gui_diff.py:10, gui_reflow.py:8,
    # main_reflow.py:13: `from ddsp.core import upsample`), one of them under an alias
    for n in ("flask_api", "flask_api_diff", "gui_diff", "gui_reflow", "main_reflow"):
        mod(n).upsample = core.upsample
    mods["gui_reflow"].core_upsample = core.upsample
    return mods


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
def test_patch_reference_rebinding_with_standins(dev, monkeypatch):
    mods = _fake_reference()
    for name in mods:
        monkeypatch.delitem(sys.modules, name, raising=False)
    for name, m in mods.items():
        monkeypatch.setitem(sys.modules, name, m)
    from ddsp_svc_amd import vocoder as V
    core, voc, dvoc = mods["ddsp.core"], mods["ddsp.vocoder"], mods["diffusion.vocoder"]
    ref_super = voc.CombSubSuperFast
    try:
        V.patch_reference()
        assert dvoc.CombSubSuperFast is V.CombSubSuperFast and voc.CombSub is V.CombSub       # names bound at import are rebound
        assert voc.upsample is core.upsample and core.upsample is not core._reference_upsample
        # ... wherever they are: no list of importing modules, the originals are found by identity over sys.modules
        for n in ("flask_api", "flask_api_diff", "gui_diff", "gui_reflow", "main_reflow"):
            assert mods[n].upsample is core.upsample, n
        assert mods["gui_reflow"].core_upsample is core.upsample
        # the cascade constructs its DDSP stage by name (diffusion/vocoder.py:282) -> the HIP-backed class, built on the
        # Unit2Control the (fake) checkout provides
        stage = dvoc.CombSubSuperFast(SR, HOP, 2048, 16, 1).to(dev).eval()
        assert type(stage) is V.CombSubSuperFast and type(stage.unit2ctrl).__name__ == "StandInUnit2Control"
        B, F = 2, 7
        g = torch.Generator().manual_seed(3)
        units, vol = torch.randn(B, F, 16, generator=g).to(dev), (torch.rand(B, F, 1, generator=g) * 0.1).to(dev)
        f0 = _bench_f0(B, F, 4).to(dev)
        with torch.no_grad():
            wav, hidden, _ = stage(units, f0, vol, infer=True)
        assert wav.shape == (B, F * HOP) and torch.isfinite(wav).all() and hidden.shape == (B, F, 256)
        # dispatch of the patched ddsp.core functions
        sig = torch.rand(1, 5, 3, generator=g)
        if dev.type == "cuda":
            out = core.upsample(sig.to(dev), 64)                  # GPU float32 -> HIP kernel, bit-equal to the interpolation
            assert torch.equal(out.cpu(), core._reference_upsample(sig, 64))
            taps = core.frequency_impulse_response(torch.rand(1, 4, 33, device=dev))
            assert torch.is_tensor(taps) and taps.shape == (1, 4, 64)
            assert core.fft_convolve(torch.rand(1, 64, dtype=torch.float64, device=dev), torch.rand(1, 1, 8, device=dev)) \
                == ("reference", "fft_convolve")                   # a dtype the kernels do not take keeps the reference's code
            stage_cpu_in = stage(units.cpu(), f0.cpu(), vol.cpu())
            assert stage_cpu_in == "reference forward"            # host tensors -> the reference's forward
        assert torch.equal(core.upsample(sig, 64), core._reference_upsample(sig, 64))          # host tensors -> reference
        assert core.frequency_filter(torch.rand(1, 8), torch.rand(1, 1, 3)) == ("reference", "frequency_filter")
    finally:
        V.unpatch_reference()
    assert dvoc.CombSubSuperFast is ref_super and not hasattr(core, "_reference_upsample")
    assert voc.upsample is core.upsample
    assert all(mods[n].upsample is core.upsample for n in ("flask_api", "gui_diff", "main_reflow"))
    assert mods["gui_reflow"].core_upsample is core.upsample
    from ddsp_svc_amd import vocoder as V2
    assert V2.CombSub.__module__ == "ddsp_svc_amd.vocoder" and V2.CombSub is V.CombSub      # this package keeps its own names
