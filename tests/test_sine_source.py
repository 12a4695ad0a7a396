"""Harmonic source of NSF-HiFiGAN (nsf_hifigan/models.py:101-204): the oracle against the
reference's own SourceModuleHnNSF output (fixture sinesrc.npz, random draws injected), the HIP kernel against both.
Tolerance: 2e-6 absolute on a tanh output of magnitude <= 0.2 (float32 sine of arguments up to ~600 rad)."""
import os
import sys
from unittest import mock
from unittest.mock import MagicMock

import numpy as np
import pytest
import torch

from oracle import ddsp_oracle as O
from tests.backends import BACKENDS, dev  # noqa: F401

SR, UPP = 44100, 512


def test_oracle_against_reference_source(golden_dir):
    g = np.load(os.path.join(golden_dir, "sinesrc.npz"))
    ref = O.sine_source(g["f0"], UPP, SR, g["weight"], g["bias"], g["rand_ini"], g["noise"])
    assert ref.shape == g["out"].shape
    assert np.abs(ref - g["out"]).max() <= 1e-7


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
def test_sine_source_golden(dev, golden_dir):
    from ddsp_svc_amd import nsf_source as S
    g = np.load(os.path.join(golden_dir, "sinesrc.npz"))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    out = S.sine_source(t(g["f0"]), UPP, SR, t(g["weight"]), t(g["bias"]), t(g["rand_ini"]), t(g["noise"]))
    assert np.abs(out.cpu().numpy() - g["out"]).max() <= 2e-6


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
@pytest.mark.parametrize("B,L,dim", [(1, 1, 9), (3, 300, 9), (2, 7, 1)])
def test_sine_source_shapes(dev, B, L, dim):
    """single frame, more frames than one scan chunk, the fundamental-only variant; unvoiced frames everywhere"""
    from ddsp_svc_amd import nsf_source as S
    rng = np.random.default_rng(L)
    f0 = O.synth_f0(B, L, SR, UPP, seed=L)[..., 0]
    f0[rng.random((B, L)) < 0.2] = 0.0
    w = rng.standard_normal(dim).astype(np.float32) * 0.3
    b = rng.standard_normal(1).astype(np.float32) * 0.1
    ri = rng.random(dim).astype(np.float32)
    ri[0] = 0
    nz = rng.standard_normal((B, L * UPP, dim)).astype(np.float32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    out = S.sine_source(t(f0), UPP, SR, t(w), t(b), t(ri), t(nz)).cpu().numpy()
    ref = O.sine_source(f0, UPP, SR, w, b, ri, nz)
    assert np.abs(out - ref).max() <= 2e-6


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
def test_source_module_dropin(dev, monkeypatch):
    from ddsp_svc_amd import nsf_source as S
    m = S.SourceModuleHnNSF(SR, harmonic_num=8).to(dev)
    assert set(m.state_dict()) == {"l_linear.weight", "l_linear.bias"}                  # models.py:195
    f0 = torch.from_numpy(O.synth_f0(2, 5, SR, UPP, seed=1)[..., 0]).to(dev)
    ri = torch.rand(1, 1, 9, generator=torch.Generator().manual_seed(2)).to(dev)
    nz = torch.randn(2, 5 * UPP, 9, generator=torch.Generator().manual_seed(3)).to(dev)
    monkeypatch.setattr(torch, "rand", lambda *a, **k: ri.clone())
    monkeypatch.setattr(torch, "randn", lambda *a, **k: nz)
    out = m(f0, UPP)
    assert out.shape == (2, 5 * UPP, 1)
    ri0 = ri.clone()
    ri0[..., 0] = 0
    ref = O.sine_source(f0.cpu().numpy(), UPP, SR, m.l_linear.weight.detach().cpu().numpy(),
                        m.l_linear.bias.detach().cpu().numpy(), ri0.cpu().numpy().reshape(-1), nz.cpu().numpy())
    assert np.abs(out.detach().cpu().numpy()[..., 0] - ref).max() <= 2e-6
    # grad enabled + trainable l_linear (models.py:198-204 is differentiable in the reference): Linear + tanh ran in torch
    # on the recovered per-harmonic waves; the fused inference path gives the same values
    assert out.requires_grad
    with torch.no_grad():
        fused = m(f0, UPP)
    assert not fused.requires_grad and np.abs(fused.cpu().numpy()[..., 0] - ref).max() <= 2e-6
    out.sum().backward()
    gw = m.l_linear.weight.grad.cpu().numpy().reshape(-1)
    assert np.isfinite(gw).all() and np.abs(gw).max() > 0
    with pytest.raises(RuntimeError):                                                # unsupported harmonic count
        S.sine_source(f0, UPP, SR, torch.zeros(4, device=dev), torch.zeros(1, device=dev), torch.zeros(4, device=dev),
                      torch.zeros(2, 5 * UPP, 4, device=dev))


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
def test_against_reference_source_module(dev):
    ref_root = os.environ.get("DDSP_REFERENCE_PATH", "/root/reference")
    if not os.path.isdir(os.path.join(ref_root, "nsf_hifigan")):
        pytest.skip("reference checkout not present (only in the build container)")
    if ref_root not in sys.path:
        sys.path.insert(0, ref_root)
    for name in ["librosa", "librosa.util", "librosa.filters", "librosa.core", "librosa.sequence", "soundfile", "torchaudio",
                 "torchaudio.transforms"]:
        sys.modules.setdefault(name, MagicMock())
    import nsf_hifigan.models as nm
    from ddsp_svc_amd import nsf_source as S
    torch.manual_seed(5)
    ref = nm.SourceModuleHnNSF(SR, harmonic_num=8)
    ours = S.SourceModuleHnNSF(SR, harmonic_num=8)
    ours.load_state_dict(ref.state_dict(), strict=True)
    f0 = torch.from_numpy(O.synth_f0(2, 12, SR, UPP, seed=6)[..., 0]).clone()
    f0[0, 3:5] = 0
    ri = torch.rand(1, 1, 9, generator=torch.Generator().manual_seed(7))
    nz = torch.randn(2, 12 * UPP, 9, generator=torch.Generator().manual_seed(8))
    with mock.patch("torch.rand", side_effect=lambda *a, **k: ri.clone()), \
            mock.patch("torch.randn_like", side_effect=lambda t: nz), \
            mock.patch("torch.randn", side_effect=lambda *a, **k: nz), torch.no_grad():
        want = ref(f0, UPP)                                  # the reference on its CPU path
    ours = ours.to(dev)
    rid, nzd = ri.to(dev), nz.to(dev)
    with mock.patch("torch.rand", side_effect=lambda *a, **k: rid.clone()), \
            mock.patch("torch.randn_like", side_effect=lambda t: nzd), \
            mock.patch("torch.randn", side_effect=lambda *a, **k: nzd), torch.no_grad():
        got = ours(f0.to(dev), UPP).cpu()
    assert got.shape == want.shape
    print("SourceModuleHnNSF on %s against the reference's CPU path: max abs error %.2e" % (dev, float((got - want).abs().max())))
    assert (got - want).abs().max() <= 2e-6
