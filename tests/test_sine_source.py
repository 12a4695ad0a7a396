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


# ---- the opt-in in-kernel standard-normal draw (models.py:168's randn_like: [B, L*upp, dim] floats that need not exist) -----------
@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
@pytest.mark.parametrize("B,T,dim,seed,offset", [(1, 512, 9, 1, 0), (3, 700, 9, 0x1234567890ABCDEF, 7), (2, 40, 1, 5, (1 << 40) + 3)])
def test_normal_noise_matches_oracle(dev, B, T, dim, seed, offset):
    """the written-out draw against the float64 restatement (Philox4x32-10 is pinned to its known answers in test_noise_rng.py):
    the hardware log2 / sine / cosine are within 2e-6 of it on values up to 5.77"""
    from ddsp_svc_amd import nsf_source as S
    z = S.normal_noise(B, T, dim, seed, offset, dev).cpu().numpy()
    ref = O.normal_noise(B, T, dim, seed, offset)
    assert z.shape == (B, T, dim) and z.dtype == np.float32
    assert np.abs(z - ref).max() <= 2e-6 * max(1.0, float(np.abs(ref).max()))


def test_normal_noise_statistics():
    z = O.normal_noise(3, 1 << 15, 9, 20260923, 0)
    n = z.size
    assert abs(z.mean()) < 4 / np.sqrt(n) and abs(z.var() - 1.0) < 6 * np.sqrt(2.0 / n)
    assert abs(float((z ** 3).mean())) < 6 * np.sqrt(15.0 / n) and abs(float((z ** 4).mean()) - 3.0) < 6 * np.sqrt(96.0 / n)
    assert np.abs(z).max() <= 5.78                                  # u1 >= 2^-24
    # a 32-bin chi-square against the normal cdf
    from math import erf, sqrt
    edges = np.linspace(-3.2, 3.2, 31)
    cdf = np.array([0.5 * (1 + erf(e / sqrt(2))) for e in edges])
    p = np.diff(np.concatenate([[0.0], cdf, [1.0]]))
    counts = np.histogram(z, bins=np.concatenate([[-np.inf], edges, [np.inf]]))[0]
    chi2 = float(((counts - n * p) ** 2 / (n * p)).sum())
    assert chi2 < 31 + 5 * np.sqrt(2 * 31), chi2
    # independence: between the harmonics of a sample (incl. the two members of a Box-Muller pair and the two pairs of a
    # counter), between neighbouring samples, utterances and offsets
    zz = z.reshape(-1, 9)
    c = np.corrcoef(zz.T)
    assert np.abs(c - np.eye(9)).max() < 5 / np.sqrt(zz.shape[0])
    for lag in (1, 2, 512):
        r = float((z[:, lag:, 0] * z[:, :-lag, 0]).mean())
        assert abs(r) < 5 / np.sqrt(z.shape[0] * z.shape[1]), (lag, r)
    assert abs(float((z[0] * z[1]).mean())) < 5 / np.sqrt(z[0].size)
    v = O.normal_noise(1, 1 << 15, 9, 20260923, 1)
    assert abs(float((z[0] * v[0]).mean())) < 5 / np.sqrt(v.size)


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
@pytest.mark.parametrize("dim", [9, 1])
def test_sine_source_with_in_kernel_noise(dev, dim):
    """``noise=None`` is the same kernel with its noise generated instead of loaded: the output equals the call fed the
    written-out draw (bit for bit), is the reference's formula on those numbers, and the module opt-in advances the stream"""
    from ddsp_svc_amd import nsf_source as S
    B, L = 3, 11
    rng = np.random.default_rng(dim)
    f0 = O.synth_f0(B, L, SR, UPP, seed=4)[..., 0]
    f0[rng.random((B, L)) < 0.3] = 0.0
    w = rng.standard_normal(dim).astype(np.float32) * 0.3
    b = rng.standard_normal(1).astype(np.float32) * 0.1
    ri = rng.random(dim).astype(np.float32)
    ri[0] = 0
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    drawn = S.sine_source(t(f0), UPP, SR, t(w), t(b), t(ri), None, noise_seed=99, noise_offset=3)
    z = S.normal_noise(B, L * UPP, dim, 99, 3, dev)
    fed = S.sine_source(t(f0), UPP, SR, t(w), t(b), t(ri), z)
    assert torch.equal(drawn, fed)
    ref = O.sine_source(f0, UPP, SR, w, b, ri, z.cpu().numpy())
    assert np.abs(drawn.cpu().numpy() - ref).max() <= 2e-6
    with pytest.raises(ValueError):
        S.sine_source(t(f0), UPP, SR, t(w), t(b), t(ri), None)
    if dim == 9:
        m = S.SourceModuleHnNSF(SR, harmonic_num=8).to(dev)
        m.in_kernel_noise_seed = 7
        torch.manual_seed(0)
        a = m(t(f0), UPP)
        torch.manual_seed(0)                                      # the same rand_ini draw; the noise stream has advanced
        c = m(t(f0), UPP)
        assert a.shape == (B, L * UPP, 1) and not torch.equal(a, c) and m._noise_calls == 2
        # training the merge layer: the per-harmonic recovery uses ONE draw for all nine passes
        for p in m.l_linear.parameters():
            p.requires_grad_(True)
        y = m(t(f0), UPP)
        y.sum().backward()
        assert m.l_linear.weight.grad is not None and m._noise_calls == 3


def test_normal_and_uniform_draws_of_one_seed_are_different_words():
    """the normal draw's key carries a domain tag: for (seed, offset) with offset_hi = 0 the raw Philox words behind the first four
    harmonics of sample t are NOT the words behind the uniform draw's samples 4 t .. (they were)"""
    seed, T = 12345, 512
    ctr = np.zeros((T, 4), dtype=np.uint64)
    ctr[:, 0] = np.arange(T)
    plain = O.philox4x32_10(ctr, np.broadcast_to(np.array([seed, 0], dtype=np.uint64), (T, 2)))
    tagged = O.philox4x32_10(ctr, np.broadcast_to(np.array([seed, 0x4E4F524D], dtype=np.uint64), (T, 2)))
    assert not np.any(plain == tagged)
    z = O.normal_noise(1, T, 4, seed, 0)[0]                         # built from the tagged words ...
    u1 = ((tagged[:, 0].astype(np.uint64) >> np.uint64(8)) + np.uint64(1)).astype(np.float64) * 2.0 ** -24
    u2 = (tagged[:, 1].astype(np.uint64) >> np.uint64(8)).astype(np.float64) * 2.0 ** -24
    assert np.allclose(z[:, 0], np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2), atol=1e-12)
    u = O.uniform_noise(1, 512, seed, 0)[0]                          # ... the uniform draw from the plain ones
    assert np.allclose(u[:128], (plain[:128, 0].astype(np.uint64) >> np.uint64(8)).astype(np.float64) * 2.0 ** -24)
    with pytest.raises(ValueError):
        O.normal_noise(1, 8, 17, seed, 0)


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
def test_normal_noise_rejects_more_than_16_harmonics(dev):
    from ddsp_svc_amd import nsf_source as S
    with pytest.raises(RuntimeError):
        S.normal_noise(1, 8, 17, 1, 0, dev)
