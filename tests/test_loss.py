"""Spectral loss of the training loop (ddsp/loss.py:9-54): the oracle against the reference's own
SSSLoss / RSSLoss values and autograd gradients (fixture sssloss.npz; torchaudio's Spectrogram replaced by a torch.stft
stand-in, see make_golden.py), the fused HIP kernels behind ``ddsp_svc_amd.loss`` against both.

Tolerances: eps = 1e-7 sits at the float32 rounding floor of the spectra (|X| / ||w|| ~ 1, FFT noise ~ 1e-7), so
log(S) of the empty bins -- and with it the loss -- carries ~1e-5 relative float32 noise in the reference itself:
loss 5e-5 relative; gradient 2e-3 of its RMS (the sign(.) / S_pred term flips on those bins)."""
import os

import numpy as np
import pytest
import torch

from oracle import ddsp_oracle as O
from tests.backends import BACKENDS, dev  # noqa: F401

LOSS_RTOL = 5e-5
GRAD_RTOL = 2e-3


def _rel_rms(a, b):
    return float(np.sqrt(np.mean((np.asarray(a, np.float64) - b) ** 2)) / np.sqrt(np.mean(np.asarray(b, np.float64) ** 2)))


def test_oracle_against_reference_loss(golden_dir):
    g = np.load(os.path.join(golden_dir, "sssloss.npz"))
    for i, (n_fft, alpha, overlap) in enumerate(g["cases"]):
        loss = O.sss_loss(g["x_true"], g["x_pred"], int(n_fft), alpha, overlap)
        grad = O.sss_loss_backward(g["x_true"], g["x_pred"], int(n_fft), alpha, overlap)
        assert abs(loss - float(g[f"loss{i}"])) <= LOSS_RTOL * abs(loss)
        assert _rel_rms(g[f"grad{i}"], grad) <= GRAD_RTOL
    rss = np.mean([O.sss_loss(g["x_true"], g["x_pred"], int(n)) for n in g["rss_sizes"]])
    assert abs(rss - float(g["rss_loss"])) <= LOSS_RTOL * rss


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
def test_sss_loss_golden(dev, golden_dir):
    from ddsp_svc_amd import loss as L
    g = np.load(os.path.join(golden_dir, "sssloss.npz"))
    xt = torch.from_numpy(g["x_true"]).to(dev)
    for i, (n_fft, alpha, overlap) in enumerate(g["cases"]):
        f = L.SSSLoss(int(n_fft), float(alpha), float(overlap)).to(dev)
        xp = torch.from_numpy(g["x_pred"]).to(dev).requires_grad_(True)
        loss = f(xt, xp)
        loss.backward()
        want = float(g[f"loss{i}"])
        assert loss.shape == () and abs(float(loss.detach()) - want) <= LOSS_RTOL * want
        assert _rel_rms(xp.grad.cpu().numpy(), g[f"grad{i}"].astype(np.float64)) <= GRAD_RTOL


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
def test_rss_loss_golden(dev, golden_dir, monkeypatch):
    from ddsp_svc_amd import loss as L
    g = np.load(os.path.join(golden_dir, "sssloss.npz"))
    rss = L.RSSLoss(256, 300, 4, device=dev)
    sizes = torch.from_numpy(g["rss_sizes"])
    monkeypatch.setattr(torch, "randint", lambda *a, **k: sizes)
    xt = torch.from_numpy(g["x_true"]).to(dev)
    xp = torch.from_numpy(g["x_pred"]).to(dev).requires_grad_(True)
    loss = rss(xp, xt)                                                    # loss.py:46: (x_pred, x_true)
    loss.backward()
    assert abs(float(loss.detach()) - float(g["rss_loss"])) <= LOSS_RTOL * float(g["rss_loss"])
    assert _rel_rms(xp.grad.cpu().numpy(), g["rss_grad"].astype(np.float64)) <= GRAD_RTOL
    assert sorted(rss.lossdict) == sorted(set(int(n) for n in g["rss_sizes"]))


def _eager(xt, xp, n_fft, hop, alpha, eps):
    """loss.py:22-31 composed from torch ops in float64 (both gradients through autograd)."""
    w = torch.hann_window(n_fft, dtype=torch.float64)
    sp = lambda x: torch.stft(x, n_fft, hop_length=hop, win_length=n_fft, window=w, center=False,
                              return_complex=True).abs() / w.pow(2).sum().sqrt() + eps
    St, Sp = sp(xt), sp(xp)
    conv = torch.mean(torch.linalg.norm(St - Sp, dim=(1, 2)) / torch.linalg.norm(St + Sp, dim=(1, 2)))
    return conv + alpha * torch.nn.functional.l1_loss(St.log(), Sp.log())


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
@pytest.mark.parametrize("B,T,n_fft,overlap", [(1, 300, 300, 0.0), (5, 4096, 64, 0.5), (2, 20000, 1531, 0.75)])
def test_both_gradients(dev, B, T, n_fft, overlap):
    """single frame; many small frames; a prime transform size.  eps raised to 1e-3 so float32 is well conditioned and
    the comparison with the float64 composition can be tight."""
    from ddsp_svc_amd import loss as L
    rng = np.random.default_rng(T)
    a = (rng.standard_normal((B, T)) * 0.1).astype(np.float32)
    b = (a * 0.7 + rng.standard_normal((B, T)) * 0.05).astype(np.float32)
    f = L.SSSLoss(n_fft, 0.8, overlap, eps=1e-3).to(dev)
    xt = torch.from_numpy(a).to(dev).requires_grad_(True)
    xp = torch.from_numpy(b).to(dev).requires_grad_(True)
    loss = f(xt, xp)
    (2.5 * loss).backward()
    rt = torch.from_numpy(a).double().requires_grad_(True)
    rp = torch.from_numpy(b).double().requires_grad_(True)
    ref = _eager(rt, rp, n_fft, f.hop_length, 0.8, 1e-3)
    (2.5 * ref).backward()
    assert abs(float(loss.detach()) - float(ref.detach())) <= 2e-6 * float(ref.detach())
    assert _rel_rms(xp.grad.cpu().numpy(), rp.grad.numpy()) <= 2e-5
    assert _rel_rms(xt.grad.cpu().numpy(), rt.grad.numpy()) <= 2e-5


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
def test_loss_edge_cases(dev):
    from ddsp_svc_amd import loss as L
    f = L.SSSLoss(128, 1.0, 0.0).to(dev)
    x = torch.randn(2, 1000, generator=torch.Generator().manual_seed(1)).to(dev)
    assert float(f(x, x)) == 0.0                                          # identical signals
    z = torch.zeros(2, 1000, device=x.device, requires_grad=True)
    loss = f(x, z)                                                        # silent prediction: S_pred = eps everywhere
    loss.backward()
    assert np.isfinite(float(loss.detach())) and float(z.grad.abs().max()) == 0.0  # d|X| at the origin is 0 (as autograd)
    with pytest.raises(ValueError):
        f(x, x[:, :900])
    with pytest.raises(RuntimeError):                                     # signal shorter than one frame (torch.stft)
        f(x[:, :100], x[:, :100])


@pytest.mark.gpu
def test_full_size_against_eager_composition():
    """B = 32 x 10 s, the sizes RSSLoss draws from (256 .. 2047): the fused kernels against the eager float32
    composition of loss.py:22-31 on the same device."""
    from ddsp_svc_amd import loss as L
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(5)
    xt = (torch.randn(32, 441344, generator=g) * 0.1).to(dev)
    xp = (xt * 0.9 + 0.02 * torch.randn(32, 441344, generator=g).to(dev)).requires_grad_(True)
    for n_fft in (256, 1153, 2047):
        f = L.SSSLoss(n_fft).to(dev)
        loss = f(xt, xp)
        grad, = torch.autograd.grad(loss, xp)
        w = f.spec.window
        sp = lambda x: torch.stft(x, n_fft, hop_length=n_fft, win_length=n_fft, window=w, center=False,
                                  return_complex=True).abs() / w.pow(2).sum().sqrt() + 1e-7
        St, Sp = sp(xt), sp(xp)
        ref = torch.mean(torch.linalg.norm(St - Sp, dim=(1, 2)) / torch.linalg.norm(St + Sp, dim=(1, 2))) \
            + torch.nn.functional.l1_loss(St.log(), Sp.log())
        rgrad, = torch.autograd.grad(ref, xp)
        assert abs(float(loss.detach()) - float(ref.detach())) <= 1e-5 * float(ref.detach())
        assert float((grad - rgrad).pow(2).mean().sqrt() / rgrad.pow(2).mean().sqrt()) <= 1e-4
