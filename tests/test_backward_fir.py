"""Adjoints of the time-varying FIR (second part: ddsp/core.py:120-182 under autograd).
Oracle pinned against the reference's own autograd through ``ddsp.core.fft_convolve`` (build container only), HIP
kernel against the oracle: gradients <= 5e-6 relative RMS."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import ddsp_oracle as O
from tests.backends import BACKENDS, dev  # noqa: F401

HOP = 512


def rms(a):
    return float(np.sqrt(np.mean(np.square(np.asarray(a, dtype=np.float64)))))


def _case(B, F, N, seed):
    rng = np.random.default_rng(seed)
    x = (rng.random((B, F * HOP)) * 2 - 1).astype(np.float32)
    ir = (rng.standard_normal((B, F, N)) / np.sqrt(N) * rng.uniform(0.05, 2.0, size=(B, F, 1))).astype(np.float32)
    R = rng.standard_normal((B, F * HOP)).astype(np.float32)
    return x, ir, R


def test_oracle_backward_against_reference_autograd():
    ref_root = os.environ.get("DDSP_REFERENCE_PATH", "/root/reference")
    if not os.path.isdir(os.path.join(ref_root, "ddsp")):
        pytest.skip("reference checkout not present (only in the build container)")
    if ref_root not in sys.path:
        sys.path.insert(0, ref_root)
    import ddsp.core as rcore
    conv = getattr(rcore, "_reference_fft_convolve", rcore.fft_convolve)
    for B, F, N in ((2, 7, 510), (1, 4, 30), (1, 1, 510)):
        x, ir, R = _case(B, F, N, 10 * F + N)
        a = torch.from_numpy(x).requires_grad_(True)
        h = torch.from_numpy(ir).requires_grad_(True)
        (conv(a, h) * torch.from_numpy(R)).sum().backward()
        dx, dh = O.ltv_fir_backward(R, x, ir)
        assert rms(dx - a.grad.numpy()) <= 2e-6 * rms(dx)
        assert rms(dh - h.grad.numpy()) <= 2e-6 * rms(dh)


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
@pytest.mark.parametrize("B,F,N,run", [(2, 7, 510, 1000), (1, 8, 510, 1), (1, 1, 510, 1000), (2, 2, 30, 1000), (1, 5, 512, 2),
                                       (1, 13, 2, 3), (1, 12, 254, 1)])
def test_fft_convolve_backward(dev, B, F, N, run, monkeypatch):
    """odd / even block counts (the held last tap row), a single frame, the largest N, several runs per utterance
    (carry rebuilt by the warm-up pair)"""
    from ddsp_svc_amd import core
    monkeypatch.setenv("DDSP_HIP_BLK_RUN", str(run))
    x, ir, R = _case(B, F, N, 100 * F + N)
    t = lambda a: torch.from_numpy(a).to(dev)
    dx, dh = core.fft_convolve_backward(t(R), t(x), t(ir))
    rx, rh = O.ltv_fir_backward(R, x, ir)
    assert rms(dx.cpu().numpy() - rx) <= 5e-6 * rms(rx)
    assert rms(dh.cpu().numpy() - rh) <= 5e-6 * rms(rh)
    none, dh2 = core.fft_convolve_backward(t(R), t(x), t(ir), need_audio_grad=False)
    assert none is None and torch.equal(dh2, dh)


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
def test_fft_convolve_autograd(dev):
    from ddsp_svc_amd import core
    x, ir, R = _case(2, 6, 254, 3)
    a = torch.from_numpy(x).to(dev).requires_grad_(True)
    h = torch.from_numpy(ir).to(dev).requires_grad_(True)
    y = core.fft_convolve(a, h)
    assert y.requires_grad
    assert rms(y.detach().cpu().numpy() - O.ltv_fir_blockfft(x, ir)) <= 2e-6 * rms(O.ltv_fir_blockfft(x, ir))
    (y * torch.from_numpy(R).to(dev)).sum().backward()
    rx, rh = O.ltv_fir_backward(R, x, ir)
    assert rms(a.grad.cpu().numpy() - rx) <= 5e-6 * rms(rx)
    assert rms(h.grad.cpu().numpy() - rh) <= 5e-6 * rms(rh)
    # only the taps need a gradient: the input-gradient half of the kernel is skipped
    h2 = torch.from_numpy(ir).to(dev).requires_grad_(True)
    (core.fft_convolve(torch.from_numpy(x).to(dev), h2) * torch.from_numpy(R).to(dev)).sum().backward()
    assert rms(h2.grad.cpu().numpy() - rh) <= 5e-6 * rms(rh)
    with pytest.raises(RuntimeError):                       # shapes outside the hop-block form are refused loudly
        core.fft_convolve_backward(torch.zeros(1, 1024, device=dev), torch.zeros(1, 1024, device=dev),
                                   torch.zeros(1, 4, 30, device=dev))
