"""The rest of the functional boundary: the two window helpers of ddsp/core.py:185-251 on time-domain
taps, and autograd through the patched-in functions that the reference's own versions support (VERDICT r1 "missing" #5,
ADVICE r1: ``upsample`` / ``remove_above_fmax`` must not cut the graph; a windowed ``torch.complex(param, 0)`` response
must be differentiable).  Gradients are checked against torch's autograd through the CPU op chain of
oracle/aten_chain.py (itself pinned to the reference's outputs)."""
import os

import numpy as np
import pytest
import torch

from oracle import aten_chain as A
from tests.backends import BACKENDS, dev  # noqa: F401

SR, HOP = 44100, 512


def rms(a):
    return float(np.sqrt(np.mean(np.square(np.asarray(a, dtype=np.float64)))))


def N_(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
@pytest.mark.parametrize("n_mag", [65, 256, 257])
def test_window_helpers_golden(dev, golden_dir, n_mag):
    """apply_window_to_impulse_response / apply_dynamic_window_to_impulse_response on irfft output == the reference's
    frequency_impulse_response taps (core.py:259-266)"""
    from ddsp_svc_amd import core
    g = np.load(os.path.join(golden_dir, f"filter_n{n_mag}.npz"))
    mag = torch.from_numpy(g["mag"]) if "mag" in g else torch.exp(torch.from_numpy(g["ctrl"]))
    ir0 = torch.fft.irfft(torch.complex(mag, torch.zeros_like(mag))).to(dev)        # zero-phase taps, as core.py:259
    hw = torch.from_numpy(g["half_width"]).unsqueeze(-1).to(dev)
    got = core.apply_window_to_impulse_response(ir0)
    assert rms(N_(got) - g["ir_hann"]) <= 2e-6 * max(rms(g["ir_hann"]), 1e-3)
    got = core.apply_dynamic_window_to_impulse_response(ir0, hw)
    assert rms(N_(got) - g["ir_dyn"]) <= 2e-6 * max(rms(g["ir_dyn"]), 1e-3)
    with pytest.raises(AttributeError):
        core.apply_window_to_impulse_response(ir0, causal=True)                      # as the reference (core.py:204)


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
@pytest.mark.parametrize("n", [129, 128, 7])
def test_hann_window_odd_size(dev, n):
    """odd tap counts in Hann mode: the reference rolls the window by N//2, multiplies, and rolls the product by N//2 again
    (core.py:209-235), so for odd N the tap that lands at j carries hann[(j + 1) mod N], not hann[j]"""
    from ddsp_svc_amd import core
    g = torch.Generator().manual_seed(11)
    ir = torch.randn(2, 3, n, generator=g)

    def ref(x):                                                     # the reference's op chain, padding == 0 branch
        w = torch.hann_window(n).roll(n // 2, -1)
        return (x * w.unsqueeze(0)).roll(n // 2, -1)
    got = core.apply_window_to_impulse_response(ir.to(dev))
    assert rms(N_(got) - ref(ir).numpy()) <= 2e-6 * rms(ref(ir).numpy())


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
def test_dynamic_window_odd_size_and_grad(dev):
    """odd tap count (the reference allows 2*n_mag-1, core.py:240) and the differentiable composition"""
    from ddsp_svc_amd import core
    g = torch.Generator().manual_seed(3)
    ir = torch.randn(2, 5, 129, generator=g)
    hw = torch.rand(2, 5, 1, generator=g) * 100 + 20

    def ref(x):
        n = x.shape[-1]
        pos = torch.arange(-(n // 2), (n + 1) // 2, dtype=x.dtype) / hw
        pos[pos > 1] = 0
        return x.roll(n // 2, -1) * ((1 + torch.cos(np.pi * pos)) / 2)
    got = core.apply_dynamic_window_to_impulse_response(ir.to(dev), hw.to(dev))
    assert rms(N_(got) - ref(ir).numpy()) <= 2e-6 * rms(ref(ir).numpy())
    x1 = ir.clone().to(dev).requires_grad_(True)
    x2 = ir.clone().requires_grad_(True)
    R = torch.randn(2, 5, 129, generator=g)
    (core.apply_dynamic_window_to_impulse_response(x1, hw.to(dev)) * R.to(dev)).sum().backward()
    (ref(x2) * R).sum().backward()
    assert rms(N_(x1.grad) - x2.grad.numpy()) <= 2e-6 * rms(x2.grad.numpy())


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
def test_upsample_and_mask_autograd(dev):
    from ddsp_svc_amd import core
    g = torch.Generator().manual_seed(5)
    sig = torch.rand(2, 7, 3, generator=g)
    R = torch.randn(2, 7 * 64, 3, generator=g)
    a = sig.clone().to(dev).requires_grad_(True)
    b = sig.clone().requires_grad_(True)
    out = core.upsample(a, 64)
    assert out.requires_grad and torch.equal(out.detach().cpu(), A.to_sample_rate(sig, 64))
    (out * R.to(dev)).sum().backward()
    (A.to_sample_rate(b, 64) * R).sum().backward()
    assert rms(N_(a.grad) - b.grad.numpy()) <= 1e-6 * rms(b.grad.numpy())
    amps = torch.rand(2, 7, 40, generator=g)
    pitch = torch.rand(2, 7, 1, generator=g) * 1500 + 60
    a = amps.clone().to(dev).requires_grad_(True)
    out = core.remove_above_fmax(a, pitch.to(dev), 22050.0, 1)
    Rm = torch.randn(2, 7, 40, generator=g)
    (out * Rm.to(dev)).sum().backward()
    mask = ((pitch * torch.arange(1, 41) < 22050.0).float() + 1e-7)
    assert np.array_equal(N_(a.grad), (Rm * mask).numpy())


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
@pytest.mark.parametrize("dynamic", [False, True])
def test_windowed_complex_response_grad(dev, dynamic):
    """frequency_filter(audio, torch.complex(param, 0), hann_window=True[, half_width]) under autograd -- how the
    reference's CombSub / Sins call it (vocoder.py:606,849,857): gradient w.r.t. the real parameter"""
    from ddsp_svc_amd import core
    g = torch.Generator().manual_seed(9)
    B, F, n = 2, 6, 65
    audio = torch.rand(B, F * HOP, generator=g) * 2 - 1
    p = torch.randn(B, F, n, generator=g)
    hw = (torch.rand(B, F, 1, generator=g) * 60 + 10) if dynamic else None
    R = torch.randn(B, F * HOP, generator=g)
    p1 = p.clone().to(dev).requires_grad_(True)
    m1 = torch.exp(p1)
    y1 = core.frequency_filter(audio.to(dev), torch.complex(m1, torch.zeros_like(m1)), hann_window=True,
                               half_width_frames=None if hw is None else hw.to(dev))
    (y1 * R.to(dev)).sum().backward()
    p2 = p.clone().requires_grad_(True)
    m2 = torch.exp(p2)
    y2 = A.filter_with_response(audio, torch.complex(m2, torch.zeros_like(m2)), True, hw)
    (y2 * R).sum().backward()
    assert rms(N_(y1) - y2.detach().numpy()) <= 2e-6 * rms(y2.detach().numpy())
    assert rms(N_(p1.grad) - p2.grad.numpy()) <= 2e-5 * rms(p2.grad.numpy())


@pytest.mark.gpu
def test_two_threads_two_streams_gpu():
    """the synthesiser tails keep no per-device state: two host threads on two torch streams of one device, calling
    concurrently (the GUI runs the model in its audio callback thread, gui.py:393), get bit-identical results to the
    serial calls"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import threading
    from ddsp_svc_amd import synth
    from oracle import ddsp_oracle as O
    dev = torch.device("cuda:0")

    def make(seed, B, F):
        f0 = torch.from_numpy(O.synth_f0(B, F, seed=seed)).to(dev)
        c = [torch.from_numpy(a).to(dev) for a in O.synth_controls(B, F, [256, 256, 256], seed=seed + 1)]
        nz = torch.from_numpy(O.synth_noise(B, F * HOP, seed=seed + 2)).to(dev)
        return f0, c, nz

    def run(inp):
        f0, c, nz = inp
        st = synth.phase(f0, SR, HOP)
        return synth.combsub_synth(f0, st, c[0], c[1], c[2], nz, SR, HOP, want_components=False)[0]
    jobs = [make(10, 8, 600), make(20, 6, 700)]            # >= 4096 frames each: the noise branch forks onto the aux stream
    serial = [run(j).clone() for j in jobs]
    torch.cuda.synchronize()
    results, errors = [None, None], []

    def worker(i):
        try:
            s = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(s):
                for _ in range(12):
                    out = run(jobs[i])
                s.synchronize()
            results[i] = out
        except Exception as e:                              # pragma: no cover
            errors.append(e)
    for _ in range(3):
        ths = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        assert not errors, errors
        for i in range(2):
            assert torch.equal(results[i], serial[i]), i
