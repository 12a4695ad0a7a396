"""Seeded random shapes through the fused tails (batch, frame count incl. 1 and odd counts, bin counts on and off the
256-column table, harmonic counts, run splits of the hop-block FIR / the spectral filter) against the oracle: the
edge cases nobody thought of.  Tolerance: 2e-5 relative RMS per output (the tails sit at 1e-6 ... 5e-6)."""
import numpy as np
import pytest
import torch

from oracle import ddsp_oracle as O
from tests.backends import BACKENDS, dev  # noqa: F401

SR, HOP = 44100, 512


def _rms(a):
    return float(np.sqrt(np.mean(np.asarray(a, np.float64) ** 2)))


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
@pytest.mark.parametrize("seed", range(8))
def test_random_harmonic_plus_noise_tails(dev, seed, knobs):
    from ddsp_svc_amd import synth
    rng = np.random.default_rng(1000 + seed)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    for _ in range(3):
        B, F = int(rng.integers(1, 4)), int(rng.integers(1, 45))
        n = int(rng.choice([256, 256, 129, 65, 33, 252, 200]))
        kind = rng.choice(["combsub", "sins"])
        run = int(rng.choice([0, 1, 2, 3, 5]))
        knobs("BLK_RUN", run or 0)
        H = int(rng.choice([256, 128, 40, 17])) if kind == "sins" else n
        f0 = O.synth_f0(B, F, SR, HOP, seed=int(rng.integers(1 << 30)))
        ctrls = [rng.standard_normal((B, F, s)).astype(np.float32) * 0.7 for s in (H, n, n)]
        noise = rng.random((B, F * HOP)).astype(np.float32) * 2 - 1
        st = synth.phase(t(f0), SR, HOP)
        fn = synth.combsub_synth if kind == "combsub" else synth.sins_synth
        ofn = O.combsub_dsp if kind == "combsub" else O.sins_dsp
        out = fn(t(f0), st, t(ctrls[0]), t(ctrls[1]), t(ctrls[2]), t(noise), SR, HOP)
        ref = ofn(f0, ctrls[0], ctrls[1], ctrls[2], noise, SR, HOP)
        for got, key in zip(out, ("signal", "harmonic", "noise")):
            e, r = _rms(got.cpu().numpy() - ref[key]), _rms(ref[key])
            assert np.isfinite(e) and e <= 2e-5 * max(r, 1e-9), (kind, B, F, n, H, run, key, e, r)


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
@pytest.mark.parametrize("seed", range(4))
def test_random_spectral_tails(dev, seed, knobs):
    from ddsp_svc_amd import synth
    rng = np.random.default_rng(2000 + seed)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    for _ in range(2):
        B, F = int(rng.integers(1, 3)), int(rng.integers(1, 30))
        kind = rng.choice(["fast", "super"])
        run = int(rng.choice([0, 1, 2, 3]))
        knobs("STFT_RUN", run or 0)
        f0 = O.synth_f0(B, F, SR, HOP, seed=int(rng.integers(1 << 30)))
        if kind == "fast":
            c = [rng.standard_normal((B, F, 513)).astype(np.float32) * 0.5 for _ in range(3)]
            noise = rng.random((B, F * HOP)).astype(np.float32) * 2 - 1
            w = torch.sqrt(torch.hann_window(1024)).to(dev)
            st = synth.phase(t(f0), SR, HOP)
            got = synth.combsubfast_synth(t(f0), st, t(c[0]), t(c[1]), t(c[2]), t(noise), w, SR, HOP)
            ref = O.combsubfast_dsp(f0, c[0], c[1], c[2], noise, SR, HOP)
        else:
            c = [rng.standard_normal((B, F, 1025)).astype(np.float32) * 0.5 for _ in range(4)]
            noise = rng.standard_normal((B, F * HOP)).astype(np.float32)
            w = torch.hann_window(2048).to(dev)
            st = synth.fast_source(t(f0), SR, HOP)
            got = synth.combsubsuperfast_synth(t(f0), st, t(c[0]), t(c[1]), t(c[2]), t(c[3]), t(noise), w, SR, HOP)
            ref = O.combsubsuperfast_dsp(f0, c[0], c[1], c[2], c[3], noise, SR, HOP)
        ref = ref["signal"] if isinstance(ref, dict) else ref
        e, r = _rms(got.cpu().numpy() - ref), _rms(ref)
        assert np.isfinite(e) and e <= 2e-5 * max(r, 1e-9), (kind, B, F, run, e, r)


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
@pytest.mark.parametrize("seed", range(4))
def test_random_scale_loss(dev, seed, knobs):
    """RSSLoss as the reference draws it (loss.py:47) on random batch sizes and lengths -- exactly one frame, odd frame
    counts, a tail shorter than a frame, overlapping frames, strided inputs, several rounds of workgroups -- against the
    oracle's float64
    loss and analytic gradient.  eps = 1e-5 keeps the comparison away from the sign flips of bins at the rounding floor
    (tests/test_loss.py has the eps = 1e-7 cases)."""
    from ddsp_svc_amd import loss as L
    rng = np.random.default_rng(3000 + seed)
    for _ in range(2):
        B = int(rng.integers(1, 4))
        sizes = [int(s) for s in rng.integers(2, 2048, 3)]
        T = int(max(sizes) * rng.choice([1, 1, 2, 3]) + rng.integers(0, 40))
        knobs("CZT_ROUNDS", int(rng.choice([0, 1, 3])))
        a = (rng.standard_normal((B, 2 * T)) * 0.1).astype(np.float32)
        b = (a * 0.6 + rng.standard_normal((B, 2 * T)) * 0.05).astype(np.float32)
        strided = bool(rng.integers(0, 2))
        xt = torch.from_numpy(a).to(dev)[:, ::2] if strided else torch.from_numpy(np.ascontiguousarray(a[:, ::2])).to(dev)
        xp = (torch.from_numpy(b).to(dev)[:, ::2] if strided else torch.from_numpy(np.ascontiguousarray(b[:, ::2])).to(dev))
        xp = xp.detach().requires_grad_(True)
        overlap = float(rng.choice([0.0, 0.0, 0.25, 0.5, 0.75]))
        if overlap:
            sizes = [max(s, 8) for s in sizes]
        rss = L.RSSLoss(2, 2048, len(sizes), overlap=overlap, eps=1e-5, device=dev)
        real = torch.randint
        torch.randint = lambda *args, **kw: torch.tensor(sizes)
        try:
            value = rss(xp, xt)
        finally:
            torch.randint = real
        assert "RandomScaleWaveLoss" in type(value.grad_fn).__name__
        grad, = torch.autograd.grad(value, xp)
        at, bt = a[:, ::2], b[:, ::2]
        want = np.mean([O.sss_loss(at, bt, n, 1.0, overlap, eps=1e-5) for n in sizes])
        gwant = np.mean([O.sss_loss_backward(at, bt, n, 1.0, overlap, eps=1e-5) for n in sizes], axis=0)
        assert abs(float(value.detach()) - want) <= 2e-5 * want, (B, T, sizes, strided, overlap)
        assert _rms(grad.cpu().numpy() - gwant) <= 2e-4 * _rms(gwant), (B, T, sizes, strided, overlap)


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
@pytest.mark.parametrize("seed", range(6))
def test_random_split_calls_and_adjoints(dev, seed, knobs):
    """round 5's paths under random shapes: a call split into sub-batches (random LANE_ROWS, one or two lanes, in-kernel or
    supplied noise) against the unsplit call bit for bit; the filter's adjoint at random tap counts up to 1022 and random run
    lengths (hop-block form, per-frame 2048-point form) and the sinusoid bank's at random hops against the oracle's float64"""
    from ddsp_svc_amd import core, synth
    rng = np.random.default_rng(3000 + seed)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    # ---- split against unsplit
    B, F = int(rng.integers(2, 7)), int(rng.integers(2, 30))
    kind = rng.choice(["combsub", "sins"])
    n = int(rng.choice([256, 129, 65]))
    H = int(rng.choice([64, 17])) if kind == "sins" else n
    f0 = t(O.synth_f0(B, F, SR, HOP, seed=int(rng.integers(1 << 30))))
    c = [t(rng.standard_normal((B, F, s)).astype(np.float32) * 0.7) for s in (H, n, n)]
    u = t(rng.random((B, F * HOP)).astype(np.float32))
    in_kernel = bool(rng.integers(2)) and n <= 257
    fn = synth.combsub_synth if kind == "combsub" else synth.sins_synth

    def call():
        st = synth.phase(f0, SR, HOP)
        if in_kernel:
            return fn(f0, st, c[0], c[1], c[2], None, SR, HOP, noise_seed=seed + 1, noise_offset=3)
        return fn(f0, st, c[0], c[1], c[2], u, SR, HOP, noise_is_u01=True)
    knobs("LANE_ROWS", 1)
    whole = call()
    knobs("LANE_ROWS", int(rng.integers(2, B * F)))
    knobs("LANES", int(rng.choice([0, 1])))
    split = call()
    for a, b in zip(whole, split):
        assert torch.equal(a, b), (kind, B, F, n, H, in_kernel)
    knobs("LANE_ROWS", 0)
    knobs("LANES", 0)
    # ---- the filter's adjoints
    B, F = int(rng.integers(1, 3)), int(rng.integers(1, 14))
    N = 2 * int(rng.integers(1, 512))
    knobs("BLK_RUN", int(rng.choice([0, 1, 2, 3])))
    knobs("FFT_RUN", int(rng.choice([0, 1, 2, 3])))
    x = (rng.random((B, F * HOP)) * 2 - 1).astype(np.float32)
    ir = (rng.standard_normal((B, F, N)) / np.sqrt(N)).astype(np.float32)
    R = rng.standard_normal((B, F * HOP)).astype(np.float32)
    dx, dh = core.fft_convolve_backward(t(R), t(x), t(ir))
    rx, rh = O.ltv_fir_backward(R, x, ir)
    assert _rms(dx.cpu().numpy() - rx) <= 5e-6 * _rms(rx) and _rms(dh.cpu().numpy() - rh) <= 5e-6 * _rms(rh), (B, F, N)
    # ---- the sinusoid bank's adjoint at another hop
    hop = int(rng.choice([64, 128, 256, 300, 1024]))
    B, F, H = 1, int(rng.integers(1, 6)), int(rng.integers(1, 50))
    f0 = O.synth_f0(B, F, SR, hop, seed=int(rng.integers(1 << 30)))
    (ca,) = O.synth_controls(B, F, [H], seed=int(rng.integers(1 << 30)))
    R = rng.standard_normal((B, F * hop)).astype(np.float32)
    st = synth.phase(t(f0), SR, hop)
    cc = t(ca).requires_grad_(True)
    (synth.SinusoidBankFunction.apply(t(f0), st, cc, SR, hop) * t(R)).sum().backward()
    xw, _ = O.wrapped_phase(f0, SR, hop)
    want = O.sinusoid_bank_backward(R, xw, f0, ca, SR, hop)
    assert _rms(cc.grad.cpu().numpy() - want) <= 1e-5 * _rms(want), (hop, F, H)
