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


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
@pytest.mark.parametrize("n_fft", [2, 97, 512, 513, 1024, 1153, 2048])
def test_in_kernel_transform_parts(dev, n_fft):
    """csrc/loss_czt.hip on its own, through the C ABI: the chirp-z spectra of both signals against numpy's float64 rfft
    (every plan: 1024 / 2048 / 4096 points, at its edges), frames that are silent or equal, the backward transform against
    the float64 adjoint of the same bin gradients, the accumulate switch and the zeroed tail."""
    from ddsp_svc_amd import loss as L, _ffi
    from ddsp_svc_amd._ffi import ptr
    lib = _ffi.lib()
    B, frames = 2, 5
    T = frames * n_fft + min(3, n_fft - 1)
    rng = np.random.default_rng(n_fft)
    a = (rng.standard_normal((B, T)) * 0.1).astype(np.float32)
    b = (a * 0.7 + rng.standard_normal((B, T)) * 0.05).astype(np.float32)
    a[0, n_fft:2 * n_fft] = 0.0                                           # a silent frame of the truth
    b[1, 2 * n_fft:3 * n_fft] = 0.0                                       # ... of the prediction
    b[1, 3 * n_fft:4 * n_fft] = a[1, 3 * n_fft:4 * n_fft]                 # an equal pair of frames
    b[0, 4 * n_fft:5 * n_fft] *= 1e-4                                     # a prediction 80 dB below its target
    xt, xp = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    f = L.SSSLoss(n_fft)
    tab = L._czt_tables(n_fft, xt)
    assert tab is not None and tab.numel() * 4 == lib.ddsp_hip_stft_loss_table_bytes(n_fft)
    bins = n_fft // 2 + 1
    assert lib.ddsp_hip_stft_loss_frames(T, n_fft, n_fft) == frames
    spec = torch.empty(2, B, frames, bins, dtype=torch.complex64, device=dev)
    nb = lib.ddsp_hip_stft_loss_scratch_bytes(B, T, n_fft, n_fft)
    scratch = torch.empty(max(nb, 8) // 8, dtype=torch.float64, device=dev)
    norms, lo = torch.empty(B, 2, device=dev), torch.empty((), device=dev)
    inv = f.spec.inv_window_norm
    _ffi.check(lib.ddsp_hip_stft_loss(ptr(xt), ptr(xp), B, T, T, n_fft, n_fft, ptr(tab), inv, 1e-7, 1.0, ptr(scratch), nb,
                                      ptr(spec[0]), ptr(spec[1]), ptr(norms), ptr(lo), _ffi.stream_of(xt)))
    w = torch.hann_window(n_fft, dtype=torch.float64).numpy()
    X64 = lambda x: np.fft.rfft(x[:, :frames * n_fft].reshape(B, frames, n_fft).astype(np.float64) * w, axis=-1)
    got_t, got_p = spec[0].cpu().numpy(), spec[1].cpu().numpy()
    for got, x in ((got_t, a), (got_p, b)):
        want = X64(x)
        assert np.abs(got - want).max() <= 5e-7 * np.abs(want).max()
    quiet = X64(b)[0, 4]                                                  # ... is accurate to ITS size, as a transform of its own
    assert np.abs(got_p[0, 4] - quiet).max() <= 2e-6 * np.abs(quiet).max()
    assert not got_t[0, 1].any() and not got_p[1, 2].any()                # silence is exact
    assert np.array_equal(got_t[1, 3], got_p[1, 3])                       # equal frames, equal spectra
    St, Sp = np.abs(X64(a)) * inv + 1e-7, np.abs(X64(b)) * inv + 1e-7
    want = np.mean(np.linalg.norm((St - Sp).reshape(B, -1), axis=1) / np.linalg.norm((St + Sp).reshape(B, -1), axis=1)) \
        + np.mean(np.abs(np.log(St) - np.log(Sp)))
    assert abs(float(lo) - want) <= 2e-5 * want
    go = torch.full((), 0.5, device=dev)
    for wrt_true in (0, 1):
        G = torch.empty_like(spec[1])
        _ffi.check(lib.ddsp_hip_spectral_loss_backward(ptr(spec[0]), ptr(spec[1]), B, frames * bins, ptr(norms), inv, 1e-7,
                                                       1.0, ptr(go), wrt_true, ptr(G), _ffi.stream_of(xt)))
        xx = torch.from_numpy(a if wrt_true else b).double().requires_grad_(True)
        Xr = torch.fft.rfft(xx[:, :frames * n_fft].reshape(B, frames, n_fft) * torch.from_numpy(w), dim=-1)
        Xr.backward(G.cpu().to(torch.complex128))
        d = torch.full((B, T), 7.0, device=dev)
        _ffi.check(lib.ddsp_hip_stft_loss_backward(ptr(spec[0]), ptr(spec[1]), B, T, n_fft, n_fft, ptr(tab), ptr(norms), inv, 1e-7,
                                                   1.0, ptr(go), wrt_true, ptr(d), T, 0, None, 0, _ffi.stream_of(xt)))
        ref = xx.grad.numpy()
        assert _rel_rms(d.cpu().numpy(), ref) <= 1e-6
        assert not d.cpu().numpy()[:, frames * n_fft:].any()              # the tail is written, with zeros
        d2 = d.clone()
        _ffi.check(lib.ddsp_hip_stft_loss_backward(ptr(spec[0]), ptr(spec[1]), B, T, n_fft, n_fft, ptr(tab), ptr(norms), inv, 1e-7,
                                                   1.0, ptr(go), wrt_true, ptr(d2), T, 1, None, 0, _ffi.stream_of(xt)))
        assert torch.equal(d2, d + d)
    assert lib.ddsp_hip_stft_loss_table_bytes(2049) == 0 and lib.ddsp_hip_stft_loss_table_bytes(1) == 0


@pytest.mark.gpu
def test_full_size_against_eager_composition():
    """B = 32 x 10 s, sizes RSSLoss draws from (256 .. 2047; 397, 1153 and 2047 = 23 * 89 with large prime factors): the
    in-kernel chirp-z STFT and its loss against float64 / eager float32 compositions of loss.py:22-31 on the same device.

    The loss's gradient is DISCONTINUOUS where S_true = S_pred (the sign of the log difference, times 1 / S_pred): bins
    within float32 rounding of that flip between any two float32 transforms (rocFFT against this one, or against itself
    on another architecture), and one flipped bin moves a whole frame's gradient.  So the gradient is checked in parts
    that are well conditioned -- the spectra against float64, the backward transform against the float64 adjoint of the
    SAME bin gradients -- and against the eager composition per frame: the median frame tightly, the flipped ones by
    count."""
    from ddsp_svc_amd import loss as L, _ffi
    from ddsp_svc_amd._ffi import ptr
    dev = torch.device("cuda:0")
    lib = _ffi.lib()
    g = torch.Generator(device="cpu").manual_seed(5)
    B, T = 32, 441344
    xt = (torch.randn(B, T, generator=g) * 0.1).to(dev)
    xp = (xt * 0.9 + 0.02 * torch.randn(B, T, generator=g).to(dev)).requires_grad_(True)
    for n_fft in (256, 397, 1153, 2047):
        f = L.SSSLoss(n_fft).to(dev)
        loss = f(xt, xp)
        assert loss.grad_fn is not None and "WaveLoss" in type(loss.grad_fn).__name__
        grad, = torch.autograd.grad(loss, xp)
        w = f.spec.window
        sp = lambda x: torch.stft(x, n_fft, hop_length=n_fft, win_length=n_fft, window=w, center=False,
                                  return_complex=True).abs() / w.pow(2).sum().sqrt() + 1e-7
        St, Sp = sp(xt), sp(xp)
        ref = torch.mean(torch.linalg.norm(St - Sp, dim=(1, 2)) / torch.linalg.norm(St + Sp, dim=(1, 2))) \
            + torch.nn.functional.l1_loss(St.log(), Sp.log())
        rgrad, = torch.autograd.grad(ref, xp)
        assert abs(float(loss.detach()) - float(ref.detach())) <= 1e-5 * float(ref.detach())
        frames, bins = T // n_fft, n_fft // 2 + 1
        per_frame = lambda x: x[:, :frames * n_fft].reshape(B * frames, n_fft)
        err = (per_frame(grad) - per_frame(rgrad)).pow(2).mean(1).sqrt() / per_frame(rgrad).pow(2).mean(1).sqrt()
        assert float(err.median()) <= 2e-5
        assert float((err > 1e-3).float().mean()) <= 2e-3            # frames holding a flipped bin
        assert float(grad[:, frames * n_fft:].abs().max()) == 0.0 if frames * n_fft < T else True
        # the parts
        tab = L._czt_tables(n_fft, xt)
        spec = torch.empty(2, B, frames, bins, dtype=torch.complex64, device=dev)
        nb = lib.ddsp_hip_stft_loss_scratch_bytes(B, T, n_fft, n_fft)
        scratch = torch.empty(nb // 8, dtype=torch.float64, device=dev)
        norms, lo = torch.empty(B, 2, device=dev), torch.empty((), device=dev)
        inv = f.spec.inv_window_norm
        _ffi.check(lib.ddsp_hip_stft_loss(ptr(xt), ptr(xp.detach()), B, T, T, n_fft, n_fft, ptr(tab), inv, 1e-7, 1.0,
                                          ptr(scratch), nb, ptr(spec[0]), ptr(spec[1]), ptr(norms), ptr(lo),
                                          _ffi.stream_of(xt)))
        X64 = lambda x: torch.fft.rfft(x.double()[:, :frames * n_fft].reshape(B, frames, n_fft) * w.double(), dim=-1)
        for got, x in ((spec[0], xt), (spec[1], xp.detach())):
            want = X64(x)
            assert float((got - want).abs().max() / want.abs().max()) <= 5e-7
        go = torch.ones((), device=dev)
        G = torch.empty_like(spec[1])
        _ffi.check(lib.ddsp_hip_spectral_loss_backward(ptr(spec[0]), ptr(spec[1]), B, frames * bins, ptr(norms), inv, 1e-7,
                                                       1.0, ptr(go), 0, ptr(G), _ffi.stream_of(xt)))
        xx = xp.detach().double().requires_grad_(True)
        X64(xx).backward(G.to(torch.complex128))
        d = torch.empty(B, T, device=dev)
        _ffi.check(lib.ddsp_hip_stft_loss_backward(ptr(spec[0]), ptr(spec[1]), B, T, n_fft, n_fft, ptr(tab), ptr(norms), inv, 1e-7,
                                                   1.0, ptr(go), 0, ptr(d), T, 0, None, 0, _ffi.stream_of(xt)))
        assert float((d - xx.grad).pow(2).mean().sqrt() / xx.grad.pow(2).mean().sqrt()) <= 1e-6
        # ... and frame by frame, with NO frame excused: against the float64 adjoint of the same bin gradients nothing is
        # discontinuous, so a frame whose gradient were wrong for any reason fails here however few such frames there are --
        # the cap the flipped-frame count above cannot give.  The autograd path returns these very numbers.
        # (The kernel sends the gradients of frames 2i, 2i + 1 through ONE inverse transform: each carries the other's rounding
        # noise, 1e-7 of the LARGER of the two -- so the yardstick of a frame is the larger gradient of its pair.)
        r = per_frame(xx.grad).pow(2).mean(1).sqrt().reshape(B, frames)
        rp = torch.nn.functional.pad(r, (0, frames % 2)).reshape(B, -1, 2).amax(dim=2, keepdim=True).expand(-1, -1, 2)
        rp = rp.reshape(B, -1)[:, :frames].reshape(B * frames)
        fe = (per_frame(d) - per_frame(xx.grad)).pow(2).mean(1).sqrt() / rp.clamp_min(1e-30)
        assert float(fe.max()) <= 2e-5, (n_fft, float(fe.max()))
        ge = (per_frame(grad) - per_frame(d)).pow(2).mean(1).sqrt() / per_frame(d).pow(2).mean(1).sqrt().clamp_min(1e-30)
        assert float(ge.max()) <= 1e-6, (n_fft, float(ge.max()))
        d2 = d.clone()                                                  # accumulate: exactly twice
        _ffi.check(lib.ddsp_hip_stft_loss_backward(ptr(spec[0]), ptr(spec[1]), B, T, n_fft, n_fft, ptr(tab), ptr(norms), inv, 1e-7,
                                                   1.0, ptr(go), 0, ptr(d2), T, 1, None, 0, _ffi.stream_of(xt)))
        assert torch.equal(d2, d + d)


@pytest.mark.gpu
def test_random_scale_loss_is_one_node_and_matches_its_scales():
    """RSSLoss with overlap 0: the fused node (all scales, one gradient buffer) against the scales one by one."""
    from ddsp_svc_amd import loss as L
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(6)
    xt = (torch.randn(8, 100000, generator=g) * 0.1).to(dev)
    xp = (xt * 0.8 + 0.05 * torch.randn(8, 100000, generator=g).to(dev)).requires_grad_(True)
    sizes = torch.tensor([1153, 397, 2011, 768])
    rss = L.RSSLoss(256, 2048, 4, device=dev)
    real = torch.randint
    torch.randint = lambda *a, **k: sizes
    try:
        loss = rss(xp, xt)
    finally:
        torch.randint = real
    assert "RandomScaleWaveLoss" in type(loss.grad_fn).__name__
    grad, = torch.autograd.grad(loss, xp)
    xq = xp.detach().clone().requires_grad_(True)
    parts = sum(L.SSSLoss(int(n)).to(dev)(xt, xq) for n in sizes) / 4
    pgrad, = torch.autograd.grad(parts, xq)
    assert abs(float(loss.detach()) - float(parts.detach())) <= 1e-6 * float(parts.detach())
    assert float((grad - pgrad).abs().max()) <= 1e-6 * float(pgrad.abs().max())


@pytest.mark.gpu
def test_full_size_properties():
    """B = 32 x 10 s, one size per transform plan (1024 / 2048 / 4096 points): what holds at any size without a second
    implementation -- identical signals give exactly 0 (equal frames, equal spectra) and a zero gradient; negating both
    signals changes neither the loss nor, up to its sign, any bit of the gradient (negation commutes with every rounding
    of the transform); the gradient behind the last whole frame is zero; two evaluations agree bit for bit."""
    from ddsp_svc_amd import loss as L
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(11)
    xt = (torch.randn(32, 441344, generator=g) * 0.1).to(dev)
    xp = (xt * 0.7 + 0.03 * torch.randn(32, 441344, generator=g).to(dev))
    for n_fft, overlap in ((509, 0.0), (1021, 0.0), (2039, 0.0), (1021, 0.5)):
        f = L.SSSLoss(n_fft, 1.0, overlap).to(dev)
        a = xp.clone().requires_grad_(True)
        same = f(a, a.detach())
        assert float(same.detach()) == 0.0
        la = f(xt, a)
        ga, = torch.autograd.grad(la, a)
        b = (-xp).requires_grad_(True)
        lb = f(-xt, b)
        gb, = torch.autograd.grad(lb, b)
        assert torch.equal(la.detach(), lb.detach()) and torch.equal(ga, -gb)
        c = xp.clone().requires_grad_(True)
        lc = f(xt, c)
        gc, = torch.autograd.grad(lc, c)
        assert torch.equal(la.detach(), lc.detach()) and torch.equal(ga, gc)
        hop = int(n_fft * (1 - overlap))
        covered = (1 + (441344 - n_fft) // hop - 1) * hop + n_fft
        assert torch.isfinite(ga).all() and float(ga[:, covered:].abs().max()) == 0.0
        assert float(ga[:, :covered].abs().max()) > 0.0


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_rss_loss_under_autocast(dev, dtype, golden_dir, monkeypatch):
    """diffusion/solver_new.py:132 evaluates the DDSP loss inside torch.autocast: the loss is the float32 one (torch.stft is on
    autocast's float32 list, and so is this path: it upcasts what it is given), also for a prediction that arrives in reduced
    precision, whose gradient comes back in ITS dtype"""
    from ddsp_svc_amd import loss as L
    g = np.load(os.path.join(golden_dir, "sssloss.npz"))
    sizes = torch.from_numpy(g["rss_sizes"])
    monkeypatch.setattr(torch, "randint", lambda *a, **k: sizes)
    rss = L.RSSLoss(256, 300, 4, device=dev)
    xt = torch.from_numpy(g["x_true"]).to(dev)
    xp = torch.from_numpy(g["x_pred"]).to(dev).requires_grad_(True)
    with torch.autocast("cuda" if dev.type == "cuda" else "cpu", dtype=dtype):
        loss = rss(xp, xt)
    assert loss.dtype == torch.float32
    loss.backward()
    assert abs(float(loss.detach()) - float(g["rss_loss"])) <= LOSS_RTOL * float(g["rss_loss"])
    assert xp.grad.dtype == torch.float32 and _rel_rms(xp.grad.cpu().numpy(), g["rss_grad"].astype(np.float64)) <= GRAD_RTOL
    # a reduced-precision prediction: the loss of its upcast values, the gradient in its dtype
    xh = xp.detach().to(dtype).requires_grad_(True)
    with torch.autocast("cuda" if dev.type == "cuda" else "cpu", dtype=dtype):
        lh = rss(xh, xt)
    lh.backward()
    xu = xh.detach().float().requires_grad_(True)
    lu = rss(xu, xt)
    lu.backward()
    assert lh.dtype == torch.float32 and float(lh) == float(lu)
    assert xh.grad.dtype == dtype and torch.equal(xh.grad, xu.grad.to(dtype))
