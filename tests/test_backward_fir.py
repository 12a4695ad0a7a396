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
def test_fft_convolve_backward(dev, B, F, N, run, knobs):
    """odd / even block counts (the held last tap row), a single frame, the largest N, several runs per utterance
    (carry rebuilt by the warm-up pair)"""
    from ddsp_svc_amd import core
    knobs("BLK_RUN", run)
    x, ir, R = _case(B, F, N, 100 * F + N)
    t = lambda a: torch.from_numpy(a).to(dev)
    dx, dh = core.fft_convolve_backward(t(R), t(x), t(ir))
    rx, rh = O.ltv_fir_backward(R, x, ir)
    assert rms(dx.cpu().numpy() - rx) <= 5e-6 * rms(rx)
    assert rms(dh.cpu().numpy() - rh) <= 5e-6 * rms(rh)
    none, dh2 = core.fft_convolve_backward(t(R), t(x), t(ir), need_audio_grad=False)
    assert none is None and torch.equal(dh2, dh)


def _case_hop(B, F, N, hop, seed):
    rng = np.random.default_rng(seed)
    x = (rng.random((B, F * hop)) * 2 - 1).astype(np.float32)
    ir = (rng.standard_normal((B, F, N)) / np.sqrt(N) * rng.uniform(0.05, 2.0, size=(B, F, 1))).astype(np.float32)
    R = rng.standard_normal((B, F * hop)).astype(np.float32)
    return x, ir, R


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
@pytest.mark.parametrize("B,F,N,hop", [(2, 5, 1022, 512), (1, 1, 1022, 512), (1, 3, 766, 512), (2, 7, 254, 256), (1, 4, 30, 100),
                                       (1, 2, 2050, 512), (1, 3, 510, 1100), (3, 1, 2, 7)])
def test_fft_convolve_backward_any_shape(dev, B, F, N, hop):
    """outside hop 512 / N <= 512 the adjoints are direct correlations (csrc/fir_bwd_direct.hip): the classic CombSub
    configuration's 512 harmonic bins (N = 1022), other block sizes, tap rows longer than a tile of the kernel (N > 1024),
    hops longer than a chunk (hop > 1024), a single frame (the held last row takes both weights of its own block)"""
    from ddsp_svc_amd import core
    x, ir, R = _case_hop(B, F, N, hop, 7 * F + N + hop)
    t = lambda a: torch.from_numpy(a).to(dev)
    dx, dh = core.fft_convolve_backward(t(R), t(x), t(ir))
    rx, rh = O.ltv_fir_backward(R, x, ir)
    assert rms(dx.cpu().numpy() - rx) <= 5e-6 * rms(rx)
    assert rms(dh.cpu().numpy() - rh) <= 5e-6 * rms(rh)
    none, dh2 = core.fft_convolve_backward(t(R), t(x), t(ir), need_audio_grad=False)
    assert none is None and torch.equal(dh2, dh)


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
def test_classic_combsub_training_step(dev):
    """CombSub at the classic bin counts 256 / 512 / 256 (N = 1022 for the harmonic filter): forward on the long-tap form,
    backward through the direct adjoints -- gradients of all three controls against the float64 adjoints of the oracle's
    operators, chained by hand (the same composition as synth._combsub_synth_train)"""
    from ddsp_svc_amd import synth
    B, F, SR = 1, 6, 44100
    f0 = O.synth_f0(B, F, SR, HOP, seed=5)
    cg, ch, cn = O.synth_controls(B, F, [256, 512, 256], seed=6)
    u = np.random.default_rng(8).random((B, F * HOP), dtype=np.float32)
    R = np.random.default_rng(9).standard_normal((B, F * HOP)).astype(np.float32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    ctrls = [t(c).requires_grad_(True) for c in (cg, ch, cn)]
    f0t = t(f0)
    st = synth.phase(f0t, SR, HOP)
    sig, harm, nz = synth.combsub_synth(f0t, st, ctrls[0], ctrls[1], ctrls[2], t(u), SR, HOP, noise_is_u01=True)
    ref = O.combsub_dsp(f0, cg, ch, cn, (u * np.float32(2) - np.float32(1)).astype(np.float32), SR, HOP)
    assert rms(sig.detach().cpu().numpy() - ref["signal"]) <= 1e-5 * rms(ref["signal"])
    (sig * t(R)).sum().backward()
    # the same chain on the CPU in float64 torch (the reference's op order, oracle/aten_chain.py) gives the expected gradients
    from oracle import aten_chain as A
    c64 = [torch.from_numpy(c.astype(np.float64)).requires_grad_(True) for c in (cg, ch, cn)]
    want = A.combsub_tail(torch.from_numpy(f0.astype(np.float64)).reshape(B, F, 1), c64[0], c64[1], c64[2],
                          torch.from_numpy((u.astype(np.float64) * 2 - 1)), SR, HOP)[0]
    (want * torch.from_numpy(R.astype(np.float64))).sum().backward()
    for got, ref_c, name in zip(ctrls, c64, ("group delay", "harmonic", "noise")):
        e = rms(got.grad.cpu().numpy() - ref_c.grad.numpy())
        assert e <= (3e-5 if name == "group delay" else 1e-5) * rms(ref_c.grad.numpy()), (name, e, rms(ref_c.grad.numpy()))


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
    with pytest.raises(RuntimeError):                       # an odd tap count is no output of core.py:254-270: refused loudly
        core.fft_convolve_backward(torch.zeros(1, 1024, device=dev), torch.zeros(1, 1024, device=dev),
                                   torch.zeros(1, 4, 31, device=dev))


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
def test_fft_convolve_add(dev):
    """signal = filter(x) + addend with the sum riding in the filter launch (core.fft_convolve_add): values of both outputs and
    gradients of all three inputs, with either output or both feeding the loss"""
    from ddsp_svc_amd import core
    x, ir, R = _case(2, 5, 510, 11)
    ad = np.random.default_rng(12).standard_normal(x.shape).astype(np.float32)
    R2 = np.random.default_rng(13).standard_normal(x.shape).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(dev)
    want = O.ltv_fir_blockfft(x, ir)
    for use_sum, use_plain in ((True, True), (True, False), (False, True)):
        a, h, d = t(x).requires_grad_(True), t(ir).requires_grad_(True), t(ad).requires_grad_(True)
        s, p = core.fft_convolve_add(a, h, d)
        assert torch.equal(p.detach(), core.fft_convolve(t(x), t(ir)))                   # the plain output: the same launch without the sum
        assert rms(s.detach().cpu().numpy() - (want + ad)) <= 2e-6 * rms(want + ad)
        loss = (s * t(R)).sum() * float(use_sum) + (p * t(R2)).sum() * float(use_plain)
        loss.backward()
        g = R * float(use_sum) + R2 * float(use_plain)
        rx, rh = O.ltv_fir_backward(g, x, ir)
        assert rms(a.grad.cpu().numpy() - rx) <= 5e-6 * rms(rx)
        assert rms(h.grad.cpu().numpy() - rh) <= 5e-6 * rms(rh)
        assert np.array_equal(d.grad.cpu().numpy(), R * np.float32(use_sum))
    with pytest.raises(ValueError):
        core.fft_convolve_add(t(x), t(ir), t(ad)[:, :-1])


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
def test_allpass_backward_256_same_bits(dev):
    """the 256-bin form of the all-pass adjoint (16-byte accesses, tanh once per bin) against the general kernel, which a row
    stride that is no multiple of four selects: the same operations in the same order -- the same bits"""
    from ddsp_svc_amd import _ffi
    from ddsp_svc_amd._ffi import ptr
    rng = np.random.default_rng(3)
    rows, n = 37, 256
    c = torch.from_numpy(rng.standard_normal((rows, n)).astype(np.float32)).to(dev)
    d_re = torch.from_numpy(rng.standard_normal((rows, n)).astype(np.float32)).to(dev)
    d_im = torch.from_numpy(rng.standard_normal((rows, n)).astype(np.float32)).to(dev)
    wide = torch.zeros(rows, n + 1, dtype=torch.float32, device=dev)
    wide[:, :n] = c
    out_a, out_b = torch.empty_like(c), torch.empty_like(c)
    lib, st = _ffi.lib(), _ffi.stream_of(c)
    _ffi.check(lib.ddsp_hip_allpass_backward(ptr(c), n, rows, n, ptr(d_re), ptr(d_im), ptr(out_a), st))
    _ffi.check(lib.ddsp_hip_allpass_backward(ptr(wide), n + 1, rows, n, ptr(d_re), ptr(d_im), ptr(out_b), st))
    assert torch.equal(out_a, out_b) and float(out_a.abs().sum()) > 0


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
def test_allpass_taps_backward_one_launch(dev, knobs):
    """ddsp_hip_allpass_taps_backward: at 256 bins the activation's adjoint runs in the tap adjoint's last stage (no scratch, any
    row stride, a ragged last batch of rows) and agrees with the two launches it replaces (knob AP_BWD_SPLIT = 1, which then need
    the scratch and say so) to rounding level and with the oracle's float64 adjoint; other bin counts need the scratch"""
    from ddsp_svc_amd import _ffi, synth
    from ddsp_svc_amd._ffi import ptr
    rng = np.random.default_rng(5)
    rows, n = 37, 256
    N = 2 * (n - 1)
    c = rng.standard_normal((rows, n)).astype(np.float32)
    R = rng.standard_normal((rows, N)).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(dev)
    wide = torch.zeros(rows, n + 3, dtype=torch.float32, device=dev)
    wide[:, :n] = t(c)
    g, tab = t(R), synth.ir_table(n, dev)
    lib, st = _ffi.lib(), _ffi.stream_of(g)
    out, out_w, out_s = (torch.empty(rows, n, dtype=torch.float32, device=dev) for _ in range(3))
    cc = t(c)
    _ffi.check(lib.ddsp_hip_allpass_taps_backward(ptr(g), ptr(cc), n, rows, n, ptr(tab), ptr(out), None, None, st))
    _ffi.check(lib.ddsp_hip_allpass_taps_backward(ptr(g), ptr(wide), n + 3, rows, n, ptr(tab), ptr(out_w), None, None, st))
    assert torch.equal(out, out_w)
    d_re, d_im = O.impulse_response_backward(R[None], O.MODE_ROLL)
    want = O.allpass_backward(c[None], d_re, d_im)[0]
    assert rms(out.cpu().numpy() - want) <= 2e-5 * rms(want)
    knobs("AP_BWD_SPLIT", 1)
    assert lib.ddsp_hip_allpass_taps_backward(ptr(g), ptr(cc), n, rows, n, ptr(tab), ptr(out_s), None, None, st) == -4
    s1, s2 = torch.empty_like(out), torch.empty_like(out)
    _ffi.check(lib.ddsp_hip_allpass_taps_backward(ptr(g), ptr(cc), n, rows, n, ptr(tab), ptr(out_s), ptr(s1), ptr(s2), st))
    assert rms((out - out_s).cpu().numpy()) <= 2e-6 * rms(want)
    knobs("AP_BWD_SPLIT", 0)
    n2 = 129
    g2 = t(rng.standard_normal((rows, 2 * (n2 - 1))).astype(np.float32))
    c2 = t(rng.standard_normal((rows, n2)).astype(np.float32))
    o2 = torch.empty(rows, n2, dtype=torch.float32, device=dev)
    assert lib.ddsp_hip_allpass_taps_backward(ptr(g2), ptr(c2), n2, rows, n2, ptr(synth.ir_table(n2, dev)), ptr(o2), None, None, st) == -4


# ---- tap synthesis backward + the CombSub training path ---------------------------------------------------------
@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
@pytest.mark.parametrize("n_mag,rows", [(256, 70), (129, 9), (65, 130), (5, 3)])
def test_impulse_response_backward(dev, n_mag, rows):
    """d_taps -> gradient of the raw magnitude control for the Hann and the dynamic window, and of the complex
    all-pass response back to the group-delay control; full and ragged tiles"""
    from ddsp_svc_amd import synth, _ffi
    rng = np.random.default_rng(n_mag + rows)
    N = 2 * (n_mag - 1)
    c = rng.standard_normal((1, rows, n_mag)).astype(np.float32)
    R = rng.standard_normal((1, rows, N)).astype(np.float32)
    hw = (rng.random((1, rows)) * 100 + 2).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(dev)
    for mode, scale, hwv in ((_ffi.MODE_HANN, 1.0 / 128.0, None), (_ffi.MODE_DYNAMIC, 1.0, hw)):
        cc = t(c).requires_grad_(True)
        taps = synth.MagnitudeTapsFunction.apply(cc, scale, mode, None if hwv is None else t(hwv))
        ref_taps = O.impulse_response(np.exp(c.astype(np.float64)) * scale, None, mode, hwv)
        assert rms(taps.detach().cpu().numpy() - ref_taps) <= 2e-6 * rms(ref_taps)
        (taps * t(R)).sum().backward()
        d_re, _ = O.impulse_response_backward(R, mode, hwv)
        want = d_re * np.exp(c.astype(np.float64)) * scale
        assert rms(cc.grad.cpu().numpy() - want) <= 5e-6 * rms(want), (mode, rms(cc.grad.cpu().numpy() - want), rms(want))
    cc = t(c).requires_grad_(True)
    taps = synth.AllpassTapsFunction.apply(cc)
    (taps * t(R)).sum().backward()
    d_re, d_im = O.impulse_response_backward(R, O.MODE_ROLL)
    want = O.allpass_backward(c, d_re, d_im)
    assert rms(cc.grad.cpu().numpy() - want) <= 2e-5 * rms(want), (rms(cc.grad.cpu().numpy() - want), rms(want))


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
def test_combsub_backward_golden(dev, golden_dir):
    """autograd through the CombSub tail reproduces the reference's control gradients (captured from the reference
    module's own backward pass, fixtures combsub_grad.npz)"""
    from ddsp_svc_amd import synth
    g = np.load(os.path.join(golden_dir, "combsub_grad.npz"))
    keys = ("group_delay", "harmonic_magnitude", "noise_magnitude")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    f0 = t(g["f0_frames"])
    sizes = [int(v) for v in g["sizes"]]
    packed = t(np.concatenate([g["ctrl_" + k] for k in keys], axis=-1)).requires_grad_(True)
    views = torch.split(packed, sizes, dim=-1)
    st = synth.phase(f0, 44100, 512)
    sig, harm, nz = synth.combsub_synth(f0, st, *views, t(g["noise"]), 44100, 512)
    assert sig.requires_grad
    assert rms(sig.detach().cpu().numpy() - g["signal"]) <= 1e-5 * rms(g["signal"])
    (sig * t(g["cotangent"])).sum().backward()
    grads = torch.split(packed.grad, sizes, dim=-1)
    for a, k in zip(grads, keys):
        ref = g["grad_" + k]
        tol = 3e-5 if k == "group_delay" else 1e-5
        assert rms(a.cpu().numpy() - ref) <= tol * rms(ref), (k, rms(a.cpu().numpy() - ref), rms(ref))


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
@pytest.mark.parametrize("H,F", [(40, 6), (256, 5), (1, 3), (17, 1), (300, 2), (264, 2)])   # 33-harmonic blocks, 8 per pass: one / two passes
def test_sinusoid_bank_backward(dev, H, F):
    """adjoint of the sinusoid bank w.r.t. the amplitude control: harmonic counts that are not a multiple of the
    16-harmonic block, a single frame (the held last row takes both parts), masked harmonics above Nyquist"""
    from ddsp_svc_amd import synth
    B = 2
    f0 = O.synth_f0(B, F, 44100, 512, seed=31 + H)
    f0[0] *= 2.5
    f0 = np.clip(f0, 65, 800).astype(np.float32)
    (c_amp,) = O.synth_controls(B, F, [H], seed=5)
    R = np.random.default_rng(H).standard_normal((B, F * 512)).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(dev)
    st = synth.phase(t(f0), 44100, 512)
    c = t(c_amp).requires_grad_(True)
    out = synth.SinusoidBankFunction.apply(t(f0), st, c, 44100, 512)
    (out * t(R)).sum().backward()
    x, _ = O.wrapped_phase(f0, 44100, 512)
    want = O.sinusoid_bank_backward(R, x, f0, c_amp, 44100, 512)
    # same documented deviation as the forward bank (no float32 rounding of k * phase) plus the rotation error
    assert rms(c.grad.cpu().numpy() - want) <= 1e-5 * rms(want), (rms(c.grad.cpu().numpy() - want), rms(want))


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
def test_sins_backward_golden(dev, golden_dir):
    """autograd through the Sins tail reproduces the reference's control gradients (fixture sins_grad.npz)"""
    from ddsp_svc_amd import synth
    g = np.load(os.path.join(golden_dir, "sins_grad.npz"))
    keys = ("amplitudes", "group_delay", "noise_magnitude")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    f0 = t(g["f0_frames"])
    sizes = [int(v) for v in g["sizes"]]
    packed = t(np.concatenate([g["ctrl_" + k] for k in keys], axis=-1)).requires_grad_(True)
    views = torch.split(packed, sizes, dim=-1)
    st = synth.phase(f0, 44100, 512)
    sig, harm, nz = synth.sins_synth(f0, st, *views, t(g["noise"]), 44100, 512)
    assert sig.requires_grad
    assert rms(sig.detach().cpu().numpy() - g["signal"]) <= 1e-5 * rms(g["signal"])
    (sig * t(g["cotangent"])).sum().backward()
    grads = torch.split(packed.grad, sizes, dim=-1)
    for a, k in zip(grads, keys):
        ref = g["grad_" + k]
        assert rms(a.cpu().numpy() - ref) <= 3e-5 * rms(ref), (k, rms(a.cpu().numpy() - ref), rms(ref))


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
@pytest.mark.parametrize("kind", ["combsub", "sins"])
def test_combsub_module_training_step_matches_reference(dev, kind):
    """one backward pass through the drop-in CombSub / Sins (reference Unit2Control inside): parameter gradients equal
    the reference module's (same weights, inputs, noise, cotangent)"""
    from unittest import mock
    from unittest.mock import MagicMock
    ref_root = os.environ.get("DDSP_REFERENCE_PATH", "/root/reference")
    if not os.path.isdir(os.path.join(ref_root, "ddsp")):
        pytest.skip("reference checkout not present (only in the build container)")
    if ref_root not in sys.path:
        sys.path.insert(0, ref_root)
    for name in ["transformers", "pyworld", "parselmouth", "torchcrepe", "resampy", "fairseq", "torchaudio",
                 "torchaudio.transforms", "gin", "local_attention", "librosa", "librosa.sequence", "librosa.util",
                 "librosa.filters", "librosa.core", "soundfile"]:
        sys.modules.setdefault(name, MagicMock())
    import ddsp.vocoder as rvoc
    from ddsp_svc_amd import vocoder as V
    name = "CombSub" if kind == "combsub" else "Sins"
    ref_cls = getattr(rvoc, "_reference_" + name, getattr(rvoc, name))
    torch.manual_seed(3)
    B, F, n_unit = 2, 6, 16
    args = (44100, 512, 65, 129, 65) if kind == "combsub" else (44100, 512, 40, 65, 33)
    ref = ref_cls(*args, n_unit=n_unit, n_spk=1).train()
    ours = getattr(V, name)(*args, n_unit=n_unit, n_spk=1).train()
    ours.load_state_dict(ref.state_dict(), strict=True)
    ours = ours.to(dev)                                         # the reference stays on its CPU path
    for m in (ref, ours):
        for sub in m.modules():
            if isinstance(sub, torch.nn.Dropout):
                sub.p = 0.0
    g = torch.Generator().manual_seed(4)
    units = torch.randn(B, F, n_unit, generator=g)
    f0 = torch.from_numpy(O.synth_f0(B, F, 44100, 512, seed=9))
    vol = torch.rand(B, F, 1, generator=g) * 0.1
    u = torch.rand(B, F * 512, generator=g)
    R = torch.randn(B, F * 512, generator=g)
    ud = u.to(dev)
    with mock.patch("torch.rand_like", side_effect=lambda t: u.reshape(t.shape)):
        r_sig, _, _ = ref(units, f0, vol, infer=True)
    with mock.patch("torch.rand", side_effect=lambda *a, **k: ud):
        o_sig, _, _ = ours(units.to(dev), f0.to(dev), vol.to(dev), infer=True)
    (r_sig * R).sum().backward()
    (o_sig * R.to(dev)).sum().backward()
    # on the MI355X Unit2Control's own float32 GEMMs (forward and backward) round differently from the CPU's
    tol = 5e-5 if dev.type == "cpu" else 2e-4                   # measured: 1.3e-5 (profiles/r04_v11_reference_on_gpu.log)
    checked, worst = 0, 0.0
    for (n1, p1), (n2, p2) in zip(ref.named_parameters(), ours.named_parameters()):
        assert n1 == n2
        if p1.grad is None:
            assert p2.grad is None
            continue
        scale = max(rms(p1.grad.numpy()), 1e-12)
        err = rms((p2.grad.cpu() - p1.grad).numpy())
        worst = max(worst, err / scale)
        assert err <= tol * scale + 1e-9, (n1, err, scale)
        checked += 1
    print("training step %s on %s: %d parameter gradients, worst relative rms error %.2e" % (kind, dev, checked, worst))
    assert checked > 10


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
@pytest.mark.parametrize("hop,H,F", [(256, 40, 5), (128, 17, 4), (1000, 33, 2), (2048, 8, 2)])
def test_sinusoid_bank_backward_other_hops(dev, hop, H, F):
    """the adjoint of the sinusoid bank at block sizes other than 512 (train.py with a config whose block_size is not 512:
    the reference back-propagates through vocoder.py:585-594 at any block size): k_sins_bank_bwd_any against the oracle"""
    from ddsp_svc_amd import synth
    B = 2
    f0 = O.synth_f0(B, F, 44100, hop, seed=3 + H)
    f0[0] *= 2.5
    f0 = np.clip(f0, 65, 800).astype(np.float32)
    (c_amp,) = O.synth_controls(B, F, [H], seed=6)
    R = np.random.default_rng(hop).standard_normal((B, F * hop)).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(dev)
    st = synth.phase(t(f0), 44100, hop)
    c = t(c_amp).requires_grad_(True)
    out = synth.SinusoidBankFunction.apply(t(f0), st, c, 44100, hop)
    (out * t(R)).sum().backward()
    x, _ = O.wrapped_phase(f0, 44100, hop)
    want = O.sinusoid_bank_backward(R, x, f0, c_amp, 44100, hop)
    assert rms(c.grad.cpu().numpy() - want) <= 1e-5 * rms(want), (rms(c.grad.cpu().numpy() - want), rms(want))


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
def test_sins_trains_at_hop_256(dev):
    """a patched Sins at block size 256 must train where the reference trains -- every control gets a gradient, equal to
    float64 autograd through the reference's op chain (oracle/aten_chain.py) on the same inputs"""
    from ddsp_svc_amd import synth
    from oracle import aten_chain as A
    hop, B, F, H, n = 256, 2, 6, 24, 65
    f0 = O.synth_f0(B, F, 44100, hop, seed=8)
    ca, cg, cn = O.synth_controls(B, F, [H, n, n], seed=9)
    u = np.random.default_rng(10).random((B, F * hop), dtype=np.float32)
    R = np.random.default_rng(11).standard_normal((B, F * hop)).astype(np.float32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    ctrls = [t(a).requires_grad_(True) for a in (ca, cg, cn)]
    st = synth.phase(t(f0), 44100, hop)
    sig = synth.sins_synth(t(f0), st, *ctrls, t(u), 44100, hop, noise_is_u01=True)[0]
    assert sig.requires_grad
    (sig * t(R)).sum().backward()
    ref_c = [torch.from_numpy(a).double().requires_grad_(True) for a in (ca, cg, cn)]
    nz = torch.from_numpy(u).double() * 2 - 1
    ref = A.sins_tail(torch.from_numpy(f0).double(), *ref_c, nz, 44100, hop, True)[0]
    assert rms(sig.detach().cpu().numpy() - ref.detach().numpy()) <= 1e-5 * rms(ref.detach().numpy())
    (ref * torch.from_numpy(R).double()).sum().backward()
    for got, want, name in zip(ctrls, ref_c, ("amplitudes", "group_delay", "noise_magnitude")):
        e, w = rms(got.grad.cpu().numpy() - want.grad.numpy()), rms(want.grad.numpy())
        assert e <= 3e-5 * w, (name, e, w)


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
@pytest.mark.parametrize("B,F,N,run", [(2, 5, 1022, 0), (1, 8, 1022, 1), (1, 7, 514, 2), (2, 1, 1022, 0), (1, 2, 600, 1), (1, 12, 766, 3),
                                       (1, 9, 1000, 1)])
def test_fft_convolve_backward_long_taps_fft_form(dev, B, F, N, run, knobs):
    """514 .. 1022 taps at hop 512 (the classic CombSub harmonic filter, n_mag 512): the per-frame 2048-point adjoint
    (csrc/fir_fft_bwd.hip) against the oracle's float64 adjoint -- odd / even frame counts (the padded last pair, the held last
    row fed by frames F-1 and F through two atomic adds), a single frame, runs of one pair (every carry rebuilt by a warm-up frame)
    -- and against the direct-correlation form it replaces there (knob FIR_BWD_DIRECT = 1): the same gradients, other bits"""
    from ddsp_svc_amd import core
    if run:
        knobs("FFT_RUN", run)
    x, ir, R = _case(B, F, N, 1000 * F + N)
    t = lambda a: torch.from_numpy(a).to(dev)
    dx, dh = core.fft_convolve_backward(t(R), t(x), t(ir))
    rx, rh = O.ltv_fir_backward(R, x, ir)
    assert rms(dx.cpu().numpy() - rx) <= 5e-6 * rms(rx), rms(dx.cpu().numpy() - rx) / rms(rx)
    assert rms(dh.cpu().numpy() - rh) <= 5e-6 * rms(rh), rms(dh.cpu().numpy() - rh) / rms(rh)
    none, dh2 = core.fft_convolve_backward(t(R), t(x), t(ir), need_audio_grad=False)
    assert none is None and torch.equal(dh2, dh)
    dx3, dh3 = core.fft_convolve_backward(t(R), t(x), t(ir))
    assert torch.equal(dx3, dx) and torch.equal(dh3, dh)                      # the atomics on the last row add two addends: same bits
    knobs("FIR_BWD_DIRECT", 1)
    dxd, dhd = core.fft_convolve_backward(t(R), t(x), t(ir))
    assert rms((dxd - dx).cpu().numpy()) <= 5e-6 * rms(rx) and rms((dhd - dh).cpu().numpy()) <= 5e-6 * rms(rh)
    assert not torch.equal(dhd, dh)                                           # (it IS another kernel)


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
@pytest.mark.parametrize("kind", ["combsub", "sins"])
def test_fused_tail_training_node(dev, kind, monkeypatch, knobs):
    """256-bin models at hop 512 train through ONE autograd node whose forward is the fused inference call (its intermediates stay
    in the workspace, ddsp_hip_tail_layout) -- against the per-operator composition of rounds 2 - 5 on the same inputs (outputs:
    the composition's filters use another run split, rounding level; gradients <= 1e-5) and, for CombSub, against the oracle's
    float64 adjoint (pinned to the reference's autograd by test_oracle_golden.py::test_combsub_tail_adjoint); cotangents on all
    three outputs, odd frame count, a uniform draw handed over as u in [0, 1)"""
    from ddsp_svc_amd import synth
    B, F, n, H = 2, 7, 256, 40
    f0 = O.synth_f0(B, F, 44100, 512, seed=11)
    sizes = [n, n, n] if kind == "combsub" else [H, n, n]
    ctrls = O.synth_controls(B, F, sizes, seed=12)
    u = np.random.default_rng(13).random((B, F * 512)).astype(np.float32)
    R = [np.random.default_rng(20 + i).standard_normal((B, F * 512)).astype(np.float32) for i in range(3)]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    fn = synth.combsub_synth if kind == "combsub" else synth.sins_synth

    def run(composed, which=(0, 1, 2)):
        monkeypatch.setattr(synth, "_TRAIN_COMPOSED", composed)
        c = [t(x).requires_grad_(True) for x in ctrls]
        st = synth.phase(t(f0), 44100, 512)
        outs = fn(t(f0), st, c[0], c[1], c[2], t(u), 44100, 512, noise_is_u01=True)
        assert all(o.requires_grad for o in outs)
        loss = sum((outs[i] * t(R[i])).sum() for i in which)
        grads = torch.autograd.grad(loss, c, allow_unused=True)
        return [o.detach().cpu().numpy() for o in outs], [None if g is None else g.cpu().numpy() for g in grads]
    # a training loop differentiates `signal` only; `noise` alone leaves the harmonic branch's controls without a gradient
    for which in ((0,), (2,), (1,)):
        _, a = run(False, which)
        _, b = run(True, which)
        for x, y in zip(a, b):
            assert (x is None) == (y is None) or (x is None and rms(y) == 0.0) or (y is None and rms(x) == 0.0), which
            if x is not None and y is not None and rms(y) > 0:
                assert rms(x - y) <= 1e-5 * rms(y), (which, rms(x - y), rms(y))
    out_f, g_f = run(False)
    out_c, g_c = run(True)
    for a, b in zip(out_f, out_c):
        assert rms(a - b) <= 2e-6 * rms(b)
    for a, b in zip(g_f, g_c):
        assert rms(a - b) <= 1e-5 * rms(b), (rms(a - b), rms(b))
    if kind == "combsub":
        # signal = harmonic + noise: the cotangent of the harmonic branch is R0 + R1, of the noise branch R0 + R2
        wh = O.combsub_dsp_backward(R[0] + R[1], f0, ctrls[0], ctrls[1], ctrls[2], 2.0 * u - 1.0)
        wn = O.combsub_dsp_backward(R[0] + R[2], f0, ctrls[0], ctrls[1], ctrls[2], 2.0 * u - 1.0)
        for got, want in zip(g_f, (wh["group_delay"], wh["harmonic_magnitude"], wn["noise_magnitude"])):
            assert rms(got - want) <= 2e-5 * rms(want), (rms(got - want), rms(want))
        # the all-pass activation's adjoint rides in the tap adjoint's last stage (three launches); knob AP_BWD_SPLIT = 1 runs it as
        # the launch of its own it was (k_allpass_backward_256: libm's tanh and a float64 phase where the fused stage has the forward
        # kernel's hardware tanh and fixed-point phase) -- the two agree far inside the bar above, the other gradients bit for bit
        knobs("AP_BWD_SPLIT", 1)
        _, g_s = run(False)
        assert rms(g_s[0] - g_f[0]) <= 2e-6 * rms(g_f[0]), (rms(g_s[0] - g_f[0]), rms(g_f[0]))
        assert np.array_equal(g_s[1], g_f[1]) and np.array_equal(g_s[2], g_f[2])


@pytest.mark.gpu
def test_sinusoid_bank_backward_beyond_the_shift_form_gpu():
    """an utterance of more than 2^24 samples at hop 512: make_upsampler has no shift form there, the forward takes the generic
    k_sins_bank (ATen's float-rounded interpolation positions) -- and the adjoint must take the matching wave-per-frame kernel, not
    the matrix-pipe one that hard-codes the shift form.  The bank is LINEAR in A = exp(c) / 128, so the gradient
    of <R, out> w.r.t. c[f, k] is A[f, k] <R, out(one-hot A at (f, k))>: the adjoint against the FORWARD kernel itself, on the
    utterance's last frames (where a float-rounded position differs most from the integer one)"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ddsp_svc_amd import synth
    dev = torch.device("cuda:0")
    B, F, H = 1, 32770, 6                                            # F * 512 = 16 778 240 > 2^24
    f0 = np.clip(O.synth_f0(B, F, 44100, 512, seed=77), 65, 800).astype(np.float32)
    (c_amp,) = O.synth_controls(B, F, [H], seed=6)
    R = np.zeros((B, F * 512), dtype=np.float32)
    tail = 6 * 512
    R[:, -tail:] = np.random.default_rng(1).standard_normal((B, tail)).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(dev)
    f0t, Rt = t(f0), t(R)
    st = synth.phase(f0t, 44100, 512)
    c = t(c_amp).requires_grad_(True)
    out = synth.SinusoidBankFunction.apply(f0t, st, c, 44100, 512)
    (out * Rt).sum().backward()
    got = c.grad.cpu().numpy()
    assert float(np.abs(got[:, :-8]).max()) == 0.0 or float(np.abs(got[:, :-8]).max()) <= 1e-6 * float(np.abs(got).max())
    for f, k in ((F - 1, 0), (F - 1, 5), (F - 2, 2), (F - 3, 1), (F - 6, 4), (F - 7, 3)):
        hot = torch.full((B, F, H), -1.0e4, device=dev)
        hot[0, f, k] = float(np.log(128.0))                          # A = 1 there, 0 elsewhere
        basis = synth.sinusoid_bank(f0t, st, hot, 44100, 512)
        want = float((basis.double() * Rt.double()).sum()) * float(np.exp(np.float64(c_amp[0, f, k])) / 128.0)
        assert abs(float(got[0, f, k]) - want) <= 2e-5 * max(abs(want), 1e-3 * float(np.abs(got[:, -8:]).max())), (f, k, got[0, f, k], want)
