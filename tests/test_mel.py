"""Log-mel front-end (nsf_hifigan/nvSTFT.py:73-117): oracle pinning against the reference's own
``STFT.get_mel`` outputs (fixtures mel_*.npz), and parity of the HIP kernel on both backends.

Tolerances: the oracle (float64) sits <= 5e-6 (log domain) from the reference's float32 pipeline.  The HIP path is
held to <= 2e-6 of the frame's largest mel value in the linear domain, and to <= 2e-4 in the log domain on every band
above 1e-4 of that maximum (bands further down sit on the float32 FFT's noise floor, where the reference's own
pocketfft and any other float32 transform legitimately differ).  The same bars hold for the chirp-z kernel of the
keyshift / speed / center variants (csrc/mel_czt.hip), whose oracle form is pinned <= 1e-5 to the reference's
outputs for ten (keyshift, speed, center) combinations, a short window and the 22.05 kHz configuration."""
import os
import sys
from unittest import mock
from unittest.mock import MagicMock

import numpy as np
import pytest
import torch

from oracle import ddsp_oracle as O
from tests.backends import BACKENDS, dev  # noqa: F401

CFG = dict(sr=44100, n_mels=128, n_fft=2048, win_size=2048, hop_length=512, fmin=40, fmax=16000)


def _basis(golden_dir):
    return np.load(os.path.join(golden_dir, "mel_a.npz"))["basis"]


def _check(mel, ref):
    lin, rlin = np.exp(mel.astype(np.float64)), np.exp(ref.astype(np.float64))
    top = rlin.max(axis=1, keepdims=True)
    assert np.abs(lin - rlin).max() <= 2e-6 * rlin.max()
    assert (np.abs(lin - rlin) <= 2e-6 * top + 1e-12).all()
    strong = rlin >= 1e-4 * top
    assert np.abs(mel - ref)[strong].max() <= 2e-4


@pytest.mark.parametrize("tag", ["a", "t1024", "t512"])
def test_oracle_against_reference_get_mel(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, f"mel_{tag}.npz"))
    m = O.get_mel(g["audio"], _basis(golden_dir))
    assert m.shape == g["mel"].shape
    assert np.abs(m - g["mel"]).max() <= 1e-5


def _shifted_cases():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from make_golden import MEL_SHIFTED_CASES
    return MEL_SHIFTED_CASES


SHIFTED = sorted(_shifted_cases())


@pytest.mark.parametrize("tag", SHIFTED)
def test_oracle_against_reference_get_mel_shifted(golden_dir, tag):
    """keyshift / speed / center (nvSTFT.py:82-116) against the reference's own outputs"""
    ks, sp, ce, which = _shifted_cases()[tag]
    g = np.load(os.path.join(golden_dir, "mel_shifted.npz"))
    m = O.get_mel(g[which], _basis(golden_dir), keyshift=ks, speed=sp, center=ce)
    assert m.shape == g["mel_" + tag].shape
    assert np.abs(m - g["mel_" + tag]).max() <= 1e-5


def test_oracle_against_reference_get_mel_other_configurations(golden_dir):
    g = np.load(os.path.join(golden_dir, "mel_shifted.npz"))
    for tag, kw in (("win1024", {}), ("win1024_ks5", {"keyshift": 5})):           # window centred in a longer transform
        m = O.get_mel(g["audio"], _basis(golden_dir), win_size=1024, **kw)
        assert m.shape == g["mel_" + tag].shape and np.abs(m - g["mel_" + tag]).max() <= 1e-5
    g = np.load(os.path.join(golden_dir, "mel_shifted_22k.npz"))                  # the class's default configuration
    for tag, kw in (("plain", {}), ("ks5", {"keyshift": 5}), ("ksm4_center", {"keyshift": -4, "center": True})):
        m = O.get_mel(g["audio"], g["basis"], 1024, 1024, 256, **kw)
        assert m.shape == g["mel_" + tag].shape and np.abs(m - g["mel_" + tag]).max() <= 1e-5


def test_filterbank_restatement_properties(golden_dir):
    """librosa is not installed, so the Slaney filterbank is checked structurally: the committed basis equals the
    oracle's, rows are single contiguous non-negative bands with their Slaney area normalisation, centre frequencies
    are monotone, and the product-side construction (ddsp_svc_amd.mel) agrees with the oracle's."""
    from ddsp_svc_amd import mel as M
    W = O.mel_filterbank_slaney(44100, 2048, 128, 40, 16000)
    assert np.array_equal(W, _basis(golden_dir))
    assert (W >= 0).all()
    freqs = np.linspace(0, 22050, 1025)
    prev_peak = -1
    for row in W:
        nz = np.nonzero(row)[0]
        assert nz.size and np.array_equal(nz, np.arange(nz[0], nz[-1] + 1))        # one contiguous band
        assert row.argmax() > prev_peak
        prev_peak = row.argmax()
    assert 40 <= freqs[np.nonzero(W[0])[0][0]] <= 80 and freqs[np.nonzero(W[-1])[0][-1]] <= 16000
    W2 = M.slaney_mel_filterbank(44100, 2048, 128, 40, 16000).numpy()
    assert np.abs(W2 - W).max() <= 1e-8
    band, packed = M._bands(torch.from_numpy(W))
    band, packed = band.numpy(), packed.numpy()
    for c, row in enumerate(W):
        nz = np.nonzero(row)[0]
        assert band[c, 0] == nz[0] and band[c, 1] == nz[-1] + 1
        assert np.array_equal(packed[band[c, 2]:band[c, 2] + band[c, 1] - band[c, 0]], row[band[c, 0]:band[c, 1]])


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
@pytest.mark.parametrize("tag", ["a", "t1024", "t512"])
def test_get_mel_golden(dev, golden_dir, tag):
    from ddsp_svc_amd import mel as M
    g = np.load(os.path.join(golden_dir, f"mel_{tag}.npz"))
    stft = M.STFT(**CFG, mel_basis=torch.from_numpy(_basis(golden_dir)))
    out = stft.get_mel(torch.from_numpy(g["audio"]).to(dev))
    assert tuple(out.shape) == g["mel"].shape                                       # [B, n_mels, frames]
    assert out.transpose(1, 2).is_contiguous()                                      # the callers' transpose is free
    _check(out.cpu().numpy(), g["mel"])


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
@pytest.mark.parametrize("tag", SHIFTED)
def test_get_mel_shifted_golden(dev, golden_dir, tag):
    """the chirp-z kernel against the reference's outputs: one and two chunks per frame, an odd transform length, zero-filled
    bins, the scaled hop, torch.stft's centring, the zero-padding branch"""
    from ddsp_svc_amd import mel as M
    ks, sp, ce, which = _shifted_cases()[tag]
    g = np.load(os.path.join(golden_dir, "mel_shifted.npz"))
    stft = M.STFT(**CFG, mel_basis=torch.from_numpy(_basis(golden_dir)))
    out = stft.get_mel(torch.from_numpy(g[which]).to(dev), keyshift=ks, speed=sp, center=ce)
    assert tuple(out.shape) == g["mel_" + tag].shape
    assert out.transpose(1, 2).is_contiguous()
    _check(out.cpu().numpy(), g["mel_" + tag])


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
def test_get_mel_other_configurations_golden(dev, golden_dir):
    """a window shorter than the transform (centred, as torch.stft places it) and the class's default 22.05 kHz configuration
    (n_fft 1024, hop 256, 80 bands: the 2048-point convolution plan), plain and shifted"""
    from ddsp_svc_amd import mel as M
    g = np.load(os.path.join(golden_dir, "mel_shifted.npz"))
    stft = M.STFT(**dict(CFG, win_size=1024), mel_basis=torch.from_numpy(_basis(golden_dir)))
    y = torch.from_numpy(g["audio"]).to(dev)
    _check(stft.get_mel(y).cpu().numpy(), g["mel_win1024"])
    _check(stft.get_mel(y, keyshift=5).cpu().numpy(), g["mel_win1024_ks5"])
    g = np.load(os.path.join(golden_dir, "mel_shifted_22k.npz"))
    stft = M.STFT(mel_basis=torch.from_numpy(g["basis"]))
    y = torch.from_numpy(g["audio"]).to(dev)
    _check(stft.get_mel(y).cpu().numpy(), g["mel_plain"])
    _check(stft.get_mel(y, keyshift=5).cpu().numpy(), g["mel_ks5"])
    _check(stft.get_mel(y, keyshift=-4, center=True).cpu().numpy(), g["mel_ksm4_center"])


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
@pytest.mark.parametrize("B,T,kw", [
    (3, 512 * 9 + 100, {"speed": 0.5}),                       # 2048 points through the chirp-z kernel: exactly one chunk
    (1, 512 * 11 + 7, {"keyshift": 0.0141}),                  # 2050 points: the second chunk holds two samples
    (2, 512 * 14, {"keyshift": 12.01}),                       # 4098 points: three chunks
    (1, 512 * 30, {"keyshift": 24}),                          # 8192 points: four chunks, the kernel's limit
    (2, 300, {"keyshift": -7, "center": True}),               # zeros instead of the reflection, then torch.stft's own
    (1, 512 * 40 + 3, {"keyshift": -2.5, "speed": 2.0}),      # 1772 points, hop 1024: several frames per workgroup run
    (2, 1, {"keyshift": 3}),                                  # one sample: everything else is zero padding
    (2, 2, {"keyshift": -3, "center": True}),
    (1, 5, {"speed": 0.3}),                                   # hop 154
])
def test_get_mel_shifted_shapes(dev, B, T, kw):
    from ddsp_svc_amd import mel as M
    rng = np.random.default_rng(T)
    t = np.arange(T) / 44100.0
    y = (0.4 * np.sin(2 * np.pi * 330.0 * t)[None] + 0.1 * rng.standard_normal((B, T))).astype(np.float32)
    out = M.STFT(**CFG).get_mel(torch.from_numpy(y).to(dev), **kw).cpu().numpy()
    ref = O.get_mel(y, O.mel_filterbank_slaney(44100, 2048, 128, 40, 16000), **kw)
    assert out.shape == ref.shape
    _check(out, ref)


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
@pytest.mark.parametrize("B,T,run", [(1, 512 * 7, 1), (3, 512 * 9 + 100, 2), (1, 700, 4), (2, 300, 4), (1, 512 * 33, 3)])
def test_get_mel_shapes(dev, B, T, run, knobs):
    """odd frame counts, lengths that are not a multiple of the hop, signals shorter than the padding (zero-padding
    branch, nvSTFT.py:99-102), several runs per utterance; the class builds its own Slaney basis here"""
    from ddsp_svc_amd import mel as M
    knobs("MEL_RUN", run)
    rng = np.random.default_rng(T)
    t = np.arange(T) / 44100.0
    y = (0.4 * np.sin(2 * np.pi * 330.0 * t)[None] + 0.1 * rng.standard_normal((B, T))).astype(np.float32)
    stft = M.STFT(**CFG)
    out = stft.get_mel(torch.from_numpy(y).to(dev)).cpu().numpy()
    ref = O.get_mel(y, O.mel_filterbank_slaney(44100, 2048, 128, 40, 16000))
    assert out.shape == ref.shape
    _check(out, ref)


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
def test_get_mel_dense_basis(dev):
    """a basis that is not banded (too many weights to stage on chip) takes the dense path"""
    from ddsp_svc_amd import mel as M
    rng = np.random.default_rng(5)
    W = (rng.random((40, 1025)) * 1e-2).astype(np.float32)
    y = (0.1 * rng.standard_normal((2, 512 * 6))).astype(np.float32)
    cfg = dict(CFG, n_mels=40)
    out = M.STFT(**cfg, mel_basis=torch.from_numpy(W)).get_mel(torch.from_numpy(y).to(dev)).cpu().numpy()
    _check(out, O.get_mel(y, W))


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
def test_get_mel_shifted_dense_basis(dev):
    """more filters than one pass of the projection takes (200 > 128), none of them banded"""
    from ddsp_svc_amd import mel as M
    rng = np.random.default_rng(6)
    W = (rng.random((200, 1025)) * 1e-2).astype(np.float32)
    y = (0.1 * rng.standard_normal((2, 512 * 6))).astype(np.float32)
    cfg = dict(CFG, n_mels=200)
    out = M.STFT(**cfg, mel_basis=torch.from_numpy(W)).get_mel(torch.from_numpy(y).to(dev), keyshift=1.3).cpu().numpy()
    _check(out, O.get_mel(y, W, keyshift=1.3))


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
def test_get_mel_contract(dev):
    from ddsp_svc_amd import mel as M
    stft = M.STFT(**CFG)
    y = torch.zeros(1, 2048, device=dev)
    out = stft.get_mel(y)                                                            # silence: sqrt(1e-9) per bin
    assert torch.isfinite(out).all() and out.shape == (1, 128, 4)
    y2 = torch.randn(1, 8192, generator=torch.Generator().manual_seed(1)).to(dev)
    assert stft.get_mel(y2, keyshift=2).shape == (1, 128, 16) and stft.get_mel(y2, speed=2).shape == (1, 128, 8)
    assert stft.get_mel(y2, center=True).shape == (1, 128, 20)
    assert len(stft._shifted) == 2                                                   # tables per (transform, window) length
    # where the reference raises (torch.stft / F.pad) or the kernel's range ends, so does the drop-in
    for kw in ({"keyshift": 24.1},                                                   # 8239 points: beyond four chunks
               {"speed": 5},                                                         # hop 2560 > window
               {"keyshift": 12, "center": True}):                                    # fine ...
        if kw == {"keyshift": 12, "center": True}:
            assert stft.get_mel(y2, **kw).shape[1] == 128
            continue
        with pytest.raises(RuntimeError):
            stft.get_mel(y2, **kw)
    with pytest.raises(RuntimeError):                                                # window longer than the transform
        M.STFT(44100, 128, 1024, 2048, 512, 40, 16000).get_mel(y2)
    with pytest.raises(RuntimeError):                                                # a basis of more than 1025 bins
        M.STFT(44100, 128, 8192, 8192, 512, 40, 16000).get_mel(y2, keyshift=1)
    with pytest.raises(ValueError):
        stft.get_mel(y2[0], keyshift=1)                                              # [T]: the kernels take [B, T]


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
def test_against_reference_stft_class(dev):
    """the reference's own STFT.get_mel (librosa's filterbank replaced by the oracle's, the only missing piece in this
    image) against the drop-in on the same audio"""
    ref_root = os.environ.get("DDSP_REFERENCE_PATH", "/root/reference")
    if not os.path.isdir(os.path.join(ref_root, "nsf_hifigan")):
        pytest.skip("reference checkout not present (only in the build container)")
    if ref_root not in sys.path:
        sys.path.insert(0, ref_root)
    for name in ["librosa", "librosa.util", "librosa.filters", "librosa.core", "librosa.sequence", "soundfile", "torchaudio",
                 "torchaudio.transforms"]:
        sys.modules.setdefault(name, MagicMock())
    import nsf_hifigan.nvSTFT as nv
    from ddsp_svc_amd import mel as M
    basis = O.mel_filterbank_slaney(44100, 2048, 128, 40, 16000)
    g = torch.Generator().manual_seed(3)
    y = torch.randn(2, 512 * 12, generator=g) * 0.2
    with mock.patch.object(nv, "librosa_mel_fn", side_effect=lambda **kw: basis):
        ref = nv.STFT(44100, 128, 2048, 2048, 512, 40, 16000).get_mel(y)
        ref_shift = nv.STFT(44100, 128, 2048, 2048, 512, 40, 16000).get_mel(y, keyshift=-1.5, speed=1.1, center=True)
    ours = M.STFT(44100, 128, 2048, 2048, 512, 40, 16000).get_mel(y.to(dev)).cpu()
    assert ours.shape == ref.shape
    _check(ours.numpy(), ref.numpy())
    ours = M.STFT(44100, 128, 2048, 2048, 512, 40, 16000).get_mel(y.to(dev), keyshift=-1.5, speed=1.1, center=True).cpu()
    assert ours.shape == ref_shift.shape
    _check(ours.numpy(), ref_shift.numpy())


def test_patch_reference_stft_keeps_cpu_calls_on_the_reference():
    ref_root = os.environ.get("DDSP_REFERENCE_PATH", "/root/reference")
    if not os.path.isdir(os.path.join(ref_root, "nsf_hifigan")):
        pytest.skip("reference checkout not present (only in the build container)")
    if ref_root not in sys.path:
        sys.path.insert(0, ref_root)
    for name in ["librosa", "librosa.util", "librosa.filters", "librosa.core", "librosa.sequence", "soundfile", "torchaudio",
                 "torchaudio.transforms"]:
        sys.modules.setdefault(name, MagicMock())
    import nsf_hifigan.nvSTFT as nv
    from ddsp_svc_amd import mel as M
    basis = O.mel_filterbank_slaney(44100, 2048, 128, 40, 16000)
    y = torch.randn(1, 4096, generator=torch.Generator().manual_seed(1)) * 0.1
    with mock.patch.object(nv, "librosa_mel_fn", side_effect=lambda **kw: basis):
        want = nv.STFT(44100, 128, 2048, 2048, 512, 40, 16000).get_mel(y)
        try:
            M.patch_reference_stft()
            got = nv.STFT(44100, 128, 2048, 2048, 512, 40, 16000).get_mel(y)          # CPU tensor -> reference path
            assert torch.equal(got, want)
        finally:
            nv.STFT.get_mel = nv.STFT._reference_get_mel
            del nv.STFT._reference_get_mel


@pytest.mark.gpu
def test_patched_reference_stft_takes_shifted_calls_on_the_gpu():
    """main_diff.py:359: ``vocoder.extract(seg_ddsp_output, sr, keyshift=formant_shift_key)`` -> ``STFT.get_mel(audio, keyshift)``
    of the patched reference class: a GPU tensor goes through csrc/mel_czt.hip (the call leaves its chirp tables on the
    instance) and agrees with the reference's own code on the same audio; a transform beyond the kernel's range stays on the
    reference's code."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    ref_root = os.environ.get("DDSP_REFERENCE_PATH", "/root/reference")
    if not os.path.isdir(os.path.join(ref_root, "nsf_hifigan")):
        pytest.skip("reference checkout not present (only in the build container)")
    if ref_root not in sys.path:
        sys.path.insert(0, ref_root)
    for name in ["librosa", "librosa.util", "librosa.filters", "librosa.core", "librosa.sequence", "soundfile", "torchaudio",
                 "torchaudio.transforms"]:
        sys.modules.setdefault(name, MagicMock())
    import nsf_hifigan.nvSTFT as nv
    from ddsp_svc_amd import mel as M
    basis = O.mel_filterbank_slaney(44100, 2048, 128, 40, 16000)
    y = torch.randn(2, 512 * 16, generator=torch.Generator().manual_seed(9)) * 0.1
    with mock.patch.object(nv, "librosa_mel_fn", side_effect=lambda **kw: basis):
        want = nv.STFT(44100, 128, 2048, 2048, 512, 40, 16000).get_mel(y, keyshift=3.0)
        try:
            M.patch_reference_stft()
            stft = nv.STFT(44100, 128, 2048, 2048, 512, 40, 16000)
            got = stft.get_mel(y.cuda(), keyshift=3.0)
            assert got.is_cuda and len(stft.__dict__.get("_hip_shifted", {})) == 1
            _check(got.cpu().numpy(), want.numpy())
            far = stft.get_mel(y.cuda(), keyshift=25.0)                              # 8680 points: the reference's operators
            assert far.is_cuda and len(stft._hip_shifted) == 1
        finally:
            nv.STFT.get_mel = nv.STFT._reference_get_mel
            del nv.STFT._reference_get_mel
