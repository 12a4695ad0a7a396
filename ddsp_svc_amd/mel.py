"""Drop-in for the waveform -> log-mel front-end of the cascade: ``nsf_hifigan.nvSTFT.STFT``
(nsf_hifigan/nvSTFT.py:60-122), the extractor behind ``Vocoder.extract`` (diffusion/vocoder.py:98-111,146-148)
that turns the DDSP waveform into the conditioning mel of the diffusion / reflow stage.

Same constructor and ``get_mel`` signature.  ``get_mel`` runs on HIP kernels only: csrc/mel.hip for the configuration the
cascades use (``keyshift == 0``, ``speed == 1``, ``center == False``, ``n_fft == win_size == 2048``, ``hop_length == 512``:
a 2048-point FFT, two frames per complex transform), csrc/mel_czt.hip for everything else -- the formant shift of the
cascade's inference (``main_diff.py:359``: ``keyshift = formant_shift_key``) and the pitch augmentation of
``preprocess.py:88-92`` (a transform of ``round(n_fft 2^(k/12))`` points: ANY integer length, a chirp-z transform), a speed
change, ``center=True``, other (n_fft, win, hop) configurations.  There is no torch-operator path.  The mel basis is what the
reference builds with ``librosa.filters.mel`` (nvSTFT.py:90): pass it as ``mel_basis`` (any dense
``[n_mels, n_fft/2+1]`` tensor), or let the class build the Slaney filterbank itself (librosa's published
algorithm, ``htk=False``, ``norm='slaney'``).
"""
import math

import torch

from . import _ffi
from ._ffi import ptr


def slaney_mel_filterbank(sr, n_fft, n_mels, fmin, fmax):
    """``librosa.filters.mel(sr=, n_fft=, n_mels=, fmin=, fmax=)`` with librosa's defaults, float32
    ``[n_mels, n_fft/2+1]`` (what nvSTFT.py:90 requests)."""
    f_sp = 200.0 / 3.0
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = math.log(6.4) / 27.0

    def hz_to_mel(f):
        return min_log_mel + math.log(f / min_log_hz) / logstep if f >= min_log_hz else f / f_sp

    mels = torch.linspace(hz_to_mel(float(fmin)), hz_to_mel(float(fmax)), n_mels + 2, dtype=torch.float64)
    mel_f = torch.where(mels >= min_log_mel, min_log_hz * torch.exp(logstep * (mels - min_log_mel)), f_sp * mels)
    fftfreqs = torch.linspace(0.0, sr / 2.0, n_fft // 2 + 1, dtype=torch.float64)
    fdiff = mel_f[1:] - mel_f[:-1]
    ramps = mel_f[:, None] - fftfreqs[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    weights = torch.clamp(torch.minimum(lower, upper), min=0.0)
    weights = weights * (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return weights.float()


def _bands(basis):
    """Band table of a mel basis: int32 ``[n_mels, 4]`` = (first, one-past-last non-zero bin, offset of the row's
    band in the packed weights, 0) and the packed band weights (all-zero rows get an empty band)."""
    nz = basis != 0
    any_ = nz.any(dim=1)
    n_bins = basis.shape[1]
    idx = torch.arange(n_bins, device=basis.device)
    lo = torch.where(nz, idx, n_bins).min(dim=1).values
    hi = torch.where(nz, idx + 1, 0).max(dim=1).values
    lo = torch.where(any_, lo, torch.zeros_like(lo))
    hi = torch.where(any_, hi, torch.zeros_like(hi))
    length = hi - lo
    off = torch.cumsum(length, 0) - length
    table = torch.stack([lo, hi, off, torch.zeros_like(lo)], dim=1).to(torch.int32).contiguous()
    inband = (idx[None, :] >= lo[:, None]) & (idx[None, :] < hi[:, None])
    packed = basis[inband].contiguous()                        # row-major order == band after band
    return table, packed


def mel_spectrogram(audio, window, mel_basis, band, hop_length, clip_val=1e-5):
    band, packed = band
    """``[B,T]`` waveform -> ``[B, n_mels, frames]`` log-mel (nvSTFT.py:97-116).  The result is laid out
    frame-major in memory, so the ``transpose(1, 2)`` every caller applies (diffusion/vocoder.py:147) is free."""
    _ffi.check_device(audio, window, mel_basis, band, packed)
    if audio.dim() != 2:
        raise ValueError("audio must be [B, T]")
    a = audio if (audio.dtype == torch.float32 and audio.is_contiguous()) else audio.float().contiguous()
    B, T = a.shape
    n_fft = window.numel()
    n_mels = mel_basis.shape[0]
    hop = int(hop_length)
    frames = _ffi.lib().ddsp_hip_mel_frames(T, n_fft, hop) if T > 0 else -1
    if frames < 1:
        raise ValueError("empty audio")
    store = torch.empty(B, frames, n_mels, dtype=torch.float32, device=a.device)
    _ffi.check(_ffi.lib().ddsp_hip_mel_spectrogram(ptr(a), B, T, ptr(window), n_fft, hop, ptr(mel_basis), ptr(band),
                                                   ptr(packed), packed.numel(), n_mels, float(clip_val), ptr(store), frames * n_mels, 1, n_mels,
                                                   _ffi.stream_of(a)))
    return store.transpose(1, 2)


def _shifted_sizes(n_fft, win_size, hop_length, keyshift, speed):
    """nvSTFT.py:82-85 (numpy's rounding, as the reference)."""
    import numpy as np
    factor = 2 ** (keyshift / 12)
    return int(np.round(n_fft * factor)), int(np.round(win_size * factor)), int(np.round(hop_length * speed))


def mel_spectrogram_shifted(audio, tables, band, n_fft_new, win_new, hop_new, center, n_bins, mag_scale, n_mels,
                            clip_val=1e-5):
    """``get_mel`` at any transform length / hop / centring (nvSTFT.py:83-116) -> ``[B, n_mels, frames]``, frame-major in
    memory like ``mel_spectrogram``.  ``tables``: ``shifted_tables(...)``."""
    band, packed = band
    _ffi.check_device(audio, tables, band, packed)
    if audio.dim() != 2:
        raise ValueError("audio must be [B, T]")
    a = audio if (audio.dtype == torch.float32 and audio.is_contiguous()) else audio.float().contiguous()
    B, T = a.shape
    frames = _ffi.lib().ddsp_hip_mel_shifted_frames(T, n_fft_new, win_new, hop_new, int(bool(center))) if T > 0 else -1
    if frames < 1:
        raise RuntimeError("get_mel: no frame (transform %d, window %d, hop %d, center %s against %d samples): torch.stft "
                           "raises here too" % (n_fft_new, win_new, hop_new, bool(center), T))
    store = torch.empty(B, frames, n_mels, dtype=torch.float32, device=a.device)
    _ffi.check(_ffi.lib().ddsp_hip_mel_shifted_spectrogram(ptr(a), B, T, ptr(tables), n_fft_new, win_new, hop_new,
                                                          int(bool(center)), n_bins, float(mag_scale), ptr(band), ptr(packed),
                                                          n_mels, float(clip_val), ptr(store), frames * n_mels, 1, n_mels,
                                                          _ffi.stream_of(a)))
    return store.transpose(1, 2)


def shifted_tables(n_fft_new, win_new, n_bins, device):
    """The chirp tables of one (transform length, window length, basis width) on ``device`` (float32 pairs)."""
    nbytes = _ffi.lib().ddsp_hip_mel_shifted_table_bytes(n_fft_new, n_bins)
    if nbytes == 0 or win_new > n_fft_new or win_new < 1:
        raise RuntimeError("get_mel: transform length %d / window %d with a basis of %d bins is outside the kernel's range "
                           "(n_fft <= 2048, shifted transform <= 8192 points, window <= transform)"
                           % (n_fft_new, win_new, n_bins))
    tab = torch.empty(nbytes // 4, dtype=torch.float32, device=device)
    _ffi.check(_ffi.lib().ddsp_hip_mel_shifted_tables(n_fft_new, win_new, n_bins, ptr(tab), _ffi.stream_of(tab)))
    return tab


class STFT:
    """nsf_hifigan/nvSTFT.py:60-122 on the MI355X."""

    def __init__(self, sr=22050, n_mels=80, n_fft=1024, win_size=1024, hop_length=256, fmin=20, fmax=11025,
                 clip_val=1e-5, mel_basis=None):
        self.target_sr = sr
        self.n_mels = n_mels
        self.n_fft = n_fft
        self.win_size = win_size
        self.hop_length = hop_length
        self.fmin = fmin
        self.fmax = fmax
        self.clip_val = clip_val
        self.mel_basis = {}
        self.hann_window = {}
        self._band = {}
        self._shifted = {}
        self._given_basis = mel_basis

    def _tables(self, device):
        key = str(self.fmax) + "_" + str(device)                                      # nvSTFT.py:87-91
        if key not in self.mel_basis:
            basis = self._given_basis if self._given_basis is not None else slaney_mel_filterbank(
                self.target_sr, self.n_fft, self.n_mels, self.fmin, self.fmax)
            basis = basis.to(device=device, dtype=torch.float32).contiguous()
            self.mel_basis[key] = basis
            self._band[key] = _bands(basis)
        wkey = "0_" + str(device)                                                     # nvSTFT.py:93-95
        if wkey not in self.hann_window:
            self.hann_window[wkey] = torch.hann_window(self.win_size).to(device)
        return self.mel_basis[key], self._band[key], self.hann_window[wkey]

    def get_mel(self, y, keyshift=0, speed=1, center=False):
        basis, band, window = self._tables(y.device)
        if (keyshift == 0 and speed == 1 and not center and self.n_fft == self.win_size == 2048 and self.hop_length == 512):
            return mel_spectrogram(y, window, basis, band, self.hop_length, self.clip_val)
        return _get_mel_shifted(self, y, keyshift, speed, center, band, self._shifted)


def _get_mel_shifted(self, y, keyshift, speed, center, band, cache):
    """nvSTFT.py:82-116 on csrc/mel_czt.hip; ``cache``: chirp tables per (transform, window) length and device (the
    reference caches its windows per keyshift the same way, :92-94)."""
    n_new, win_new, hop_new = _shifted_sizes(self.n_fft, self.win_size, self.hop_length, keyshift, speed)
    n_bins = self.n_fft // 2 + 1
    key = (n_new, win_new, str(y.device))
    if key not in cache:
        cache[key] = shifted_tables(n_new, win_new, n_bins, y.device)
    scale = self.win_size / win_new if keyshift != 0 else 1.0                       # nvSTFT.py:109-114
    return mel_spectrogram_shifted(y, cache[key], band, n_new, win_new, hop_new, center, n_bins, scale, self.n_mels,
                                   self.clip_val)


def _shifted_in_range(n_fft, win_size, hop_length, keyshift, speed):
    n_new, win_new, hop_new = _shifted_sizes(n_fft, win_size, hop_length, keyshift, speed)
    return (1 <= win_new <= n_new and 1 <= hop_new <= win_new and
            _ffi.lib().ddsp_hip_mel_shifted_table_bytes(n_new, n_fft // 2 + 1) > 0)


def patch_reference_stft():
    """Route ``nsf_hifigan.nvSTFT.STFT.get_mel`` of an importable reference checkout through the HIP kernels for every
    call on a ``[B, T]`` GPU tensor the kernels take (the cascade's configuration on csrc/mel.hip; keyshift / speed /
    center / other sizes on csrc/mel_czt.hip); CPU tensors and sizes outside the kernels' range keep the reference
    code.  The reference's own mel basis (librosa) is used as it is."""
    import nsf_hifigan.nvSTFT as nv
    if hasattr(nv.STFT, "_reference_get_mel"):
        return nv
    ref_get_mel = nv.STFT.get_mel
    nv.STFT._reference_get_mel = ref_get_mel

    def get_mel(self, y, keyshift=0, speed=1, center=False):
        plain = (keyshift == 0 and speed == 1 and not center and self.n_fft == 2048 and self.win_size == 2048 and
                 self.hop_length == 512)
        hip_ok = (getattr(y, "is_cuda", False) and y.dim() == 2 and
                  (plain or _shifted_in_range(self.n_fft, self.win_size, self.hop_length, keyshift, speed)))
        if not hip_ok:
            return ref_get_mel(self, y, keyshift=keyshift, speed=speed, center=center)
        key = str(self.fmax) + "_" + str(y.device)
        if key not in self.mel_basis:                                                # nvSTFT.py:87-91
            mel = nv.librosa_mel_fn(sr=self.target_sr, n_fft=self.n_fft, n_mels=self.n_mels, fmin=self.fmin,
                                    fmax=self.fmax)
            self.mel_basis[key] = torch.from_numpy(mel).float().to(y.device)
        wkey = "0_" + str(y.device)
        if wkey not in self.hann_window:                                             # nvSTFT.py:93-95
            self.hann_window[wkey] = torch.hann_window(self.win_size).to(y.device)
        bands = self.__dict__.setdefault("_hip_bands", {})
        if key not in bands:
            bands[key] = _bands(self.mel_basis[key])
        if not plain:
            return _get_mel_shifted(self, y, keyshift, speed, center, bands[key], self.__dict__.setdefault("_hip_shifted", {}))
        return mel_spectrogram(y, self.hann_window[wkey], self.mel_basis[key].contiguous(), bands[key],
                               self.hop_length, self.clip_val)

    nv.STFT.get_mel = get_mel
    return nv
