"""Drop-in for the waveform -> log-mel front-end of the cascade: ``nsf_hifigan.nvSTFT.STFT``
(nsf_hifigan/nvSTFT.py:60-122), the extractor behind ``Vocoder.extract`` (diffusion/vocoder.py:98-111,146-148)
that turns the DDSP waveform into the conditioning mel of the diffusion / reflow stage.

Same constructor and ``get_mel`` signature.  ``get_mel`` runs on the HIP kernel (csrc/mel.hip) for the
configuration the cascades use (``keyshift == 0``, ``speed == 1``, ``center == False``, ``n_fft == win_size == 2048``,
``hop_length == 512``).  The augmentation variants (key shift: a transform of ``round(n_fft 2^(k/12))`` points, any length
from 1024 to 4096; speed change; ``center``) are off the inference path (training-time augmentation, the enhancer's adaptive
key): the stand-alone class raises ``NotImplementedError`` for them, and ``patch_reference_stft()`` leaves them on the
reference's own code (this package holds no torch-operator restatement of the reference).  The mel basis is what the
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
        if keyshift != 0 or speed != 1 or center or self.n_fft != self.win_size:
            # training-time augmentation / the enhancer's adaptive key (nvSTFT.py:83-85,109-114): a transform of
            # round(n_fft 2^(k/12)) points, a scaled hop, `center`.  No kernel of this package takes them and the package
            # carries no second, torch-operator implementation of the reference: a patched reference class
            # (patch_reference_stft) keeps such calls on the reference's own code.
            raise NotImplementedError("ddsp_svc_amd.mel.STFT.get_mel: keyshift / speed / center / n_fft != win_size are not "
                                      "taken by the HIP kernel; use nsf_hifigan.nvSTFT.STFT (patch_reference_stft() routes only "
                                      "the cascade's inference configuration to the kernel and leaves these to the reference)")
        basis, band, window = self._tables(y.device)
        return mel_spectrogram(y, window, basis, band, self.hop_length, self.clip_val)


def patch_reference_stft():
    """Route ``nsf_hifigan.nvSTFT.STFT.get_mel`` of an importable reference checkout through the HIP kernel when
    the call is the cascade's inference configuration on a GPU tensor (keyshift 0, speed 1, center False,
    n_fft == win == 2048, hop 512); every other call -- CPU tensors, augmentation, other sizes -- keeps the
    reference code.  The reference's own mel basis (librosa) and Hann window caches are used as they are."""
    import nsf_hifigan.nvSTFT as nv
    if hasattr(nv.STFT, "_reference_get_mel"):
        return nv
    ref_get_mel = nv.STFT.get_mel
    nv.STFT._reference_get_mel = ref_get_mel

    def get_mel(self, y, keyshift=0, speed=1, center=False):
        hip_ok = (getattr(y, "is_cuda", False) and keyshift == 0 and speed == 1 and not center and
                  self.n_fft == 2048 and self.win_size == 2048 and self.hop_length == 512 and y.dim() == 2)
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
        return mel_spectrogram(y, self.hann_window[wkey], self.mel_basis[key].contiguous(), bands[key],
                               self.hop_length, self.clip_val)

    nv.STFT.get_mel = get_mel
    return nv
