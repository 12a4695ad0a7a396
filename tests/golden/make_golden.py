"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE on CPU.

Only runs where ``/root/reference`` is mounted (the build container); the GPU box has no
reference, which is why the outputs are committed as small ``.npz`` files.  Nothing here is
imported by the product.  Usage:  ``python tests/golden/make_golden.py``

Fixtures (all float32 unless noted):
  upsample.npz      ddsp.core.upsample                                   core.py:66
  phase.npz         the inlined phase accumulation of Sins/CombSub       vocoder.py:564-575
  filter_*.npz      ddsp.core.frequency_filter, three window modes       core.py:273
  sins_*.npz        Sins.forward with captured controls + injected noise vocoder.py:556-611
  combsub_*.npz     CombSub.forward, same                                vocoder.py:811-862
  fastsrc.npz       CombSubSuperFast.fast_source_gen                     vocoder.py:639-651
  csfast_*.npz      CombSubFast.forward, captured controls, injected uniform noise      vocoder.py:735-786
  cssuper_*.npz     CombSubSuperFast.forward, captured controls, injected normal noise  vocoder.py:653-710
  mel_*.npz         nsf_hifigan.nvSTFT.STFT.get_mel with the oracle's Slaney filterbank injected  nvSTFT.py:73-117
  mel_shifted*.npz  the same with keyshift / speed / center, the 22.05 kHz default configuration, a window shorter than the
                    transform (--mel-shifted regenerates these alone)                          nvSTFT.py:82-116
  sinesrc.npz       nsf_hifigan.models.SourceModuleHnNSF.forward with its two random draws injected  models.py:140-204
  sssloss.npz       ddsp.loss.SSSLoss / RSSLoss forward + autograd w.r.t. x_pred, with a torch.stft stand-in for the absent
                    torchaudio.transforms.Spectrogram (documented semantics restated)                  loss.py:9-54
  cfg1_*.npz        BASELINE cfg 1 exactly (B=1, 5 s, F=431): CombSub 256/128/256, Sins 128/256/256, infer True/False; inputs from
                    seeds, outputs decimated                                                  vocoder.py:556-611, :811-862
  phase_10s.npz     the 10 s (F=862, 441 344-term) phase scan, infer True/False                        vocoder.py:564-575
  filter_n{128,257,512}.npz, combsub_mixed.npz, sins_mixed.npz   tap synthesis, filters and tails at the larger bin counts
  *_grad.npz        autograd of Sins / CombSub / CombSubFast / CombSubSuperFast.forward w.r.t. the controls Unit2Control produced,
                    for a random cotangent R: d(sum(signal * R)) / d ctrl
"""
import os
import sys
from unittest import mock
from unittest.mock import MagicMock

import numpy as np
import torch

REF = os.environ.get("DDSP_REFERENCE_PATH", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def import_reference():
    sys.path.insert(0, REF)
    for m in ["transformers", "pyworld", "parselmouth", "torchcrepe", "resampy", "fairseq",
              "torchaudio", "torchaudio.transforms", "gin", "local_attention", "librosa",
              "librosa.sequence", "librosa.util", "librosa.filters", "librosa.core", "soundfile"]:
        sys.modules.setdefault(m, MagicMock())
    import ddsp.core as core
    import ddsp.vocoder as vocoder
    return core, vocoder


class SpectrogramStandIn(torch.nn.Module):
    """torchaudio.transforms.Spectrogram as ddsp/loss.py:20 uses it (torchaudio is not installed here): periodic Hann
    window of n_fft, torch.stft without centring or padding, one-sided, ``normalized=True`` = division by the
    window's L2 norm, ``power=1`` = magnitude  (torchaudio/functional/functional.py, spectrogram())."""

    def __init__(self, n_fft, hop_length, power, normalized, center):
        super().__init__()
        assert power == 1 and normalized is True and center is False
        self.n_fft, self.hop_length = n_fft, hop_length
        self.register_buffer("window", torch.hann_window(n_fft))

    def forward(self, x):
        z = torch.stft(x, self.n_fft, hop_length=self.hop_length, win_length=self.n_fft, window=self.window,
                       center=False, normalized=False, onesided=True, return_complex=True)
        return (z / self.window.pow(2.).sum().sqrt()).abs()


def loss_fixture():
    """ddsp/loss.py run as is, on top of the Spectrogram stand-in."""
    import_reference()
    sys.modules["torchaudio"].transforms.Spectrogram = SpectrogramStandIn
    sys.modules.pop("ddsp.loss", None)
    import ddsp.loss as L
    g = torch.Generator().manual_seed(71)
    t = torch.arange(6000) / 44100.0
    tone = sum(0.2 / k * torch.sin(2 * np.pi * 196.0 * k * t + 0.7 * k) for k in range(1, 20))
    x_true = torch.stack([tone + 0.02 * torch.randn(6000, generator=g), 0.1 * torch.randn(6000, generator=g),
                          torch.roll(tone, 37) * 0.5])
    x_pred = x_true * 0.8 + 0.05 * torch.randn(3, 6000, generator=g)
    x_pred[2, 1000:1400] = 0.0
    out = {"x_true": x_true.numpy(), "x_pred": x_pred.numpy()}
    cases = [(111, 1.0, 0.0), (256, 1.0, 0.75), (777, 0.5, 0.0), (2047, 1.0, 0.0), (1024, 1.0, 0.5)]
    out["cases"] = np.array(cases, np.float64)
    for i, (n_fft, alpha, overlap) in enumerate(cases):
        f = L.SSSLoss(int(n_fft), alpha, overlap)
        xp = x_pred.clone().requires_grad_(True)
        loss = f(x_true, xp)
        loss.backward()
        out[f"loss{i}"] = loss.detach().numpy()
        out[f"grad{i}"] = xp.grad.numpy()
    torch.manual_seed(72)
    rss = L.RSSLoss(256, 300, 4, device="cpu")
    drawn = torch.randint(256, 300, (4,), generator=torch.Generator().manual_seed(73))
    with mock.patch("torch.randint", side_effect=lambda *a, **k: drawn):
        xp = x_pred.clone().requires_grad_(True)
        loss = rss(xp, x_true)
        loss.backward()
    out["rss_sizes"] = drawn.numpy()
    out["rss_loss"] = loss.detach().numpy()
    out["rss_grad"] = xp.grad.numpy()
    np.savez(os.path.join(HERE, "sssloss.npz"), **out)


DECIM = 97          # prime: the decimated samples walk through every position inside a hop block


def input_checks(a):
    """A few numbers that identify a regenerated input array (sum, sum of squares, first / last values): the BASELINE-
    shape fixtures keep their large inputs as seeds, and the tests refuse to compare if the regenerated array differs."""
    a = np.asarray(a, np.float64).reshape(-1)
    return np.array([a.sum(), np.square(a).sum(), a[0], a[1], a[a.size // 2], a[-1]], np.float64)


def summarise(y, hop=512, windows=((0, 1536), (100 * 512 - 300, 100 * 512 + 900))):
    """Small-on-disk view of a [B,T] waveform: every DECIM-th sample, the per-frame RMS, two contiguous stretches (the
    start with its zero-state transient, one interior stretch across frame boundaries)."""
    y = np.asarray(y)
    B, T = y.shape
    out = {"dec": y[:, ::DECIM].copy(), "frame_rms": np.sqrt(np.mean(np.square(y.astype(np.float64)).reshape(B, T // hop, hop), -1)).astype(np.float32)}
    for i, (a, b) in enumerate(windows):
        a, b = max(0, min(a, T)), max(0, min(b, T))
        out[f"win{i}"] = y[:, a:b].copy()
        out[f"win{i}_range"] = np.array([a, b], np.int64)
    return out


class FixedControls(torch.nn.Module):
    """stands in for Unit2Control inside the UNMODIFIED reference module: returns the drawn controls (the DSP tail is
    what is being pinned; ddsp/unit2control.py:107-109 returns ``(controls dict, hidden)``)"""

    def __init__(self, ctrls):
        super().__init__()
        self.ctrls = ctrls

    def forward(self, units, f0, phase, volume, spk_id=None, spk_mix_dict=None):
        self.phase_frames = phase
        return self.ctrls, torch.zeros(units.shape[0], units.shape[1], 1)


def baseline_shape_fixtures():
    """Reference-pinned fixtures at the BASELINE.json shapes that round 1 covered only transitively (missing #2
    and #7): cfg 1 exactly (B = 1, 5 s, F = 431: CombSub 256/128/256 and Sins 128/256/256 = configs/sins.yaml:20-22), the
    10 s phase scan (F = 862, vocoder.py:564-575, infer True / False), and the tap synthesis + tails at n_mag 128 / 257 /
    512.  Large inputs are regenerated from seeds (oracle.synth_*: numpy PCG64 streams) and identified by input_checks;
    large outputs are stored decimated."""
    from oracle import ddsp_oracle as O
    core, V = import_reference()
    sr, hop = 44100, 512

    # ---- cfg 1: B = 1, 5 s ------------------------------------------------------------------------------------------
    F1 = 5 * sr // hop + 1                                    # vocoder.py:222 frame rule -> 431
    assert F1 == 431
    for kind, sizes, seed in (("combsub", (256, 128, 256), 910), ("sins", (128, 256, 256), 920)):
        for infer in (True, False):
            f0 = O.synth_f0(1, F1, sr, hop, seed=seed)
            if kind == "combsub":
                f0 = np.clip(f0 * np.float32(1.7), 65, 800).astype(np.float32)     # crosses 259 Hz: dynamic-window clamp quirk
            ctrls = O.synth_controls(1, F1, sizes, seed=seed + 1, scale=0.7)
            noise = O.synth_noise(1, F1 * hop, seed=seed + 2)
            if kind == "sins":
                model = V.Sins(sr, hop, *sizes, n_unit=8, n_spk=1).eval()
                names = ("amplitudes", "group_delay", "noise_magnitude")
            else:
                model = V.CombSub(sr, hop, *sizes, n_unit=8, n_spk=1).eval()
                names = ("group_delay", "harmonic_magnitude", "noise_magnitude")
            model.unit2ctrl = FixedControls({k: torch.from_numpy(c) for k, c in zip(names, ctrls)})
            u01 = torch.from_numpy((noise + np.float32(1)) / np.float32(2))            # exact inverse of 2u-1 on this grid
            assert np.array_equal((u01 * 2 - 1).numpy(), noise)
            with torch.no_grad(), mock.patch("torch.rand_like", side_effect=lambda t: u01.to(t)):
                signal, _, (harm, nz) = model(torch.zeros(1, F1, 8), torch.from_numpy(f0), torch.zeros(1, F1, 1), infer=infer)
            out = dict(f0_frames=f0, sizes=np.array(sizes), seeds=np.array([seed + 1, seed + 2]), ctrl_scale=np.float64(0.7),
                       decim=np.int64(DECIM), phase_frames=model.unit2ctrl.phase_frames.numpy()[..., 0],
                       noise_check=input_checks(noise))
            for k, c in zip(names, ctrls):
                out["check_" + k] = input_checks(c)
            for key, y in (("signal", signal), ("harmonic", harm), ("noise_out", nz)):
                for kk, vv in summarise(y.numpy()).items():
                    out[f"{key}_{kk}"] = vv
            np.savez_compressed(os.path.join(HERE, f"cfg1_{kind}_infer{int(infer)}.npz"), **out)

    # ---- 10 s phase scan: F = 862, 441 344-term cumsum ---------------------------------------------------------------
    F2 = 10 * sr // hop + 1
    assert F2 == 862
    f0 = O.synth_f0(3, F2, sr, hop, seed=930)
    f0[1] = 800.0                                             # largest accumulated phase: 8 000 cycles
    f0[2] = np.clip(f0[2] * np.float32(0.4), 65, 800)
    f0t = torch.from_numpy(f0)
    out = {"f0_frames": f0, "decim": np.int64(DECIM)}
    for infer in (True, False):
        f0u = core.upsample(f0t, hop)
        x = torch.cumsum(f0u.double() / sr, axis=1) if infer else torch.cumsum(f0u / torch.tensor(sr), axis=1)
        x = x - torch.round(x)
        x = x.to(f0u)
        phase = 2 * np.pi * x
        out[f"x_dec_infer{int(infer)}"] = x[:, ::DECIM, 0].numpy()
        out[f"x_tail_infer{int(infer)}"] = x[:, -2048:, 0].numpy()
        out[f"phase_frames_infer{int(infer)}"] = phase[:, ::hop, 0].numpy()
    np.savez_compressed(os.path.join(HERE, "phase_10s.npz"), **out)

    # ---- tap synthesis + filters at n_mag 128 / 257 / 512 ------------------------------------------------------------
    for n_mag, Fr in ((128, 7), (257, 6), (512, 5)):
        g = torch.Generator().manual_seed(300 + n_mag)
        B = 2
        T = Fr * hop
        audio = torch.rand(B, T, generator=g) * 2 - 1
        c = torch.randn(B, Fr, n_mag, generator=g)
        f0f = torch.from_numpy(O.synth_f0(B, Fr, sr, hop, seed=n_mag))
        f0f[1] = f0f[1] * 0 + 640.0
        gd = np.pi * torch.tanh(c)
        ap = torch.exp(1.j * torch.cumsum(gd, axis=-1))
        mag = torch.exp(c)
        zmag = torch.complex(mag, torch.zeros_like(mag))
        hw = 1.5 * sr / (f0f + 1e-3)
        np.savez_compressed(os.path.join(HERE, f"filter_n{n_mag}.npz"),
                            audio=audio.numpy(), ctrl=c.numpy(), f0_frames=f0f.numpy(), half_width=hw.numpy()[..., 0],
                            y_roll=core.frequency_filter(audio, ap, hann_window=False).numpy(),
                            y_hann=core.frequency_filter(audio, zmag, hann_window=True).numpy(),
                            y_dyn=core.frequency_filter(audio, zmag, hann_window=True, half_width_frames=hw).numpy(),
                            ir_roll=core.frequency_impulse_response(ap, hann_window=False).numpy(),
                            ir_hann=core.frequency_impulse_response(zmag).numpy(),
                            ir_dyn=core.frequency_impulse_response(zmag, half_width_frames=hw).numpy())

    # ---- module tails with mixed bin counts, through the real Unit2Control (captured controls) ------------------------
    def run_module(kind, sizes, B, Fr, seed):
        torch.manual_seed(seed)
        model = (V.Sins if kind == "sins" else V.CombSub)(sr, hop, sizes[0], sizes[1], sizes[2], n_unit=64, n_spk=1).eval()
        with torch.no_grad():
            model.unit2ctrl.dense_out.weight_g.mul_(4.0)
        g = torch.Generator().manual_seed(seed + 1)
        units = torch.randn(B, Fr, 64, generator=g)
        f0f = torch.from_numpy(O.synth_f0(B, Fr, sr, hop, seed=seed + 2))
        f0f[0] = torch.clamp(f0f[0] * 2.2, 65, 800)
        vol = torch.rand(B, Fr, 1, generator=g) * 0.1
        u01 = torch.rand(B, Fr * hop, generator=g)
        cap = {}
        hk = model.unit2ctrl.register_forward_hook(lambda mod, i, o: cap.update(ctrls=o[0]))
        with torch.no_grad(), mock.patch("torch.rand_like", side_effect=lambda t: u01.to(t)):
            signal, hidden, (harm, nz) = model(units, f0f, vol, infer=True)
        hk.remove()
        return dict(f0_frames=f0f.numpy(), noise=(u01 * 2 - 1).numpy(), signal=signal.numpy(), harmonic=harm.numpy(),
                    noise_out=nz.numpy(), sizes=np.array(sizes),
                    **{"ctrl_" + k: v.detach().numpy() for k, v in cap["ctrls"].items()})

    np.savez_compressed(os.path.join(HERE, "combsub_mixed.npz"), **run_module("combsub", (257, 128, 512), 2, 9, 940))
    np.savez_compressed(os.path.join(HERE, "sins_mixed.npz"), **run_module("sins", (64, 512, 128), 1, 9, 950))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz") and (f.startswith(("cfg1_", "phase_10s", "combsub_mixed", "sins_mixed")) or f[8:-4] in ("128", "257", "512")):
            print(f, os.path.getsize(os.path.join(HERE, f)))


MEL_SHIFTED_CASES = {                    # tag: (keyshift, speed, center, which audio)
    "ks3p7": (3.7, 1, False, "audio"),          # 2536 points: two chunks of the chirp-z kernel
    "ksm5": (-5, 1, False, "audio"),            # 1534 points: 768 bins, the rest zero-filled (nvSTFT.py:111-113)
    "ks2": (2, 1, False, "audio"),              # 2299 points: an odd length
    "ks12": (12, 1, False, "audio"),            # 4096 points
    "ksm12": (-12, 1, False, "audio"),          # 1024 points
    "sp1p25": (0, 1.25, False, "audio"),        # hop 640
    "center": (0, 1, True, "audio"),
    "mix": (-3.3, 0.8, True, "audio"),          # 1693 points, hop 410, centred
    "short": (4, 1, False, "audio_short"),      # signal shorter than the padding: zeros instead of the reflection (:99-102)
    "short_center": (-2, 1.5, True, "audio_short"),
}


def mel_shifted_fixtures():
    """mel_shifted*.npz: nsf_hifigan.nvSTFT.STFT.get_mel with keyshift / speed / center (nvSTFT.py:82-116), the oracle's
    Slaney filterbank injected for the absent librosa; the 44.1 kHz configuration, the class's default 22.05 kHz one
    (n_fft 1024, hop 256, 80 bands) and a window shorter than the transform."""
    from oracle import ddsp_oracle as O
    import_reference()
    import nsf_hifigan.nvSTFT as nv

    def audio(Tn, seed):
        g = torch.Generator().manual_seed(seed)
        t = torch.arange(Tn) / 44100.0
        y = sum(0.3 / k * torch.sin(2 * np.pi * 220.0 * k * t + k) for k in range(1, 40))
        return torch.stack([y + 0.05 * torch.randn(Tn, generator=g), 0.2 * torch.randn(Tn, generator=g)])

    ys = {"audio": audio(20 * 512, 77), "audio_short": audio(700, 78)}
    basis = O.mel_filterbank_slaney(44100, 2048, 128, 40, 16000)
    out = {k: v.numpy() for k, v in ys.items()}
    with mock.patch.object(nv, "librosa_mel_fn", side_effect=lambda **kw: basis):
        stft = nv.STFT(44100, 128, 2048, 2048, 512, 40, 16000)
        for tag, (ks, sp, ce, which) in MEL_SHIFTED_CASES.items():
            out["mel_" + tag] = stft.get_mel(ys[which], keyshift=ks, speed=sp, center=ce).numpy()
        short_win = nv.STFT(44100, 128, 2048, 1024, 512, 40, 16000)               # window centred in the transform
        out["mel_win1024"] = short_win.get_mel(ys["audio"]).numpy()
        out["mel_win1024_ks5"] = short_win.get_mel(ys["audio"], keyshift=5).numpy()
    np.savez_compressed(os.path.join(HERE, "mel_shifted.npz"), **out)
    basis22 = O.mel_filterbank_slaney(22050, 1024, 80, 20, 11025)
    out = {"audio": ys["audio"].numpy()[:, :12 * 256], "basis": basis22}
    y22 = ys["audio"][:, :12 * 256]
    with mock.patch.object(nv, "librosa_mel_fn", side_effect=lambda **kw: basis22):
        stft = nv.STFT()                                                           # 22050, 80, 1024, 1024, 256, 20, 11025
        out["mel_plain"] = stft.get_mel(y22).numpy()
        out["mel_ks5"] = stft.get_mel(y22, keyshift=5).numpy()                     # 1367 points
        out["mel_ksm4_center"] = stft.get_mel(y22, keyshift=-4, center=True).numpy()
    np.savez_compressed(os.path.join(HERE, "mel_shifted_22k.npz"), **out)
    for f in ("mel_shifted.npz", "mel_shifted_22k.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)))


def main():
    from oracle import ddsp_oracle as O
    if "--only-loss" in sys.argv:
        return loss_fixture()
    if "--mel-shifted" in sys.argv:
        return mel_shifted_fixtures()
    if "--baseline-shapes" in sys.argv:
        return baseline_shape_fixtures()
    core, V = import_reference()
    torch.manual_seed(0)
    sr, hop = 44100, 512

    # ---- upsample -----------------------------------------------------------------
    g = torch.Generator().manual_seed(11)
    sig = torch.rand(2, 9, 3, generator=g) * 700 + 65
    np.savez(os.path.join(HERE, "upsample.npz"), sig=sig.numpy(), hop=np.int64(hop),
             out=core.upsample(sig, hop).numpy())
    sig = torch.randn(1, 5, 1, generator=g)
    np.savez(os.path.join(HERE, "upsample_hop64.npz"), sig=sig.numpy(), hop=np.int64(64),
             out=core.upsample(sig, 64).numpy())

    # ---- phase ----------------------------------------------------------------------
    f0 = torch.from_numpy(O.synth_f0(3, 40, sr, hop, seed=5))
    ip = torch.tensor([0.3, -2.0, 5.5]).reshape(3, 1, 1)
    out = {"f0_frames": f0.numpy(), "initial_phase": ip.numpy().reshape(3)}
    for infer in (True, False):
        for use_ip in (False, True):
            f0u = core.upsample(f0, hop)
            if infer:
                x = torch.cumsum(f0u.double() / sr, axis=1)
            else:
                x = torch.cumsum(f0u / torch.tensor(sr), axis=1)
            if use_ip:
                x += ip.to(x) / 2 / np.pi
            x = x - torch.round(x)
            x = x.to(f0u)
            phase = 2 * np.pi * x
            tag = f"infer{int(infer)}_ip{int(use_ip)}"
            out["x_" + tag] = x.squeeze(-1).numpy()
            out["phase_frames_" + tag] = phase[:, ::hop, 0].numpy()
    np.savez(os.path.join(HERE, "phase.npz"), **out)

    # ---- frequency_filter, three modes -------------------------------------------------
    for n_mag, Fr, h in ((65, 10, 512), (129, 7, 256), (256, 12, 512)):
        g = torch.Generator().manual_seed(100 + n_mag)
        B = 2
        T = Fr * h
        audio = torch.rand(B, T, generator=g) * 2 - 1
        c = torch.randn(B, Fr, n_mag, generator=g)
        f0f = torch.from_numpy(O.synth_f0(B, Fr, sr, h, seed=n_mag))
        f0f[1] = f0f[1] * 0 + 500.0          # constant high f0: exercises the w>1 clamp quirk
        gd = np.pi * torch.tanh(c)
        ap = torch.exp(1.j * torch.cumsum(gd, axis=-1))
        mag = torch.exp(c)
        hw = 1.5 * sr / (f0f + 1e-3)
        np.savez(os.path.join(HERE, f"filter_n{n_mag}.npz"),
                 audio=audio.numpy(), ctrl=c.numpy(), f0_frames=f0f.numpy(),
                 half_width=hw.numpy()[..., 0],
                 resp_re=ap.real.numpy(), resp_im=ap.imag.numpy(), mag=mag.numpy(),
                 y_roll=core.frequency_filter(audio, ap, hann_window=False).numpy(),
                 y_hann=core.frequency_filter(audio, torch.complex(mag, torch.zeros_like(mag)),
                                              hann_window=True).numpy(),
                 y_dyn=core.frequency_filter(audio, torch.complex(mag, torch.zeros_like(mag)),
                                             hann_window=True, half_width_frames=hw).numpy(),
                 ir_roll=core.frequency_impulse_response(ap, hann_window=False).numpy(),
                 ir_hann=core.frequency_impulse_response(torch.complex(mag, torch.zeros_like(mag))).numpy(),
                 ir_dyn=core.frequency_impulse_response(torch.complex(mag, torch.zeros_like(mag)),
                                                        half_width_frames=hw).numpy())

    # ---- module forwards with captured controls and injected noise ------------------------
    def run_module(kind, sizes, B, Fr, seed, infer=True):
        torch.manual_seed(seed)
        if kind == "sins":
            model = V.Sins(sr, hop, sizes[0], sizes[1], sizes[2], n_unit=64, n_spk=1).eval()
        else:
            model = V.CombSub(sr, hop, sizes[0], sizes[1], sizes[2], n_unit=64, n_spk=1).eval()
        # random-init Unit2Control emits small controls; widen them so exp()/tanh() are exercised
        with torch.no_grad():
            model.unit2ctrl.dense_out.weight_g.mul_(4.0)
        g = torch.Generator().manual_seed(seed + 1)
        units = torch.randn(B, Fr, 64, generator=g)
        f0f = torch.from_numpy(O.synth_f0(B, Fr, sr, hop, seed=seed + 2))
        f0f[0] = torch.clamp(f0f[0] * 2.2, 65, 800)      # one utterance well above 259 Hz
        vol = torch.rand(B, Fr, 1, generator=g) * 0.1
        u01 = torch.rand(B, Fr * hop, generator=g)
        cap = {}
        hk = model.unit2ctrl.register_forward_hook(lambda mod, i, o: cap.update(ctrls=o[0]))
        with torch.no_grad(), mock.patch("torch.rand_like", side_effect=lambda t: u01.to(t)):
            signal, hidden, (harm, nz) = model(units, f0f, vol, infer=infer)
        hk.remove()
        ctrls = {k: v.detach().numpy() for k, v in cap["ctrls"].items()}
        return dict(f0_frames=f0f.numpy(), noise=(u01 * 2 - 1).numpy(),
                    signal=signal.numpy(), harmonic=harm.numpy(), noise_out=nz.numpy(),
                    sizes=np.array(sizes), **{"ctrl_" + k: v for k, v in ctrls.items()})

    np.savez(os.path.join(HERE, "sins_h256.npz"), **run_module("sins", (256, 256, 256), 2, 24, 7))
    np.savez(os.path.join(HERE, "sins_h128.npz"), **run_module("sins", (128, 256, 256), 1, 16, 8))
    np.savez(os.path.join(HERE, "sins_h40_train.npz"), **run_module("sins", (40, 65, 129), 2, 10, 9, infer=False))
    np.savez(os.path.join(HERE, "combsub_256.npz"), **run_module("combsub", (256, 256, 256), 2, 24, 17))
    np.savez(os.path.join(HERE, "combsub_128.npz"), **run_module("combsub", (256, 128, 256), 1, 16, 18))
    np.savez(os.path.join(HERE, "combsub_small_train.npz"), **run_module("combsub", (65, 129, 33), 2, 10, 19, infer=False))

    # ---- CombSubFast / CombSubSuperFast ----------------------------------
    def run_fast(kind, B, Fr, seed, infer=True, gain=4.0):
        torch.manual_seed(seed)
        if kind == "fast":
            model = V.CombSubFast(sr, hop, n_unit=64, n_spk=1).eval()
        else:
            model = V.CombSubSuperFast(sr, hop, 2048, n_unit=64, n_spk=1).eval()
        with torch.no_grad():
            model.unit2ctrl.dense_out.weight_g.mul_(gain)
        g = torch.Generator().manual_seed(seed + 1)
        units = torch.randn(B, Fr, 64, generator=g)
        f0f = torch.from_numpy(O.synth_f0(B, Fr, sr, hop, seed=seed + 2))
        f0f[0] = torch.clamp(f0f[0] * 2.2, 65, 800)
        vol = torch.rand(B, Fr, 1, generator=g) * 0.1
        if kind == "fast":
            draw = torch.rand(B, Fr * hop, generator=g)
            patch = mock.patch("torch.rand_like", side_effect=lambda t: draw.to(t))
            noise = draw * 2 - 1
        else:
            draw = torch.randn(B, Fr * hop, generator=g)
            patch = mock.patch("torch.randn_like", side_effect=lambda t: draw.to(t))
            noise = draw
        cap = {}
        hk = model.unit2ctrl.register_forward_hook(lambda mod, i, o: cap.update(ctrls=o[0], phase=i[2]))
        with torch.no_grad(), patch:
            signal, hidden, _ = model(units, f0f, vol, infer=infer)
        hk.remove()
        ctrls = {k: v.detach().numpy() for k, v in cap["ctrls"].items()}
        out = dict(f0_frames=f0f.numpy(), noise=noise.numpy(), signal=signal.numpy(),
                   phase_frames=cap["phase"].detach().numpy()[..., 0], window=model.window.numpy(),
                   **{"ctrl_" + k: v for k, v in ctrls.items()})
        if kind == "super":
            with torch.no_grad():
                comb, pf = model.fast_source_gen(f0f)
            out["combtooth"] = comb.numpy()
        return out

    f0f = torch.from_numpy(O.synth_f0(3, 60, sr, hop, seed=31))
    f0f[1] = torch.clamp(f0f[1] * 2.5, 65, 800)
    f0f[2, 20:30] = 65.0
    torch.manual_seed(3)
    m = V.CombSubSuperFast(sr, hop, 2048, n_unit=64, n_spk=1)
    with torch.no_grad():
        comb, pf = m.fast_source_gen(f0f)
    np.savez(os.path.join(HERE, "fastsrc.npz"), f0_frames=f0f.numpy(), combtooth=comb.numpy(),
             phase_frames=pf.numpy()[..., 0])
    np.savez(os.path.join(HERE, "csfast_a.npz"), **run_fast("fast", 2, 20, 41))
    np.savez(os.path.join(HERE, "csfast_train.npz"), **run_fast("fast", 1, 9, 42, infer=False))
    np.savez(os.path.join(HERE, "cssuper_a.npz"), **run_fast("super", 2, 20, 43))
    np.savez(os.path.join(HERE, "cssuper_short.npz"), **run_fast("super", 1, 2, 44))     # T <= win/2: zero padding
    np.savez(os.path.join(HERE, "cssuper_f3.npz"), **run_fast("super", 1, 3, 45))        # shortest reflect case
    # ---- gradients of the spectral tails w.r.t. the controls -----------------
    def run_grad(kind, B, Fr, seed):
        torch.manual_seed(seed)
        model = (V.CombSubFast(sr, hop, n_unit=64, n_spk=1) if kind == "fast"
                 else V.CombSubSuperFast(sr, hop, 2048, n_unit=64, n_spk=1)).eval()
        with torch.no_grad():
            model.unit2ctrl.dense_out.weight_g.mul_(2.0)
        g = torch.Generator().manual_seed(seed + 1)
        units = torch.randn(B, Fr, 64, generator=g)
        f0f = torch.from_numpy(O.synth_f0(B, Fr, sr, hop, seed=seed + 2))
        vol = torch.rand(B, Fr, 1, generator=g) * 0.1
        draw = torch.rand(B, Fr * hop, generator=g) if kind == "fast" else torch.randn(B, Fr * hop, generator=g)
        R = torch.randn(B, Fr * hop, generator=g)
        cap = {}

        def hook(mod, i, o):
            for v in o[0].values():
                v.retain_grad()
            cap.update(ctrls=o[0])
        hk = model.unit2ctrl.register_forward_hook(hook)
        name = "torch.rand_like" if kind == "fast" else "torch.randn_like"
        with mock.patch(name, side_effect=lambda t: draw.to(t)):
            signal, _, _ = model(units, f0f, vol, infer=True)
        hk.remove()
        (signal * R).sum().backward()
        out = dict(f0_frames=f0f.numpy(), noise=(draw * 2 - 1 if kind == "fast" else draw).numpy(), cotangent=R.numpy(),
                   signal=signal.detach().numpy(), window=model.window.numpy())
        for k, v in cap["ctrls"].items():
            out["ctrl_" + k] = v.detach().numpy()
            out["grad_" + k] = v.grad.numpy()
        return out

    def run_grad_combsub(B, Fr, seed, sizes=(65, 129, 65)):
        torch.manual_seed(seed)
        model = V.CombSub(sr, hop, sizes[0], sizes[1], sizes[2], n_unit=64, n_spk=1).eval()
        with torch.no_grad():
            model.unit2ctrl.dense_out.weight_g.mul_(3.0)
        g = torch.Generator().manual_seed(seed + 1)
        units = torch.randn(B, Fr, 64, generator=g)
        f0f = torch.from_numpy(O.synth_f0(B, Fr, sr, hop, seed=seed + 2))
        f0f[0] = torch.clamp(f0f[0] * 2.2, 65, 800)
        vol = torch.rand(B, Fr, 1, generator=g) * 0.1
        u01 = torch.rand(B, Fr * hop, generator=g)
        R = torch.randn(B, Fr * hop, generator=g)
        cap = {}

        def hook(mod, i, o):
            for v in o[0].values():
                v.retain_grad()
            cap.update(ctrls=o[0])
        hk = model.unit2ctrl.register_forward_hook(hook)
        with mock.patch("torch.rand_like", side_effect=lambda t: u01.to(t)):
            signal, _, _ = model(units, f0f, vol, infer=True)
        hk.remove()
        (signal * R).sum().backward()
        out = dict(f0_frames=f0f.numpy(), noise=(u01 * 2 - 1).numpy(), cotangent=R.numpy(),
                   signal=signal.detach().numpy(), sizes=np.array(sizes))
        for k, v in cap["ctrls"].items():
            out["ctrl_" + k] = v.detach().numpy()
            out["grad_" + k] = v.grad.numpy()
        return out

    def run_grad_sins(B, Fr, seed, sizes=(48, 65, 33)):
        torch.manual_seed(seed)
        model = V.Sins(sr, hop, sizes[0], sizes[1], sizes[2], n_unit=64, n_spk=1).eval()
        with torch.no_grad():
            model.unit2ctrl.dense_out.weight_g.mul_(3.0)
        g = torch.Generator().manual_seed(seed + 1)
        units = torch.randn(B, Fr, 64, generator=g)
        f0f = torch.from_numpy(O.synth_f0(B, Fr, sr, hop, seed=seed + 2))
        f0f[0] = torch.clamp(f0f[0] * 2.2, 65, 800)
        vol = torch.rand(B, Fr, 1, generator=g) * 0.1
        u01 = torch.rand(B, Fr * hop, generator=g)
        R = torch.randn(B, Fr * hop, generator=g)
        cap = {}

        def hook(mod, i, o):
            for v in o[0].values():
                v.retain_grad()
            cap.update(ctrls=o[0])
        hk = model.unit2ctrl.register_forward_hook(hook)
        with mock.patch("torch.rand_like", side_effect=lambda t: u01.to(t)):
            signal, _, _ = model(units, f0f, vol, infer=True)
        hk.remove()
        (signal * R).sum().backward()
        out = dict(f0_frames=f0f.numpy(), noise=(u01 * 2 - 1).numpy(), cotangent=R.numpy(),
                   signal=signal.detach().numpy(), sizes=np.array(sizes))
        for k, v in cap["ctrls"].items():
            out["ctrl_" + k] = v.detach().numpy()
            out["grad_" + k] = v.grad.numpy()
        return out

    np.savez(os.path.join(HERE, "sins_grad.npz"), **run_grad_sins(2, 8, 54))
    np.savez(os.path.join(HERE, "combsub_grad.npz"), **run_grad_combsub(2, 8, 53))
    np.savez(os.path.join(HERE, "cssuper_grad.npz"), **run_grad("super", 2, 9, 51))
    np.savez(os.path.join(HERE, "csfast_grad.npz"), **run_grad("fast", 2, 8, 52))

    # ---- harmonic source of NSF-HiFiGAN --------------------------------------
    import nsf_hifigan.models as nm
    torch.manual_seed(61)
    src = nm.SourceModuleHnNSF(44100, harmonic_num=8)
    Bs, Ls = 2, 24
    f0s = torch.from_numpy(O.synth_f0(Bs, Ls, sr, hop, seed=62))[..., 0].clone()
    f0s[0, 5:9] = 0.0                                          # unvoiced stretches (uv mask, noise amplitude switch)
    f0s[1, 0:2] = 0.0
    f0s[1, -1] = 0.0
    g = torch.Generator().manual_seed(63)
    ri = torch.rand(1, 1, 9, generator=g)
    nzs = torch.randn(Bs, Ls * hop, 9, generator=g)
    with mock.patch("torch.rand", side_effect=lambda *a, **k: ri.clone()), \
            mock.patch("torch.randn_like", side_effect=lambda t: nzs), torch.no_grad():
        merged = src(f0s, hop)
    ri0 = ri.clone()
    ri0[..., 0] = 0
    np.savez(os.path.join(HERE, "sinesrc.npz"), f0=f0s.numpy(), rand_ini=ri0.numpy().reshape(-1), noise=nzs.numpy(),
             weight=src.l_linear.weight.detach().numpy(), bias=src.l_linear.bias.detach().numpy(),
             out=merged.numpy()[..., 0])

    # ---- log-mel front-end ---------------------------------------------------
    import nsf_hifigan.nvSTFT as nv
    basis = O.mel_filterbank_slaney(44100, 2048, 128, 40, 16000)
    with mock.patch.object(nv, "librosa_mel_fn", side_effect=lambda **kw: basis):
        for tag, Tn in (("a", 20 * 512), ("t1024", 1024), ("t512", 512)):
            g = torch.Generator().manual_seed(len(tag) + Tn)
            t = torch.arange(Tn) / 44100.0
            y = sum(0.3 / k * torch.sin(2 * np.pi * 220.0 * k * t + k) for k in range(1, 40))
            y = torch.stack([y + 0.05 * torch.randn(Tn, generator=g), 0.2 * torch.randn(Tn, generator=g)])
            stft = nv.STFT(44100, 128, 2048, 2048, 512, 40, 16000)
            mel = stft.get_mel(y)
            np.savez(os.path.join(HERE, f"mel_{tag}.npz"), audio=y.numpy(), mel=mel.numpy(),
                     basis=basis if tag == "a" else np.zeros(0, np.float32))

    loss_fixture()
    baseline_shape_fixtures()
    mel_shifted_fixtures()

    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
