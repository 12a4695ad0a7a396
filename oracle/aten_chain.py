"""CPU baseline that walks the reference's OP CHAIN with torch CPU operators (TEST / BENCH INFRASTRUCTURE ONLY).

``oracle/ddsp_oracle.py`` restates the algorithm in numpy with its own formulations (closed-form weights, a direct
definition of the time-varying filter); its run time therefore says little about what the reference costs on a host.
This file restates the same path as the sequence of ATen kernels the reference dispatches -- linear
``interpolate``, float64 ``cumsum``, ``sinc``, ``irfft`` + ``roll`` + window, zero ``pad`` + ``unfold`` + periodic
Bartlett window, ``rfft`` / ``irfft`` at ``2 hop + N - 1`` points (1533 for the BASELINE shapes), ``fold`` overlap-add,
crop -- so that ``bench.py`` can time "the reference's cost on this box's cores" without the reference checkout (which
cannot travel to the GPU box).  Only ``tests/`` and ``bench.py``'s ``cpu_baseline`` leg import it (which also times the
same chain with the tensors on the GPU: what running the reference's DSP under PyTorch-ROCm costs there); it is pinned to
the same reference-generated fixtures as the numpy oracle (tests/test_aten_chain.py).

Each function names the reference lines whose op sequence it follows.
"""
import math

import torch
import torch.nn.functional as F


def to_sample_rate(ctrl, hop):
    """core.py:66-70 -- [B,F,C] -> [B,F*hop,C]: linear interpolation with the last frame held (one repeated frame
    appended, ``align_corners`` interpolation to F*hop+1 points, last point dropped)."""
    c = ctrl.transpose(1, 2)
    c = torch.cat([c, c[..., -1:]], dim=-1)
    up = F.interpolate(c, size=(c.shape[-1] - 1) * hop + 1, mode="linear", align_corners=True)
    return up[..., :-1].transpose(1, 2)


def wrapped_cycles(f0_frames, sr, hop, infer=True):
    """vocoder.py:564-575 / :819-829 -- per-sample f0, cumulative phase in cycles (float64 when ``infer``), wrapped to
    [-0.5, 0.5], back in float32; also ``phase_frames``."""
    f0 = to_sample_rate(f0_frames, hop)
    acc = torch.cumsum(f0.double() / sr, dim=1) if infer else torch.cumsum(f0 / sr, dim=1)
    acc = acc - torch.round(acc)
    x = acc.to(f0.dtype)
    return f0, x, 2 * math.pi * x[:, ::hop, :]


def taps_from_response(resp, window=True, half_width=None):
    """core.py:254-270 (+ :185-251) -- one-sided response [B,F,n] (complex) -> causal taps [B,F,2(n-1)]."""
    ir = torch.fft.irfft(resp)
    n_taps = ir.shape[-1]
    if not window:
        return ir.roll(n_taps // 2, -1)
    if half_width is None:                                   # periodic Hann of the full length, zero-phase then causal
        w = torch.hann_window(n_taps, dtype=ir.dtype, device=ir.device).roll(n_taps // 2, -1)
        return (ir * w).roll(n_taps // 2, -1)
    pos = torch.arange(-(n_taps // 2), (n_taps + 1) // 2, dtype=ir.dtype, device=ir.device) / half_width      # core.py:244
    pos[pos > 1] = 0                                                                        # core.py:245 (one-sided)
    return ir.roll(n_taps // 2, -1) * ((1 + torch.cos(math.pi * pos)) / 2)


def framewise_convolve(audio, taps):
    """core.py:120-182 -- 50 %-overlapping Bartlett frames, product of spectra at ``2 hop + N - 1`` points, overlap-add
    with ``fold``, delay compensation of N/2."""
    B, T = audio.shape
    Fr, n_taps = taps.shape[1], taps.shape[2]
    hop = T // Fr
    frames = F.pad(audio, (hop, hop)).unfold(1, 2 * hop, hop) * torch.bartlett_window(2 * hop, dtype=audio.dtype, device=audio.device)
    size = 2 * hop + n_taps - 1
    spec = torch.fft.rfft(frames, size) * torch.fft.rfft(torch.cat([taps, taps[:, -1:]], dim=1), size)
    pieces = torch.fft.irfft(spec, size)                      # [B, F+1, size]
    total = Fr * hop + size
    ola = F.fold(pieces.transpose(1, 2), output_size=(1, total), kernel_size=(1, size), stride=(1, hop))
    ola = ola.reshape(B, total)[:, hop:]
    start = n_taps // 2
    return ola[:, start:start + T]


def filter_with_response(audio, resp, window=True, half_width=None):
    """core.py:273-280"""
    return framewise_convolve(audio, taps_from_response(resp, window, half_width))


def combsub_tail(f0_frames, c_gd, c_harm, c_noise, noise, sr=44100, hop=512, infer=True):
    """CombSub.forward without Unit2Control (vocoder.py:819-862): raw controls in, (signal, harmonic, noise) out.
    ``noise`` is the ``2u-1`` draw."""
    f0, x, _ = wrapped_cycles(f0_frames, float(sr), hop, infer)
    comb = torch.sinc(sr * x / (f0 + 1e-3)).squeeze(-1)                                            # :839-840
    allpass = torch.exp(1j * torch.cumsum(math.pi * torch.tanh(c_gd), dim=-1))                     # :834, :845
    harm = filter_with_response(comb, allpass, window=False)
    src = torch.exp(c_harm)
    harm = filter_with_response(harm, torch.complex(src, torch.zeros_like(src)), True, 1.5 * sr / (f0_frames + 1e-3))
    nzp = torch.exp(c_noise) / 128
    nz = filter_with_response(noise, torch.complex(nzp, torch.zeros_like(nzp)), True)
    return harm + nz, harm, nz


def sins_tail(f0_frames, c_amp, c_gd, c_noise, noise, sr=44100, hop=512, infer=True, chunk=32):
    """Sins.forward without Unit2Control (vocoder.py:564-611)."""
    f0, x, _ = wrapped_cycles(f0_frames, float(sr), hop, infer)
    phase = 2 * math.pi * x
    amp = torch.exp(c_amp) / 128
    H = amp.shape[-1]
    order = torch.arange(1, H + 1, dtype=phase.dtype, device=phase.device)
    amp = amp * ((f0_frames * order < sr / 2).float() + 1e-7)                                      # core.py:73-77
    sinus = 0.
    for lo in range(0, H, chunk):                                                                  # :588-594
        sinus = sinus + (torch.sin(phase * order[lo:lo + chunk]) * to_sample_rate(amp[:, :, lo:lo + chunk], hop)).sum(-1)
    allpass = torch.exp(1j * torch.cumsum(math.pi * torch.tanh(c_gd), dim=-1))
    harm = filter_with_response(sinus, allpass, window=False)
    nzp = torch.exp(c_noise) / 128
    nz = filter_with_response(noise, torch.complex(nzp, torch.zeros_like(nzp)), True)
    return harm + nz, harm, nz
