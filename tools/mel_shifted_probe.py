#!/usr/bin/env python
"""get_mel with a key shift (nsf_hifigan/nvSTFT.py:82-116; main_diff.py:359's formant shift, preprocess.py:88-92's augmentation) on
csrc/mel_czt.hip: ms per call at B x 10 s for a few shifts, the one-off table build, and -- as the yardstick, not a product path --
the same quantity out of torch-ROCm's own operators (pad, torch.stft at the shifted length, magnitude, matmul, log), whose FFT
library takes lengths like 2536 = 2^3 317 through Bluestein passes in HBM.  The largest difference between the two is printed
(float32 both; the parity tests hold the kernel to the float64 oracle)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ddsp_svc_amd import mel as M

dev = torch.device("cuda:0")
B = int(os.environ.get("B", "32"))
T = 862 * 512
y = (0.3 * torch.randn(B, T, generator=torch.Generator().manual_seed(1))).to(dev)
stft = M.STFT(44100, 128, 2048, 2048, 512, 40, 16000)
basis = stft._tables(dev)[0]


def timeit(fn, reps=40, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def torch_operators(y, keyshift):
    n_new, win_new, hop_new = M._shifted_sizes(2048, 2048, 512, keyshift, 1)
    window = torch.hann_window(win_new, device=y.device)
    left = (win_new - hop_new) // 2
    right = max((win_new - hop_new + 1) // 2, win_new - y.size(-1) - left)
    yp = torch.nn.functional.pad(y.unsqueeze(1), (left, right), mode="reflect" if right < y.size(-1) else "constant").squeeze(1)
    z = torch.stft(yp, n_new, hop_length=hop_new, win_length=win_new, window=window, center=False, return_complex=True)
    mag = torch.sqrt(z.real ** 2 + z.imag ** 2 + 1e-9)
    if keyshift != 0:
        if mag.size(1) < 1025:
            mag = torch.nn.functional.pad(mag, (0, 0, 0, 1025 - mag.size(1)))
        mag = mag[:, :1025] * 2048 / win_new
    return torch.log(torch.clamp(basis @ mag, min=1e-5))


print("B = %d x %d samples (%d frames each)" % (B, T, T // 512))
t_plain = timeit(lambda: stft.get_mel(y))
print("keyshift 0 (k_mel, 2048-point FFT, two frames per transform): %.3f ms" % t_plain)
for ks in (-5, -1, 2, 3.7, 7, 12):
    n_new, win_new, _ = M._shifted_sizes(2048, 2048, 512, ks, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = stft.get_mel(y, keyshift=ks)
    torch.cuda.synchronize()
    t_first = (time.perf_counter() - t0) * 1e3
    t_k = timeit(lambda: stft.get_mel(y, keyshift=ks))
    ref = torch_operators(y, ks)
    t_t = timeit(lambda: torch_operators(y, ks), reps=10, warm=3)
    lin, rlin = torch.exp(out.double()), torch.exp(ref.double())
    d = float(((lin - rlin).abs() / rlin.amax(dim=1, keepdim=True)).max())
    print("keyshift %5.1f: %4d points, %d chunk(s): kernel %.3f ms (first call with its tables %.2f ms) | torch operators %.2f ms "
          "(%.1fx) | largest difference %.1e of a frame's peak band" % (ks, n_new, (n_new + 2047) // 2048, t_k, t_first, t_t, t_t / t_k, d))
