#!/usr/bin/env python
"""Time the spectral-tail forward / backward kernels and k_mel through the ctypes entry points (events on the launch
stream, no Python work inside the timed loop besides the call itself)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ddsp_svc_amd import _ffi, mel as M

dev = torch.device("cuda:0")
B, F, HOP = 32, 862, 512
T = F * HOP
L = _ffi.lib()
st = torch.cuda.current_stream().cuda_stream


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / reps, 4)


res = {}
for win in (2048, 1024):
    n = win // 2 + 1
    exc, nz, R = (torch.randn(B, T, device=dev) for _ in range(3))
    c = torch.randn(B, F, 4 * n, device=dev)
    hm, hp, nm, nph = torch.split(c, [n] * 4, dim=-1)
    w = torch.hann_window(win, device=dev)
    out = torch.empty(B, T, device=dev)
    d = [torch.empty(B, F, n, device=dev) for _ in range(4)]
    ld = c.stride(1)
    sup = win == 2048
    res["win%d_forward_ms" % win] = timeit(lambda: _ffi.check(L.ddsp_hip_stft_filter(
        exc.data_ptr(), nz.data_ptr(), 0, hm.data_ptr(), ld, hp.data_ptr(), ld, nm.data_ptr(), ld,
        nph.data_ptr() if sup else None, ld, 1 / 128, w.data_ptr(), win, int(sup), int(sup), B, F, HOP, out.data_ptr(), st)))
    res["win%d_backward_ms" % win] = timeit(lambda: _ffi.check(L.ddsp_hip_stft_filter_backward(
        exc.data_ptr(), nz.data_ptr(), 0, hm.data_ptr(), ld, hp.data_ptr(), ld, nm.data_ptr(), ld,
        nph.data_ptr() if sup else None, ld, 1 / 128, w.data_ptr(), win, int(sup), int(sup), R.data_ptr(), B, F, HOP,
        d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr() if sup else None, st)))
stft = M.STFT(44100, 128, 2048, 2048, 512, 40, 16000)
y = torch.randn(B, T, device=dev) * 0.1
basis, (band, packed), window = stft._tables(dev)
frames = L.ddsp_hip_mel_frames(T, 2048, 512)
store = torch.empty(B, frames, 128, device=dev)
for wps in (2, 3):
    _ffi.set_tuning("MEL_WPS", wps)
    res["mel_wps%d_ms" % wps] = timeit(lambda: _ffi.check(L.ddsp_hip_mel_spectrogram(
        y.data_ptr(), B, T, window.data_ptr(), 2048, 512, basis.data_ptr(), band.data_ptr(), packed.data_ptr(),
        packed.numel(), 128, 1e-5, store.data_ptr(), frames * 128, 1, 128, st)))
print(json.dumps(res, indent=1))
