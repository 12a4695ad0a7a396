#!/usr/bin/env python
"""Time the FIR kernel variants with events on the launch stream (B=32 x 10 s, N=510), incl. env-selected
sub-variants of the hop-block FFT form (DDSP_HIP_BLK_WPS, DDSP_HIP_BLK_RUN)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ddsp_svc_amd import _ffi

dev = torch.device("cuda:0")
B, F, n, HOP = int(os.environ.get("B", 32)), 862, 256, 512
T, N = F * HOP, 2 * (n - 1)
L = _ffi.lib()
st = torch.cuda.current_stream().cuda_stream
x = torch.rand(B, T, device=dev) * 2 - 1
taps = torch.randn(B, F, N, device=dev) / N ** 0.5
y = torch.empty(B, T, device=dev)


def timeit(impl, reps=20):
    def once():
        _ffi.check(L.ddsp_hip_fft_convolve(x.data_ptr(), 0, taps.data_ptr(), None, y.data_ptr(), None, B, F, HOP, N, impl, st))
    for _ in range(3):
        once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        once()
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / reps, 4)


res = {"impl3_mfma8_ms": timeit(3), "impl4_fft2048_ms": timeit(4)}
y4 = y.clone()
for wps in os.environ.get("WPS", "3").split():            # (2: only in a -DDDSP_AB_GENERATIONS build; the product library refuses the knob)
    try:
        _ffi.set_tuning("BLK_WPS", int(wps) if wps != "3" else 0)
    except RuntimeError:
        continue
    for run in os.environ.get("RUNS", "0").split():
        if run != "0":
            _ffi.set_tuning("BLK_RUN", int(run))
        else:
            _ffi.set_tuning("BLK_RUN", 0)
        res["impl5_blk_wps%s_run%s_ms" % (wps, run)] = timeit(5)
res["rel_rms_impl5_vs_impl4"] = float(((y - y4).double().pow(2).mean().sqrt() / y4.double().pow(2).mean().sqrt()))
print(json.dumps(res, indent=1))
