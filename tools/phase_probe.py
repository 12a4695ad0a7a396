#!/usr/bin/env python
"""Why does k_phase_frame_sums take 34 us inside the CombSub step and 18 us inside the Sins step?  Time the phase
call alone, and right behind each kind of kernel of the steps."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from ddsp_svc_amd import _ffi, core, synth

dev = torch.device("cuda:0")
SR, HOP, n = 44100, 512, 256
N = 2 * (n - 1)
B, F = 32, 862
T = F * HOP
L = _ffi.lib()
st = torch.cuda.current_stream().cuda_stream
f0, ctrls, noise = bench.make_inputs("combsub", B, F, (n, n, n), dev, 1234)
rows = B * F
tab = core.ir_table(n, dev)
taps = torch.randn(B, F, N, device=dev) / N ** 0.5
y = torch.empty(B, T, device=dev)
tp = torch.empty(rows, N, device=dev)
hw = ((1.5 * SR) / (f0.reshape(-1) + 1e-3)).contiguous()
c = ctrls[1]


def fir():
    _ffi.check(L.ddsp_hip_fft_convolve(noise.data_ptr(), 0, taps.data_ptr(), None, y.data_ptr(), None, B, F, HOP, N, 0, st))


def gemm_dyn():
    _ffi.check(L.ddsp_hip_impulse_response(c.data_ptr(), c.stride(1), None, 0, 1, 1.0, 2, hw.data_ptr(), rows, n, tab.data_ptr(), tp.data_ptr(), st))


def gemm_hann():
    _ffi.check(L.ddsp_hip_impulse_response(c.data_ptr(), c.stride(1), None, 0, 1, 1.0, 1, None, rows, n, tab.data_ptr(), tp.data_ptr(), st))


def phase():
    return synth.phase(f0, SR, HOP)


def time_after(pre, what, reps=30):
    tot = 0.0
    for i in range(reps + 3):
        if pre:
            pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        what()
        e1.record()
        torch.cuda.synchronize()
        if i >= 3:
            tot += e0.elapsed_time(e1)
    return round(tot / reps * 1e3, 1)


for _ in range(300):
    fir()
torch.cuda.synchronize()
res = {}
for name, pre in (("alone", None), ("after_fir", fir), ("after_gemm_dyn", gemm_dyn), ("after_gemm_hann", gemm_hann)):
    res["phase_us_" + name] = time_after(pre, phase)
    res["gemm_dyn_us_" + name] = time_after(pre, gemm_dyn)
    res["gemm_hann_us_" + name] = time_after(pre, gemm_hann)
    res["fir_us_" + name] = time_after(pre, fir)
print(json.dumps(res, indent=1))
