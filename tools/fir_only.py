#!/usr/bin/env python
"""Run only the FIR kernel variants a few times (for rocprofv3 --pmc passes)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ddsp_svc_amd import _ffi

dev = torch.device("cuda:0")
B, F, n, HOP = 32, 862, 256, 512
T, N = F * HOP, 2 * (n - 1)
L = _ffi.lib()
st = torch.cuda.current_stream().cuda_stream
x = torch.rand(B, T, device=dev) * 2 - 1
taps = torch.randn(B, F, N, device=dev) / N ** 0.5
y = torch.empty(B, T, device=dev)
impls = [int(a) for a in sys.argv[1:]] or [2, 3]
for impl in impls:
    for _ in range(5):
        _ffi.check(L.ddsp_hip_fft_convolve(x.data_ptr(), 0, taps.data_ptr(), None, y.data_ptr(), None, B, F, HOP, N, impl, st))
torch.cuda.synchronize()
