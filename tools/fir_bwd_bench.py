#!/usr/bin/env python
"""Time the adjoint of the hop-block FIR (ddsp_hip_fft_convolve_backward, B = 32 x 10 s, N = 510) with and without
the input gradient."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ddsp_svc_amd import _ffi

dev = torch.device("cuda:0")
B, F, n, HOP = 32, 862, int(os.environ.get("NBINS", 256)), 512
T, N = F * HOP, 2 * (n - 1)
L = _ffi.lib()
st = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
x = torch.rand(B, T, device=dev) * 2 - 1
g = torch.randn(B, T, device=dev)
taps = torch.randn(B, F, N, device=dev) / N ** 0.5
dx = torch.empty(B, T, device=dev)
dt = torch.empty(B, F, N, device=dev)


def timeit(with_dx, reps=int(os.environ.get("REPS", 400))):      # long enough to be past the clocks' transient (~100 launches)
    def once():
        _ffi.check(L.ddsp_hip_fft_convolve_backward(x.data_ptr(), 0, taps.data_ptr(), g.data_ptr(),
                                                    dx.data_ptr() if with_dx else None, dt.data_ptr(), B, F, HOP, N, st))
    for _ in range(max(3, reps // 2)):
        once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        once()
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / reps, 4)


print(json.dumps({"bwd_taps_only_ms": timeit(False), "bwd_taps_and_input_ms": timeit(True),
                  "checksum": float(dt.double().abs().sum()), "checksum_dx": float(dx.double().abs().sum())}))
