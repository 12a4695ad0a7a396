#!/usr/bin/env python
"""Repeat the forms-agree check of tests/test_fullsize_gpu.py many times in one process (a timing-dependent fault of one
of the three FIR forms would show as an occasional mismatch) -- exit status 1 and the offending iteration on a mismatch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ddsp_svc_amd import _ffi, core

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda:0")
B, F, HOP, N = 32, 862, 512, 510
rms = lambda a: float(torch.sqrt(torch.mean(torch.square(a.double()))))
g = torch.Generator().manual_seed(4)
x = (torch.rand(B, F * HOP, generator=g) * 2 - 1).to(dev)
taps = (torch.randn(B, F, N, generator=g) / N ** 0.5 * torch.rand(B, F, 1, generator=g) * 4).to(dev)
ref3 = core.fft_convolve(x, taps, impl=3)
bad = 0
for it in range(n):
    for impl in (3, 4, 5):
        y = core.fft_convolve(x, taps, impl=impl)
        e = rms(y - ref3) / rms(ref3)
        if not e <= 1.5e-6:
            bad += 1
            print("iteration", it, "impl", impl, "relative rms", e, "max abs", float((y - ref3).abs().max()),
                  "bad samples", int(((y - ref3).abs() > 1e-3).sum()))
print("iterations", n, "mismatches", bad)
sys.exit(1 if bad else 0)
