#!/usr/bin/env python
"""The dense tap synthesis k_ir_gemm (every n_mag other than 256) at B = 32 x 10 s: time per launch, fraction of the f32 MFMA
roof (2 n N multiply-adds per frame for a real response over the mirrored half, twice that for a complex one) and of the HBM
roof (control in, taps out), beside the prime-factor kernel at 256 bins.  VERDICT r2 #7 asks for these where k_ir_gemm stays."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ddsp_svc_amd import core

dev = torch.device("cuda:0")
B, F = 32, 862
g = torch.Generator().manual_seed(3)
out = {}
for n in (128, 256, 257, 512):
    N = 2 * (n - 1)
    c = torch.randn(B, F, n, generator=g).to(dev)
    mag = torch.exp(c)
    z = torch.complex(torch.randn(B, F, n, generator=g), torch.randn(B, F, n, generator=g)).to(dev)
    hw = (torch.rand(B, F, 1, generator=g) * 300 + 20).to(dev)
    cases = {"hann (real)": lambda: core.frequency_impulse_response(mag),
             "dynamic (real)": lambda: core.frequency_impulse_response(mag, half_width_frames=hw),
             "roll (complex)": lambda: core.frequency_impulse_response(z, hann_window=False)}
    for name, fn in cases.items():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        cplx = 2 if "complex" in name else 1
        flops = 2.0 * n * (N / 2 + 1) * cplx * B * F          # contraction over the mirrored half of the taps
        byts = 4.0 * (cplx * n + N) * B * F
        out["n_mag %d, %s" % (n, name)] = {"ms": round(ms, 4), "kernel": "k_taps_pfa510" if n == 256 else "k_ir_gemm",
                                          "mfma_frac": round(flops / (ms * 1e-3) / 157.3e12, 4),
                                          "hbm_frac": round(byts / (ms * 1e-3) / 8e12, 4), "us_per_1000_frames": round(ms * 1e3 / (B * F / 1000.0), 3)}
print(json.dumps(out, indent=1))
